// ps_sort.hpp — device-side canonical sort of full result lists (K4, ps_sort.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace ps {

struct SortBuffers {
  uint32_t* doc;         // in/out: doc ids of all runs, [n_items]
  uint64_t* score_bits;  // in/out: f64 scores as bits
  uint32_t* doc_tmp;     // scratch, [n_items]
  uint64_t* score_tmp;   // scratch, [n_items]
  uint32_t* seg_begin;   // scratch, [n_segments]
  uint32_t* seg_end;     // scratch, [n_segments]
};

// Sorts every segment [off[q], off[q] + cnt[q]) by (score desc, doc asc), in place (result in
// doc / score_bits).  Call with temp == nullptr to get the required temp_bytes.
hipError_t sort_results(SortBuffers& b, unsigned n_items, unsigned n_segments, const uint64_t* d_off,
                        const uint32_t* d_cnt, void* temp, size_t& temp_bytes, hipStream_t st);

// Same ordering for ONE run [offset, offset + n) with device-wide sorts (large single queries).
hipError_t sort_run(SortBuffers& b, size_t offset, unsigned n, void* temp, size_t& temp_bytes, hipStream_t st);

// Few / huge runs (a query with ~10^6 matches): ONE set of device-wide sorts for all runs of the batch instead of
// sorts per run (36 launches of 5-8 us per run were 9 ms for 24 runs of 0.55 M).  The runs are copied next to each
// other, then three stable LSD sorts of (key, permutation index): doc id ascending (only the bits n_docs needs),
// score descending (64 bits), query ascending (only the bits n_runs needs; skipped for one run) - after the last one
// run q occupies [cmp_off[q], cmp_off[q+1]) again, in canonical order.  pack_sorted() then writes the first
// out_off[q+1] - out_off[q] results of every run as {key, score} records.
struct GlobalSort {
  // inputs (device)
  const uint32_t* doc;         // doc ids, run q at run_off[q]
  const uint64_t* score_bits;  // f64 scores as bits, same positions
  const uint64_t* run_off;     // [n_runs_total]
  const uint64_t* cmp_off;     // [n_runs_total + 1] offsets of the runs laid next to each other (sum of counts)
  uint32_t n_docs;
  // the part of the batch this call sorts: runs [run0, run0 + n_runs), compact indices [first, first + n)
  // (several parts of one batch share the scratch arrays: a part only touches its own index range, so the
  // download of one part can overlap the sorts of the next)
  uint32_t run0;
  uint32_t n_runs;
  uint32_t first;              // cmp_off[run0]
  uint32_t n;                  // cmp_off[run0 + n_runs] - first
  // scratch (device), one element per result of the whole batch
  uint32_t* kd;      // doc ids, compact (kept for pack_sorted)
  uint64_t* sc;      // scores, compact (kept for pack_sorted)
  uint32_t* k32;     // sort scratch
  uint32_t* ia;      // permutation ping (values: compact indices of the whole batch)
  uint32_t* ib;      // permutation pong
  uint64_t* ks;      // score keys in doc order
  uint64_t* ks_out;  // sort scratch
  // output of sort_runs_global: which of ia / ib holds the final permutation
  const uint32_t* perm;
};
hipError_t sort_runs_global(GlobalSort& g, void* temp, size_t& temp_bytes, hipStream_t st);
// out[o] for o in [out_off[q], out_off[q+1]) = {keys[doc], score} of run q's (o - out_off[q])-th result, for the
// runs of the part; out_first / n_out = out_off[run0] and the part's number of records (host-known).
hipError_t pack_sorted(const GlobalSort& g, const uint64_t* out_off, uint64_t out_first, uint64_t n_out,
                       const uint64_t* keys, void* out, hipStream_t st);

}  // namespace ps
