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

}  // namespace ps
