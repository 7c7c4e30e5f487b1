// ps_kernels_common.hpp — what every scoring kernel shares: build-time knobs, the kernel parameter block (KParams), the K1d work
// descriptors, work counters, wave-level helpers and the wave top-K.  Part of ps_kernels.hpp (one translation unit: ps_engine.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cstdint>

#include "../../include/probly_search_amd.h"

namespace ps {

constexpr int MAX_F = 8;
constexpr int WAVE = 64;
// Build-time shape of K1 (the defaults are the measured optimum on C2..C5; K1 must stay within
// 128 VGPRs for 4 waves per SIMD - tools/kernel_resources.py):
#ifndef PS_UNROLL
#define PS_UNROLL 2          // 64-posting trips per lane in flight in the streaming loop
#endif
#ifndef PS_WG_WAVES
#define PS_WG_WAVES 4        // waves per workgroup when two 8-wave workgroups do not fit a CU
#endif
#ifndef PS_G
#define PS_G 3               // plan entries whose ranges + first trips are requested together
#endif
#ifndef PS_FU
#define PS_FU 1              // postings per lane in a prefetched first trip
#endif
#ifndef PS_Z21_HARVEST_UNROLL_1F
#define PS_Z21_HARVEST_UNROLL_1F 4  // zero_to_one harvest with one live field: 16-byte LDS reads in flight per lane
#endif
#ifndef PS_DAAT_UM
#define PS_DAAT_UM 2             // K1d, multi-expansion arm: postings per lane in flight
#endif
#ifndef PS_DAAT_MQ
#define PS_DAAT_MQ 1             // K1d, multi-expansion arm: survivors of the first lookup level wait in a wave-private LDS queue (0: the round-2 arm)
#endif
#ifndef PS_DAAT_UMQ
#define PS_DAAT_UMQ 3            // ... postings per lane in flight in its scan stage (4: 12 KB of reach ring, 4.5 waves per SIMD by LDS: C5 1.715 ms against 1.60 at 3 and 1.61 at 2)
#endif
#ifndef PS_DAAT_MRQ
#define PS_DAAT_MRQ 1            // ... the postings that pass the first bound test wait in a reach queue until 64 are together (the first lookup with every lane busy)
#endif
#ifndef PS_HARVEST_UNROLL
#define PS_HARVEST_UNROLL 8  // 16-byte LDS reads in flight per lane while a tile is harvested
#endif
#ifndef PS_FUSED_UNROLL
#define PS_FUSED_UNROLL 4    // ... when a dense row is added during the harvest (row loads fly too)
#endif
#ifndef PS_DAAT_U
#define PS_DAAT_U 4          // K1d: postings per lane whose lookups are in flight together
#endif
#ifndef PS_DAAT_WGW
#define PS_DAAT_WGW 2        // K1d: waves per workgroup (they share the LUT copy; 8 / 4 / 2 measured 0.51 / 0.46 / 0.44 ms on C2: a workgroup holds its slots until its slowest wave ends)
#endif
#ifndef PS_ABLATE_BUILD
#define PS_ABLATE_BUILD 0    // profiling builds only: honour KParams::ablate in the hot loops
#endif
constexpr int DAAT_WGW = PS_DAAT_WGW;
constexpr int UNROLL = PS_UNROLL;
constexpr int WG_WAVES = PS_WG_WAVES;   // each wave owns its own LDS tile
constexpr int MERGE_WAVES = 16;         // most waves per workgroup of K3 (the host sizes it to the candidates)
constexpr int LUT_TF = 16;              // LUT columns: term frequency 0..15

struct RowDesc {  // one hot (list, idf, expansion_boost) combination K0b has to score into its row slot
  uint64_t post_off;
  uint32_t len;
  uint32_t _pad;
  double idf, eb;
  uint32_t slot;     // row slot in the snapshot's row slab
  uint32_t tbl_off;  // the list's tile-offset table (one slot per tile), or NO_TABLE: the host zero-fills the row
};
constexpr uint32_t NO_TABLE = 0xFFFFFFFFu;

// K1d work descriptors (built on the device: ps_prep_kernels.hpp)
struct DEntry {        // per plan entry
  double skip_thr;     // upper bound of any document that only occurs in this list and lists with lower bounds
  double others;       // upper bound of what every OTHER entry of the query can add to a document of this list
  double ub;           // upper bound of any posting score of this list
  uint32_t rank;       // position in the query's processing order (0 = highest upper bound); the dedupe order
  uint32_t q;          // query of the entry
};
struct DGroup {        // per plan entry, for queries with several expansions per query term
  double ub_s;         // this entry's bound, inflated (1e-9)
  double nxt_s;        // inflated bound of the next entry (rank order) of the SAME query term, 0 = none
  uint32_t grp;        // dense ordinal of the entry's query term within the query; 0xFFFFFFFF = more than 4 terms
  uint32_t _pad[3];
};
struct DItem {         // a chunk of one list
  uint32_t entry;      // plan entry
  uint32_t begin;      // first posting of the chunk within the list
  uint32_t count;
  uint32_t slot;       // candidate slot (query-major)
  // copies of DEntry::skip_thr / q of the list: nine workgroups in ten only exist to find their list non-essential and
  // leave, and with these here that costs two dependent loads (item -> threshold) instead of three
  double skip_thr;
  uint32_t q;
  uint32_t _pad;
};

struct DItemGen {      // per plan entry: k_prep_items expands it into its DItems on the device
  uint32_t entry;      // plan entry
  uint32_t item_at;    // its first item
  uint32_t chunk;      // postings per item
  uint32_t first_slot; // candidate slot of its first item
};

constexpr uint32_t DENSE_FLAG = 0x80000000u;  // ps_plan_entry::shift bit 31: entry reads dense row `node`
constexpr uint32_t DENSE_ASSIGN_FLAG = 0x40000000u;  // ... as the tile's first contribution: written, not added
constexpr uint32_t DENSE_FUSE_FLAG = 0x20000000u;    // ... as the query's last one: added while harvesting

struct KParams {
  const uint32_t* doc;
  const uint32_t* tf;
  const uint32_t* fl;
  const uint32_t* table;
  const uint64_t* keys;
  const ps_plan_entry* plan;
  const uint32_t* qbeg;
  const uint32_t* qterms_len;  // zero_to_one
  const uint32_t* qorder;      // [B] queries in the order K1 hands them out within a run (heaviest first)
  const uint32_t* gen_queries; // zero_to_one: the n_general queries k_z21 has to run
  const uint32_t* qflags;      // zero_to_one: bit 0 = "simple" query (k_score<MODE_Z21S> owns it)
  uint32_t slice_bytes;        // per-wave LDS for the table slices (0 = look ranges up in global memory)
  const uint32_t* zorder;      // zero_to_one: per query, entry indices sorted by (score desc, plan order)
  const double* zfub;          // zero_to_one: [B][F] upper bound of any document's pool of field x for query q (null: none)
  uint64_t P;
  uint32_t B, n_tiles, T, S, n_super, K, n_docs, F, max_qterms, z_nodes, z_tile, z_qwords;
  double k1, k1p1, one_minus_b, b;
  double avg[MAX_F], boost[MAX_F];
  // saturated-tf LUT (see k_bm25_lut): rows of LUT_TF doubles, row = lut_base[x] + field_length
  const double* lut;
  uint32_t lut_rows, lut_stride;  // entry (tf, row) lives at tf * lut_stride + row; stride is odd
  uint32_t lut_cap[MAX_F], lut_base[MAX_F];
  // Dense rows (see k_dense_rows): per-document f64 score of the batch's hot lists, one row each
  const double* rows;
  const RowDesc* row_desc;
  uint64_t row_stride;  // doubles per row plane = n_tiles * T
  uint32_t n_rows;
  uint32_t row_mode;    // MODE_BM25 | MODE_Z21S: what k_dense_rows evaluates
  uint32_t row_planes;  // 1 (BM25 score) | F (zero_to_one: one contribution plane per field)
  uint64_t layout_bytes;         // host-side bookkeeping: bytes of the layout actually streamed
  uint32_t z_masked;             // host-side: some simple query needs the consumed-query-term masks
  uint32_t n_simple, n_general;  // host-side bookkeeping (zero_to_one query classes in this batch)
  uint32_t ablate;  // PS_ABLATE debug bit mask (profiling only): 1 = no top-k offer, 2 = no scoring
  // K1d k_daat (exact dynamic pruning, see there)
  const struct DEntry* dentry;  // [n_plan_entries], parallel to plan[]
  const struct DItem* ditems;   // [n_ditems] in processing order (highest upper bound first)
  const uint32_t* qslot;        // [B] first candidate slot (= item) of query q
  const uint32_t* qslot_n;      // [B] its candidate slots
  const uint32_t* n_ditems_dev; // the batch's item count as k_prep_finish wrote it (n_ditems below is the host's upper bound = the grid)
  unsigned long long* item_trace; // profiling builds (PS_ITEM_TRACE): [n_ditems][4] = {start, end (s_memrealtime, 100 MHz), trips | rank << 32, scanned}
  uint32_t* prep_ctl;           // the preparation's control words (ps_prep_kernels.hpp: PrepCtl), zeroed behind k_merge_items
  uint32_t prep_ctl_words;
  const uint32_t* rorder;       // [n_plan_entries] per query: its entries in rank order (highest bound first)
  const struct DGroup* dgroup;  // [n_plan_entries] (multi-expansion batches)
  // Bloom filters of the lists without a bitmap (k_build_bloom): "is document d in this sparse list" is
  // one 8-byte load of a few-KB filter - and the answer is no for > 98 % of the documents asked - instead
  // of two table words and a handful of doc ids
  const unsigned long long* bloom;        // filter words
  const unsigned long long* layer_bloom;  // [n_layers] first word (low 40 bits) | log2(words) << 58; ~0 = none
  const double* splane;         // [P][F] score plane (k_list_bounds): (tfn * idf) * boost_x of every (posting, field), 0.0 where tf_x == 0 - what K1d reads instead of re-deriving it per visit
  const uint32_t* tfl;          // [P][F] packed {tf (8 bits, 255 = see the tf plane), field length (24 bits, all ones = see the fl plane)}: what the hot loops read
  const uint32_t* bits;         // membership bitmaps of the denser lists (ps_plan_entry::bm_off)
  const uint32_t* alive;        // one bit per doc id, cleared by a delta removal; null = every document alive
  uint32_t n_ditems, t_log2;
  uint32_t item_base;           // first item of this launch (the batch may be split into two launches)
  const uint32_t* item_split_dev; // k_daat of a batch split between k_daat_small and k_daat (queries of <= 4 lists, one per query term /
                                // the others): the first item of the second part as the preparation counted it (null: item_base)
  uint32_t* cand_cnt;           // [n_ditems] candidates an item left in its slot
  uint32_t* work_counter;    // next (query, run) item for the persistent waves of k_score
  unsigned long long* wstats;  // [WS_SLOTS][WS_WORDS] work counters (always on; see WorkStats)
  unsigned long long* gthr;  // [B] bits of the best published local K-th score per query (0 = none)
  unsigned long long* gtie;  // K1dz (ps_z21_daat.hpp): [B] the same among chunks that lie below doc id D0; zeroed by k_merge_items
  const double* z_ubnum;     // K1dz: [n_plan_entries] largest record numerator of the list
  uint32_t z_dl[3];          // K1dz: the doc ids D_0 < D_1 < D_2 of the tie-threshold levels (0xFFFFFFFF: the level does not exist)
  uint32_t z_tstride;        // K1dz: words between gtie[l] and gtie[l + 1]
  double* cand_score;  // [B * n_super * K]
  uint32_t* cand_doc;
  // full-result mode
  uint32_t* full_doc;
  double* full_score;
  const uint64_t* full_off;  // [B+1]
  uint32_t* full_cnt;        // [B]
  // final outputs
  uint64_t* out_keys;
  double* out_scores;
  uint32_t* out_counts;
  const uint32_t* out_row;   // [B] the output row of query q; nullptr: q itself (set when a batch is split between two scoring kernels)
};

// ------------------------------------------------------------------------------------------
// Work counters (ps_work_counters): what the scoring kernels really read, counted by the kernels.
// A wave keeps wave-uniform counts in scalar registers (ballot + s_bcnt1, no vector registers) and
// lane 0 adds them to one of WS_SLOTS cache lines when the item ends; the host sums the slots.
// ------------------------------------------------------------------------------------------
#ifndef PS_WORK_COUNTERS
#define PS_WORK_COUNTERS 1   // 0: a build without the counters (A/B of their cost only)
#endif
constexpr uint32_t WS_SLOTS = 64, WS_WORDS = 16;  // one 128-byte line per slot
enum { WS_ITEMS_RUN = 0, WS_SCANNED, WS_REACHED, WS_ROW, WS_CELL, WS_PROBE, WS_HIT, WS_OFFER, WS_K1_ITEMS, WS_K1_POSTINGS,
       WS_K1_ROWSLICES, WS_ROWS_BUILT, WS_ROWS_USED, WS_ITEMS, WS_Z_SCANNED, WS_Z_HIT };  // (WS_Z_*: K1dz reads packed words, 4 bytes per field)
struct WorkStats {  // K1d, per item
  uint32_t scanned = 0, reached = 0, row = 0, cell = 0, probe = 0, hit = 0, offer = 0;
};
__device__ __forceinline__ uint32_t lanes_on(const bool b) {  // wave-uniform count of lanes where b holds
  return PS_WORK_COUNTERS ? (uint32_t)__popcll(__ballot(b)) : 0u;
}
#ifndef PS_REQ_TRACE
#define PS_REQ_TRACE 0   // profiling builds only (tools/build_variant.sh): k_daat_small's counters count distinct 128-byte LINES per wave-level
#endif                   // first-level load instead of lookups: probe = row lines at 8 B / doc, hit = at 2 B / doc, offer = bitmap-cell lines, reached = filter words
// Distinct lines among the lanes where `on` holds; the lanes of a trip hold ascending doc ids, so equal lines are neighbours.
__device__ __forceinline__ uint32_t distinct_lines(const bool on, const uint32_t line, const int lane) {
  const uint32_t prev = (uint32_t)__shfl_up((int)line, 1);
  const unsigned long long S = __ballot(lane == 0 || line != prev);  // run starts
  const unsigned long long A = __ballot(on);
  const unsigned long long upto = S & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
  const int start = 63 - __clzll((long long)upto);
  const unsigned long long before = ((1ull << lane) - 1ull) & ~((1ull << start) - 1ull);
  return (uint32_t)__popcll(__ballot(on && !(A & before)));
}

// ------------------------------------------------------------------------------------------
// wave-level helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int l) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// canonical order of test_util::test_score (src/lib.rs:54-58): score desc, then key asc
// (doc ids are assigned in ascending key order, so doc asc == key asc).
__device__ __forceinline__ bool better(double as, uint32_t ad, double bs, uint32_t bd) {
  return as > bs || (as == bs && ad < bd);
}

// Documents removed by a delta snapshot keep their postings (tombstones): every emission site drops them.
__device__ __forceinline__ bool doc_alive(const KParams& p, const uint32_t d) {
  return p.alive == nullptr || ((p.alive[d >> 5] >> (d & 31u)) & 1u);
}

struct TopK {
  double s;      // lane i: score of the i-th best so far (valid for i < n)
  uint32_t d;    // its doc id
  uint32_t n;    // wave-uniform fill
  double thr_s;  // K-th best (valid when n == K)
  uint32_t thr_d;
};

// Offer one candidate per lane (`has`), keep the best K.  All lanes must call.
// `gt` is a lower bound of the query's final K-th best score published by other waves of the same
// query (0 = none yet): anything strictly below it cannot be in the final top-K.
__device__ __forceinline__ void topk_offer(TopK& tk, const uint32_t K, const int lane, bool has, double v,
                                           uint32_t d, const double gt = 0.0) {
  bool cand = has && v >= gt && (tk.n < K || better(v, d, tk.thr_s, tk.thr_d));
  unsigned long long m = __ballot(cand);
  while (m) {
    const int src = __ffsll(m) - 1;
    m &= m - 1;
    const double cs = readlane_f64(v, src);
    const uint32_t cd = readlane_u32(d, src);
    if (tk.n == K && !better(cs, cd, tk.thr_s, tk.thr_d)) continue;
    const bool lb = ((uint32_t)lane < tk.n) && better(tk.s, tk.d, cs, cd);
    const uint32_t pos = (uint32_t)__popcll(__ballot(lb));
    const double us = __shfl_up(tk.s, 1);
    const uint32_t ud = __shfl_up(tk.d, 1);
    if ((uint32_t)lane > pos) { tk.s = us; tk.d = ud; }
    else if ((uint32_t)lane == pos) { tk.s = cs; tk.d = cd; }
    if (tk.n < K) tk.n++;
    if (tk.n == K) {
      tk.thr_s = readlane_f64(tk.s, (int)K - 1);
      tk.thr_d = readlane_u32(tk.d, (int)K - 1);
    }
  }
}

// Full-result mode: append this wave's present documents to the query's output run.
__device__ __forceinline__ void full_emit(const KParams& p, uint32_t q, int lane, bool has, double v, uint32_t d) {
  unsigned long long m = __ballot(has);
  if (m == 0) return;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(&p.full_cnt[q], (uint32_t)__popcll(m));
  base = readlane_u32(base, 0);
  if (has) {
    uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    uint64_t o = p.full_off[q] + base + rank;
    p.full_doc[o] = d;
    p.full_score[o] = v;
  }
}

// The same for N wave-wide groups of documents at once: ONE reservation (atomic) for all of them.  The counter of a
// query is a single address that every wave of the query adds to; device-scope atomics on one address serialise
// (~170 ns each measured: 2.7 ms for the 24 x 15 k reservations of a C2 full-result batch when every 64 documents
// made their own).
template <int N>
__device__ __forceinline__ void full_emit_group(const KParams& p, uint32_t q, int lane, const bool (&has)[N],
                                                const double (&v)[N], const uint32_t (&d)[N]) {
  unsigned long long m[N];
  uint32_t tot = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    m[i] = __ballot(has[i]);
    tot += (uint32_t)__popcll(m[i]);
  }
  if (tot == 0) return;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(&p.full_cnt[q], tot);
  uint64_t o = p.full_off[q] + readlane_u32(base, 0);
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (has[i]) {
      const uint64_t at = o + (uint32_t)__popcll(m[i] & below);
      p.full_doc[at] = d[i];
      p.full_score[at] = v[i];
    }
    o += (uint32_t)__popcll(m[i]);
  }
}

}  // namespace ps
