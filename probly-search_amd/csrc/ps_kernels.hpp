// ps_kernels.hpp — device code of the query-scoring path (gfx950 / CDNA4): kernel parameter
// block, wave-level top-K, K0 k_bm25_lut, K0b k_dense_rows, K1 k_score, K1d k_daat, K2 k_z21,
// K3 k_merge / K3d k_merge_items, the device planner k_plan, k_pack_tfl, k_upload, k_pack_results (the device-side
// preparation of a K1d batch: ps_prep_kernels.hpp).  Included by ps_engine.hip only (one translation unit); see that file's header
// comment for the kernel overview and DESIGN.md section 3 for the design.
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

// ------------------------------------------------------------------------------------------
// K1: BM25 posting accumulate + merge + per-run top-K   (bm25.rs:60-93, query.rs:61-89,150-164)
// ------------------------------------------------------------------------------------------
// The saturated term frequency bm25.rs:78-82 computes per posting-field,
//   tfn(tf, fl) = ((k1+1)*tf) / (k1*((1-b) + b*(fl/avg_x)) + tf),
// depends only on (field, tf, fl).  Each batch, k_bm25_lut evaluates THE SAME f64 expression once
// per (field, fl < lut_cap[x], tf < 16) and K1 stages the table in LDS, so the common small-integer
// case costs one LDS read instead of two IEEE f64 divisions; everything else takes the inline
// expression.  Same operations on the same operands -> bit-identical values.
__device__ __forceinline__ double bm25_tfn(const KParams& p, uint32_t x, uint32_t tfu, uint32_t flu) {
  const double tfd = (double)tfu;
  const double fld = (double)flu;
  // bm25.rs:78-82, evaluated left to right, no contraction
  return (p.k1p1 * tfd) / (p.k1 * (p.one_minus_b + p.b * (fld / p.avg[x])) + tfd);
}

// Out-of-line copy for K1's rare beyond-the-LUT path: keeps ~100 inlined IEEE division
// sequences out of the hot kernel's instruction stream.
__device__ __noinline__ double bm25_tfn_cold(double k1, double k1p1, double one_minus_b, double b, double avg,
                                             uint32_t tfu, uint32_t flu) {
  const double tfd = (double)tfu;
  const double fld = (double)flu;
  return (k1p1 * tfd) / (k1 * (one_minus_b + b * (fld / avg)) + tfd);
}

__global__ __launch_bounds__(256) void k_bm25_lut(const KParams p, double* out) {
  for (uint32_t i = threadIdx.x + blockIdx.x * blockDim.x; i < p.lut_stride * LUT_TF; i += blockDim.x * gridDim.x) {
    const uint32_t tfu = i / p.lut_stride, row = i % p.lut_stride;
    uint32_t x = 0;
    while (x + 1 < p.F && row >= p.lut_base[x] + p.lut_cap[x]) ++x;
    out[i] = row < p.lut_rows ? bm25_tfn(p, x, tfu, row - p.lut_base[x]) : 0.0;
  }
}

// ------------------------------------------------------------------------------------------
// K1: posting accumulate + merge + per-run top-K, one kernel skeleton for two scorers
//   MODE_BM25  bm25.rs:60-93 + max_score_merger (query.rs:61-89,150-164)
//   MODE_Z21S  zero_to_one (zero_to_one.rs:44-126) for "simple" queries: every entry of the
//              query has its own trie node and its own query term, so finalize's greedy scan
//              never skips a record and a (doc, field) pool is just the f64 sum of its records'
//              contributions in sorted order (score desc, stable) — the host uploads the entries
//              of such queries already in that order.  Anything else goes to k_z21.
// ------------------------------------------------------------------------------------------
// K0b: batch-level common-subexpression elimination.  A BM25 posting's score
// s(list, doc) = sum_x ((tfn*idf)*boost_x)*expansion_boost does not depend on the query, and in a
// Zipf batch a handful of head lists is visited by hundreds of queries (top-12 terms ~ 90 % of all
// posting visits in C2).  For the (list, idf, eb) combinations the host found hot and dense, this
// kernel evaluates s ONCE per posting — the very same f64 expression, so the bits are the same —
// into a dense per-document row (0.0 = no posting).  K1 then adds row values in plan order
// instead of re-streaming 20-byte postings and re-deriving the score per query.  Runs inside the
// timed step, once per batch.
// (one workgroup's share of one row: block `blk` of `nblk`)
__device__ __forceinline__ void dense_row_block(const KParams& p, double* rows, const RowDesc rd, const uint32_t blk, const uint32_t nblk) {
  double* row = rows + (uint64_t)rd.slot * p.row_planes * p.row_stride;
  // Each workgroup owns a range of tiles of the row: it zero-fills that range (coalesced 16-byte
  // stores), then scatters the scores of the postings that fall into it - found through the list's
  // tile-offset table - so the row needs no separate memset pass and every line is written while
  // it is still in L2.  (A list without a per-tile table is zero-filled by the host instead.)
  uint32_t pb = 0, pe = rd.len;
  if (rd.tbl_off != NO_TABLE) {
    const uint32_t tpb = (p.n_tiles + nblk - 1) / nblk;
    const uint32_t t0 = min(p.n_tiles, blk * tpb), t1 = min(p.n_tiles, t0 + tpb);
    if (t0 == t1) return;
    for (uint32_t x = 0; x < p.row_planes; ++x) {
      double2* z = reinterpret_cast<double2*>(row + (uint64_t)x * p.row_stride + (uint64_t)t0 * p.T);
      for (uint32_t i = threadIdx.x; i < (t1 - t0) * p.T / 2; i += blockDim.x) z[i] = make_double2(0.0, 0.0);
    }
    pb = p.table[rd.tbl_off + t0];
    pe = p.table[rd.tbl_off + t1];
    __syncthreads();  // the zeros are in place before any score of this range is stored
  } else {
    const uint32_t per = (rd.len + nblk - 1) / nblk;
    pb = min(rd.len, blk * per);
    pe = min(rd.len, pb + per);
  }
  if (p.row_mode != 0) {
    // zero_to_one.rs:117-120 per field: (min(score/tf, 1)*tf) / max(field_length, all_query_terms_len)
    for (uint32_t i = pb + threadIdx.x; i < pe; i += blockDim.x) {
      const uint64_t pi = rd.post_off + i;
      const uint32_t d = p.doc[pi];
      const uint32_t qtl = rd._pad & 0xFFFFu, need = rd._pad >> 16;
      for (uint32_t x = 0; x < p.F; ++x) {
        const uint32_t w = p.tfl[pi * p.F + x];  // packed {tf, field length} (tfl_pack); saturated sub-fields -> the exact planes
        uint32_t tfu = w >> 24, flu = w & 0xFFFFFFu;
        if (tfu == 255u) tfu = p.tf[(uint64_t)x * p.P + pi];
        if (tfu >= need) {
          if (flu == 0xFFFFFFu) flu = p.fl[(uint64_t)x * p.P + pi];
          const double df = (double)tfu;
          row[(uint64_t)x * p.row_stride + d] = fmin(rd.idf / df, 1.0) * df / (double)(flu > qtl ? flu : qtl);
        }
      }
    }
    return;
  }
  if (p.splane != nullptr) {
    // K1d batches: tfn * idf of every (posting, field) already sits in the boost-free score plane (k_list_bounds, the list's own
    // idf = rd.idf) - the row is the rest of the same expression, ((tfn * idf) * boost_x) * expansion_boost summed over the fields
    // in order (a field with tf == 0 adds +0.0): the same bits without the two f64 divisions per field
    for (uint32_t i = pb + threadIdx.x; i < pe; i += blockDim.x) {
      const uint64_t pi = rd.post_off + i;
      const uint32_t d = p.doc[pi];
      double s = 0.0;
      if (p.F == 2u) {
        const double2 v = reinterpret_cast<const double2*>(p.splane)[pi];
        s = (v.x * p.boost[0]) * rd.eb;
        s += (v.y * p.boost[1]) * rd.eb;
      } else {
        for (uint32_t x = 0; x < p.F; ++x) s += (p.splane[pi * p.F + x] * p.boost[x]) * rd.eb;
      }
      row[d] = s;
    }
    return;
  }
  for (uint32_t i = pb + threadIdx.x; i < pe; i += blockDim.x) {
    const uint64_t pi = rd.post_off + i;
    double s = 0.0;
    for (uint32_t x = 0; x < p.F; ++x) {
      const uint32_t w = p.tfl[pi * p.F + x];
      uint32_t tfu = w >> 24, flu = w & 0xFFFFFFu;
      if (tfu == 255u) tfu = p.tf[(uint64_t)x * p.P + pi];
      if (flu == 0xFFFFFFu) flu = p.fl[(uint64_t)x * p.P + pi];
      if (tfu > 0) s += bm25_tfn(p, x, tfu, flu) * rd.idf * p.boost[x] * rd.eb;
    }
    row[p.doc[pi]] = s;
  }
}
__global__ __launch_bounds__(256) void k_dense_rows(const KParams p, double* rows) {
  dense_row_block(p, rows, p.row_desc[blockIdx.y], blockIdx.x, gridDim.x);
}

// zero_to_one rows: plane x of the row goes to accumulator plane x of the tile ([F][T] in LDS).
// mask_bit != 0: the query has several expansions per query term; a (doc, field) takes the row
// value only if its consumed-query-term mask does not hold the bit yet (zero_to_one.rs:101-103).
// zero_to_one: one batch of CH x 128 documents of one field plane of a dense row
template <bool MASKS, bool ASSIGN, int CH>
__device__ __forceinline__ void dense_chunk_z(const double* r, double* accx, uint32_t* zmaskx, const int lane,
                                              const uint32_t c0, const uint32_t mask_bit) {
  double2 v[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) v[k] = *reinterpret_cast<const double2*>(r + c0 + k * 2 * WAVE + 2 * lane);
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const uint32_t i = c0 + k * 2 * WAVE + 2 * lane;
    if (ASSIGN) {  // first contribution to a zeroed tile (no masks on this path)
      *reinterpret_cast<double2*>(&accx[i]) = make_double2(v[k].x, v[k].y);
      continue;
    }
    bool t0 = v[k].x > 0.0, t1 = v[k].y > 0.0;
    if (MASKS && mask_bit) {
      uint2* zm = reinterpret_cast<uint2*>(zmaskx + i);
      const uint2 mk = *zm;
      t0 = t0 && !(mk.x & mask_bit);
      t1 = t1 && !(mk.y & mask_bit);
      if (t0 || t1) *zm = make_uint2(mk.x | (t0 ? mask_bit : 0u), mk.y | (t1 ? mask_bit : 0u));
    }
    // wave-private tile, in-order LDS: plain 16-byte read / add / write
    double2* slot = reinterpret_cast<double2*>(&accx[i]);
    double2 a = *slot;
    a.x += t0 ? v[k].x : 0.0; a.y += t1 ? v[k].y : 0.0;
    *slot = a;
  }
}

template <bool MASKS, bool ASSIGN = false>
__device__ __forceinline__ void dense_apply_z(const KParams& p, double* acc, uint32_t* zmask, const int lane,
                                              const uint32_t row, const uint32_t tile_base, const uint32_t mask_bit,
                                              const uint32_t fmask = 0xFFFFFFFFu) {
  for (uint32_t x = 0; x < p.F; ++x) {
    if (!((fmask >> x) & 1u)) continue;  // a field whose pool cannot reach the query's threshold any more
    const double* r = p.rows + ((uint64_t)row * p.F + x) * p.row_stride + tile_base;
    double* accx = acc + x * p.T;
    uint32_t* zmx = zmask + x * p.T;
    if (p.T >= 8 * 2 * WAVE) {
      for (uint32_t c0 = 0; c0 < p.T; c0 += 8 * 2 * WAVE) dense_chunk_z<MASKS, ASSIGN, 8>(r, accx, zmx, lane, c0, mask_bit);
    } else if (p.T >= 4 * 2 * WAVE) {
      for (uint32_t c0 = 0; c0 < p.T; c0 += 4 * 2 * WAVE) dense_chunk_z<MASKS, ASSIGN, 4>(r, accx, zmx, lane, c0, mask_bit);
    } else {
      dense_chunk_z<MASKS, ASSIGN, 2>(r, accx, zmx, lane, 0, mask_bit);
    }
  }
}

// One batch of CH x 128 documents of a dense row: CH 16-byte global loads in flight, then the merge.
template <bool TAGS, bool ASSIGN, int CH>
__device__ __forceinline__ void dense_chunk(const double* r, double* acc, uint16_t* tag, const int lane,
                                            const uint32_t c0, const uint16_t mytag) {
  double2 v[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) v[k] = *reinterpret_cast<const double2*>(r + c0 + k * 2 * WAVE + 2 * lane);
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const uint32_t i = c0 + k * 2 * WAVE + 2 * lane;
    if (TAGS) {
      // two documents per lane: one 16-byte accumulator access and one 4-byte tag access each way
      double2* slot = reinterpret_cast<double2*>(&acc[i]);
      uint32_t* tslot = reinterpret_cast<uint32_t*>(&tag[i]);
      double2 a = *slot;
      const uint32_t tg = *tslot;
      uint32_t t0 = tg & 0xFFFFu, t1 = tg >> 16;
      if (v[k].x > 0.0) {
        a.x = (a.x > 0.0) ? (t0 == mytag ? fmax(a.x, v[k].x) : a.x + v[k].x) : v[k].x;
        t0 = mytag;
      }
      if (v[k].y > 0.0) {
        a.y = (a.y > 0.0) ? (t1 == mytag ? fmax(a.y, v[k].y) : a.y + v[k].y) : v[k].y;
        t1 = mytag;
      }
      *slot = a;
      *tslot = t0 | (t1 << 16);
    } else if (ASSIGN) {
      // the tile is all zeros: 0.0 + v == v.  (Member-wise: copying the whole HIP vector struct out of
      // the array keeps the array in scratch.)
      *reinterpret_cast<double2*>(&acc[i]) = make_double2(v[k].x, v[k].y);
    } else {
      // plain 16-byte read / add / write: the tile is wave-private and LDS operations of a wave
      // execute in order; adding the 0.0 of a document without a posting changes nothing.  (Two
      // f64 LDS atomics per lane measured ~2x the LDS time of one b128 read + write.)
      double2* slot = reinterpret_cast<double2*>(&acc[i]);
      double2 a = *slot;
      a.x += v[k].x; a.y += v[k].y;
      *slot = a;
    }
  }
}

// Tile slice of a dense row -> accumulators.  T is a power of two >= 256: batches of 512 documents
// (1 KiB per load instruction, four in flight), or the single 256-document batch of the smallest tile.
template <bool TAGS, bool ASSIGN = false>
__device__ __forceinline__ void dense_apply(const KParams& p, double* acc, uint16_t* tag, const int lane,
                                            const uint32_t row, const uint32_t tile_base, const uint16_t mytag) {
  const double* r = p.rows + (uint64_t)row * p.row_stride + tile_base;
  if (p.T >= 8 * 2 * WAVE) {
    for (uint32_t c0 = 0; c0 < p.T; c0 += 8 * 2 * WAVE) dense_chunk<TAGS, ASSIGN, 8>(r, acc, tag, lane, c0, mytag);
  } else if (p.T >= 4 * 2 * WAVE) {
    for (uint32_t c0 = 0; c0 < p.T; c0 += 4 * 2 * WAVE) dense_chunk<TAGS, ASSIGN, 4>(r, acc, tag, lane, c0, mytag);
  } else {
    dense_chunk<TAGS, ASSIGN, 2>(r, acc, tag, lane, 0, mytag);
  }
}

enum { MODE_BM25 = 0, MODE_Z21S = 1 };

struct EntryC {      // wave-uniform per-entry constants (SGPRs)
  uint64_t post_off;
  uint32_t shift;
  uint32_t tag;      // BM25: visited tag of the entry's query term for the current tile
  double w0;         // BM25: idf              | Z21S: ScoreByTerm::score
  double w1;         // BM25: expansion_boost  | Z21S: unused
  uint32_t fmask;    // Z21S: fields still worth accumulating for this item (bit x; see k_score)
};

// The packed posting words: tf and field length of one (posting, field) in one u32, the fields of a
// posting next to each other - a posting costs one 4*F-byte load next to its doc id instead of 2F
// four-byte ones from 2F planes (12 instead of 20 bytes for two fields).  Saturated sub-fields (tf >= 255,
// field length >= 2^24 - 1) send the reader to the exact planes; k_pack_tfl builds the words.
constexpr uint32_t TFL_TF_ESC = 255u, TFL_FL_ESC = 0xFFFFFFu;
__device__ __forceinline__ uint32_t tfl_pack(const uint32_t tf, const uint32_t fl) {
  return (min(tf, TFL_TF_ESC) << 24) | min(fl, TFL_FL_ESC);
}
template <int F_>
__device__ __forceinline__ void tfl_load(const KParams& p, const uint64_t pi, uint32_t (&w)[F_ ? F_ : MAX_F]) {
  if (F_ == 1) {
    w[0] = p.tfl[pi];
  } else if (F_ == 2) {
    const uint2 v = reinterpret_cast<const uint2*>(p.tfl)[pi];
    w[0] = v.x; w[1] = v.y;
  } else {
#pragma unroll
    for (int x = 0; x < (F_ ? F_ : MAX_F); ++x)
      if ((uint32_t)x < p.F) w[x] = p.tfl[pi * p.F + x];
  }
}
// Unpacks U postings per lane.  Saturated sub-fields stay saturated: every reader already has a cold arm
// that such a value falls into (tf 255 is off the saturated-tf table and above any exact-numerator limit,
// a field length of 2^24 - 1 is past any table), and fetches the exact value there with tfl_exact.
template <int F_, int U>
__device__ __forceinline__ void tfl_unpack(const KParams& p, const uint32_t (&w)[U][F_ ? F_ : MAX_F],
                                           uint32_t (&tfv)[U][F_ ? F_ : MAX_F], uint32_t (&flv)[U][F_ ? F_ : MAX_F]) {
  constexpr int FA = F_ ? F_ : MAX_F;
  const uint32_t F = F_ ? (uint32_t)F_ : p.F;
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int x = 0; x < FA; ++x) {
      tfv[u][x] = 0; flv[u][x] = 0;
      if ((uint32_t)x < F) {
        tfv[u][x] = w[u][x] >> 24;
        flv[u][x] = w[u][x] & TFL_FL_ESC;
      }
    }
}
__device__ __forceinline__ void tfl_exact(const KParams& p, const uint32_t x, const uint64_t pi, uint32_t& tf, uint32_t& fl) {
  if (tf == TFL_TF_ESC) tf = p.tf[(uint64_t)x * p.P + pi];
  if (fl == TFL_FL_ESC) fl = p.fl[(uint64_t)x * p.P + pi];
}

template <int F_, int U>
__device__ __forceinline__ void load_trip(const KParams& p, const int lane, const uint64_t post_off, const uint32_t i0,
                                          const uint32_t re, uint32_t (&dv)[U], uint32_t (&wv)[U][F_ ? F_ : MAX_F]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t i = i0 + u * WAVE + lane;
    const uint64_t pi = post_off + (i < re ? i : re - 1);  // clamp: always a valid posting
    dv[u] = p.doc[pi];
    tfl_load<F_>(p, pi, wv[u]);
  }
}

// Score U postings per lane and merge them into the wave's LDS tile.  Written branch-free on
// purpose: all LUT gathers of the trip are issued back to back, then all arithmetic, then all
// LDS updates, so the wave never sits on one LDS round trip per posting-field.  `+ 0.0` for a
// field with tf == 0 leaves the f64 sum bit-identical to skipping it.
template <int MODE, int F_, bool TAGS, int U>
__device__ __forceinline__ void score_trip(const KParams& p, const double* lut, double* acc, uint16_t* tag,
                                           const int lane, const uint32_t tile_base, const uint32_t i0,
                                           const uint32_t re, const uint32_t (&dv)[U],
                                           const uint32_t (&wv)[U][F_ ? F_ : MAX_F], const EntryC& ec,
                                           const uint32_t qtl) {
  constexpr int FA = F_ ? F_ : MAX_F;
  const uint32_t F = F_ ? (uint32_t)F_ : p.F;
  if (PS_ABLATE_BUILD && (p.ablate & 2u)) {  // profiling only: loads stay alive, no scoring
#pragma unroll
    for (int u = 0; u < U; ++u)
      if ((dv[u] ^ wv[u][0]) == 0xFFFFFFF1u) acc[0] = 1.0;
    return;
  }
  uint32_t tfv[U][FA], flv[U][FA];
  tfl_unpack<F_, U>(p, wv, tfv, flv);
  auto posting_of = [&](int u) {  // cold arms only: the posting slot u was loaded from (load_trip's clamp)
    const uint32_t i = i0 + u * WAVE + lane;
    return ec.post_off + (i < re ? i : re - 1);
  };
  bool ok[U];
  uint32_t local[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t i = i0 + u * WAVE + lane;
    local[u] = dv[u] - tile_base;
    // coarse table slots (shift != 0) span several tiles: keep only this tile's documents
    ok[u] = i < re && (ec.shift == 0 || local[u] < p.T);
  }
  if (MODE == MODE_BM25) {
    double tfn[U][FA];
    bool slow = false;
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int x = 0; x < FA; ++x) {
        if ((uint32_t)x < F) {
          const uint32_t tfu = tfv[u][x], flu = flv[u][x];
          const bool in_lut = tfu < (uint32_t)LUT_TF && flu < p.lut_cap[x];
          // transposed, odd-stride table: lanes with different field lengths hit different LDS banks
          // (24-bit multiply: full rate, a 32-bit v_mul_lo_u32 is quarter rate; tfu < 16 whenever the index is used)
          tfn[u][x] = lut[in_lut ? __umul24(tfu, p.lut_stride) + p.lut_base[x] + flu : 0u];
          slow |= ok[u] && tfu > 0 && !in_lut;
        }
      }
    }
    if (__any(slow)) {  // wave-uniform; rare once the LUT covers the corpus' field lengths
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int x = 0; x < FA; ++x) {
          if ((uint32_t)x < F) {
            uint32_t tfu = tfv[u][x], flu = flv[u][x];
            if (!(tfu < (uint32_t)LUT_TF && flu < p.lut_cap[x])) {
              tfl_exact(p, (uint32_t)x, posting_of(u), tfu, flu);
              tfn[u][x] = bm25_tfn_cold(p.k1, p.k1p1, p.one_minus_b, p.b, p.avg[x], tfu, flu);
            }
          }
        }
      }
    }
    double s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s[u] = 0.0;
#pragma unroll
      for (int x = 0; x < FA; ++x) {
        if ((uint32_t)x < F) {
          const double term = tfn[u][x] * ec.w0 * p.boost[x] * ec.w1;  // bm25.rs:83-86: ((tfn*idf)*boost)*expansion_boost
          s[u] += (tfv[u][x] > 0) ? term : 0.0;
        }
      }
    }
    if (TAGS) {
      double cur[U];
      uint16_t tg[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        cur[u] = ok[u] ? acc[local[u]] : 0.0;
        tg[u] = ok[u] ? tag[local[u]] : (uint16_t)0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ok[u]) {
          if (s[u] > 0.0)  // Some(score) iff score > 0 (bm25.rs:89-92)
            // max_score_merger (query.rs:150-164); present <=> cur > 0 for BM25
            acc[local[u]] = (cur[u] > 0.0) ? (tg[u] == (uint16_t)ec.tag ? fmax(cur[u], s[u]) : cur[u] + s[u]) : s[u];
          tag[local[u]] = (uint16_t)ec.tag;  // visited even when the score was None (query.rs:87)
        }
      }
    } else {
      // one list per query term: always the `+` / assign arm (absent == +0.0).  A list holds a
      // document once, so the LDS f64 add is uncontended; issuing it as a no-return DS op keeps
      // the read-modify-write latency off the wave's critical path.
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (ok[u] && s[u] > 0.0)
          __hip_atomic_fetch_add(&acc[local[u]], s[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  } else {
    // zero_to_one.rs:117-120: (min(score / tf, 1.) * tf) / max(field_length, all_query_terms_len)
    // The numerator only depends on (score, tf), and for small tf it is the score itself - exactly, in
    // f64: the host found the largest L with fmin(score / t, 1.) * t == score for every t <= L (48 for
    // score 1.0, the exact-match expansion) and left it in the entry (ec.w1's bit pattern).  A trip whose
    // term frequencies are all <= L - nearly every trip - takes one f64 division per (posting, field)
    // instead of two; otherwise the whole wave evaluates the full expression.
    const uint32_t tf_exact = (uint32_t)__double2loint(ec.w1);
    bool wide = false;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int x = 0; x < FA; ++x)
        if ((uint32_t)x < F) wide = wide || (ok[u] && (tfv[u][x] > tf_exact || flv[u][x] == TFL_FL_ESC));  // (tf_exact <= 254)
    const bool full_expr = __builtin_amdgcn_ballot_w64(wide) != 0ull;  // wave-uniform
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int x = 0; x < FA; ++x) {
        if ((uint32_t)x < F) {
          uint32_t tfu = tfv[u][x], flu = flv[u][x];
          double num = ec.w0;
          if (full_expr) {
            tfl_exact(p, (uint32_t)x, posting_of(u), tfu, flu);
            const double df = (double)tfu;
            num = fmin(ec.w0 / df, 1.0) * df;
          }
          const uint32_t den = flu > qtl ? flu : qtl;
          const double c = num / (double)den;
          // ec.tag = occurrence rank of the node (low 16 bits, >= 1: the pool rule) | query-term ordinal
          bool take = ok[u] && tfu >= (ec.tag & 0xFFFFu) && ((ec.fmask >> x) & 1u);
          if (TAGS && (ec.tag >> 31)) {  // bit 31: this query has query terms with several expansions
            // consumed_index (zero_to_one.rs:101-103): the first record of a query term (in sorted
            // order, which is the order entries are processed in) that hits this (doc, field)
            // consumes the term; its later expansions are skipped
            uint32_t* zm = reinterpret_cast<uint32_t*>(tag) + (uint32_t)x * p.T + local[u];
            const uint32_t bit = 1u << ((ec.tag >> 16) & 31u);
            const uint32_t mk = take ? *zm : 0u;
            take = take && !(mk & bit);
            if (take) *zm = mk | bit;
          }
          if (take)
            __hip_atomic_fetch_add(&acc[(uint32_t)x * p.T + local[u]], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
      }
    }
  }
}

// Stream postings [rb, re) of one list through the tile, UNROLL*64 per trip; the next trip's
// loads are in flight while the current one is scored.
template <int MODE, int F_, bool TAGS>
__device__ __forceinline__ void score_stream(const KParams& p, const double* lut, double* acc, uint16_t* tag,
                                             const int lane, const uint32_t tile_base, const uint32_t rb,
                                             const uint32_t re, const EntryC& ec, const uint32_t qtl) {
  constexpr int FA = F_ ? F_ : MAX_F;
  constexpr int UN = F_ ? UNROLL : 1;
  const uint32_t F = F_ ? (uint32_t)F_ : p.F;
  uint32_t i0 = rb;
  if (re - i0 >= (uint32_t)(UN * WAVE)) {
    // full trips, double-buffered
    uint32_t dv[UN], wv[UN][FA];
    uint32_t dn[UN], wnx[UN][FA];
    load_trip<F_, UN>(p, lane, ec.post_off, i0, re, dv, wv);
    while (re - i0 >= (uint32_t)(UN * WAVE)) {
      const uint32_t nx = i0 + UN * WAVE;
      const bool more = re - nx >= (uint32_t)(UN * WAVE);
      if (more) load_trip<F_, UN>(p, lane, ec.post_off, nx, re, dn, wnx);
      score_trip<MODE, F_, TAGS, UN>(p, lut, acc, tag, lane, tile_base, i0, re, dv, wv, ec, qtl);
      if (more) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          dv[u] = dn[u];
#pragma unroll
          for (int x = 0; x < FA; ++x)
            if ((uint32_t)x < F) wv[u][x] = wnx[u][x];
        }
      }
      i0 = nx;
    }
  }
  // tail (< UN*64 postings): one masked trip when it is long (all loads in flight together), one
  // 64-wide trip when it is short (no empty lane slots to pay for)
  if (i0 < re) {
    if (re - i0 > (uint32_t)WAVE) {
      uint32_t dv[UN], wv[UN][FA];
      load_trip<F_, UN>(p, lane, ec.post_off, i0, re, dv, wv);
      score_trip<MODE, F_, TAGS, UN>(p, lut, acc, tag, lane, tile_base, i0, re, dv, wv, ec, qtl);
    } else {
      uint32_t dv[1], wv[1][FA];
      load_trip<F_, 1>(p, lane, ec.post_off, i0, re, dv, wv);
      score_trip<MODE, F_, TAGS, 1>(p, lut, acc, tag, lane, tile_base, i0, re, dv, wv, ec, qtl);
    }
  }
}

template <int MODE, int F_, bool TAGS, bool FULL, int WGW>
__global__ __launch_bounds__(WAVE * WGW) void k_score(const KParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int FA = F_ ? F_ : MAX_F;
  constexpr int G = F_ ? PS_G : 1;    // plan entries whose first trips are in flight together
  constexpr int FU = F_ ? PS_FU : 1;  // postings per lane in a prefetched first trip
  const int lane = threadIdx.x & (WAVE - 1);
  // readfirstlane: tell the compiler the wave index is wave-uniform, so everything derived from
  // it (item, query, plan entries, table ranges) lives in SGPRs and is fetched with scalar loads
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t F = F_ ? (uint32_t)F_ : p.F;
  const uint32_t T = p.T;
  const uint32_t AW = MODE == MODE_Z21S ? F : 1u;  // f64 accumulators per document
  // LDS: [LUT, shared by the workgroup][wave 0: tile, tags, table slices][wave 1: ...]...
  const double* lut = reinterpret_cast<const double*>(smem);
  const uint32_t lut_bytes = MODE == MODE_BM25 ? p.lut_stride * LUT_TF * 8 : 0u;
  // TAGS: BM25 = u16 visited tag per document; Z21S = u32 consumed-query-term mask per (field, document)
  const uint32_t tile_bytes = T * AW * 8 + (TAGS ? (MODE == MODE_Z21S ? T * AW * 4 : T * 2) : 0);
  const uint32_t wave_bytes = tile_bytes + p.slice_bytes;
  unsigned char* wbase = smem + lut_bytes + (size_t)wave * wave_bytes;
  double* acc = reinterpret_cast<double*>(wbase);
  uint16_t* tag = reinterpret_cast<uint16_t*>(wbase + (size_t)T * AW * 8);
  uint32_t* slice = reinterpret_cast<uint32_t*>(wbase + tile_bytes);  // [entry][2][S]: rb, re per tile of the run
  if (MODE == MODE_BM25) {
    double* l = reinterpret_cast<double*>(smem);
    for (uint32_t i = threadIdx.x; i < p.lut_stride * LUT_TF; i += WAVE * WGW) l[i] = p.lut[i];
    __syncthreads();  // the only workgroup-level synchronisation: waves are independent from here on
  }
  // Persistent waves: the grid only fills the chip; every wave keeps pulling (query, run) items
  // from one device-scope counter until none are left.  Items are numbered run-major so waves
  // that are resident together work on the same document range (posting slices stay in L2), and
  // a heavy head-term item never leaves LDS-holding sibling waves idle.
  for (uint32_t i = lane; i < T * AW; i += WAVE) acc[i] = 0.0;
  if (TAGS) {
    if (MODE == MODE_Z21S)
      for (uint32_t i = lane; i < T * AW; i += WAVE) reinterpret_cast<uint32_t*>(tag)[i] = 0u;
    else
      for (uint32_t i = lane; i < T; i += WAVE) tag[i] = 0xFFFFu;
  }
  uint32_t tagbase = 0;
  const uint32_t n_items = p.B * p.n_super;
  // A grid that covers every item with its own wave (a single query: ~1000 waves that would
  // otherwise all queue on one L2 word before doing anything) assigns them by index; otherwise
  // items come from the shared counter.
  const bool by_index = n_items <= gridDim.x * WGW;
  bool first = true;
  for (;;) {
  uint32_t item = 0;
  if (by_index) {
    if (!first) break;
    first = false;
    item = blockIdx.x * WGW + (uint32_t)wave;
  } else {
    if (lane == 0) item = atomicAdd(p.work_counter, 1u);
    item = __builtin_amdgcn_readfirstlane(item);
  }
  if (item >= n_items) break;
  const uint32_t q = p.qorder[item % p.B];
  const uint32_t sup = item / p.B;
  const uint32_t e0 = p.qbeg[q], e1 = p.qbeg[q + 1];
  const uint32_t ne = e1 - e0;
  const bool mine = MODE == MODE_BM25 || (p.qflags[q] & 1u);  // Z21S: only "simple" queries
  if (MODE == MODE_Z21S && !mine) continue;                   // k_z21 owns this query's candidate slots
  const uint32_t qtl = MODE == MODE_Z21S ? p.qterms_len[q] : 0u;
  // zero_to_one, top-k: a document scores the best of its per-field pools, and the pool of field x is at
  // most zfub[q][x] (sum over the query's lists of score / max(shortest field x holding the term, query
  // terms)).  Once the query's threshold - a lower bound of its final K-th best score, published by the
  // runs that finished - exceeds that, field x cannot decide any top-K score: it is not accumulated,
  // not loaded and not harvested for this item (exact: such pools lose the max against any score that
  // can still be returned).  With every field out the item is skipped whole.
  uint32_t fmask = 0xFFFFFFFFu;
  if (MODE == MODE_Z21S && !FULL && p.zfub != nullptr) {
    const double th = __longlong_as_double((long long)__hip_atomic_load(&p.gthr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const double thu = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(th)), __builtin_amdgcn_readfirstlane(__double2loint(th)));
    for (uint32_t x = 0; x < F; ++x)
      if (thu > p.zfub[(uint64_t)q * F + x]) fmask &= ~(1u << x);
  }
  const bool item_dead = MODE == MODE_Z21S && (fmask & ((1u << F) - 1u)) == 0u;
  const bool q_assign = !TAGS && ne != 0 && (p.plan[e0].shift & DENSE_ASSIGN_FLAG);

  TopK tk;
  tk.s = -1.0; tk.d = 0xFFFFFFFFu; tk.n = 0; tk.thr_s = 0.0; tk.thr_d = 0;
  uint32_t ws_post = 0, ws_rows = 0;  // work counters (wave-uniform): postings streamed, dense-row tile slices read

  if (ne != 0 && !item_dead) {
    const uint32_t t_begin = sup * p.S;
    const uint32_t t_end = min(p.n_tiles, t_begin + p.S);
    // Table slices: the [rb, re) range of every (entry, tile of this run), fetched once with
    // coalesced vector loads into LDS, so the per-tile lookup is an LDS broadcast read instead of
    // a dependent scalar-memory round trip per (entry, tile).
    const bool sliced = p.slice_bytes != 0;
    if (sliced) {
      for (uint32_t e = 0; e < ne; ++e) {
        const uint32_t tbl_off = p.plan[e0 + e].tbl_off;
        const uint32_t shift = p.plan[e0 + e].shift & 0xFFu;
        if ((uint32_t)lane < p.S) {
          const uint32_t slot = min(t_begin + (uint32_t)lane, p.n_tiles - 1) >> shift;
          slice[(e * 2 + 0) * p.S + lane] = p.table[tbl_off + slot];
          slice[(e * 2 + 1) * p.S + lane] = p.table[tbl_off + slot + 1];
        }
      }
    }

    EntryC ec[G];
    uint32_t ec_qterm[G], ec_tbl[G], ec_row[G], ec_flags[G];
    uint32_t fuse_row = 0xFFFFFFFFu;  // dense row of the query's last entry, added during the harvest
    uint32_t rb[G], re[G];
    uint32_t dv[G][FU], wv[G][FU][FA];
    // phase 1 of a visit (tile VT, entries EG..EG+G): ranges + first trips, all loads in flight together
#define PS_PHASE1(VT, EG, FIRST)                                                                                \
  _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                               \
    rb[g] = 0; re[g] = 0;                                                                                       \
    if ((EG) + g < ne) { /* wave-uniform */                                                                     \
      if (ne > (uint32_t)G || (FIRST)) { /* a plan of <= G entries stays in SGPRs for the whole run */          \
        const ps_plan_entry& en = p.plan[e0 + (EG) + g];                                                        \
        ec[g].post_off = en.post_off;                                                                           \
        ec[g].shift = en.shift & 0xFFu;                                                                         \
        ec[g].w0 = MODE == MODE_BM25 ? en.idf : en.boost;                                                       \
        ec[g].w1 = MODE == MODE_BM25 ? en.boost : en.idf; /* Z21S: bits = exact-numerator tf limit */           \
        ec[g].fmask = fmask;                                                                                    \
        ec_qterm[g] = MODE == MODE_Z21S ? en.qterm_index : en.qterm;                                            \
        ec_tbl[g] = en.tbl_off;                                                                                 \
        ec_row[g] = (en.shift & DENSE_FLAG) ? en.node : 0xFFFFFFFFu;                                            \
        ec_flags[g] = en.shift;                                                                                 \
      }                                                                                                         \
      if (ec_row[g] != 0xFFFFFFFFu) { /* dense row: nothing to fetch up front */                                \
      } else if (sliced) {                                                                                             \
        rb[g] = __builtin_amdgcn_readfirstlane(slice[(((EG) + g) * 2 + 0) * p.S + ((VT) - t_begin)]);           \
        re[g] = __builtin_amdgcn_readfirstlane(slice[(((EG) + g) * 2 + 1) * p.S + ((VT) - t_begin)]);           \
      } else {                                                                                                  \
        const uint32_t slot = (VT) >> ec[g].shift;                                                              \
        rb[g] = p.table[ec_tbl[g] + slot];                                                                      \
        re[g] = p.table[ec_tbl[g] + slot + 1];                                                                  \
      }                                                                                                         \
      if (rb[g] < re[g]) load_trip<F_, FU>(p, lane, ec[g].post_off, rb[g], re[g], dv[g], wv[g]);                 \
    }                                                                                                           \
  }                                                                                                             \
  if (!FULL) gt_req = __hip_atomic_load(&p.gthr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The query's threshold for a tile's harvest is requested with that tile's postings (the last load of
    // the visit) and moved to SGPRs before the next visit's loads are issued: a vector-memory load issued
    // after them would only return behind them (loads return in order), and the harvest - which is meant
    // to run while they fly - would start by waiting for all of them.  A slightly stale threshold is still a
    // lower bound of the final K-th best.
    unsigned long long gt_req = 0ull;
    PS_PHASE1(t_begin, 0u, true)
    uint32_t t = t_begin, eg = 0;
    bool dirty = false;
    for (;;) {
      const uint32_t tile_base = t * T;
      // phase 2: consume the visit in plan order
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (eg + g < ne && ec_row[g] != 0xFFFFFFFFu) {
          dirty = true;
          if (PS_WORK_COUNTERS) ws_rows += MODE == MODE_Z21S ? (uint32_t)__popc(fmask & ((1u << F) - 1u)) : 1u;
          if (PS_ABLATE_BUILD && (p.ablate & 8u)) {
          } else if (MODE == MODE_BM25 && !TAGS && (ec_flags[g] & DENSE_FUSE_FLAG)) fuse_row = ec_row[g];
          else if (MODE == MODE_BM25 && !TAGS && (ec_flags[g] & DENSE_ASSIGN_FLAG)) dense_apply<false, true>(p, acc, tag, lane, ec_row[g], tile_base, 0);
          else if (MODE == MODE_BM25) dense_apply<TAGS>(p, acc, tag, lane, ec_row[g], tile_base, (uint16_t)(tagbase + ec_qterm[g]));
          else if (!TAGS && F_ != 0 && (ec_flags[g] & DENSE_FUSE_FLAG)) fuse_row = ec_row[g];
          else if (!TAGS && F_ != 0 && (ec_flags[g] & DENSE_ASSIGN_FLAG))
            dense_apply_z<false, true>(p, acc, reinterpret_cast<uint32_t*>(tag), lane, ec_row[g], tile_base, 0u, fmask);
          else dense_apply_z<TAGS>(p, acc, reinterpret_cast<uint32_t*>(tag), lane, ec_row[g], tile_base,
                                   (ec_qterm[g] >> 31) ? (1u << ((ec_qterm[g] >> 16) & 31u)) : 0u, fmask);
        } else if (rb[g] < re[g]) {
          dirty = true;
          if (PS_WORK_COUNTERS) ws_post += re[g] - rb[g];
          ec[g].tag = MODE == MODE_Z21S ? ec_qterm[g] : tagbase + ec_qterm[g];
          score_trip<MODE, F_, TAGS, FU>(p, lut, acc, tag, lane, tile_base, rb[g], re[g], dv[g], wv[g], ec[g], qtl);
          if (rb[g] + FU * WAVE < re[g])
            score_stream<MODE, F_, TAGS>(p, lut, acc, tag, lane, tile_base, rb[g] + FU * WAVE, re[g], ec[g], qtl);
        }
      }
      // Request the next visit's ranges and first trips now: the registers are free again, and the
      // loads then fly while this tile is harvested below.
      uint32_t neg = eg + G, nt = t;
      bool last = false;
      if (neg >= ne) { neg = 0; nt = t + 1; last = true; }
      const bool more = nt < t_end;
      const double gt_tile = __hiloint2double(__builtin_amdgcn_readfirstlane((int)(gt_req >> 32)),
                                              __builtin_amdgcn_readfirstlane((int)(uint32_t)gt_req));
      if (more) { PS_PHASE1(nt, neg, false) }
      const bool harvest = last && dirty && !(PS_ABLATE_BUILD && (p.ablate & 4u));
      if (last) dirty = false;
      t = nt; eg = neg;
      if (harvest) {
      // tile epilogue: harvest + reset (two f64 per lane per LDS access where the layout allows)
      const double gt = FULL ? 0.0 : gt_tile;
      const bool zero_tile = TAGS || !q_assign || !more;
      if (MODE == MODE_BM25) {
        // several 16-byte LDS reads in flight per lane: chunks of PS_HARVEST_UNROLL x 128 documents,
        // then (tiles of 256 documents) chunks of 2 x 128
        // a query whose first entry is a WRITTEN dense row overwrites the whole tile at the start of
        // its next visit: only the item's last visit has to leave zeros behind
        auto harvest = [&](auto hu_tag, auto fused_tag, const uint32_t c) {
          constexpr int HU = decltype(hu_tag)::value;
          constexpr bool FUSED = decltype(fused_tag)::value;
          double2 vv[HU], rv[FUSED ? HU : 1];
          if (FUSED) {
            const double* r = p.rows + (uint64_t)fuse_row * p.row_stride + tile_base;
#pragma unroll
            for (int u = 0; u < HU; ++u) rv[u] = *reinterpret_cast<const double2*>(r + c + u * 2 * WAVE + 2 * lane);
          }
#pragma unroll
          for (int u = 0; u < HU; ++u) vv[u] = *reinterpret_cast<double2*>(&acc[c + u * 2 * WAVE + 2 * lane]);
          bool fh[FULL ? 2 * HU : 1];
          double fv[FULL ? 2 * HU : 1];
          uint32_t fd[FULL ? 2 * HU : 1];
#pragma unroll
          for (int u = 0; u < HU; ++u) {
            double2 v = vv[u];
            if (zero_tile && (v.x > 0.0 || v.y > 0.0))
              *reinterpret_cast<double2*>(&acc[c + u * 2 * WAVE + 2 * lane]) = make_double2(0.0, 0.0);
            if (FUSED) { v.x += rv[u].x; v.y += rv[u].y; }  // the query's last entry, in plan order
            const uint32_t d = tile_base + c + u * 2 * WAVE + 2 * lane;
            bool h0 = v.x > 0.0, h1 = v.y > 0.0;
            if (p.alive != nullptr) {  // delta removals (wave-uniform branch; d is even: both bits sit in one word)
              const uint32_t aw = p.alive[d >> 5] >> (d & 31u);
              h0 = h0 && (aw & 1u);
              h1 = h1 && (aw & 2u);
            }
            if (FULL) {
              fh[FULL ? 2 * u : 0] = h0; fv[FULL ? 2 * u : 0] = v.x; fd[FULL ? 2 * u : 0] = d;
              fh[FULL ? 2 * u + 1 : 0] = h1; fv[FULL ? 2 * u + 1 : 0] = v.y; fd[FULL ? 2 * u + 1 : 0] = d + 1;
            } else if (!(PS_ABLATE_BUILD && (p.ablate & 1u))) {
              // one wave-wide test against the best known lower bound skips the insert logic for
              // the (usual) chunks that cannot contribute
              const double lo = (tk.n == p.K && tk.thr_s > gt) ? tk.thr_s : gt;
              if (__any(fmax(v.x, v.y) >= lo && (h0 || h1))) {
                topk_offer(tk, p.K, lane, h0, v.x, d, gt);
                topk_offer(tk, p.K, lane, h1, v.y, d + 1, gt);
              }
            }
          }
          if (FULL) full_emit_group(p, q, lane, fh, fv, fd);
        };
        uint32_t c = 0;
        if (!TAGS && fuse_row != 0xFFFFFFFFu) {
          for (; c + 2 * WAVE * PS_FUSED_UNROLL <= T; c += 2 * WAVE * PS_FUSED_UNROLL)
            harvest(std::integral_constant<int, PS_FUSED_UNROLL>{}, std::true_type{}, c);
          for (; c < T; c += 2 * WAVE * 2) harvest(std::integral_constant<int, 2>{}, std::true_type{}, c);
          fuse_row = 0xFFFFFFFFu;
        } else {
          for (; c + 2 * WAVE * PS_HARVEST_UNROLL <= T; c += 2 * WAVE * PS_HARVEST_UNROLL)
            harvest(std::integral_constant<int, PS_HARVEST_UNROLL>{}, std::false_type{}, c);
          for (; c + 2 * WAVE * 4 <= T; c += 2 * WAVE * 4) harvest(std::integral_constant<int, 4>{}, std::false_type{}, c);
          for (; c < T; c += 2 * WAVE * 2) harvest(std::integral_constant<int, 2>{}, std::false_type{}, c);
        }
      } else {
        // accumulators are planar ([field][T]); two documents per lane per 16-byte LDS access, the
        // reads of all fields of ZU chunks in flight together
        // FM: compile-time copy of the item's field mask (F_ == 2: one loop body per mask, so a pruned field
        // costs no LDS read, no row fetch and no compare), or all ones = test the run-time mask per field
        auto harvest_z = [&](auto fm_tag, auto zu_tag) {
          constexpr uint32_t FM = decltype(fm_tag)::value;
          constexpr int ZU = decltype(zu_tag)::value;  // chunks of 128 documents whose LDS reads are in flight together
          auto live = [&](const int x) { return FM != 0xFFFFFFFFu ? ((FM >> x) & 1u) != 0u : ((fmask >> x) & 1u) != 0u; };
          for (uint32_t c = 0; c < T; c += 2 * WAVE * ZU) {
            double2 vv[ZU][FA], rv[ZU][F_ ? FA : 1];
            const bool fused = !TAGS && F_ != 0 && fuse_row != 0xFFFFFFFFu;  // the query's last entry is a dense row
            if (fused) {
#pragma unroll
              for (int u = 0; u < ZU; ++u)
#pragma unroll
                for (int x = 0; x < (F_ ? FA : 1); ++x) {
                  // (run-time mask: a field that is out re-reads plane 0 of the row - same lines, no branch in
                  // the load burst - and its value is dropped below)
                  const uint32_t xs = live(x) ? (uint32_t)x : 0u;
                  if (FM == 0xFFFFFFFFu || ((FM >> x) & 1u))
                    rv[u][x] = *reinterpret_cast<const double2*>(p.rows + ((uint64_t)fuse_row * F + xs) * p.row_stride + tile_base +
                                                                 c + u * 2 * WAVE + 2 * lane);
                }
            }
#pragma unroll
            for (int u = 0; u < ZU; ++u)
#pragma unroll
              for (int x = 0; x < FA; ++x)
                if (F_ && (uint32_t)x < F && (FM == 0xFFFFFFFFu || ((FM >> x) & 1u)))  // (a field that is out is never written: its plane reads zero)
                  vv[u][x] = *reinterpret_cast<double2*>(&acc[(uint32_t)x * T + c + u * 2 * WAVE + 2 * lane]);
            bool fh[FULL ? 2 * ZU : 1];
            double fv[FULL ? 2 * ZU : 1];
            uint32_t fd[FULL ? 2 * ZU : 1];
#pragma unroll
            for (int u = 0; u < ZU; ++u) {
              // result.score = max(score_by_pool, result.score) over fields, from the dummy 0. (zero_to_one.rs:81,122)
              double b0 = 0.0, b1 = 0.0;
              bool h0 = false, h1 = false;
#pragma unroll
              for (int x = 0; x < FA; ++x) {
                if ((uint32_t)x < F && (FM == 0xFFFFFFFFu || ((FM >> x) & 1u))) {
                  const uint32_t at = (uint32_t)x * T + c + u * 2 * WAVE + 2 * lane;
                  // (any number of fields: one plane at a time, 8 preloaded planes would cost 32 VGPRs)
                  double2 v = F_ ? vv[u][x] : *reinterpret_cast<double2*>(&acc[at]);
                  if (zero_tile && (v.x > 0.0 || v.y > 0.0)) {
                    *reinterpret_cast<double2*>(&acc[at]) = make_double2(0.0, 0.0);
                    if (TAGS) *reinterpret_cast<uint2*>(reinterpret_cast<uint32_t*>(tag) + at) = make_uint2(0u, 0u);
                  }
                  if (F_ != 0 && fused && live(x)) { v.x += rv[u][F_ ? x : 0].x; v.y += rv[u][F_ ? x : 0].y; }  // last record, in sorted order
                  h0 |= v.x > 0.0; h1 |= v.y > 0.0;
                  b0 = fmax(v.x, b0); b1 = fmax(v.y, b1);
                }
              }
              const uint32_t d = tile_base + c + u * 2 * WAVE + 2 * lane;
              if (p.alive != nullptr) {  // delta removals
                const uint32_t aw = p.alive[d >> 5] >> (d & 31u);
                h0 = h0 && (aw & 1u);
                h1 = h1 && (aw & 2u);
              }
              if (FULL) {
                fh[FULL ? 2 * u : 0] = h0; fv[FULL ? 2 * u : 0] = b0; fd[FULL ? 2 * u : 0] = d;
                fh[FULL ? 2 * u + 1 : 0] = h1; fv[FULL ? 2 * u + 1 : 0] = b1; fd[FULL ? 2 * u + 1 : 0] = d + 1;
              } else {
                const double lo = (tk.n == p.K && tk.thr_s > gt) ? tk.thr_s : gt;
                if (__any((h0 && b0 >= lo) || (h1 && b1 >= lo))) {
                  topk_offer(tk, p.K, lane, h0, b0, d, gt);
                  topk_offer(tk, p.K, lane, h1, b1, d + 1, gt);
                }
              }
            }
            if (FULL) full_emit_group(p, q, lane, fh, fv, fd);
          }
        };
        const uint32_t fm2 = fmask & 3u;  // wave-uniform
        constexpr int ZU1 = PS_Z21_HARVEST_UNROLL_1F;  // one live field: half the registers per chunk
        const bool wide_ok = (T % (2 * WAVE * ZU1)) == 0u;
        if (F_ == 2 && !FULL && !TAGS && fm2 == 1u && wide_ok) harvest_z(std::integral_constant<uint32_t, 1u>{}, std::integral_constant<int, ZU1>{});
        else if (F_ == 2 && !FULL && !TAGS && fm2 == 2u && wide_ok) harvest_z(std::integral_constant<uint32_t, 2u>{}, std::integral_constant<int, ZU1>{});
        else if (FULL && F_ != 0 && (T % (2 * WAVE * 4)) == 0u) harvest_z(std::integral_constant<uint32_t, 0xFFFFFFFFu>{}, std::integral_constant<int, 4>{});
        else harvest_z(std::integral_constant<uint32_t, 0xFFFFFFFFu>{}, std::integral_constant<int, (F_ ? 2 : 1)>{});
      }
      fuse_row = 0xFFFFFFFFu;
      if (!FULL && tk.n == p.K && tk.thr_s > gt) {
        // publish this run's K-th best: the final K-th best of the query can only be higher
        if (lane == 0) atomicMax(&p.gthr[q], (unsigned long long)__double_as_longlong(tk.thr_s));
      }
      if (TAGS && MODE == MODE_BM25) {
        tagbase += p.max_qterms;
        if (tagbase + p.max_qterms >= 0xFFFFu) {
          for (uint32_t i = lane; i < T; i += WAVE) tag[i] = 0xFFFFu;
          tagbase = 0;
        }
      }
      }  // harvest
      if (!more) break;
    }
#undef PS_PHASE1
  }
  if (!FULL && (uint32_t)lane < p.K) {
    const uint64_t o = ((uint64_t)sup * p.B + q) * p.K + lane;  // the slot K2 / K3 expect: (run, query)
    const bool ok = (uint32_t)lane < tk.n;
    p.cand_score[o] = ok ? tk.s : 0.0;
    p.cand_doc[o] = ok ? tk.d : 0xFFFFFFFFu;
  }
  if (PS_WORK_COUNTERS && lane == 0) {
    unsigned long long* w = p.wstats + (size_t)(blockIdx.x & (WS_SLOTS - 1u)) * WS_WORDS;
    atomicAdd(&w[WS_K1_ITEMS], 1ull);
    if (ws_post) atomicAdd(&w[WS_K1_POSTINGS], (unsigned long long)ws_post);
    if (ws_rows) atomicAdd(&w[WS_K1_ROWSLICES], (unsigned long long)ws_rows);
  }
  }  // item loop
}

// ------------------------------------------------------------------------------------------
// K2: zero_to_one   (zero_to_one.rs:44-126)
//
// LDS per wave: rec[z_tile][z_nodes][F] u32 = term frequency of distinct node n in field x for
// the tile's documents (0 = no hit).  ScoreByTerm's other members are per-entry constants in the
// plan (score, query_term_index, node) or per-query (all_query_terms_len); field_length comes
// with the posting and is kept in fls[z_tile][F].  Deduplicated postings are equivalent to the
// reference's per-occurrence records (identical adjacent records: the first is either consumed,
// after which the rest are skipped via consumed_index, or skipped for a reason that skips the
// rest as well).  finalize per (doc, field): walk the query's entries in
// zorder = stable sort by score desc (zero_to_one.rs:98), greedy-consume one record per query
// term with the per-node pool (:101-120); doc score = max over fields (:122).
// ------------------------------------------------------------------------------------------
template <bool FULL>
__global__ __launch_bounds__(WAVE) void k_z21(const KParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t F = p.F, ZN = p.z_nodes, ZT = p.z_tile, QW = p.z_qwords;
  const uint32_t stride = ZN * F;
  // rec word = term frequency (low 16 bits) | records of this node consumed so far in the field being
  // finalised (high 16 bits: the per-node pool of zero_to_one.rs:104-113, any number of entries)
  uint32_t* rec = reinterpret_cast<uint32_t*>(smem);  // [ZT][ZN][F]
  uint32_t* fls = rec + (size_t)ZT * stride;           // [ZT][F]  field lengths
  uint32_t* cq = fls + (size_t)ZT * F;                 // [ZT][QW] consumed_index: one bit per query term with entries
  const int lane = threadIdx.x;
  // grid = n_general x n_super: only the queries the simple path could not take
  const uint32_t q = p.gen_queries[blockIdx.x % p.n_general];
  const uint32_t sup = blockIdx.x / p.n_general;
  const uint32_t item = sup * p.B + q;  // candidate slot, as k_merge expects it
  const uint32_t e0 = p.qbeg[q], e1 = p.qbeg[q + 1];

  TopK tk;
  tk.s = -1.0; tk.d = 0xFFFFFFFFu; tk.n = 0; tk.thr_s = 0.0; tk.thr_d = 0;
  uint32_t ws_post = 0;  // work counter: postings streamed (exact planes: 4 + 8F bytes each here)

  if (e0 != e1) {
    for (uint32_t i = lane; i < ZT * stride; i += WAVE) rec[i] = 0;
    const uint32_t qtl = p.qterms_len[q];
    const uint32_t sub_per_tile = p.T / ZT;  // ZT is a power of two <= T
    const uint32_t t_begin = sup * p.S;
    const uint32_t t_end = min(p.n_tiles, t_begin + p.S);
    for (uint32_t t = t_begin; t < t_end; ++t) {
      for (uint32_t sub = 0; sub < sub_per_tile; ++sub) {
        const uint32_t tile_base = t * p.T + sub * ZT;
        if (tile_base >= p.n_docs) break;
        for (uint32_t e = e0; e < e1; ++e) {
          const uint64_t post_off = p.plan[e].post_off;
          const uint32_t tbl_off = p.plan[e].tbl_off;
          const uint32_t shift = p.plan[e].shift & 0xFFu;
          const uint32_t layer = p.plan[e].shift >> 8;
          const uint32_t node = p.plan[e].node;
          const uint32_t slot = t >> shift;
          const uint32_t rb = p.table[tbl_off + slot];
          const uint32_t re = p.table[tbl_off + slot + 1];
          if (PS_WORK_COUNTERS) ws_post += __builtin_amdgcn_readfirstlane(re - rb);
          for (uint32_t i = rb + lane; i < re; i += WAVE) {
            const uint64_t pi = post_off + i;
            const uint32_t local = p.doc[pi] - tile_base;
            if (local >= ZT) continue;  // table slot wider than this sub-tile
            for (uint32_t x = 0; x < F; ++x) {
              const uint32_t tfu = p.tf[(uint64_t)x * p.P + pi];
              uint32_t* r = &rec[local * stride + node * F + x];
              // layer 0 = newest version of a re-added key; older versions only fill fields
              // the newer ones left empty (the first record per (entry, doc, field) decides)
              if (tfu > 0 && (layer == 0 || *r == 0)) *r = tfu;
              fls[local * F + x] = p.fl[(uint64_t)x * p.P + pi];
            }
          }
        }
        // finalize (zero_to_one.rs:84-126): one lane per document of the sub-tile
        for (uint32_t c = 0; c < ZT; c += WAVE) {
          const uint32_t local = c + lane;
          const bool mine = local < ZT;  // sub-tiles narrower than a wave leave lanes idle
          bool has = false;
          double best = 0.0;  // the merged dummy Some(0.) (zero_to_one.rs:81,122)
          for (uint32_t x = 0; x < F && mine; ++x) {
            for (uint32_t w = 0; w < QW; ++w) cq[local * QW + w] = 0u;
            double pool = 0.0;  // score_by_pool
            bool any = false;
            for (uint32_t z = e0; z < e1; ++z) {
              const uint32_t e = p.zorder[z];
              const uint32_t node = p.plan[e].node;
              uint32_t* r = &rec[local * stride + node * F + x];
              const uint32_t word = *r;
              const uint32_t tfu = word & 0xFFFFu;
              if (tfu == 0) continue;  // no record for this (entry, doc, field)
              any = true;
              const uint32_t qt = p.plan[e].qterm;  // dense ordinal among the query's terms that have entries
              uint32_t* cw = &cq[local * QW + (qt >> 5)];
              if ((*cw >> (qt & 31u)) & 1u) continue;  // :101-103
              // df_pool_by_id (:104-113): a node may be consumed term_frequency times in total
              if ((word >> 16) >= tfu) continue;
              *r = word + 0x10000u;
              *cw |= 1u << (qt & 31u);
              const double sc = p.plan[e].boost;
              const double df = (double)tfu;
              const uint32_t fl = fls[local * F + x];
              const uint32_t den = fl > qtl ? fl : qtl;  // usize::max(field_length, all_query_terms_len)
              pool += fmin(sc / df, 1.0) * df / (double)den;  // :117-120
            }
            if (any) { has = true; best = fmax(pool, best); }  // :122
          }
          if (has)
            for (uint32_t w = 0; w < stride; ++w) rec[local * stride + w] = 0;
          const uint32_t d = tile_base + local;
          has = has && (d < p.n_docs) && doc_alive(p, d);
          if (FULL) full_emit(p, q, lane, has, best, d);
          else topk_offer(tk, p.K, lane, has, best, d);
        }
      }
    }
  }
  if (!FULL && (uint32_t)lane < p.K) {
    const uint64_t o = (uint64_t)item * p.K + lane;
    const bool ok = (uint32_t)lane < tk.n;
    p.cand_score[o] = ok ? tk.s : 0.0;
    p.cand_doc[o] = ok ? tk.d : 0xFFFFFFFFu;
  }
  if (PS_WORK_COUNTERS && lane == 0) {
    unsigned long long* w = p.wstats + (size_t)(blockIdx.x & (WS_SLOTS - 1u)) * WS_WORDS;
    atomicAdd(&w[WS_K1_ITEMS], 1ull);
    if (ws_post) atomicAdd(&w[WS_K1_POSTINGS], (unsigned long long)ws_post);
  }
}

// ------------------------------------------------------------------------------------------
// K3: merge per-run top-K lists -> final top-K per query, doc id -> key   (query.rs:97-105)
// ------------------------------------------------------------------------------------------
// One workgroup of MERGE_WAVES waves per query.  The n_super*K candidates are split over the
// waves; each keeps several 64-candidate loads in flight and drops everything strictly below the
// query's published threshold (a lower bound of its final K-th best) before the insert logic.  The
// waves' lists meet in LDS and wave 0 folds them.  Last, the query's control words are zeroed
// again, so the next batch needs no memset.
__global__ __launch_bounds__(WAVE * MERGE_WAVES) void k_merge(const KParams p) {
  __shared__ double sh_s[MERGE_WAVES][WAVE];
  __shared__ uint32_t sh_d[MERGE_WAVES][WAVE];
  __shared__ uint32_t sh_n[MERGE_WAVES];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t q = blockIdx.x;
  TopK tk;
  tk.s = -1.0; tk.d = 0xFFFFFFFFu; tk.n = 0; tk.thr_s = 0.0; tk.thr_d = 0;
  const uint32_t K = p.K;
  const double gt = __longlong_as_double((long long)p.gthr[q]);
  // candidate c of the query = (run c / K, rank c % K); run `sup` lives at item = sup * B + q
  const uint32_t n_c = p.n_super * K;
  const uint32_t n_waves = blockDim.x >> 6;
  constexpr int U = 4;
  for (uint32_t c0 = (uint32_t)wave * WAVE * U; c0 < n_c; c0 += n_waves * WAVE * U) {
    double v[U];
    uint32_t d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t c = c0 + u * WAVE + lane;
      v[u] = 0.0; d[u] = 0xFFFFFFFFu;
      if (c < n_c) {
        const uint64_t o = ((uint64_t)(c / K) * p.B + q) * K + c % K;
        d[u] = p.cand_doc[o];
        v[u] = p.cand_score[o];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool has = d[u] != 0xFFFFFFFFu;
      if (__any(has && v[u] >= gt)) topk_offer(tk, K, lane, has, v[u], d[u], gt);
    }
  }
  sh_s[wave][lane] = tk.s;
  sh_d[wave][lane] = tk.d;
  if (lane == 0) sh_n[wave] = tk.n;
  __syncthreads();
  if (wave != 0) return;
  for (uint32_t w = 1; w < n_waves; ++w) {
    const bool has = (uint32_t)lane < sh_n[w];
    topk_offer(tk, K, lane, has, sh_s[w][lane], sh_d[w][lane]);
  }
  const uint32_t row = p.out_row != nullptr ? p.out_row[q] : q;
  if ((uint32_t)lane < K) {
    const bool ok = (uint32_t)lane < tk.n;
    const uint64_t o = (uint64_t)row * K + lane;
    p.out_keys[o] = ok ? p.keys[tk.d] : ~0ull;
    p.out_scores[o] = ok ? tk.s : 0.0;
  }
  if (lane == 0) {
    p.out_counts[row] = tk.n;
    p.gthr[q] = 0ull;
    if (p.gtie != nullptr)
      for (uint32_t l = 0; l < 3u; ++l) p.gtie[(size_t)l * p.z_tstride + q] = 0ull;
    if (q == 0) *p.work_counter = 0u;
  }
}

// ------------------------------------------------------------------------------------------
// K1d: k_daat — exact top-K with dynamic pruning (BM25, positive boosts).   [same results as
//      query.rs:61-105 + bm25.rs:60-93 restricted to the first K of the canonical order]
//
// The reference scores every posting of every list.  For a top-K answer most of that work cannot
// matter: with U(e) an upper bound of any posting score of list e (host: exact per-list maxima of
// the saturated term frequency, pushed through THE SAME f64 expression, so it bounds the computed
// value, rounding included) and theta a lower bound of the query's final K-th best score,
//   * a document that only occurs in lists whose bounds sum to less than theta cannot enter the
//     top-K (strictly below the K-th best, so ties are unaffected): with the lists sorted by U,
//     the longest such prefix is "non-essential" and is never traversed (MaxScore);
//   * every other document occurs in at least one essential list: it is evaluated exactly once,
//     from the posting of its highest-bound list (the "rank" order), by looking its other
//     contributions up (dense row read, or binary search in the list's tile slice) and folding
//     them IN PLAN ORDER through the same add / max state machine as k_score - same operands, same
//     order, same bits;
//   * a posting whose own score plus everything the other lists could add is below theta is
//     dropped before any lookup.
// theta is the running K-th best of any wave of the query, shared through the same device-scope
// word k_score uses; items are handed out highest-bound lists first, so by the time the long
// low-idf lists come up most of them are skipped whole.  No LDS tiles, no harvest over N documents.
// ------------------------------------------------------------------------------------------
// ---- Bloom filters of the sparse lists --------------------------------------------------------------
constexpr unsigned long long NO_BLOOM = ~0ull;
constexpr uint32_t BLOOM_BITS_PER_KEY = 16;
#ifndef PS_BLOOM_DOC_ORDER
// 1: the filter word of a document is chosen by its DOC ID (d >> shift: the filter is laid out in document order, like the list
// itself), only the three bits inside the word by a hash.  The documents a wave asks about are consecutive postings of its own
// doc-sorted list, i.e. a narrow range of the document space: their filter words then share a handful of 128-byte lines
// instead of 64 lines scattered over the whole filter (C2: 3.65 M filter words per launch = a quarter of all line requests of
// k_daat_small, profiles/r06_request_lines.txt).  A list whose documents cluster in id space loads some words more than others
// - more "maybe" answers there, never a wrong one.  0: round 5's hashed word.
#define PS_BLOOM_DOC_ORDER 1
#endif
// filter descriptor: bits 0-39 first word, 40-45 shift (doc-ordered layout), 58-63 log2(words)
__device__ __host__ __forceinline__ void bloom_probe(const uint32_t d, const unsigned long long desc, uint64_t& word, unsigned long long& mask) {
  const unsigned long long h = (unsigned long long)d * 0x9E3779B97F4A7C15ull;
#if PS_BLOOM_DOC_ORDER
  word = (desc & ((1ull << 40) - 1ull)) + (uint64_t)(d >> (uint32_t)((desc >> 40) & 63u));
  mask = (1ull << (h >> 58)) | (1ull << ((h >> 52) & 63u)) | (1ull << ((h >> 46) & 63u));  // (the product's high bits are the mixed ones)
#else
  const uint32_t lg = (uint32_t)(desc >> 58);
  word = (desc & ((1ull << 40) - 1ull)) + ((h >> 36) & ((1ull << lg) - 1ull));
  mask = (1ull << (h & 63u)) | (1ull << ((h >> 6) & 63u)) | (1ull << ((h >> 12) & 63u));
#endif
}
// one wave per sparse list: every posting sets its three bits (blocked filter: all three in one 64-bit word)
__global__ __launch_bounds__(256) void k_build_bloom(const uint32_t* doc, const uint4* layer_a, const unsigned long long* layer_bloom,
                                                     const uint32_t n_layers, unsigned long long* bloom) {
  const uint32_t l = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const uint32_t lane = threadIdx.x % WAVE;
  if (l >= n_layers) return;
  const unsigned long long desc = layer_bloom[l];
  if (desc == NO_BLOOM) return;
  const uint4 la = layer_a[l];
  const uint64_t off = (uint64_t)la.x | ((uint64_t)la.y << 32);
  for (uint32_t i = lane; i < la.z; i += WAVE) {
    uint64_t w;
    unsigned long long m;
    bloom_probe(doc[off + i], desc, w, m);
    atomicOr(&bloom[w], m);
  }
}

// ---- score planes --------------------------------------------------------------------------------
// (tfn * idf) of a (posting, field) depends on the list (idf) and the scorer parameters, not on the query and - since
// round 5 - not on fields_boost either (src/query.rs:26: a per-call argument): k_list_bounds evaluates it ONCE per
// posting - the first multiplication of the f64 expression, left to right (bm25.rs:78-86) - into a plane next to the
// postings, and K1d's per-visit work shrinks to `sum_x (plane_x * boost_x) * expansion_boost` (the remaining
// multiplications and the additions of the same expression, in the same order: bit-identical).  A new boost vector
// therefore rewrites nothing and waits for nobody.  item traces showed k_daat bound by VALU issue - ~800 wave instructions
// per 256 postings, most of them unpacking words and gathering the saturated-tf table - not by latency.
template <int F_>
__device__ __forceinline__ void plane_load(const KParams& p, const uint64_t pi, double (&t)[F_ ? F_ : MAX_F]) {
  if (F_ == 1) {
    t[0] = p.splane[pi];
  } else if (F_ == 2) {
    const double2 v = reinterpret_cast<const double2*>(p.splane)[pi];
    t[0] = v.x; t[1] = v.y;
  } else {
#pragma unroll
    for (int x = 0; x < (F_ ? F_ : MAX_F); ++x)
      if ((uint32_t)x < p.F) t[x] = p.splane[pi * p.F + x];
  }
}
template <int F_, int U>
__device__ __forceinline__ void scores_from_plane(const KParams& p, const double (&t)[U][F_ ? F_ : MAX_F], const bool (&on)[U],
                                                  const double eb, double (&s)[U]) {
  constexpr int FA = F_ ? F_ : MAX_F;
  const uint32_t F = F_ ? (uint32_t)F_ : p.F;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    double acc = 0.0;
#pragma unroll
    for (int x = 0; x < FA; ++x)
      if ((uint32_t)x < F) acc += (t[u][x] * p.boost[x]) * eb;  // ((tfn*idf)*boost)*expansion_boost: the plane holds tfn*idf; a field with tf == 0 adds +0.0
    s[u] = on[u] ? acc : 0.0;
  }
}
template <int F_, int U>
__device__ __forceinline__ void plane_scores(const KParams& p, const uint64_t (&pi)[U], const bool (&on)[U], const double eb, double (&s)[U]) {
  constexpr int FA = F_ ? F_ : MAX_F;
  double t[U][FA];
#pragma unroll
  for (int u = 0; u < U; ++u) {
#pragma unroll
    for (int x = 0; x < FA; ++x) t[u][x] = 0.0;
    if (on[u]) plane_load<F_>(p, pi[u], t[u]);
  }
  scores_from_plane<F_, U>(p, t, on, eb, s);
}

// BM25 scores of U postings per lane from their packed {tf, field length} words (already loaded).
template <int F_, int U>
__device__ __forceinline__ void scores_from_words(const KParams& p, const double* lut, const uint64_t (&pi)[U], const bool (&on)[U],
                                                  const uint32_t (&wv)[U][F_ ? F_ : MAX_F], const double idf, const double eb,
                                                  double (&s)[U]) {
  constexpr int FA = F_ ? F_ : MAX_F;
  const uint32_t F = F_ ? (uint32_t)F_ : p.F;
  uint32_t tfv[U][FA], flv[U][FA];
  tfl_unpack<F_, U>(p, wv, tfv, flv);
  {  // saturated sub-fields: fetch the exact values now, while the posting indices are still live (rare; the whole wave goes)
    bool esc = false;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int x = 0; x < FA; ++x)
        if ((uint32_t)x < F) esc = esc || (on[u] && (tfv[u][x] == TFL_TF_ESC || flv[u][x] == TFL_FL_ESC));
    if (__any(esc)) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int x = 0; x < FA; ++x)
          if ((uint32_t)x < F && on[u]) tfl_exact(p, (uint32_t)x, pi[u], tfv[u][x], flv[u][x]);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    double acc = 0.0;
#pragma unroll
    for (int x = 0; x < FA; ++x) {
      if ((uint32_t)x < F) {
        const uint32_t tfu = tfv[u][x], flu = flv[u][x];
        const bool in_lut = tfu < (uint32_t)LUT_TF && flu < p.lut_cap[x];
        double tfn = lut[in_lut ? __umul24(tfu, p.lut_stride) + p.lut_base[x] + flu : 0u];
        if (!in_lut && tfu > 0) tfn = bm25_tfn_cold(p.k1, p.k1p1, p.one_minus_b, p.b, p.avg[x], tfu, flu);
        const double term = tfn * idf * p.boost[x] * eb;  // bm25.rs:83-86: ((tfn*idf)*boost)*expansion_boost
        acc += (tfu > 0) ? term : 0.0;
      }
    }
    s[u] = on[u] ? acc : 0.0;
  }
}

// Scores of U postings per lane (indices pi[u]); all loads of the trip are issued before the arithmetic.
template <int F_, int U>
__device__ __forceinline__ void posting_scores(const KParams& p, const double* lut, const uint64_t (&pi)[U], const bool (&on)[U],
                                               const double idf, const double eb, double (&s)[U]) {
  constexpr int FA = F_ ? F_ : MAX_F;
  uint32_t wv[U][FA];
#pragma unroll
  for (int u = 0; u < U; ++u) {
#pragma unroll
    for (int x = 0; x < FA; ++x) wv[u][x] = 0;
    if (on[u]) tfl_load<F_>(p, pi[u], wv[u]);
  }
  scores_from_words<F_, U>(p, lut, pi, on, wv, idf, eb, s);
}

// Scores of documents d[u] (where on[u]) in list `en`; 0.0 = the list does not hold the document.
// The U lookups advance together: every step issues U independent loads.
template <int F_, int U>
__device__ __forceinline__ void lookup_scores(const KParams& p, const double* lut, const ps_plan_entry& en, const uint32_t (&d)[U],
                                              const bool (&on)[U], double (&s)[U], WorkStats& ws) {
  if (en.shift & DENSE_FLAG) {  // a dense score row: the value itself
#pragma unroll
    for (int u = 0; u < U; ++u) { s[u] = on[u] ? p.rows[(uint64_t)en.node * p.row_stride + d[u]] : 0.0; ws.row += lanes_on(on[u]); }
    return;
  }
  bool found[U];
  uint64_t pi[U];
#pragma unroll
  for (int u = 0; u < U; ++u) s[u] = 0.0;
  if (en.bm_off != 0xFFFFFFFFu) {
    // denser lists carry a bitmap of {bits, postings before} cells: one 8-byte load answers
    // "is d in the list" (usually no) and, if so, where its posting is
    uint2 cell[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      cell[u] = on[u] ? *reinterpret_cast<const uint2*>(p.bits + (uint64_t)en.bm_off + 2 * (uint64_t)(d[u] >> 5)) : make_uint2(0u, 0u);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ws.cell += lanes_on(on[u]);
      const uint32_t bit = d[u] & 31u;
      found[u] = on[u] && ((cell[u].x >> bit) & 1u);
      pi[u] = en.post_off + cell[u].y + (uint32_t)__popc(cell[u].x & ((1u << bit) - 1u));
    }
  } else {
    // sparse lists: first the list's Bloom filter (one 8-byte load; nearly every document asked is not in
    // the list), then, for a "maybe", the tile-offset table slot - a handful of postings - short binary search
    bool may[U];
    {
      const unsigned long long desc = p.layer_bloom ? p.layer_bloom[en.node] : NO_BLOOM;
      bool any_may = false;
      if (desc != NO_BLOOM) {
        unsigned long long w[U], mk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          uint64_t wi;
          bloom_probe(d[u], desc, wi, mk[u]);
          w[u] = on[u] ? p.bloom[wi] : 0ull;
          ws.cell += lanes_on(on[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { may[u] = on[u] && (w[u] & mk[u]) == mk[u]; any_may |= may[u]; }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) { may[u] = on[u]; any_may |= may[u]; }
      }
      if (!__any(any_may)) return;
    }
    const uint32_t* docs = p.doc + en.post_off;
    uint32_t lo[U], hi[U], end[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      lo[u] = 0; hi[u] = 0; end[u] = 0;
      if (may[u]) {
        const uint32_t slot = (d[u] >> p.t_log2) >> (en.shift & 0xFFu);
        lo[u] = p.table[en.tbl_off + slot];
        end[u] = p.table[en.tbl_off + slot + 1];
        hi[u] = end[u];
      }
      ws.probe += 2u * lanes_on(may[u]);
    }
    bool more = true;  // wave-uniform
    while (more) {
      uint32_t v[U], mid[U];
      bool act[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        act[u] = lo[u] < hi[u];
        mid[u] = (lo[u] + hi[u]) >> 1;
        v[u] = act[u] ? docs[mid[u]] : 0u;
        ws.probe += lanes_on(act[u]);
      }
      bool any_act = false;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (act[u]) { if (v[u] < d[u]) lo[u] = mid[u] + 1; else hi[u] = mid[u]; }
        any_act |= lo[u] < hi[u];
      }
      more = __any(any_act);
    }
    uint32_t chk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { chk[u] = (may[u] && lo[u] < end[u]) ? docs[lo[u]] : 0xFFFFFFFFu; ws.probe += lanes_on(may[u] && lo[u] < end[u]); }
#pragma unroll
    for (int u = 0; u < U; ++u) { found[u] = may[u] && lo[u] < end[u] && chk[u] == d[u]; pi[u] = en.post_off + (found[u] ? lo[u] : 0u); }
  }
  bool any_found = false;
#pragma unroll
  for (int u = 0; u < U; ++u) { any_found |= found[u]; ws.hit += lanes_on(found[u]); }
  if (__any(any_found)) plane_scores<F_, U>(p, pi, found, en.boost, s);
}

#ifndef PS_DAAT_MULTI_WAVES
#define PS_DAAT_MULTI_WAVES 5  // waves per SIMD the multi-expansion arm is compiled for (its register budget)
#endif
template <int F_, bool MULTI>
__global__ __launch_bounds__(WAVE * DAAT_WGW) __attribute__((amdgpu_waves_per_eu(MULTI ? PS_DAAT_MULTI_WAVES : 4))) void k_daat(const KParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int U = (F_ && !MULTI) ? PS_DAAT_U : (F_ ? PS_DAAT_UM : 2);  // postings per lane in flight (the multi-expansion arm keeps per-term maxima per posting)
  const int lane = threadIdx.x & (WAVE - 1);
  const double* lut = reinterpret_cast<const double*>(smem);
  // A grid that covers every item with its own wave assigns them by index (workgroups are dispatched
  // in index order, so the processing order still holds approximately); otherwise the waves are
  // persistent and pull items from the device-scope counter.
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // (the grid is sized by the host's upper bound of the item count; the device-built count is exact)
  const uint32_t item_base = p.item_split_dev ? *p.item_split_dev : p.item_base;
  const uint32_t n_all = p.n_ditems_dev ? *p.n_ditems_dev : 0xFFFFFFFFu;
  const uint32_t n_ditems = min(p.n_ditems, n_all > item_base ? n_all - item_base : 0u);
  const bool by_index = p.n_ditems <= gridDim.x * (uint32_t)DAAT_WGW;
  if (by_index) {
    // most waves of a launch only hold a chunk of a list that is already non-essential: they leave at once (every wave
    // for itself - the waves of a workgroup share nothing -, so none waits for its neighbour's two loads)
    const uint32_t id = blockIdx.x * (uint32_t)DAAT_WGW + (uint32_t)wave;
    if (id >= n_ditems) return;
    const DItem it0 = p.ditems[item_base + id];
    const double theta0 = __longlong_as_double((long long)__hip_atomic_load(&p.gthr[it0.q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (__builtin_amdgcn_readfirstlane((int)(it0.skip_thr < theta0))) {
      if (lane == 0) p.cand_cnt[it0.slot] = 0u;
      return;
    }
  }
  bool first = true;
  for (;;) {
    uint32_t id = 0;
    if (by_index) {
      if (!first) break;
      first = false;
      id = blockIdx.x * (uint32_t)DAAT_WGW + (uint32_t)wave;
    } else {
      if (lane == 0) id = atomicAdd(p.work_counter, 1u);
      id = __builtin_amdgcn_readfirstlane(id);
    }
    if (id >= n_ditems) break;
    const DItem it = p.ditems[item_base + id];
    const uint32_t e_own = __builtin_amdgcn_readfirstlane(it.entry);
    const ps_plan_entry& own = p.plan[e_own];
    const DEntry de = p.dentry[e_own];
    const uint32_t q = __builtin_amdgcn_readfirstlane(de.q);
    const uint32_t e0 = p.qbeg[q], e1 = p.qbeg[q + 1];
    const double own_eb = own.boost;
    const uint64_t own_off = own.post_off;
    const uint32_t own_rank = de.rank;
    const double skip_thr = de.skip_thr;
    // What the OTHER lists can add to a document evaluated here.  A document is evaluated from its highest-ranked list
    // only, so one that is evaluated here sits in no list ranked above the own one: for plans with one list per query
    // term that is the sum of the bounds of the lists ranked BELOW it (the preparation's `others` counts every other list:
    // still what the plan-order fallback arms use).
    double others = de.others;
    if (!MULTI && e1 - e0 <= 64u) {
      others = 0.0;
      for (uint32_t r = e0 + own_rank + 1u; r < e1; ++r) others += p.dentry[p.rorder[r]].ub;
      others *= 1.0 + 1e-9;
    }
    // multi-expansion queries: the query term of this list, and per query term the bound of its best
    // OTHER list (what pass 1 starts from)
    uint32_t own_grp = 0xFFFFFFFFu;
    double rem0[4] = {0.0, 0.0, 0.0, 0.0};
    if (MULTI && p.dgroup != nullptr && e1 - e0 <= 64u) {
      own_grp = p.dgroup[e_own].grp;
      if (own_grp < 4u) {
        // (only the lists ranked BELOW the own one: a document evaluated here sits in no higher-ranked list - it would be
        // evaluated there -, so those can add nothing; they are only asked, last, whether they cancel a survivor)
        for (uint32_t r = e1; r-- > e0 + own_rank + 1u;) {  // ascending bound: the last write per term is its best list
          const uint32_t j = p.rorder[r];
          const DGroup gj = p.dgroup[j];
#pragma unroll
          for (int g = 0; g < 4; ++g)
            if ((uint32_t)g == gj.grp) rem0[g] = gj.ub_s;
        }
      }
    }
    TopK tk;
    tk.s = -1.0; tk.d = 0xFFFFFFFFu; tk.n = 0; tk.thr_s = 0.0; tk.thr_d = 0;
    double published = 0.0;
    const uint32_t end = (p.ablate & 16u) ? it.begin : it.begin + it.count;  // (debug: 16 = no postings)
    bool essential = true;  // wave-uniform
    WorkStats ws;
    bool handled = false;
    if constexpr (MULTI && PS_DAAT_MQ != 0) {
      if (e1 - e0 <= 64u && own_grp < 4u) {
        // Several expansions per query term (the expansions of one term merge by max, query.rs:150-164: a document
        // scores at most the sum over query terms of the best of its lists of that term), in two stages.
        // The walk over the other lists, highest bound first, is a chain of dependent lookups that a wave follows
        // as long as ANY of its postings is alive - yet a posting survives 1.7 lookups on average (C5).  So the
        // scan stage only does the FIRST lookup (the highest-bound other list) for the postings of a trip, UA per
        // lane in flight; what is still alive - a fraction of the lanes - waits in a wave-private LDS queue until 64
        // are together, and the rest of the walk (pass 1 from the second list on, pass 2 = the add / max state
        // machine in plan order) runs with every lane busy.
        handled = true;
        constexpr int UA = F_ ? PS_DAAT_UMQ : 2;
        constexpr uint32_t QCAP = 128;  // a push adds <= 64 to < 64
        __shared__ uint32_t mq_d[DAAT_WGW][QCAP];
        __shared__ double mq_so[DAAT_WGW][QCAP];
        __shared__ double mq_s1[DAAT_WGW][QCAP];
#if PS_DAAT_MRQ
        // Reach queue: the postings that pass the first bound test (about one in seven on C5) wait here until 64 are
        // together; the first lookup then runs with every lane busy instead of once per trip over four sparse slots
        // (512 entries: a trip adds up to UA x 64 to < 64.  level1 and process each have ONE call site, at the top of the
        // loop: inlined at several sites the two bodies - every lookup_scores in them - no longer fit the instruction
        // cache, 6.7 ms instead of 1.7)
        constexpr uint32_t RCAP = (UA + 1) * 64 <= 256 ? 256 : 512;  // (a trip adds up to UA x 64 to < 64)
        __shared__ uint32_t rq_d[DAAT_WGW][RCAP];
        __shared__ double rq_so[DAAT_WGW][RCAP];
        uint32_t rq_head = 0, rq_n = 0;  // wave-uniform
#endif
        uint32_t q_head = 0, q_n = 0;  // wave-uniform
        const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
        const uint32_t r1 = min(e1, e0 + own_rank + 1u);  // the highest-bound list ranked below the own one
        const bool has1 = r1 < e1;
        const uint32_t j1 = has1 ? p.rorder[r1] : e_own;
        const ps_plan_entry& en1 = p.plan[j1];
        const uint32_t g1 = has1 ? p.dgroup[j1].grp : 0xFFFFFFFFu;
        const uint32_t j1_rank = p.dentry[j1].rank;
        double rem1[4] = {rem0[0], rem0[1], rem0[2], rem0[3]};  // per query term: the best list not looked at after level 1
        if (has1) {
          const double nxt = p.dgroup[j1].nxt_s;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            if ((uint32_t)g == g1) rem1[g] = nxt;
        }
        double theta = 0.0;
        double alt0 = 0.0, others0 = 0.0;  // the first bound test: the own term's best other list, the other terms' best lists (all ranked below)
#pragma unroll
        for (int g = 0; g < 4; ++g) { if ((uint32_t)g == own_grp) alt0 = rem0[g]; else others0 += rem0[g]; }
#ifdef PS_MQ_TIME
        unsigned long long mq_tb1 = 0, mq_tb2 = 0, mq_cnt = 0;
#endif
        // the rest of the walk for the first `count` (<= 64) queued documents, one per lane
        auto process = [&](const uint32_t count) {
          const uint32_t at = (q_head + (uint32_t)lane) & (QCAP - 1u);
          const bool ok = (uint32_t)lane < count;
          const uint32_t d1[1] = {ok ? mq_d[wave][at] : 0u};
          const double so = ok ? mq_so[wave][at] : 0.0, s1v = ok ? mq_s1[wave][at] : 0.0;
          q_head = (q_head + count) & (QCAP - 1u);
          q_n -= count;
#ifdef PS_MQ_TIME  // profiling builds only: time in this stage -> the `probe` counter, survivors -> `offer`, 100 ns units of the whole arm -> `row`
          const unsigned long long t_b0 = __builtin_amdgcn_s_memrealtime();
          mq_cnt += count;
#endif
          bool alive1[1] = {ok};
          double act[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            act[g] = (uint32_t)g == own_grp ? so : 0.0;
            if ((uint32_t)g == g1 && s1v > 0.0) act[g] = fmax(act[g], s1v);
          }
          unsigned long long hits = s1v > 0.0 ? 1ull << (j1 - e0) : 0ull;
          double rem[4] = {rem1[0], rem1[1], rem1[2], rem1[3]};
          bool any_alive = true;
          for (uint32_t r = r1 + 1; r < e1 && any_alive; ++r) {
            const uint32_t j = p.rorder[r];
            if (j != e_own) {
              const ps_plan_entry& en = p.plan[j];
              const DGroup gj = p.dgroup[j];
              const uint32_t j_rank = p.dentry[j].rank;
              double s[1];
              lookup_scores<F_, 1>(p, lut, en, d1, alive1, s, ws);
#pragma unroll
              for (int g = 0; g < 4; ++g)
                if ((uint32_t)g == gj.grp) rem[g] = gj.nxt_s;
              if (alive1[0]) {
                if (s[0] > 0.0) {
                  hits |= 1ull << (j - e0);
#pragma unroll
                  for (int g = 0; g < 4; ++g)
                    if ((uint32_t)g == gj.grp) act[g] = fmax(act[g], s[0]);
                }
                double bound = 0.0;
#pragma unroll
                for (int g = 0; g < 4; ++g) bound += fmax(act[g], rem[g]);
                // (j_rank < own_rank: the document is evaluated from its highest-bound list only)
                if (bound < theta || (s[0] > 0.0 && j_rank < own_rank)) alive1[0] = false;
              }
              any_alive = __any(alive1[0]);
            }
          }
          // the survivors: is the document in a list ranked above the own one?  Then it is evaluated there, not here.
          for (uint32_t r = e0; r < e0 + own_rank && any_alive; ++r) {
            const ps_plan_entry& en = p.plan[p.rorder[r]];
            double s[1];
            lookup_scores<F_, 1>(p, lut, en, d1, alive1, s, ws);
            if (s[0] > 0.0) alive1[0] = false;
            any_alive = __any(alive1[0]);
          }
#ifdef PS_MQ_TIME
          mq_tb1 += __builtin_amdgcn_s_memrealtime() - t_b0;
#endif
          if (!any_alive) return;
          // pass 2, the survivors: the add / max state machine in PLAN order (query.rs:33-89,150-164)
#ifdef PS_MQ_TIME
          const unsigned long long t_b2 = __builtin_amdgcn_s_memrealtime();
#endif
          double P = 0.0;
          bool present = false, visited = false;
          uint32_t cur_qterm = 0xFFFFFFFFu;
          for (uint32_t j = e0; j < e1; ++j) {
            const ps_plan_entry& en = p.plan[j];
            if (en.qterm != cur_qterm) {  // query.rs:37
              cur_qterm = en.qterm;
              visited = false;
            }
            double s[1] = {0.0};
            if (j == e_own) {
              s[0] = so;
            } else if (has1 && j == j1) {
              s[0] = s1v;  // (looked up by the scan stage)
            } else {
              // (keeping what pass 1 found in registers instead - 7 lists - cost 14 VGPRs and the fourth wave per
              // SIMD: 2.64 ms against 2.44 on C5)
              bool want[1] = {alive1[0] && ((hits >> (j - e0)) & 1ull)};
              if (__any(want[0])) lookup_scores<F_, 1>(p, lut, en, d1, want, s, ws);
            }
            if (alive1[0] && s[0] > 0.0) {
              P = present ? (visited ? fmax(P, s[0]) : P + s[0]) : s[0];
              visited = true;
              present = true;
            }
          }
          const bool offer = alive1[0] && P >= theta;
          ws.offer += lanes_on(offer);
          if (__any(offer)) topk_offer(tk, p.K, lane, alive1[0], P, d1[0], theta);
          if (tk.n == p.K && tk.thr_s > published && tk.thr_s > theta) {
            // this wave's K-th best so far: the final K-th best of the query can only be higher
            published = tk.thr_s;
            if (lane == 0) atomicMax(&p.gthr[q], (unsigned long long)__double_as_longlong(tk.thr_s));
          }
#ifdef PS_MQ_TIME
          mq_tb2 += __builtin_amdgcn_s_memrealtime() - t_b2;
#endif
        };
#if PS_DAAT_MRQ
        // the first lookup (the highest-bound list ranked below the own one) for the first `count` (<= 64) documents of the
        // reach queue, one per lane; what is still alive moves on to the survivor queue
        auto level1 = [&](const uint32_t count) {
          const uint32_t rat = (rq_head + (uint32_t)lane) & (RCAP - 1u);
          bool on[1] = {(uint32_t)lane < count};
          const uint32_t dq[1] = {on[0] ? rq_d[wave][rat] : 0u};
          const double so = on[0] ? rq_so[wave][rat] : 0.0;
          rq_head = (rq_head + count) & (RCAP - 1u);
          rq_n -= count;
          double s1[1] = {0.0};
          if (has1) lookup_scores<F_, 1>(p, lut, en1, dq, on, s1, ws);
          if (on[0]) {
            double bound = 0.0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              double a = (uint32_t)g == own_grp ? so : 0.0;
              if ((uint32_t)g == g1 && s1[0] > 0.0) a = fmax(a, s1[0]);
              bound += fmax(a, rem1[g]);
            }
            if (bound < theta || (s1[0] > 0.0 && j1_rank < own_rank)) on[0] = false;
          }
          const unsigned long long m = __ballot(on[0]);
          if (m) {
            if (on[0]) {
              const uint32_t at = (q_head + q_n + (uint32_t)__popcll(m & lt)) & (QCAP - 1u);
              mq_d[wave][at] = dq[0];
              mq_so[wave][at] = so;
              mq_s1[wave][at] = s1[0];
            }
            q_n += (uint32_t)__popcll(m);  // (a push adds <= 64 to < 64: the survivor queue is drained first at the top of the loop)
          }
        };
#endif
        uint32_t i0 = it.begin;
#ifdef PS_MQ_TIME
        const unsigned long long t_a0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (;;) {
          const bool scanning = i0 < end && essential;
#if PS_DAAT_MRQ
          const uint32_t rq_left = rq_n;
#else
          const uint32_t rq_left = 0u;
#endif
          // (the survivor queue first, so that it holds < 64 whenever the reach queue hands it up to 64 more; its rest last)
          if (q_n >= (uint32_t)WAVE || (!scanning && !rq_left && q_n)) { process(min(q_n, (uint32_t)WAVE)); continue; }
#if PS_DAAT_MRQ
          if (rq_n >= (uint32_t)WAVE || (!scanning && rq_n)) { level1(min(rq_n, (uint32_t)WAVE)); continue; }
#endif
          if (!scanning) break;
          const unsigned long long tbits = __hip_atomic_load(&p.gthr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          theta = __hiloint2double(__builtin_amdgcn_readfirstlane((int)(tbits >> 32)),
                                   __builtin_amdgcn_readfirstlane((int)(uint32_t)tbits));
          essential = !(skip_thr < theta);  // false: the whole list has become non-essential
          uint32_t d[UA];
          uint64_t pi[UA];
          bool alive[UA];
          double s_own[UA];
#pragma unroll
          for (int u = 0; u < UA; ++u) {
            const uint32_t i = i0 + u * WAVE + lane;
            alive[u] = essential && i < end;
            pi[u] = own_off + (i < end ? i : end - 1);
            d[u] = p.doc[pi[u]];
          }
          if (p.alive != nullptr) {  // delta removals
            uint32_t aw[UA];  // (every d[u] is a real doc id - out-of-range lanes re-read the last posting: all words requested together, no branch per posting)
#pragma unroll
            for (int u = 0; u < UA; ++u) aw[u] = p.alive[d[u] >> 5];
#pragma unroll
            for (int u = 0; u < UA; ++u) alive[u] = alive[u] & (bool)((aw[u] >> (d[u] & 31u)) & 1u);
          }
          plane_scores<F_, UA>(p, pi, alive, own_eb, s_own);
          bool any_alive = false;
#pragma unroll
          for (int u = 0; u < UA; ++u) {
            // everything the lower-ranked lists could add, at most (per query term the best of them): below theta the document is out
            alive[u] = alive[u] && (fmax(s_own[u], alt0) + others0 >= theta) && !(p.ablate & 32u);  // (debug: 32 = no lookups)
            any_alive |= alive[u];
            ws.reached += lanes_on(alive[u]);
          }
          if (essential) ws.scanned += min(end - i0, (uint32_t)(WAVE * UA)); else ws.probe += min(end - i0, (uint32_t)(WAVE * UA));
          i0 += WAVE * UA;
          if (!__any(any_alive)) continue;
#if PS_DAAT_MRQ
#pragma unroll
          for (int u = 0; u < UA; ++u) {
            const unsigned long long m = __ballot(alive[u]);
            if (m) {
              if (alive[u]) {
                const uint32_t at = (rq_head + rq_n + (uint32_t)__popcll(m & lt)) & (RCAP - 1u);
                rq_d[wave][at] = d[u];
                rq_so[wave][at] = s_own[u];
              }
              rq_n += (uint32_t)__popcll(m);
            }
          }
        }
#else
          double s1[UA];
#pragma unroll
          for (int u = 0; u < UA; ++u) s1[u] = 0.0;
          if (has1) lookup_scores<F_, UA>(p, lut, en1, d, alive, s1, ws);
#pragma unroll
          for (int u = 0; u < UA; ++u) {
            if (alive[u]) {
              double bound = 0.0;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                double a = (uint32_t)g == own_grp ? s_own[u] : 0.0;
                if ((uint32_t)g == g1 && s1[u] > 0.0) a = fmax(a, s1[u]);
                bound += fmax(a, rem1[g]);
              }
              if (bound < theta || (s1[u] > 0.0 && j1_rank < own_rank)) alive[u] = false;
            }
            const unsigned long long m = __ballot(alive[u]);
            if (m) {
              if (alive[u]) {
                const uint32_t at = (q_head + q_n + (uint32_t)__popcll(m & lt)) & (QCAP - 1u);
                mq_d[wave][at] = d[u];
                mq_so[wave][at] = s_own[u];
                mq_s1[wave][at] = s1[u];
              }
              q_n += (uint32_t)__popcll(m);
              if (q_n >= (uint32_t)WAVE) process((uint32_t)WAVE);
            }
          }
        }
#endif
#ifdef PS_MQ_TIME
        ws.probe = (uint32_t)mq_tb1; ws.offer = (uint32_t)mq_tb2; ws.row = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_a0); ws.hit = (uint32_t)mq_cnt;
#endif
      }
    }
    if (!handled)
    for (uint32_t i0 = it.begin; i0 < end && essential; i0 += WAVE * U) {
      // the query's current threshold: a lower bound of its final K-th best score (0 = none yet).
      // One load instruction returns one value to the whole wave; readfirstlane tells the compiler.
      const unsigned long long tbits = __hip_atomic_load(&p.gthr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double theta = __hiloint2double(__builtin_amdgcn_readfirstlane((int)(tbits >> 32)),
                                            __builtin_amdgcn_readfirstlane((int)(uint32_t)tbits));
      essential = !(skip_thr < theta);  // false: the whole list has become non-essential
      uint32_t d[U];
      uint64_t pi[U];
      bool alive[U];
      double s_own[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t i = i0 + u * WAVE + lane;
        alive[u] = essential && i < end;
        pi[u] = own_off + (i < end ? i : end - 1);
        d[u] = p.doc[pi[u]];
      }
      if (p.alive != nullptr) {  // delta removals
        uint32_t aw[U];  // (every d[u] is a real doc id: all words requested together, no branch per posting)
#pragma unroll
        for (int u = 0; u < U; ++u) aw[u] = p.alive[d[u] >> 5];
#pragma unroll
        for (int u = 0; u < U; ++u) alive[u] = alive[u] & (bool)((aw[u] >> (d[u] & 31u)) & 1u);
      }
      plane_scores<F_, U>(p, pi, alive, own_eb, s_own);
      bool any_alive = false;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // everything the other entries could add, at most: below theta the document is out
        alive[u] = alive[u] && (s_own[u] + others >= theta) && !(p.ablate & 32u);  // (debug: 32 = no lookups)
        any_alive |= alive[u];
        ws.reached += lanes_on(alive[u]);
      }
      // (the doc ids of a trip are requested together with the threshold: a trip that finds its list
      // non-essential has read them - 4 bytes each, booked as probes - but not the packed words)
      if (essential) ws.scanned += min(end - i0, (uint32_t)(WAVE * U)); else ws.probe += min(end - i0, (uint32_t)(WAVE * U));
      any_alive = __any(any_alive);
      double P[U];
#pragma unroll
      for (int u = 0; u < U; ++u) P[u] = 0.0;
      if (any_alive) {
        const uint32_t ne = e1 - e0;
        if (!MULTI && ne <= 64u) {
          // Pass 1, highest-bound lists first: every lookup replaces a list's bound by what it really
          // adds (usually nothing), and a document is dropped as soon as what is left cannot reach
          // theta.  `others` is inflated by 1e-9, far above the rounding of this running sum.
          double bound[U];
          unsigned long long hits[U];
#pragma unroll
          for (int u = 0; u < U; ++u) { bound[u] = s_own[u] + others; hits[u] = 0ull; }
          for (uint32_t r = e0 + own_rank + 1u; r < e1 && any_alive; ++r) {  // (the lists ranked below the own one: see others_low)
            const uint32_t j = p.rorder[r];
            {
              const ps_plan_entry& en = p.plan[j];
              const DEntry dj = p.dentry[j];
              double s[U];
              lookup_scores<F_, U>(p, lut, en, d, alive, s, ws);
              bool any = false;
#pragma unroll
              for (int u = 0; u < U; ++u) {
                if (alive[u]) {
                  bound[u] = (bound[u] - dj.ub) + s[u];
                  if (s[u] > 0.0) hits[u] |= 1ull << (j - e0);
                  // (dj.rank < own_rank: the document is evaluated from its highest-bound list only)
                  if (bound[u] < theta || (s[u] > 0.0 && dj.rank < own_rank)) alive[u] = false;
                }
                any |= alive[u];
              }
              any_alive = __any(any);
            }
          }
          // the survivors: a document that sits in a list ranked above the own one is evaluated there, not here
          for (uint32_t r = e0; r < e0 + own_rank && any_alive; ++r) {
            const ps_plan_entry& en = p.plan[p.rorder[r]];
            double s[U];
            lookup_scores<F_, U>(p, lut, en, d, alive, s, ws);
            bool any = false;
#pragma unroll
            for (int u = 0; u < U; ++u) { if (s[u] > 0.0) alive[u] = false; any |= alive[u]; }
            any_alive = __any(any);
          }
          // Pass 2, the few survivors: the sum in PLAN order (query.rs:33-89; one list per query term:
          // always the `+` / assign arm, 0.0 + s == s), same operands, same order, same bits
          if (any_alive) {
            for (uint32_t j = e0; j < e1; ++j) {
              const ps_plan_entry& en = p.plan[j];
              double s[U];
              if (j == e_own) {
#pragma unroll
                for (int u = 0; u < U; ++u) s[u] = s_own[u];
              } else {
                bool want[U];
                bool any = false;
#pragma unroll
                for (int u = 0; u < U; ++u) { want[u] = alive[u] && ((hits[u] >> (j - e0)) & 1ull); any |= want[u]; s[u] = 0.0; }
                if (__any(any)) lookup_scores<F_, U>(p, lut, en, d, want, s, ws);
              }
#pragma unroll
              for (int u = 0; u < U; ++u)
                if (alive[u] && s[u] > 0.0) P[u] += s[u];
            }
          }
        } else if (MULTI && PS_DAAT_MQ == 0 && ne <= 64u && own_grp < 4u) {
          // Several expansions per query term: the expansions of one term merge by max
          // (query.rs:150-164), so a document scores at most the sum over query terms of the best of
          // its lists of that term.  Pass 1 (highest-bound lists first) keeps, per query term, the best
          // contribution found so far (per posting) and the bound of the best list not looked at yet
          // (wave-uniform); the posting is dropped when their sum cannot reach theta.
          double act[U][4];
          unsigned long long hits[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            hits[u] = 0ull;
#pragma unroll
            for (int g = 0; g < 4; ++g) act[u][g] = (uint32_t)g == own_grp ? s_own[u] : 0.0;
          }
          double rem[4] = {rem0[0], rem0[1], rem0[2], rem0[3]};
          for (uint32_t r = e0; r < e1 && any_alive; ++r) {
            const uint32_t j = p.rorder[r];
            if (j != e_own) {
              const ps_plan_entry& en = p.plan[j];
              const DGroup gj = p.dgroup[j];
              const uint32_t j_rank = p.dentry[j].rank;
              double s[U];
              lookup_scores<F_, U>(p, lut, en, d, alive, s, ws);
#pragma unroll
              for (int g = 0; g < 4; ++g)
                if ((uint32_t)g == gj.grp) rem[g] = gj.nxt_s;
              bool any = false;
#pragma unroll
              for (int u = 0; u < U; ++u) {
                if (alive[u]) {
                  if (s[u] > 0.0) {
                    hits[u] |= 1ull << (j - e0);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                      if ((uint32_t)g == gj.grp) act[u][g] = fmax(act[u][g], s[u]);
                  }
                  double bound = 0.0;
#pragma unroll
                  for (int g = 0; g < 4; ++g) bound += fmax(act[u][g], rem[g]);
                  // (j_rank < own_rank: the document is evaluated from its highest-bound list only)
                  if (bound < theta || (s[u] > 0.0 && j_rank < own_rank)) alive[u] = false;
                }
                any |= alive[u];
              }
              any_alive = __any(any);
            }
          }
          // Pass 2, the survivors: the add / max state machine in PLAN order (query.rs:33-89,150-164)
          if (any_alive) {
            bool present[U], visited[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { present[u] = false; visited[u] = false; }
            uint32_t cur_qterm = 0xFFFFFFFFu;
            for (uint32_t j = e0; j < e1; ++j) {
              const ps_plan_entry& en = p.plan[j];
              if (en.qterm != cur_qterm) {  // query.rs:37
                cur_qterm = en.qterm;
#pragma unroll
                for (int u = 0; u < U; ++u) visited[u] = false;
              }
              double s[U];
              if (j == e_own) {
#pragma unroll
                for (int u = 0; u < U; ++u) s[u] = s_own[u];
              } else {
                bool want[U];
                bool any = false;
#pragma unroll
                for (int u = 0; u < U; ++u) { want[u] = alive[u] && ((hits[u] >> (j - e0)) & 1ull); any |= want[u]; s[u] = 0.0; }
                if (__any(any)) lookup_scores<F_, U>(p, lut, en, d, want, s, ws);
              }
#pragma unroll
              for (int u = 0; u < U; ++u) {
                if (alive[u] && s[u] > 0.0) {
                  P[u] = present[u] ? (visited[u] ? fmax(P[u], s[u]) : P[u] + s[u]) : s[u];
                  visited[u] = true;
                  present[u] = true;
                }
              }
            }
          }
        } else {
          bool present[U], visited[U], dup[U];
#pragma unroll
          for (int u = 0; u < U; ++u) { present[u] = false; visited[u] = false; dup[u] = false; }
          uint32_t cur_qterm = 0xFFFFFFFFu;
          for (uint32_t j = e0; j < e1; ++j) {  // plan order (query.rs:33-89)
            const ps_plan_entry& en = p.plan[j];
            if (MULTI && en.qterm != cur_qterm) {  // query.rs:37
              cur_qterm = en.qterm;
#pragma unroll
              for (int u = 0; u < U; ++u) visited[u] = false;
            }
            double s[U];
            if (j == e_own) {
#pragma unroll
              for (int u = 0; u < U; ++u) s[u] = s_own[u];
            } else {
              lookup_scores<F_, U>(p, lut, en, d, alive, s, ws);
            }
            const uint32_t j_rank = p.dentry[j].rank;
#pragma unroll
            for (int u = 0; u < U; ++u) {
              if (alive[u] && s[u] > 0.0) {
                // the document is evaluated from its highest-bound list only
                if (j != e_own && j_rank < own_rank) dup[u] = true;
                if (MULTI) {
                  // max_score_merger (query.rs:150-164)
                  P[u] = present[u] ? (visited[u] ? fmax(P[u], s[u]) : P[u] + s[u]) : s[u];
                  visited[u] = true;
                } else {
                  P[u] += s[u];
                }
                present[u] = true;
              }
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) alive[u] = alive[u] && !dup[u] && present[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool offer = alive[u] && P[u] >= theta;
          ws.offer += lanes_on(offer);
          if (__any(offer)) topk_offer(tk, p.K, lane, alive[u], P[u], d[u], theta);
        }
        if (tk.n == p.K && tk.thr_s > published && tk.thr_s > theta) {
          // this wave's K-th best so far: the final K-th best of the query can only be higher
          published = tk.thr_s;
          if (lane == 0) atomicMax(&p.gthr[q], (unsigned long long)__double_as_longlong(tk.thr_s));
        }
      }
    }
    if ((uint32_t)lane < p.K) {
      const uint64_t o = (uint64_t)it.slot * p.K + lane;
      const bool ok = (uint32_t)lane < tk.n;
      p.cand_score[o] = ok ? tk.s : 0.0;
      p.cand_doc[o] = ok ? tk.d : 0xFFFFFFFFu;
      if (lane == 0) p.cand_cnt[it.slot] = tk.n;
    }
    if (PS_WORK_COUNTERS && lane == 0) {  // (an item skipped whole by its workgroup never gets here: it read nothing)
      unsigned long long* w = p.wstats + (size_t)(blockIdx.x & (WS_SLOTS - 1u)) * WS_WORDS;
      atomicAdd(&w[WS_ITEMS_RUN], 1ull);
      if (ws.scanned) atomicAdd(&w[WS_SCANNED], (unsigned long long)ws.scanned);
      if (ws.reached) atomicAdd(&w[WS_REACHED], (unsigned long long)ws.reached);
      if (ws.row) atomicAdd(&w[WS_ROW], (unsigned long long)ws.row);
      if (ws.cell) atomicAdd(&w[WS_CELL], (unsigned long long)ws.cell);
      if (ws.probe) atomicAdd(&w[WS_PROBE], (unsigned long long)ws.probe);
      if (ws.hit) atomicAdd(&w[WS_HIT], (unsigned long long)ws.hit);
      if (ws.offer) atomicAdd(&w[WS_OFFER], (unsigned long long)ws.offer);
    }
  }
}

// ------------------------------------------------------------------------------------------
// K1d for small plans: k_daat_small - the same exact dynamic pruning as k_daat for batches whose queries
// have one list per query term and at most 4 lists (BASELINE C2 / C4: 3), with the dependent-load chain of a
// trip cut from ~15 levels to ~4.  k_daat walks the other lists one after the other, twice (pass 1 prunes,
// pass 2 re-looks the hits up in plan order), every lookup hanging on the previous one's outcome; its
// launch time is the number of trips per wave slot times that chain (the kernel moves ~0.6 GB: no
// throughput roof is near).  Here a trip issues, as soon as its doc ids are known, the FIRST-level load of
// every other list together - dense-row value, {bits, rank} bitmap cell, or the two table words of a sparse
// list's slot - next to the own postings' packed words; bounds are then tightened with what is already
// exact (row values, bitmap membership), the survivors fetch what is left (packed words of bitmap hits; up
// to 4 doc ids of a sparse slot at once, then the packed words of a match), and the contributions are
// summed in PLAN order as they complete: same operands, same order of additions, same bits as k_daat / k_score.
// ------------------------------------------------------------------------------------------
#ifndef PS_DAAT_US
#define PS_DAAT_US 4   // postings per lane in flight
#endif
#ifndef PS_EXP
#define PS_EXP 0       // profiling builds only (wrong results): 1 = no top-K offers, 2 = no second level, 4 = no first-level loads
#endif
constexpr int DAAT_SMALL_MAX = 4;  // most lists per query

// WC: keep the work counters (ps_work_counters).  The serving instantiation (PS_WORK_COUNTERS=0 at run time) carries none
// of the ballots / popcounts / atomics they cost (4 % of the kernel on C2).
#ifndef PS_DAAT_SMALL_BARRIER
// 1: the waves of a workgroup decide together whether to leave at once (one __syncthreads_or); 0: every wave for itself,
// as k_daat and k_daat_z do.  Nothing is shared either way - but without the barrier this kernel compiles to 78 VGPRs and
// 145-165 SGPR spills instead of 123 / 114, and that code is slower: C2 0.273 -> 0.276 ms, C4 1.095 -> 1.212 (same box).
#define PS_DAAT_SMALL_BARRIER 1
#endif
// NL: most lists of a query of the launch (3 or DAAT_SMALL_MAX = 4).  The per-list words of the OTHER lists are wave-uniform state
// (scalar registers, spilled to VGPR lanes beyond ~100) and every one of them unrolls another copy of the lookup code: the launch
// of three-list queries (BASELINE configs 2 and 4) instantiated for three lists instead of four takes 0.288 -> 0.275 ms per step
// on C2, its counting instantiation 0.268 -> 0.228 ms per launch (round 5, A/B/A/B on one box).
#ifndef PS_DAAT_SMALL_WAVES3
#define PS_DAAT_SMALL_WAVES3 4  // waves per SIMD the register allocation of the three-list instantiation aims at (102 VGPRs as is; 5 needs <= 96)
#endif
template <int F_, bool WC, int NL = DAAT_SMALL_MAX>
__global__ __launch_bounds__(WAVE * DAAT_WGW) __attribute__((amdgpu_waves_per_eu(NL <= 3 ? PS_DAAT_SMALL_WAVES3 : 4))) void k_daat_small(const KParams p) {
  static_assert(NL >= 2 && NL <= DAAT_SMALL_MAX, "k_daat_small is instantiated for 3 or 4 lists per query");
  auto cnt = [](const bool b) -> uint32_t { return WC ? (uint32_t)__popcll(__ballot(b)) : 0u; };  // wave-uniform count of lanes where b holds
  constexpr int U = PS_DAAT_US;
  constexpr int NO = NL - 1;              // other lists of a query
  constexpr int FA = F_ ? F_ : MAX_F;
  constexpr uint32_t QCAP = 128;          // survivor queue entries per wave (a push adds <= 64 to < 64)
  constexpr double SLACK = 1.0 + 1e-9;    // bounds are summed in another order than the scores
  // Survivor queue (wave-private LDS ring): the documents of a trip that are still alive after the first
  // level - a few percent of the lanes - wait here until 64 of them are together; their second level
  // (packed / plane words of bitmap hits, the doc ids of a sparse slot, the plan-order sum, the top-K offer)
  // then runs with every lane busy instead of once per trip for a handful of lanes.
  __shared__ uint32_t q_d[DAAT_WGW][QCAP];
  __shared__ double q_s[DAAT_WGW][QCAP];
  __shared__ unsigned long long q_loc[NO][DAAT_WGW][QCAP];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t n_ditems = p.n_ditems_dev ? min(p.n_ditems, *p.n_ditems_dev) : p.n_ditems;
  const uint32_t id = blockIdx.x * (uint32_t)DAAT_WGW + (uint32_t)wave;
#if PS_DAAT_SMALL_BARRIER
  {
    // most workgroups of a launch only hold chunks of lists that are already non-essential: they leave at once
    int need = 0;
    if (id < n_ditems) {
      const DItem it0 = p.ditems[id];
      const double theta = __longlong_as_double((long long)__hip_atomic_load(&p.gthr[it0.q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      need = !(it0.skip_thr < theta);
    }
    if (!__syncthreads_or(need)) {
      if (id < n_ditems && lane == 0) p.cand_cnt[p.ditems[id].slot] = 0u;
      return;
    }
  }
  if (id >= n_ditems) return;
  const DItem it = p.ditems[id];
#else
  if (id >= n_ditems) return;
  const DItem it = p.ditems[id];
  {
    // most waves of a launch only hold a chunk of a list that is already non-essential: they leave at once (every wave
    // for itself - the waves of a workgroup share nothing -, so none waits for its neighbour's two loads)
    const double theta0 = __longlong_as_double((long long)__hip_atomic_load(&p.gthr[it.q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (__builtin_amdgcn_readfirstlane((int)(it.skip_thr < theta0))) {
      if (lane == 0) p.cand_cnt[it.slot] = 0u;
      return;
    }
  }
#endif
  const uint32_t e_own = __builtin_amdgcn_readfirstlane(it.entry);
  const ps_plan_entry& own = p.plan[e_own];
  const DEntry de = p.dentry[e_own];
  const uint32_t q = __builtin_amdgcn_readfirstlane(de.q);
  const uint32_t e0 = p.qbeg[q], ne = p.qbeg[q + 1] - e0;  // ne <= NL (host: the launch's instantiation covers its longest plan)
  const uint32_t own_pos = e_own - e0;
  const double own_eb = own.boost;
  const uint64_t own_off = own.post_off;
  const uint32_t own_rank = de.rank;
  const double skip_thr = de.skip_thr;
  // the other lists, in plan order (wave-uniform: scalar registers)
  uint64_t o_off[NO];
  uint32_t o_shift[NO], o_bm[NO], o_tbl[NO], o_row[NO], o_rank[NO];
  unsigned long long o_bloom[NO];
  double o_eb[NO], o_ub[NO];
#pragma unroll
  for (int k = 0; k < NO; ++k) {
    o_off[k] = 0; o_shift[k] = 0; o_bm[k] = 0xFFFFFFFFu; o_tbl[k] = 0; o_row[k] = 0; o_rank[k] = 0xFFFFFFFFu;
    o_eb[k] = 0.0; o_ub[k] = 0.0; o_bloom[k] = NO_BLOOM;
    if ((uint32_t)k + 1u < ne) {
      const uint32_t j = e0 + (uint32_t)k + ((uint32_t)k >= own_pos ? 1u : 0u);
      const ps_plan_entry& en = p.plan[j];
      const DEntry dj = p.dentry[j];
      o_off[k] = en.post_off; o_shift[k] = en.shift; o_bm[k] = en.bm_off; o_tbl[k] = en.tbl_off; o_row[k] = en.node;
      o_eb[k] = en.boost; o_ub[k] = dj.ub; o_rank[k] = dj.rank;
      if (!(en.shift & DENSE_FLAG) && en.bm_off == 0xFFFFFFFFu && p.layer_bloom) o_bloom[k] = p.layer_bloom[en.node];
    }
  }
  // A document is evaluated from its highest-ranked list only, so one that is evaluated HERE sits in no list ranked
  // above the own one: only the lists ranked BELOW can add to it.  (A document that does sit in a higher-ranked list
  // is cancelled further down if it gets that far; it is evaluated by that list's items, under that list's bounds.)
  double others = 0.0;
#pragma unroll
  for (int k = 0; k < NO; ++k)
    if ((uint32_t)k + 1u < ne && o_rank[k] > own_rank) others += o_ub[k];
  others *= SLACK;
  TopK tk;
  tk.s = -1.0; tk.d = 0xFFFFFFFFu; tk.n = 0; tk.thr_s = 0.0; tk.thr_d = 0;
  double published = 0.0;
  const uint32_t end = it.begin + it.count;
  bool essential = true;  // wave-uniform
  WorkStats ws;
  uint32_t q_head = 0, q_n = 0;  // wave-uniform
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#ifdef PS_ITEM_TRACE
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  uint32_t n_trips = 0;
#endif

  // Second level + the sum in PLAN order (query.rs:33-89; one list per query term: always the `+` / assign
  // arm) + the top-K offer for the first `count` (<= 64) queued documents, one per lane.
  auto process = [&](const uint32_t count, const double theta) {
    const uint32_t at = (q_head + (uint32_t)lane) & (QCAP - 1u);
    bool ok = (uint32_t)lane < count;
    const uint32_t d = ok ? q_d[wave][at] : 0u;
    const double s_own = ok ? q_s[wave][at] : 0.0;
    double P = 0.0;
#pragma unroll
    for (int k = 0; k <= NO; ++k) {
      if ((uint32_t)k == own_pos && ok && s_own > 0.0) P += s_own;
      if (k < NO && (uint32_t)k + 1u < ne && !(PS_EXP & 2)) {
        const unsigned long long loc = ok ? q_loc[k < NO ? k : 0][wave][at] : ~0ull;
        double sk = 0.0;
        if (o_shift[k] & DENSE_FLAG) {
          sk = ok ? __longlong_as_double((long long)loc) : 0.0;
        } else {
          bool found = false;
          uint64_t pk = o_off[k];
          if (o_bm[k] != 0xFFFFFFFFu) {
            found = ok && loc != ~0ull;
            if (found) pk = loc;
          } else {
            // a sparse list whose filter said "maybe": its table slot holds a handful of postings - up to 4 doc ids
            // per step, all requested at once
            const uint32_t* docs = p.doc + o_off[k];
            bool open = ok && loc != ~0ull;
            uint32_t lo = 0, hi = 0;
            if (open) {
              const uint32_t slot = (d >> p.t_log2) >> (o_shift[k] & 0xFFu);
              lo = p.table[o_tbl[k] + slot];
              hi = p.table[o_tbl[k] + slot + 1];
            }
            if (!PS_REQ_TRACE) ws.probe += 2u * cnt(open);
            open = open && lo < hi;
            while (__any(open)) {
              uint32_t v[4];
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const bool rd = open && lo + t < hi;
                v[t] = rd ? docs[lo + t] : 0xFFFFFFFFu;
                if (!PS_REQ_TRACE) ws.probe += cnt(rd);
              }
              if (open) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                  if (v[t] == d) { found = true; pk = o_off[k] + lo + t; }
                // ascending doc ids: past the document, or past the slot, the search is over
                open = !found && v[3] < d && lo + 4 < hi;
                lo += 4;
              }
            }
          }
          if (!PS_REQ_TRACE) ws.hit += cnt(found);
          if (__any(found)) {
            double t[FA];
#pragma unroll
            for (int x = 0; x < FA; ++x) t[x] = 0.0;
            if (found) plane_load<F_>(p, pk, t);
            double acc = 0.0;
#pragma unroll
            for (int x = 0; x < FA; ++x)
              if ((uint32_t)x < (F_ ? (uint32_t)F_ : p.F)) acc += (t[x] * p.boost[x]) * o_eb[k];
            sk = found ? acc : 0.0;
          }
        }
        if (ok && sk > 0.0) {
          if (o_rank[k] < own_rank) ok = false;  // evaluated from its highest-bound list only
          P += sk;
        }
      }
    }
    const bool offer = ok && P >= theta;
    if (!PS_REQ_TRACE) ws.offer += cnt(offer);
    if (!(PS_EXP & 1) && __any(offer)) topk_offer(tk, p.K, lane, ok, P, d, theta);
    q_head = (q_head + count) & (QCAP - 1u);
    q_n -= count;
    if (tk.n == p.K && tk.thr_s > published && tk.thr_s > theta) {
      // this wave's K-th best so far: the final K-th best of the query can only be higher
      published = tk.thr_s;
      if (lane == 0) atomicMax(&p.gthr[q], (unsigned long long)__double_as_longlong(tk.thr_s));
    }
  };

  double theta = 0.0;
  for (uint32_t i0 = it.begin; i0 < end && essential; i0 += WAVE * U) {
#ifdef PS_ITEM_TRACE
    ++n_trips;
#endif
    const unsigned long long tbits = __hip_atomic_load(&p.gthr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t d[U];
    uint64_t pi[U];
    double tw[U][FA];
    bool inr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = i0 + u * WAVE + lane;
      inr[u] = i < end;
      pi[u] = own_off + (i < end ? i : end - 1);
      d[u] = p.doc[pi[u]];
#pragma unroll
      for (int x = 0; x < FA; ++x) tw[u][x] = 0.0;
      plane_load<F_>(p, pi[u], tw[u]);
    }
    theta = __hiloint2double(__builtin_amdgcn_readfirstlane((int)(tbits >> 32)), __builtin_amdgcn_readfirstlane((int)(uint32_t)tbits));
    essential = !(skip_thr < theta);  // false: the whole list has become non-essential
    const uint32_t n_in = min(end - i0, (uint32_t)(WAVE * U));
    if (!essential) {  // (its doc ids and plane values were requested with the threshold: booked, then out)
      if (WC && !PS_REQ_TRACE) ws.probe += n_in * (1u + 2u * (F_ ? (uint32_t)F_ : p.F));
      break;
    }
    // ---- own scores; first bound test: everything the other entries could add, at most - below theta the
    // document is out before anything is asked of another list ----
    if (p.alive != nullptr) {  // delta removals
      uint32_t aw[U];  // (every d[u] is a real doc id: all words requested together, no branch per posting)
#pragma unroll
      for (int u = 0; u < U; ++u) aw[u] = p.alive[d[u] >> 5];
#pragma unroll
      for (int u = 0; u < U; ++u) inr[u] = inr[u] & (bool)((aw[u] >> (d[u] & 31u)) & 1u);
    }
    double s_own[U];
    scores_from_plane<F_, U>(p, tw, inr, own_eb, s_own);
    bool rch[U];
    if (WC) ws.scanned += n_in;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rch[u] = inr[u] && (s_own[u] + others >= theta);
      const uint32_t nr = cnt(rch[u]);  // every document that passed asks every other list's first level
      if (!PS_REQ_TRACE) ws.reached += nr;
    }
    // ---- first level of the other lists for the documents that passed: dense-row value, {bits, rank} bitmap cell, or
    // the sparse list's Bloom-filter word - every list at once, all loads in flight together.  (Asking the highest-bound lower-ranked
    // list first and the rest only for what it leaves alive halves the row lookups and was measured slower twice, rounds 4 and 5:
    // the extra dependency level costs more than the requests it saves) ----
    uint2 fl[NO][U];
#pragma unroll
    for (int k = 0; k < NO; ++k)
#pragma unroll
      for (int u = 0; u < U; ++u) fl[k][u] = make_uint2(0u, 0u);
    auto first_level = [&](const int k, const bool (&on)[U]) {
      if ((uint32_t)k + 1u < ne && !(PS_EXP & 4)) {
        if (o_shift[k] & DENSE_FLAG) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (on[u]) fl[k][u] = *reinterpret_cast<const uint2*>(p.rows + (uint64_t)o_row[k] * p.row_stride + d[u]);
            ws.row += cnt(on[u]);
            if (PS_REQ_TRACE) { ws.probe += distinct_lines(on[u], d[u] >> 4, lane); ws.hit += distinct_lines(on[u], d[u] >> 6, lane); }
          }
        } else if (o_bm[k] != 0xFFFFFFFFu) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (on[u]) fl[k][u] = *reinterpret_cast<const uint2*>(p.bits + (uint64_t)o_bm[k] + 2 * (uint64_t)(d[u] >> 5));
            ws.cell += cnt(on[u]);
            if (PS_REQ_TRACE) ws.offer += distinct_lines(on[u], d[u] >> 9, lane);
          }
        } else if (o_bloom[k] != NO_BLOOM) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            uint64_t wi;
            unsigned long long mk;
            bloom_probe(d[u], o_bloom[k], wi, mk);
            const unsigned long long w = on[u] ? p.bloom[wi] : 0ull;
            fl[k][u].x = (on[u] && (w & mk) == mk) ? 1u : 0u;  // maybe
            ws.cell += cnt(on[u]);
            if (PS_REQ_TRACE) ws.reached += cnt(on[u]);
          }
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) fl[k][u].x = on[u] ? 1u : 0u;  // no filter: ask the table
        }
      }
    };
#pragma unroll
    for (int k = 0; k < NO; ++k) first_level(k, rch);
    // ---- what the first level already tells: exact row values, bitmap membership, filter misses ----
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bool alive = rch[u];
      double bound = s_own[u];
      unsigned long long loc[NO];
#pragma unroll
      for (int k = 0; k < NO; ++k) {
        loc[k] = ~0ull;
        if ((uint32_t)k + 1u < ne) {
          const bool dense = (o_shift[k] & DENSE_FLAG) != 0, bitmap = !dense && o_bm[k] != 0xFFFFFFFFu;
          double c;
          bool hit;
          if (dense) {
            c = __hiloint2double((int)fl[k][u].y, (int)fl[k][u].x);
            hit = c > 0.0;
            loc[k] = (unsigned long long)fl[k][u].x | ((unsigned long long)fl[k][u].y << 32);
          } else if (bitmap) {
            const uint32_t bit = d[u] & 31u;
            hit = (fl[k][u].x >> bit) & 1u;
            c = hit ? o_ub[k] : 0.0;
            if (hit) loc[k] = o_off[k] + fl[k][u].y + (uint32_t)__popc(fl[k][u].x & ((1u << bit) - 1u));
          } else {
            hit = fl[k][u].x != 0u;  // the filter (or its absence) says maybe
            c = hit ? o_ub[k] : 0.0;
            if (hit) loc[k] = 0ull;
          }
          if (o_rank[k] > own_rank) bound += c;  // (a higher-ranked list adds nothing to a document evaluated here)
          // (a document is evaluated from its highest-bound list only: known here for rows and bitmaps)
          if ((dense || bitmap) && hit && o_rank[k] < own_rank) alive = false;
        }
      }
      alive = alive && (bound * SLACK >= theta);
      // ---- survivors wait in the queue until 64 are together ----
      const unsigned long long m = __ballot(alive);
      if (m) {
        if (alive) {
          const uint32_t at = (q_head + q_n + (uint32_t)__popcll(m & lt)) & (QCAP - 1u);
          q_d[wave][at] = d[u];
          q_s[wave][at] = s_own[u];
#pragma unroll
          for (int k = 0; k < NO; ++k)
            if ((uint32_t)k + 1u < ne) q_loc[k][wave][at] = loc[k];
        }
        q_n += (uint32_t)__popcll(m);
        if (q_n >= (uint32_t)WAVE) process((uint32_t)WAVE, theta);
      }
    }
  }
  while (q_n) process(min(q_n, (uint32_t)WAVE), theta);
  if ((uint32_t)lane < p.K) {
    const uint64_t o = (uint64_t)it.slot * p.K + lane;
    const bool ok = (uint32_t)lane < tk.n;
    p.cand_score[o] = ok ? tk.s : 0.0;
    p.cand_doc[o] = ok ? tk.d : 0xFFFFFFFFu;
    if (lane == 0) p.cand_cnt[it.slot] = tk.n;
  }
#ifdef PS_ITEM_TRACE
  if (p.item_trace != nullptr && lane == 0) {
    unsigned long long* tr = p.item_trace + (size_t)id * 4;
    tr[0] = t_start; tr[1] = __builtin_amdgcn_s_memrealtime(); tr[2] = (unsigned long long)n_trips | ((unsigned long long)own_rank << 32);
    tr[3] = ws.scanned | ((unsigned long long)ws.reached << 32);
  }
#endif
  if (WC && lane == 0) {
    unsigned long long* w = p.wstats + (size_t)(blockIdx.x & (WS_SLOTS - 1u)) * WS_WORDS;
    atomicAdd(&w[WS_ITEMS_RUN], 1ull);
    if (ws.scanned) atomicAdd(&w[WS_SCANNED], (unsigned long long)ws.scanned);
    if (ws.reached) atomicAdd(&w[WS_REACHED], (unsigned long long)ws.reached);
    if (ws.row) atomicAdd(&w[WS_ROW], (unsigned long long)ws.row);
    if (ws.cell) atomicAdd(&w[WS_CELL], (unsigned long long)ws.cell);
    if (ws.probe) atomicAdd(&w[WS_PROBE], (unsigned long long)ws.probe);
    if (ws.hit) atomicAdd(&w[WS_HIT], (unsigned long long)ws.hit);
    if (ws.offer) atomicAdd(&w[WS_OFFER], (unsigned long long)ws.offer);
  }
}

// K3d: merge of the items' candidate lists of a query -> final top-K, doc id -> key.  A document is
// evaluated by exactly one item, so the lists are disjoint.  Leaves the control words zeroed.
__global__ __launch_bounds__(WAVE * MERGE_WAVES) void k_merge_items(const KParams p) {
  __shared__ double sh_s[MERGE_WAVES][WAVE];
  __shared__ uint32_t sh_d[MERGE_WAVES][WAVE];
  __shared__ uint32_t sh_n[MERGE_WAVES];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t q = blockIdx.x;
  TopK tk;
  tk.s = -1.0; tk.d = 0xFFFFFFFFu; tk.n = 0; tk.thr_s = 0.0; tk.thr_d = 0;
  const uint32_t K = p.K;
  const double gt = __longlong_as_double((long long)p.gthr[q]);
  const uint32_t s0 = p.qslot[q], s1 = s0 + p.qslot_n[q];
  const uint32_t n_waves = blockDim.x >> 6;
  constexpr int U = 4;
  for (uint32_t sb = s0 + (uint32_t)wave * U; sb < s1; sb += n_waves * U) {
    double v[U];
    uint32_t d[U];
    bool has[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t sl = sb + u;
      const uint32_t cnt = sl < s1 ? p.cand_cnt[sl] : 0u;
      has[u] = (uint32_t)lane < cnt;
      const uint64_t o = (uint64_t)sl * K + lane;
      v[u] = has[u] ? p.cand_score[o] : 0.0;
      d[u] = has[u] ? p.cand_doc[o] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (__any(has[u] && v[u] >= gt)) topk_offer(tk, K, lane, has[u], v[u], d[u], gt);
  }
  sh_s[wave][lane] = tk.s;
  sh_d[wave][lane] = tk.d;
  if (lane == 0) sh_n[wave] = tk.n;
  __syncthreads();
  if (wave != 0) return;
  for (uint32_t w = 1; w < n_waves; ++w) {
    const bool has = (uint32_t)lane < sh_n[w];
    topk_offer(tk, K, lane, has, sh_s[w][lane], sh_d[w][lane]);
  }
  const uint32_t row = p.out_row != nullptr ? p.out_row[q] : q;
  if ((uint32_t)lane < K) {
    const bool ok = (uint32_t)lane < tk.n;
    const uint64_t o = (uint64_t)row * K + lane;
    p.out_keys[o] = ok ? p.keys[tk.d] : ~0ull;
    p.out_scores[o] = ok ? tk.s : 0.0;
  }
  if (lane == 0) {
    p.out_counts[row] = tk.n;
    p.gthr[q] = 0ull;
    if (p.gtie != nullptr)
      for (uint32_t l = 0; l < 3u; ++l) p.gtie[(size_t)l * p.z_tstride + q] = 0ull;
    if (q == 0) *p.work_counter = 0u;
  }
  // the preparation's control words (bucket counts, row uses, ...) are consumed: clean for the next batch
  if (q == 0 && p.prep_ctl != nullptr)
    for (uint32_t i = (uint32_t)lane; i < p.prep_ctl_words; i += WAVE) p.prep_ctl[i] = 0u;
}

// ------------------------------------------------------------------------------------------
// N2: device-side query planner - tokenise, term lookup, prefix expansion, before_each
//     (query.rs:29-60,109-147; index.rs:300-337; bm25.rs:35-58) for a whole batch, BM25.
//
// The frozen trie lives in HBM as it does on the host (nodes in DFS pre-order with children
// newest-first, so expand_term(prefix) is the contiguous ordinal range [term_begin, term_end) of the
// prefix's node; a node's children sorted by char for binary search).  `ln` never runs on the device:
// idf depends on the term only and is tabulated per term by the host (same libm call as the host
// planner), expansion_boost depends on the byte-length difference only and is tabulated per
// difference.  One thread plans one query; pass 1 counts, a scan places, pass 2 writes - the entries
// come out exactly as Snapshot::plan_query writes them (tests compare the bytes).
// ------------------------------------------------------------------------------------------
struct DevTrie {
  const uint4* fnodes;       // {child_begin, child_count, term_begin, term_end}
  const uint32_t* fchar;
  const uint32_t* fchild;
  const uint64_t* term_df;   // live df_raw per term ordinal
  const uint32_t* term_meta; // [4 per term] byte_len, first_layer, n_layers, fnode
  const uint32_t* term_delta;// delta_head per term
  const double* term_idf;
  const uint4* layer_a;      // {post_off lo, post_off hi, len, tbl_off}
  const uint4* layer_b;      // {shift, bm_off, next, -}
  const double* eb_table;    // [EB_TABLE] expansion_boost by (len_expanded - len_query)
  uint32_t eb_n;
  const uint4* fbits;        // [2 per node] 256-bit set of the node's child characters below U+0100 (null: binary search only)
};

struct PlanTotals {  // written by k_plan_scan
  uint32_t n_entries, max_entries, max_qterms, multi;
  unsigned long long postings;
  unsigned long long n_items;  // K1d work items of the batch under the chunking rule (chunk_min, split_div)
  unsigned long long n_items_big;  // ... of them, the items of the queries k_daat_small does not take (PLAN_BIG)
};

// q_multi / PlanTotals::multi bits.  PLAN_Z_NOT_SIMPLE: the query is not "simple" in classify_zero_to_one's sense (a term with
// several version layers, or several expansions of a query term AND a term reached by two query terms) - decided
// conservatively (two query terms whose expansion ranges intersect count as sharing a term even if the shared terms are dead).
// PLAN_BIG: the query has more than DAAT_SMALL_MAX lists or several lists under one query term - its items go to k_daat, the
// others' to k_daat_small (a BM25 batch that holds both kinds is split between the two kernels: k_prep_query applies the same rule).
constexpr uint32_t PLAN_MULTI = 1u, PLAN_Z_NOT_SIMPLE = 2u, PLAN_BIG = 4u;
constexpr uint32_t PLAN_SMALL_MAX = 4u;  // (== DAAT_SMALL_MAX, defined with k_daat_small)

__device__ __forceinline__ uint32_t utf8_next(const char* s, uint32_t& i, const uint32_t end) {
  const unsigned char c = (unsigned char)s[i++];
  if (c < 0x80) return c;
  const int extra = (c >> 5) == 0x6 ? 1 : (c >> 4) == 0xE ? 2 : 3;
  uint32_t cp = extra == 1 ? (c & 0x1Fu) : extra == 2 ? (c & 0x0Fu) : (c & 0x07u);
  for (int k = 0; k < extra && i < end; ++k) cp = (cp << 6) | ((unsigned char)s[i++] & 0x3Fu);
  return cp;
}

// find_inverted_index_node (index.rs:300-337) on the frozen trie: -1 if the path does not exist
__device__ __forceinline__ int64_t dev_find_node(const DevTrie& t, const char* s, uint32_t b, const uint32_t e) {
  uint32_t n = 0;
  while (b < e) {
    const uint32_t ch = utf8_next(s, b, e);
    const uint4 fn = t.fnodes[n];
    if (t.fbits != nullptr && ch < 256u) {
      // children are sorted by character: the child's position is the number of set bits below it - three
      // independent loads and one dependent one per level instead of a binary search's chain
      const uint4 lo = t.fbits[2 * (size_t)n], hi = t.fbits[2 * (size_t)n + 1];
      const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      const uint32_t wi = ch >> 5, bit = ch & 31u;
      uint32_t below = 0, mine = 0;
#pragma unroll
      for (uint32_t k = 0; k < 8; ++k) {
        below += k < wi ? (uint32_t)__popc(w[k]) : 0u;
        mine = k == wi ? w[k] : mine;
      }
      if (!((mine >> bit) & 1u)) return -1;
      n = t.fchild[fn.x + below + (uint32_t)__popc(mine & ((1u << bit) - 1u))];
      continue;
    }
    uint32_t lo = 0, hi = fn.y;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (t.fchar[fn.x + mid] < ch) lo = mid + 1; else hi = mid;
    }
    if (lo >= fn.y || t.fchar[fn.x + lo] != ch) return -1;
    n = t.fchild[fn.x + lo];
  }
  return (int64_t)n;
}

// One non-empty query token whose trie node is `fn` (-1: no such path): its expansions in expand_term order
// (query.rs:130-147), one entry per (expanded term, version / delta layer).  FILL writes the entries at
// entries[w...]; both passes return the counts.
template <bool FILL>
__device__ __forceinline__ void plan_token(const DevTrie& t, const int64_t fn, const uint32_t tok_bytes, const uint32_t qord, const uint32_t qi,
                                           const uint32_t chunk_min, const uint32_t split_div, ps_plan_entry* entries, uint32_t w,
                                           uint32_t& here, unsigned long long& postings, uint32_t& items, const uint32_t zmode = 0u,
                                           uint32_t* layered = nullptr) {
  here = 0; postings = 0; items = 0;
  if (layered) *layered = 0u;
  if (fn < 0) return;
  const uint4 node = t.fnodes[fn];
  for (uint32_t o = node.z; o < node.w; ++o) {  // == expand_term order (query.rs:130-147)
    const uint64_t df = t.term_df[o];
    const uint32_t byte_len = t.term_meta[4 * o], first_layer = t.term_meta[4 * o + 1], n_layers = t.term_meta[4 * o + 2];
    const uint32_t delta_head = t.term_delta[o];
    if (df == 0 || (n_layers == 0 && delta_head == 0xFFFFFFFFu)) continue;  // query.rs:47-48
    uint32_t l = 0, li = n_layers ? first_layer : delta_head;
    while (li != 0xFFFFFFFFu) {
      const uint4 la = t.layer_a[li], lb = t.layer_b[li];
      if (FILL) {
        ps_plan_entry e;
        e.post_off = (uint64_t)la.x | ((uint64_t)la.y << 32);
        e.len = la.z;
        e.tbl_off = la.w;
        e.shift = lb.x | (l << 8);
        e.qterm = qord;
        e.idf = t.term_idf[o];
        // bm25.rs:45-53: 1 for the query term itself, else ln(1 + 1/((1 + len_exp) - len_q)), tabulated
        const uint32_t delta = byte_len - tok_bytes;
        e.boost = (t.term_meta[4 * o + 3] == (uint32_t)fn) ? 1.0 : t.eb_table[delta < t.eb_n ? delta : 0];
        e.node = li;
        if (zmode) {
          // ScoreByTerm::score (zero_to_one.rs:57-73): 1 - |len_expanded - len_query| / len_expanded, the host planner's
          // expression; records pool per expanded term: `node` names the term (its trie node), as K1dz's arrangement needs
          const double el = (double)byte_len, tl = (double)tok_bytes;
          e.boost = 1.0 - fabs(el - tl) / el;
          e.idf = 0.0;
          e.node = t.term_meta[4 * o + 3];
        }
        e.qterm_index = qi;
        e.bm_off = lb.y;
        e.layer = li;
        entries[w++] = e;
      } else {  // K1d work items of this list (the rule of k_prep_batch)
        uint32_t c = ((la.z + split_div - 1) / split_div + 255u) & ~255u;
        c = c > chunk_min ? c : chunk_min;
        items += (la.z + c - 1) / c;
      }
      postings += la.z;
      ++here;
      if (l && layered) *layered = 1u;  // a second version / delta layer of one term
      ++l;
      // base layers are contiguous, then the delta chain
      if (l < n_layers) li = first_layer + l;
      else if (l == n_layers) li = delta_head;
      else li = lb.z;
    }
  }
}

// A whole query by one thread (queries of more than 64 tokens; k_plan's wave hands them to its lane 0).
template <bool FILL>
__device__ __noinline__ void plan_query_seq(const DevTrie& t, const char* s, const uint32_t qb, const uint32_t qe, const uint32_t q,
                                            const uint32_t* qbeg, ps_plan_entry* entries, uint32_t* q_cnt, uint32_t* q_terms_len,
                                            uint32_t* q_nterms, uint32_t* q_multi, unsigned long long* q_postings, uint32_t* q_items,
                                            const uint32_t chunk_min, const uint32_t split_div, const uint32_t zmode) {
  uint32_t n_tokens = 0, qord = 0, n_ent = 0, multi = PLAN_Z_NOT_SIMPLE, items = 0;  // (K1dz does not take these queries: not classified here)
  unsigned long long postings = 0;
  uint32_t w = FILL ? qbeg[q] : 0u;
  // s.split(' ') (lib.rs:42-44): k separators -> k + 1 tokens; empty ones are skipped but counted (query.rs:32-35)
  uint32_t tb = qb;
  for (uint32_t i = qb; i <= qe; ++i) {
    if (i != qe && s[i] != ' ') continue;
    const uint32_t te = i;
    const uint32_t qi = n_tokens++;
    if (te > tb) {
      uint32_t here, it;
      unsigned long long po;
      plan_token<FILL>(t, dev_find_node(t, s, tb, te), te - tb, qord, qi, chunk_min, split_div, entries, w, here, po, it, zmode);
      w += here; postings += po; items += it;
      if (here > 1) multi |= PLAN_MULTI;
      n_ent += here;
      ++qord;
    }
    tb = i + 1;
  }
  if (!FILL) {
    if (n_ent > PLAN_SMALL_MAX || (multi & PLAN_MULTI)) multi |= PLAN_BIG;
    q_cnt[q] = n_ent;
    q_terms_len[q] = n_tokens;
    q_nterms[q] = qord;
    q_multi[q] = multi;
    q_postings[q] = postings;
    q_items[q] = items;
  }
}

// One WAVE per query: the lanes find the token boundaries together (a ballot of the separators per 64
// bytes of text), then lane i plans token i - the trie walks of a query's terms, which are chains of
// dependent loads, run side by side instead of one after the other.  The count pass leaves every token's
// trie node in `tok_node` ([B][64]; -2 = empty token), so the fill pass walks nothing.
constexpr int PLAN_WAVES = 1;  // queries per workgroup (one-wave workgroups slip into the wave slots a running k_daat launch frees; fat ones wait)
template <bool FILL>
__device__ __forceinline__ void plan_wave(const DevTrie& t, const char* text, const uint64_t* offsets, const uint32_t B,
                                          const uint32_t* qbeg, ps_plan_entry* entries, uint32_t* q_cnt,
                                          uint32_t* q_terms_len, uint32_t* q_nterms, uint32_t* q_multi,
                                          unsigned long long* q_postings, uint32_t* qorder, uint32_t* q_items,
                                          const uint32_t chunk_min, const uint32_t split_div, int32_t* tok_node, const uint32_t zmode) {
  __shared__ uint32_t sh_tb[PLAN_WAVES][WAVE], sh_te[PLAN_WAVES][WAVE];
  const uint32_t wv = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  const uint32_t q = blockIdx.x * PLAN_WAVES + wv;
  if (q >= B) return;
  const uint32_t qb = (uint32_t)offsets[q], qe = (uint32_t)offsets[q + 1];
  const char* s = text;
  // token boundaries: every ' ' ends a token and starts the next (s.split(' '), lib.rs:42-44)
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  uint32_t n_tokens = 0, start = qb;  // wave-uniform
  bool overflow = false;
  for (uint32_t pos = qb; pos < qe; pos += WAVE) {
    const bool in = pos + lane < qe;
    const bool sp = in && s[pos + lane] == ' ';
    const unsigned long long m = __ballot(sp);
    if (sp) {
      const unsigned long long before = m & lt;
      const uint32_t idx = n_tokens + (uint32_t)__popcll(before);
      const uint32_t tb = before ? pos + (63u - (uint32_t)__clzll(before)) + 1u : start;
      if (idx < (uint32_t)WAVE) { sh_tb[wv][idx] = tb; sh_te[wv][idx] = pos + lane; }
    }
    if (m) {
      n_tokens += (uint32_t)__popcll(m);
      start = pos + (63u - (uint32_t)__clzll(m)) + 1u;
    }
  }
  if (n_tokens < (uint32_t)WAVE) {
    if (lane == 0) { sh_tb[wv][n_tokens] = start; sh_te[wv][n_tokens] = qe; }
  } else {
    overflow = true;
  }
  ++n_tokens;  // the last token (k separators -> k + 1 tokens)
  if (overflow) {  // more than 64 tokens: one lane walks the query
    if (lane == 0) {
      plan_query_seq<FILL>(t, s, qb, qe, q, qbeg, entries, q_cnt, q_terms_len, q_nterms, q_multi, q_postings, q_items, chunk_min, split_div, zmode);
      if (FILL) qorder[q] = q;
    }
    return;
  }
  // (the wave's LDS writes above are visible to its own lanes in program order)
  const bool mine = lane < n_tokens;
  const uint32_t tb = mine ? sh_tb[wv][lane] : 0u, te = mine ? sh_te[wv][lane] : 0u;
  const bool nonempty = mine && te > tb;
  const unsigned long long ne_mask = __ballot(nonempty);
  const uint32_t qord = (uint32_t)__popcll(ne_mask & lt);  // ordinal among the non-empty tokens (query.rs:33-37)
  int64_t fn = -1;
  if (nonempty) {
    if (FILL) fn = tok_node[(size_t)q * WAVE + lane];
    else { fn = dev_find_node(t, s, tb, te); tok_node[(size_t)q * WAVE + lane] = (int32_t)fn; }
  }
  uint32_t here = 0, items = 0;
  unsigned long long postings = 0;
  if (!FILL) {
    uint32_t layered = 0;
    if (nonempty) plan_token<false>(t, fn, te - tb, qord, lane, chunk_min, split_div, nullptr, 0u, here, postings, items, 0u, &layered);
    // zero_to_one's K1dz takes "simple" queries only (PLAN_Z_NOT_SIMPLE): no term with several layers; and, if any query
    // term has several expansions, no term reached by two query terms - the expansions of a node are a contiguous range
    // of term ordinals, so two query terms can share a term only where their ranges intersect
    bool not_simple = layered != 0u;
    if (__any(here > 1)) {
      uint32_t rz = 0, rw = 0;
      if (here) { const uint4 nd = t.fnodes[fn]; rz = nd.z; rw = nd.w; }
      for (uint32_t j = 0; j < n_tokens; ++j) {
        const uint32_t oz = (uint32_t)__shfl((int)rz, (int)j), ow = (uint32_t)__shfl((int)rw, (int)j);
        if (j != lane && rz < rw && oz < ow && rz < ow && oz < rw) not_simple = true;
      }
    }
    // per-query totals
    uint32_t n_ent = here, multi = (here > 1 ? PLAN_MULTI : 0u) | (not_simple ? PLAN_Z_NOT_SIMPLE : 0u), it = items;
    unsigned long long po = postings;
    for (int o = 32; o > 0; o >>= 1) {
      n_ent += __shfl_xor(n_ent, o); multi |= __shfl_xor(multi, o); it += __shfl_xor(it, o); po += __shfl_xor(po, o);
    }
    if (n_ent > PLAN_SMALL_MAX || (multi & PLAN_MULTI)) multi |= PLAN_BIG;
    if (lane == 0) {
      q_cnt[q] = n_ent;
      q_terms_len[q] = n_tokens;
      q_nterms[q] = (uint32_t)__popcll(ne_mask);
      q_multi[q] = multi;
      q_postings[q] = po;
      q_items[q] = it;
    }
  } else {
    // entries of token i go behind those of the tokens before it: the counts again (cheap: no trie walk),
    // an exclusive scan over the lanes, then the writes
    uint32_t cnt = 0, dummy_i;
    unsigned long long dummy_p;
    if (nonempty) plan_token<false>(t, fn, te - tb, qord, lane, chunk_min, split_div, nullptr, 0u, cnt, dummy_p, dummy_i);
    uint32_t inc = cnt;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if ((int)lane >= o) inc += v; }
    const uint32_t w = qbeg[q] + inc - cnt;
    if (nonempty && cnt) plan_token<true>(t, fn, te - tb, qord, lane, chunk_min, split_div, entries, w, here, postings, items, zmode);
    if (lane == 0) qorder[q] = q;
  }
}

// one wave: exclusive scan of the per-query entry counts + the batch totals (a lane takes B / 64 consecutive
// queries; one shuffle scan; a single wave finds a slot at once even while a k_daat launch owns the chip).
// (Folding it into the count pass behind a last-wave ticket was tried: 1024 fences + atomics on one word made the
// count pass 224 us instead of 30-160.)
__global__ __launch_bounds__(WAVE) void k_plan_scan(const uint32_t* q_cnt, const uint32_t* q_nterms, const uint32_t* q_multi,
                                                     const unsigned long long* q_postings, const uint32_t* q_items, const uint32_t B,
                                                     uint32_t* qbeg, PlanTotals* tot) {
  const uint32_t lane = threadIdx.x % WAVE, per = (B + WAVE - 1) / WAVE;
  const uint32_t b = min(B, lane * per), e = min(B, b + per);
  uint32_t sum = 0, me = 0, mt = 0, mm = 0;
  unsigned long long ps = 0, it = 0, itb = 0;
  for (uint32_t i = b; i < e; ++i) {
    sum += q_cnt[i]; me = max(me, q_cnt[i]); mt = max(mt, q_nterms[i]); mm |= q_multi[i]; ps += q_postings[i]; it += q_items[i];
    if (q_multi[i] & PLAN_BIG) itb += q_items[i];
  }
  uint32_t inc = sum;
  for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if ((int)lane >= o) inc += v; }
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
  for (int o = 32; o > 0; o >>= 1) {
    me = max(me, (uint32_t)__shfl_xor((int)me, o)); mt = max(mt, (uint32_t)__shfl_xor((int)mt, o)); mm |= (uint32_t)__shfl_xor((int)mm, o);
    ps += __shfl_xor(ps, o); it += __shfl_xor(it, o); itb += __shfl_xor(itb, o);
  }
  uint32_t run = inc - sum;  // exclusive prefix of this lane's first query
  for (uint32_t i = b; i < e; ++i) { qbeg[i] = run; run += q_cnt[i]; }
  if (lane == 0) {
    qbeg[B] = total;
    tot->max_entries = me; tot->max_qterms = mt; tot->multi = mm; tot->postings = ps; tot->n_items = it; tot->n_items_big = itb;
    __threadfence_system();
    tot->n_entries = total;
  }
}

template <bool FILL>
__global__ __launch_bounds__(WAVE * PLAN_WAVES) void k_plan(const DevTrie t, const char* text, const uint64_t* offsets, const uint32_t B,
                                                          const uint32_t* qbeg, ps_plan_entry* entries, uint32_t* q_cnt,
                                                          uint32_t* q_terms_len, uint32_t* q_nterms, uint32_t* q_multi,
                                                          unsigned long long* q_postings, uint32_t* qorder, uint32_t* q_items,
                                                          const uint32_t chunk_min, const uint32_t split_div, int32_t* tok_node,
                                                          const uint32_t zmode) {
  plan_wave<FILL>(t, text, offsets, B, qbeg, entries, q_cnt, q_terms_len, q_nterms, q_multi, q_postings, qorder, q_items, chunk_min,
                  split_div, tok_node, zmode);
}

// Plan upload without the copy engine: the staged batch is read from the pinned, device-mapped
// slot with coalesced 16-byte loads.  (An SDMA copy between two kernels costs a 20-30 us hand-over
// per batch; this is a few microseconds for the ~150 KB of a 1024-query plan.)
__global__ __launch_bounds__(256) void k_upload(const uint4* __restrict__ src, uint4* __restrict__ dst, const size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// The packed {tf, field length} words of postings [begin, end) from the exact planes (engine creation,
// and the appended range after a delta).
__global__ __launch_bounds__(256) void k_pack_tfl(const uint32_t* __restrict__ tf, const uint32_t* __restrict__ fl,
                                                  uint32_t* __restrict__ tfl, const uint64_t P, const uint32_t F,
                                                  const uint64_t begin, const uint64_t end) {
  const uint64_t n = (end - begin) * F;
  for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = begin + k / F;
    const uint32_t x = (uint32_t)(k % F);
    tfl[i * F + x] = tfl_pack(tf[(uint64_t)x * P + i], fl[(uint64_t)x * P + i]);
  }
}

// Full-result mode: the first (out_off[q+1] - out_off[q]) sorted results of run q -> {key, score}.
// grid (chunks, B): a run of 10^6 results is not one workgroup's job.
__global__ __launch_bounds__(256) void k_pack_results(const uint32_t* doc, const double* score, const uint64_t* run_off,
                                                      const uint64_t* out_off, const uint64_t* keys, ps_result* out) {
  const uint32_t q = blockIdx.y;
  const uint64_t src = run_off[q], dst = out_off[q], n = out_off[q + 1] - out_off[q];
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    out[dst + i] = ps_result{keys[doc[src + i]], score[src + i]};
}

}  // namespace ps
