// ps_prep_kernels.hpp — device-side preparation of a K1d (k_daat) batch: everything between "the plan
// exists in HBM" (uploaded by the host planner, or written by the device planner k_plan) and the
// scoring launch.  Nothing here touches the host:
//
//   k_list_bounds  per (k1, b, avg[, boosts]): exact per-list maxima of the saturated term frequency
//                  per field (M) and of the boosted per-posting sum (J), by the kernels' own f64
//                  expression (bm25.rs:78-86)
//   k_prep_query   a thread per query: upper bound of every entry, the query's processing order (rank), what
//                  the other entries can add (`others`), the whole-list skip thresholds, per-query-term data of
//                  multi-expansion queries, chunking, candidate slots, item-bucket totals, dense-row uses
//                  (wave-aggregated atomics)
//                  ; its last wave closes the counters: bucket starts (rank-major: every query's highest-bound
//                  list first, longest lists first), the batch's item count, which dense rows are read / scored
//   k_prep_items   a thread per list: its place in the item order, its dense-row flag, its work items
//
// What used to be Engine::plan_daat + select_dense_rows on the host (≈0.2 ms per 1024 queries on the
// critical path of a 0.5 ms step).  Orders within a bucket and the placement of candidate slots come
// from atomics: they are scheduling decisions, not results (every document is evaluated exactly once
// whatever the order, and the merge is order-independent).
#pragma once
#include "ps_kernels.hpp"

namespace ps {

constexpr uint32_t PREP_CLASSES = 64;                 // length classes of the rank-0 and of the rank-1 lists (log2 with one fractional bit)
constexpr uint32_t PREP_RANKS = 8;                    // ranks 2..7 get a bucket each, everything above shares the last
constexpr uint32_t PREP_SAMPLE_BUCKETS = 4;           // K1d's sample phase: ranks 0, 1, 2, 3+ (in front of everything else)
constexpr uint32_t PREP_SET_BUCKETS = PREP_SAMPLE_BUCKETS + 2 * PREP_CLASSES + PREP_RANKS;  // one set of K1d item buckets
// K1d: two sets - the items of the queries k_daat_small takes, then those of the others (a batch that holds both kinds is scored by
// two launches over the two parts of one item array; the boundary is bucket_start[PREP_SET_BUCKETS]); K1dz: 4 per phase + one set
constexpr uint32_t PREP_BUCKETS = 2 * PREP_SET_BUCKETS;
constexpr uint32_t PREP_MAX_ROWS = 64;                // dense-row candidates per snapshot
constexpr uint32_t NO_CAND = 0xFFu;

struct PrepCtl {  // per-batch control words; zeroed again behind k_merge_items
  uint32_t total_slots;
  uint32_t n_items;
  uint32_t n_rows_build;   // rows k_dense_rows_dyn has to score for this batch
  uint32_t n_rows_used;    // rows the batch reads
  uint32_t ticket;         // k_prep_query workgroups that are through (the last one closes the counters)
  uint32_t _pad;
  uint32_t bucket_total[PREP_BUCKETS];
  uint32_t bucket_start[PREP_BUCKETS];
  uint32_t bucket_fill[PREP_BUCKETS];
  uint32_t row_use[PREP_MAX_ROWS];
  unsigned long long row_first[PREP_MAX_ROWS];  // ~(lowest plan-entry index that uses the candidate) through atomicMax; 0 = none
};

struct RowState {  // per dense-row candidate: what its slot currently holds (device-resident across batches)
  unsigned long long idf_bits, eb_bits;
  uint32_t valid;   // the slot holds the row scored with (idf, eb) under the current parameters
  uint32_t use_now; // this batch reads it
};

struct PrepParams {
  ps_plan_entry* plan;       // [ne] (k_prep_items sets the dense-row flag / slot)
  const uint32_t* qbeg;      // [B + 1]
  uint32_t B, ne, F;
  uint32_t multi;            // some query term has several entries (expansions / version layers)
  uint32_t chunk_min, split_div;
  uint32_t sample_tile;      // PS_DAAT_SAMPLE_DIV: tile index of the doc id D0 ~ N / div (0: off) - the chunks that lie entirely below D0 are
                             // the launch's first items, whatever their list's rank (a sample of the document space that publishes
                             // thresholds before the long lists start; K1dz's phase order, ps_z21_daat.hpp)
  const uint32_t* table;     // tile-offset tables (the sample phase finds a list's first posting at or above D0 there)
  uint32_t sample_small;     // the sample phase also for the queries k_daat_small takes in a split batch (PS_DAAT_SAMPLE_ALL)
  uint32_t split_kinds;      // the batch is split between k_daat_small and k_daat: the items of the PLAN_BIG queries take the second bucket set
  double boost[MAX_F];
  // per-list bounds (k_list_bounds)
  const double* bound_m;     // [n_layers][F]
  const double* bound_j;     // [n_layers] joint maximum under THESE boosts (F >= 3; null otherwise)
  const double* bound_h;     // [n_layers][PREP_NDIR] support values of the list's (tfn_0, tfn_1) points (F == 2; null otherwise)
  uint32_t h_lo;             // boosts = h_a * dir[h_lo] + h_b * dir[h_lo + 1], h_a, h_b >= 0 (host: boost_cone)
  double h_a, h_b;
  // outputs
  DEntry* dentry;            // [ne]
  uint32_t* rorder;          // [ne]
  DGroup* dgroup;            // [ne] (multi)
  uint8_t* gord;             // [ne] scratch: dense ordinal of the entry's query term within its query
  DItemGen* gen;             // [ne], indexed by ENTRY (item_at filled by k_prep_items)
  uint32_t* qslot;           // [B] first candidate slot of the query
  uint32_t* qslot_n;         // [B] its candidate slots
  DItem* items;
  uint32_t items_cap;
  PrepCtl* ctl;
  // dense rows
  const uint8_t* cand_of_layer;  // [n_layers] candidate ordinal or NO_CAND
  uint32_t n_cand, min_uses, rows_resident;
  const uint4* layer_a;          // {post_off lo, hi, len, tbl_off}
  RowState* row_state;           // [PREP_MAX_ROWS]
  RowDesc* row_desc;             // [PREP_MAX_ROWS] rows to score this batch
  unsigned long long* wstats;    // work counters (rows built / used)
  // threshold priming (k_list_kth): a query's final K-th best score is at least the K-th best posting score of any ONE of its lists
  const double* kth;             // [n_layers][F + 1][KTH_RANKS] (plane units; null: off)
  uint32_t kth_rank;             // index of the smallest stored rank >= K
  unsigned long long* gthr;      // [B] the batch's threshold words: start at theta0 instead of 0
  uint32_t prime_keep_items;     // debugging (PS_DAAT_PRIME=2): thresholds are primed but every list keeps its work items
  // consistency of the three copies of the "which kernel takes this query" rule (planner count pass / host, k_plan, k_prep_query):
  uint32_t host_items, host_items_big;  // what the launch grids are sized from
  uint32_t* fault;                      // engine-wide, host-mapped: {code, device items, device split, host items}; 0 = fine
};
constexpr uint32_t FAULT_ITEM_COUNTS = 1;  // the device counted more items (of one kind) than the host sized the launch for: items would be skipped

// ---- per-list bounds --------------------------------------------------------------------------
struct BoundUnit { uint32_t layer, begin, count; };  // a segment of one list

// Two fields: the joint maximum J(b) = max over a list's postings of b_0 * tfn_0 + b_1 * tfn_1 is the support function of the
// list's point set {(tfn_0, tfn_1)} in direction b.  Its values H_d in PREP_NDIR fixed directions w_d (angles 0 .. 90 degrees
// in equal steps; the ends are the per-field maxima, the middle one is fields_boost = [1, 1]) are computed once per (k1, b,
// averages); for any positive b = alpha * w_d + beta * w_(d+1) (the two directions around it, alpha, beta >= 0) every point
// satisfies b . v = alpha * (w_d . v) + beta * (w_(d+1) . v) <= alpha * H_d + beta * H_(d+1): a bound for EVERY boost vector
// without another pass over the postings - exact when b lies on a stored direction, at most a few percent loose between two
// (a list whose postings hold the term in one field or the other has a kink there).
constexpr int PREP_NDIR = 17;

// One wave per unit (a list, or a 16 Ki-posting segment of a long one).  M[l][x] = max over the list's
// postings of tfn_x; H[l][d] (two fields) = max of w_d . (tfn_0, tfn_1); J[l] (three fields and more, per boost vector) = max
// of sum_x boost_x * tfn_x; all through bm25_tfn, the expression the scoring kernels evaluate, so they bound the COMPUTED
// values.  Positive doubles order like their bit patterns: the segments of a list meet through 64-bit atomicMax.
// The pass that computes M also writes the score plane the K1d kernels read (ps_kernels.hpp, "score planes"): per (posting,
// field) the value tfn * idf, the list's idf from `layer_idf` (the planner's own number) - boost-free, so a new fields_boost
// rewrites nothing.
__global__ __launch_bounds__(256) void k_list_bounds(const KParams p, const BoundUnit* units, const uint32_t n_units,
                                                     const uint4* layer_a, unsigned long long* M, unsigned long long* J,
                                                     unsigned long long* H, const double2* dirs, const int with_m, double* plane,
                                                     const double* layer_idf) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int lane = threadIdx.x & (WAVE - 1);
  if (wave >= n_units) return;
  const BoundUnit u = units[wave];
  const uint4 la = layer_a[u.layer];
  const uint64_t off = ((uint64_t)la.x | ((uint64_t)la.y << 32)) + u.begin;
  const double idf = layer_idf[u.layer];
  double mj = 0.0, mm[MAX_F], hh[PREP_NDIR];
  for (uint32_t x = 0; x < p.F; ++x) mm[x] = 0.0;
#pragma unroll
  for (int d = 0; d < PREP_NDIR; ++d) hh[d] = 0.0;
  const bool with_h = H != nullptr && p.F == 2u;
  for (uint32_t i = lane; i < u.count; i += WAVE) {
    const uint64_t pi = off + i;
    double sum = 0.0, t01[2] = {0.0, 0.0};
    for (uint32_t x = 0; x < p.F; ++x) {
      const uint32_t w = p.tfl[pi * p.F + x];
      uint32_t tfu = w >> 24, flu = w & TFL_FL_ESC;
      if (tfu == 0) { if (plane) plane[pi * p.F + x] = 0.0; continue; }
      tfl_exact(p, x, pi, tfu, flu);
      const double t = bm25_tfn(p, x, tfu, flu);
      if (plane) plane[pi * p.F + x] = t * idf;  // bm25.rs:83-85, left to right; boost_x and expansion_boost are the reader's
      if (t > mm[x]) mm[x] = t;
      if (x < 2u) t01[x] = t;
      sum += p.boost[x] * t;
    }
    if (sum > mj) mj = sum;
    if (with_h) {
#pragma unroll
      for (int d = 0; d < PREP_NDIR; ++d) hh[d] = fmax(hh[d], dirs[d].x * t01[0] + dirs[d].y * t01[1]);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    mj = fmax(mj, __shfl_down(mj, o));
    for (uint32_t x = 0; x < p.F; ++x) mm[x] = fmax(mm[x], __shfl_down(mm[x], o));
  }
  if (with_h) {
#pragma unroll
    for (int d = 0; d < PREP_NDIR; ++d)
      for (int o = 32; o > 0; o >>= 1) hh[d] = fmax(hh[d], __shfl_down(hh[d], o));
  }
  if (lane == 0) {
    if (J != nullptr && mj > 0.0) atomicMax(&J[u.layer], (unsigned long long)__double_as_longlong(mj));
    if (with_m)
      for (uint32_t x = 0; x < p.F; ++x)
        if (mm[x] > 0.0) atomicMax(&M[(size_t)u.layer * p.F + x], (unsigned long long)__double_as_longlong(mm[x]));
    if (with_h) {
#pragma unroll
      for (int d = 0; d < PREP_NDIR; ++d)
        if (hh[d] > 0.0) atomicMax(&H[(size_t)u.layer * PREP_NDIR + d], (unsigned long long)__double_as_longlong(hh[d]));
    }
  }
}

// ---- threshold priming ---------------------------------------------------------------------------
// Every contribution to a document's score is >= 0 (K1d's gates: positive boosts, k1 >= 0, 0 <= b <= 1) and contributions meet
// through `+` and `max` (query.rs:150-164), so a document scores at least what ANY one of its postings gives it.  Hence the
// query's final K-th best score theta_K >= the K-th best posting score of any single list of the query - a number that does
// not depend on the query and is known before the launch ("threshold priming" of MaxScore / WAND: the rank-safe top-K needs
// no document strictly below it).  Stored per list: the r-th largest value of each score plane x (plane_x = tfn_x * idf) and of
// plane_0 + plane_1, for 16 ranks r up to 64.  With it (k_prep_query): gthr[q] starts at theta0 = max over the query's lists,
// and a list whose skip threshold is already below theta0 gets NO work items at all - round 5 dispatched ~60 k workgroups per
// C2 launch (of 66 k) only to have them read the threshold and leave.  A long list is scanned in segments whose tables meet
// through atomicMax: the r-th best of a segment is a lower bound of the list's r-th best.
constexpr int KTH_RANKS = 16;
__device__ __forceinline__ uint32_t kth_rank_of(const int j) {
  constexpr uint32_t R[KTH_RANKS] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 40, 50, 64};
  return R[j];
}
// `top` = the wave's 64 largest values so far, sorted descending over the lanes; merges 64 new values (one per lane) into it
__device__ __forceinline__ void top64_merge(double& top, double v, const int lane) {
  for (int k = 2; k <= WAVE; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {  // bitonic sort of v, descending
      const double o = __shfl_xor(v, j);
      v = (((lane & j) == 0) == ((lane & k) == 0)) ? fmax(v, o) : fmin(v, o);
    }
  double w = fmax(top, __shfl(v, WAVE - 1 - lane));  // descending against ascending: a bitonic sequence holding the 64 largest of both
  for (int j = WAVE / 2; j > 0; j >>= 1) {
    const double o = __shfl_xor(w, j);
    w = ((lane & j) == 0) ? fmax(w, o) : fmin(w, o);
  }
  top = w;
}
template <int F_>
__global__ __launch_bounds__(256) void k_list_kth(const BoundUnit* units, const uint32_t n_units, const uint4* layer_a, const double* plane,
                                                  const uint32_t* doc, const uint32_t* alive, unsigned long long* kth) {
  static_assert(F_ == 1 || F_ == 2, "threshold priming tables exist for one and two fields");
  constexpr int D = F_ == 1 ? 1 : 3;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int lane = threadIdx.x & (WAVE - 1);
  if (wave >= n_units) return;
  const BoundUnit u = units[wave];
  const uint4 la = layer_a[u.layer];
  const uint64_t off = ((uint64_t)la.x | ((uint64_t)la.y << 32)) + u.begin;
  double top[D];
#pragma unroll
  for (int d = 0; d < D; ++d) top[d] = 0.0;
  for (uint32_t i0 = 0; i0 < u.count; i0 += WAVE) {
    const uint32_t i = i0 + (uint32_t)lane;
    double v[D];
#pragma unroll
    for (int d = 0; d < D; ++d) v[d] = 0.0;
    // (documents a delta removed keep their postings until the next flatten: they are not among a list's K best - `alive` is null
    // while the snapshot carries no tombstones, and every delta recomputes these tables)
    if (i < u.count && (alive == nullptr || ((alive[doc[off + i] >> 5] >> (doc[off + i] & 31u)) & 1u))) {
      if (F_ == 1) v[0] = plane[off + i];
      else {
        const double2 t = reinterpret_cast<const double2*>(plane)[off + i];
        v[0] = t.x; v[1] = t.y; v[D - 1] = t.x + t.y;
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (__any(v[d] > readlane_f64(top[d], WAVE - 1))) top64_merge(top[d], v[d], lane);
  }
#pragma unroll
  for (int d = 0; d < D; ++d)
    for (int j = 0; j < KTH_RANKS; ++j)
      if ((uint32_t)lane + 1u == kth_rank_of(j) && top[d] > 0.0)
        atomicMax(&kth[((size_t)u.layer * D + d) * KTH_RANKS + j], (unsigned long long)__double_as_longlong(top[d]));
}
// The same two numbers from values that are already in registers (one or two fields): k_prep_query requests the tables of all
// of a query's entries together - a thread per query is a chain of dependent loads, and under a running k_daat every level of it
// costs 3-5 us - and evaluates afterwards.  Same operations in the same order as prep_entry_ub / prep_entry_theta0.
__device__ __forceinline__ double prep_ub_from(const PrepParams& pp, const double idf, const double eb, const double m0, const double m1,
                                               const double ha, const double hb) {
  double ub_m = 0.0;
  if (m0 > 0.0) ub_m += m0 * idf * pp.boost[0] * eb;
  if (pp.F == 2u && m1 > 0.0) ub_m += m1 * idf * pp.boost[1] * eb;
  if (pp.F != 2u || pp.bound_h == nullptr) return ub_m;
  const double jb = pp.h_a * ha + pp.h_b * hb;
  const double ub_j = (idf * eb) * jb * (1.0 + 1e-12) + 0x1p-1066;
  return fmin(ub_m, ub_j);
}
__device__ __forceinline__ double prep_theta0_from(const PrepParams& pp, const double eb, const double k0, const double k1, const double ks) {
  double th = fmax(0.0, (k0 * pp.boost[0]) * eb);
  if (pp.F == 2u) {
    th = fmax(th, (k1 * pp.boost[1]) * eb);
    const double sdir = ks * fmin(pp.boost[0], pp.boost[1]) * eb * (1.0 - 1e-12);
    if (sdir > 1e-290) th = fmax(th, sdir);
  }
  return th;
}

// theta0 of plan entry e: at least K of its postings score at least this much - through the readers' own expression
// ((plane_x * boost_x) * expansion_boost, monotone in plane_x; the other field adds >= 0), or, for two fields, through the
// K-th best plane sum with the smaller boost (a real-number inequality, deflated past the few roundings between the two).
__device__ __forceinline__ double prep_entry_theta0(const PrepParams& pp, const ps_plan_entry& e) {
  if (pp.kth == nullptr) return 0.0;
  const uint32_t D = pp.F == 1u ? 1u : 3u;
  const double* k = pp.kth + ((size_t)e.node * D) * KTH_RANKS + pp.kth_rank;
  double th = 0.0;
  for (uint32_t x = 0; x < pp.F; ++x) th = fmax(th, (k[x * KTH_RANKS] * pp.boost[x]) * e.boost);
  if (pp.F == 2u) {
    const double sdir = k[2 * KTH_RANKS] * fmin(pp.boost[0], pp.boost[1]) * e.boost * (1.0 - 1e-12);
    if (sdir > 1e-290) th = fmax(th, sdir);  // (not among subnormal products: a rounding there is absolute)
  }
  return th;
}

// Upper bound of any posting score of plan entry `e`, rounding included: the per-field form pushes the
// maxima through the kernels' own expression (every operation is monotone: ((tfn * idf) * boost) * eb,
// summed over the fields in order), the joint form bounds the real-number value and is inflated past the
// few roundings between them.
__device__ __forceinline__ double prep_entry_ub(const PrepParams& pp, const ps_plan_entry& e) {
  double ub_m = 0.0;
  for (uint32_t x = 0; x < pp.F; ++x) {
    const double t = pp.bound_m[(size_t)e.node * pp.F + x];
    if (t > 0.0) ub_m += t * e.idf * pp.boost[x] * e.boost;
  }
  // the joint bound: the largest boosted sum of saturated term frequencies any posting of the list has (exactly for these
  // boosts, F >= 3; from the direction supports, F == 2), times the list's weights
  double jb;
  if (pp.bound_j != nullptr) jb = pp.bound_j[e.node];
  else if (pp.bound_h != nullptr) {
    const double* h = pp.bound_h + (size_t)e.node * PREP_NDIR;
    jb = pp.h_a * h[pp.h_lo] + pp.h_b * h[pp.h_lo + 1];
  } else return ub_m;  // (one field: the per-field form is the maximum itself)
  // (the relative inflation covers the few roundings between the real-number bound and the computed scores; among SUBNORMAL
  // products - boosts of 1e-320 are admitted - a rounding is an absolute 2^-1074 that no factor restores: 256 of them on top)
  const double ub_j = (e.idf * e.boost) * jb * (1.0 + 1e-12) + 0x1p-1066;
  return fmin(ub_m, ub_j);
}

__device__ __forceinline__ uint32_t prep_chunk_of(const uint32_t split_div, const uint32_t chunk_min, const uint32_t len) {
  const uint32_t c = ((len + split_div - 1) / split_div + 255u) & ~255u;
  return c > chunk_min ? c : chunk_min;
}
__device__ __forceinline__ uint32_t prep_chunk(const PrepParams& pp, const uint32_t len) {
  return prep_chunk_of(pp.split_div, pp.chunk_min, len);
}
__device__ __forceinline__ uint32_t prep_bucket_in_set(const uint32_t rank, const uint32_t len, const bool short_first) {
  if (rank <= 1) {  // 64 length classes, log2 with one fractional bit
    // First-ranked lists (every chunk runs): longest first.  Second-ranked lists: longest first as well for plans with one
    // list per query term (the ones that stay essential are what a launch ends on: the long ones must not start last),
    // SHORTEST first for multi-expansion batches (`short_first`) - there the short lists are the ones that stay essential
    // while the chunks of the long ones mostly leave at once, and a chunk that runs cannot start before the dispatch front
    // has passed everything in front of it.  Same box, serving kernels, shortest first against longest first:
    // C5 1.600 -> 1.550 ms; C2 0.273 -> 0.273, C4 1.095 -> 1.113 (hence not there).
    const uint32_t l = len ? len : 1u;
    const uint32_t lg = 31u - (uint32_t)__clz((int)l);
    const uint32_t cls = 2u * lg + (lg ? ((l >> (lg - 1)) & 1u) : 0u);
    return PREP_SAMPLE_BUCKETS + (rank == 1u && short_first ? PREP_CLASSES + cls : rank * PREP_CLASSES + 63u - cls);
  }
  return PREP_SAMPLE_BUCKETS + 2 * PREP_CLASSES + (rank < PREP_RANKS ? rank : PREP_RANKS) - 1u;
}
__device__ __forceinline__ uint32_t prep_bucket(const uint32_t rank, const uint32_t len, const bool short_first = false, const bool big = false) {
  return (big ? PREP_SET_BUCKETS : 0u) + prep_bucket_in_set(rank, len, short_first);
}
// Chunks [0, result) of a list cut into chunks of `c` postings lie entirely below the sample boundary (0: no sample phase, or
// the list's table is too coarse to tell).  Scheduling only: any value in [0, chunks] is correct.
__device__ __forceinline__ uint32_t prep_sample_chunks(const PrepParams& pp, const ps_plan_entry& en, const uint32_t c, const uint32_t nc) {
  if (!pp.sample_tile) return 0u;
  const uint32_t sh = en.shift & 0xFFu;
  if (pp.sample_tile & ((1u << sh) - 1u)) return 0u;
  const uint32_t p0 = min(en.len, pp.table[en.tbl_off + (pp.sample_tile >> sh)]);  // postings with doc id < D0
  return p0 >= en.len ? nc : p0 / c;
}

// ---- descriptors of one query ---------------------------------------------------------------------
// Result of the per-query part: candidate slots the query needs (its items), and per entry the chunking.
// Two implementations with the same outputs: plans of <= 4 entries are handled in registers (every global
// load of the query is issued up front, all the O(n^2) logic is ALU work), wider ones (<= 64) walk their
// arrays in HBM.
constexpr double PREP_SLACK = 1.0 + 1e-9;  // the bounds are summed in another order than the scores

__device__ __forceinline__ bool prep_before(const double ua, const uint32_t la, const uint32_t ia, const double ub,
                                            const uint32_t lb, const uint32_t ib) {
  // processing order: bound descending; equal bounds: the LONGER list ranks lower (it is the one that
  // becomes non-essential); then plan order (stable)
  return ua > ub || (ua == ub && (la < lb || (la == lb && ia < ib)));
}

// (what the entry loop of k_prep_query needs of a small query's entries, kept in registers: no second round of loads)
template <int NMAX>
struct PrepSmall { uint32_t len[NMAX], node[NMAX], rank[NMAX], cand[NMAX], dead; };

template <int NMAX>
__device__ __forceinline__ uint32_t prep_query_small(const PrepParams& pp, const uint32_t q, const uint32_t b, const uint32_t n, uint32_t& groups, double& th0,
                                                     PrepSmall<NMAX>& ps) {
  double ub[NMAX], t0[NMAX];
  th0 = 0.0;
  ps.dead = 0u;
  uint32_t len[NMAX], grp[NMAX], rank[NMAX], qt[NMAX];
  if (pp.F <= 2u) {
    // level 1: the entries' own words (an entry past the plan's end re-reads the last one: no branch between the loads);
    // level 2: everything that hangs on the list ordinal - per-field maxima, two direction supports, priming ranks, row candidate
    uint32_t node[NMAX];
    double idf[NMAX], eb[NMAX];
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
      const ps_plan_entry& en = pp.plan[b + ((uint32_t)i < n ? (uint32_t)i : n - 1u)];
      len[i] = en.len; qt[i] = en.qterm; node[i] = en.node; idf[i] = en.idf; eb[i] = en.boost;
    }
    double m0[NMAX], m1[NMAX], ha[NMAX], hb[NMAX], k0[NMAX], k1[NMAX], ks[NMAX];
    const bool two = pp.F == 2u, with_h = two && pp.bound_h != nullptr, with_k = pp.kth != nullptr;
    const uint32_t D = two ? 3u : 1u;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
      const size_t l = node[i];
      m0[i] = pp.bound_m[l * pp.F];
      m1[i] = two ? pp.bound_m[l * 2 + 1] : 0.0;
      ha[i] = with_h ? pp.bound_h[l * PREP_NDIR + pp.h_lo] : 0.0;
      hb[i] = with_h ? pp.bound_h[l * PREP_NDIR + pp.h_lo + 1] : 0.0;
      const double* k = pp.kth + (l * D) * KTH_RANKS + pp.kth_rank;
      k0[i] = with_k ? k[0] : 0.0;
      k1[i] = with_k && two ? k[KTH_RANKS] : 0.0;
      ks[i] = with_k && two ? k[2 * KTH_RANKS] : 0.0;
      ps.cand[i] = pp.n_cand ? (uint32_t)pp.cand_of_layer[l] : NO_CAND;
      ps.node[i] = node[i];
    }
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
      const bool on = (uint32_t)i < n;
      ub[i] = on ? prep_ub_from(pp, idf[i], eb[i], m0[i], m1[i], ha[i], hb[i]) : 0.0;
      t0[i] = on && with_k ? prep_theta0_from(pp, eb[i], k0[i], k1[i], ks[i]) : 0.0;
      if (!on) { len[i] = 0; qt[i] = 0xFFFFFFFFu; }
      th0 = fmax(th0, t0[i]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
      ub[i] = 0.0; len[i] = 0; qt[i] = 0xFFFFFFFFu; ps.node[i] = 0; ps.cand[i] = NO_CAND;
      if ((uint32_t)i < n) {
        const ps_plan_entry& en = pp.plan[b + i];
        ub[i] = prep_entry_ub(pp, en);
        t0[i] = prep_entry_theta0(pp, en);
        len[i] = en.len;
        qt[i] = en.qterm;
        ps.node[i] = en.node;
        ps.cand[i] = pp.n_cand ? (uint32_t)pp.cand_of_layer[en.node] : NO_CAND;
        th0 = fmax(th0, t0[i]);
      }
    }
  }
  uint32_t n_groups = 0;
#pragma unroll
  for (int i = 0; i < NMAX; ++i) {
    if ((uint32_t)i < n && (i == 0 || qt[i] != qt[i > 0 ? i - 1 : 0])) ++n_groups;
    grp[i] = n_groups - 1;
  }
  groups = n_groups;
#pragma unroll
  for (int i = 0; i < NMAX; ++i) {
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < NMAX; ++j)
      if ((uint32_t)j < n && j != i && prep_before(ub[j], len[j], (uint32_t)j, ub[i], len[i], (uint32_t)i)) ++r;
    rank[i] = r;
  }
  // per entry: the maximum of its query term's bounds, and whether it is the first entry of its term
  double gm[NMAX];
#pragma unroll
  for (int i = 0; i < NMAX; ++i) {
    double m = 0.0;
#pragma unroll
    for (int j = 0; j < NMAX; ++j)
      if ((uint32_t)j < n && grp[j] == grp[i]) m = fmax(m, ub[j]);
    gm[i] = m;
  }
  uint32_t slots = 0;
#pragma unroll
  for (int i = 0; i < NMAX; ++i) {
    if ((uint32_t)i < n) {
      double rest = 0.0, alt = 0.0, skip = 0.0;
#pragma unroll
      for (int j = 0; j < NMAX; ++j) {
        if ((uint32_t)j < n) {
          const bool first_of_group = j == 0 || grp[j] != grp[j > 0 ? j - 1 : 0];
          if (first_of_group && grp[j] != grp[i]) rest += gm[j];
          if (grp[j] == grp[i] && j != i) alt = fmax(alt, ub[j]);
          // skip threshold: per query term the largest bound among its entries of rank >= rank[i]
          if (first_of_group) {
            double m = 0.0;
#pragma unroll
            for (int k = 0; k < NMAX; ++k)
              if ((uint32_t)k < n && grp[k] == grp[j] && rank[k] >= rank[i]) m = fmax(m, ub[k]);
            skip += m;
          }
        }
      }
      DEntry d;
      d.ub = ub[i];
      d.q = q;
      d.rank = rank[i];
      d.others = (rest + alt) * PREP_SLACK;
      if (!(d.others >= 0.0)) d.others = INFINITY;
      d.skip_thr = skip * PREP_SLACK;
      if (!(d.skip_thr >= 0.0)) d.skip_thr = INFINITY;
      pp.dentry[b + i] = d;
      pp.rorder[b + rank[i]] = b + i;
      pp.gord[b + i] = (uint8_t)grp[i];
      if (pp.multi) {
        DGroup dg;
        dg.grp = n_groups <= 4 ? grp[i] : 0xFFFFFFFFu;
        dg.ub_s = ub[i] * PREP_SLACK;
        // the next list of the same term in rank order: the largest bound among its entries of higher rank number
        double nx = 0.0;
        uint32_t nr = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < NMAX; ++j)
          if ((uint32_t)j < n && grp[j] == grp[i] && rank[j] > rank[i] && rank[j] < nr) { nr = rank[j]; nx = ub[j]; }
        dg.nxt_s = nx * PREP_SLACK;
        dg._pad[0] = dg._pad[1] = dg._pad[2] = 0;
        pp.dgroup[b + i] = dg;
      }
      const uint32_t c = prep_chunk(pp, len[i]);
      const bool dead = !pp.prime_keep_items && d.skip_thr < th0;  // (a list that is non-essential from the start gets no items)
      if (!dead) slots += (len[i] + c - 1) / c;
      ps.dead |= dead ? 1u << i : 0u;
    }
    ps.len[i] = len[i];
    ps.rank[i] = rank[i];
  }
  return slots;
}

__device__ __noinline__ uint32_t prep_query_general(const PrepParams& pp, const uint32_t q, const uint32_t b, const uint32_t n, double& th0) {
  // bounds; dense ordinal of every entry's query term (the entries of a term are adjacent in plan order)
  uint32_t n_groups = 0, cur = 0xFFFFFFFFu;
  th0 = 0.0;
  for (uint32_t i = 0; i < n; ++i) {
    const ps_plan_entry& en = pp.plan[b + i];
    DEntry& d = pp.dentry[b + i];
    d.ub = prep_entry_ub(pp, en);
    th0 = fmax(th0, prep_entry_theta0(pp, en));
    d.q = q;
    if (en.qterm != cur) { cur = en.qterm; ++n_groups; }
    pp.gord[b + i] = (uint8_t)(n_groups - 1);
  }
  for (uint32_t i = 0; i < n; ++i) {  // insertion sort into the processing order
    const double ui = pp.dentry[b + i].ub;
    const uint32_t li = pp.plan[b + i].len;
    uint32_t j = i;
    while (j > 0) {
      const uint32_t pj = pp.rorder[b + j - 1];
      if (prep_before(ui, li, b + i, pp.dentry[pj].ub, pp.plan[pj].len, pj)) { pp.rorder[b + j] = pj; --j; } else break;
    }
    pp.rorder[b + j] = b + i;
  }
  for (uint32_t r = 0; r < n; ++r) pp.dentry[pp.rorder[b + r]].rank = r;
  // others: what every OTHER entry can add to a document of this list = the other query terms' maxima +
  // the best other entry of its own term (expansions of a term merge by max, query.rs:150-164)
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t gi = pp.gord[b + i];
    double rest = 0.0, alt = 0.0, run_max = 0.0;
    uint32_t run = 0xFFFFFFFFu;
    for (uint32_t j = 0; j < n; ++j) {
      const uint32_t gj = pp.gord[b + j];
      const double uj = pp.dentry[b + j].ub;
      if (gj != run) {
        if (run != 0xFFFFFFFFu && run != gi) rest += run_max;
        run = gj;
        run_max = 0.0;
      }
      run_max = fmax(run_max, uj);
      if (gj == gi && j != i) alt = fmax(alt, uj);
    }
    if (run != gi) rest += run_max;
    double o = (rest + alt) * PREP_SLACK;
    if (!(o >= 0.0)) o = INFINITY;
    pp.dentry[b + i].others = o;
  }
  // skip thresholds: a document that only occurs in the lists of rank >= r scores at most the sum over
  // query terms of the largest bound among those of its lists - i.e. of the first entry of each term met
  // when walking the ranks from r upwards (the bounds descend along the ranks)
  for (uint32_t r = 0; r < n; ++r) {
    unsigned long long seen = 0ull;
    double bound = 0.0;
    for (uint32_t r2 = r; r2 < n; ++r2) {
      const uint32_t j = pp.rorder[b + r2];
      const unsigned long long bit = 1ull << (pp.gord[j] & 63u);
      if (!(seen & bit)) { seen |= bit; bound += pp.dentry[j].ub; }
    }
    double sk = bound * PREP_SLACK;
    if (!(sk >= 0.0)) sk = INFINITY;
    pp.dentry[pp.rorder[b + r]].skip_thr = sk;
  }
  if (pp.multi) {
    for (uint32_t r = 0; r < n; ++r) {
      const uint32_t i = pp.rorder[b + r];
      DGroup dg;
      dg.grp = n_groups <= 4 ? (uint32_t)pp.gord[i] : 0xFFFFFFFFu;
      dg.ub_s = pp.dentry[i].ub * PREP_SLACK;
      dg.nxt_s = 0.0;
      dg._pad[0] = dg._pad[1] = dg._pad[2] = 0;
      for (uint32_t r2 = r + 1; r2 < n; ++r2) {
        const uint32_t j = pp.rorder[b + r2];
        if (pp.gord[j] == pp.gord[i]) { dg.nxt_s = pp.dentry[j].ub * PREP_SLACK; break; }
      }
      pp.dgroup[i] = dg;
    }
  }
  uint32_t slots = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t len = pp.plan[b + i].len;
    const uint32_t c = prep_chunk(pp, len);
    if (pp.prime_keep_items || !(pp.dentry[b + i].skip_thr < th0)) slots += (len + c - 1) / c;
  }
  return slots;
}

// ---- wave-aggregated atomics ------------------------------------------------------------------------
// Thousands of increments per batch go to a few dozen counters (item buckets, candidate slots, row uses): lanes
// of a wave that hit the same counter are summed first and one lane adds the sum, so a counter sees one
// atomic per wave instead of one per entry.  Returns the value the counter had before this lane's share.
__device__ __forceinline__ uint32_t wave_add_by_key(uint32_t* counters, const uint32_t key, const uint32_t val, bool active) {
  const int lane = threadIdx.x & (WAVE - 1);
  uint32_t result = 0;
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
    const bool same = active && key == k;
    const uint32_t v = same ? val : 0u;
    uint32_t inc = v;  // inclusive scan over the lanes
    for (int o = 1; o < WAVE; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
    uint32_t base = 0;
    if (lane == first) base = atomicAdd(&counters[k], total);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
    if (same) result = base + inc - v;
    todo &= ~__ballot(same);
    active = active && !same;
  }
  return result;
}

// The same without a result: nobody waits for the atomic's round trip.
__device__ __forceinline__ void wave_add_by_key_noret(uint32_t* counters, const uint32_t key, const uint32_t val, bool active) {
  const int lane = threadIdx.x & (WAVE - 1);
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
    const bool same = active && key == k;
    uint32_t v = same ? val : 0u;
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    if (lane == first) atomicAdd(&counters[k], v);
    todo &= ~__ballot(same);
    active = active && !same;
  }
}

// (run by the last wave of k_prep_query) bucket starts (rank-major, longest rank-0 lists first), the item count, and the dense rows of this
// batch: a candidate used >= min_uses times is read as a row, scored with the (idf, expansion_boost) of its
// FIRST user in plan order (deterministic); entries with other weights keep their bitmap lookups.  A resident
// row with the same weights is not scored again.
// Bucket starts from the bucket totals: an exclusive scan over PREP_BUCKETS counters by ONE wave - a lane takes a block of
// consecutive buckets (its loads go out together), the blocks meet through a shuffle scan.  (One thread walking the 280 counters
// with a dependent add per atomic load took ~150 us of every batch's preparation while a k_daat launch owned the chip - most of
// k_prep_query, the kernel that paced the pipeline once the scoring kernels got faster; round 5.)
__device__ __forceinline__ void prep_scan_buckets(PrepCtl& c) {
  constexpr uint32_t PER = (PREP_BUCKETS + WAVE - 1) / WAVE;
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  uint32_t v[PER], sum = 0;
#pragma unroll
  for (uint32_t j = 0; j < PER; ++j) {
    const uint32_t k = lane * PER + j;
    v[j] = k < PREP_BUCKETS ? __hip_atomic_load(&c.bucket_total[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  }
#pragma unroll
  for (uint32_t j = 0; j < PER; ++j) sum += v[j];
  uint32_t inc = sum;
  for (int o = 1; o < WAVE; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if ((int)lane >= o) inc += t; }
  uint32_t at = inc - sum;
#pragma unroll
  for (uint32_t j = 0; j < PER; ++j) {
    const uint32_t k = lane * PER + j;
    if (k < PREP_BUCKETS) c.bucket_start[k] = at;
    at += v[j];
  }
  if (lane == (uint32_t)WAVE - 1u) c.n_items = inc;
}

__device__ __forceinline__ void prep_finish(const PrepParams& pp) {
  PrepCtl& c = *pp.ctl;
  prep_scan_buckets(c);
  // dense rows: a lane per candidate (<= PREP_MAX_ROWS = 64); the rows to score are compacted in candidate order
  const uint32_t cd = threadIdx.x & (WAVE - 1);
  const bool have = cd < pp.n_cand;
  bool use = false, build = false;
  RowDesc rd;
  rd.post_off = 0; rd.len = 0; rd._pad = 0; rd.idf = 0.0; rd.eb = 0.0; rd.slot = cd; rd.tbl_off = 0;
  if (have) {
    RowState& rs = pp.row_state[cd];
    const uint32_t uses = __hip_atomic_load(&c.row_use[cd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long first = __hip_atomic_load(&c.row_first[cd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    use = uses >= pp.min_uses && first != 0ull;
    rs.use_now = use ? 1u : 0u;
    if (use) {
      const ps_plan_entry& en = pp.plan[~first];
      const unsigned long long ib = (unsigned long long)__double_as_longlong(en.idf), eb = (unsigned long long)__double_as_longlong(en.boost);
      build = !(pp.rows_resident && rs.valid && rs.idf_bits == ib && rs.eb_bits == eb);
      if (build) {
        rs.valid = 1; rs.idf_bits = ib; rs.eb_bits = eb;
        const uint4 la = pp.layer_a[en.node];
        rd.post_off = (uint64_t)la.x | ((uint64_t)la.y << 32);
        rd.len = la.z;
        rd.idf = en.idf;
        rd.eb = en.boost;
        rd.tbl_off = la.w;  // candidates are lists with one table slot per tile (host: shift == 0)
      }
    }
  }
  const unsigned long long mb = __ballot(build), mu = __ballot(use);
  const unsigned long long lt = cd ? (~0ull >> (64 - cd)) : 0ull;
  if (build) pp.row_desc[__popcll(mb & lt)] = rd;
  if (cd == 0) {
    const uint32_t n_build = (uint32_t)__popcll(mb), n_used = (uint32_t)__popcll(mu);
    c.n_rows_build = n_build;
    c.n_rows_used = n_used;
    if (PS_WORK_COUNTERS && (n_build | n_used)) {
      atomicAdd(&pp.wstats[WS_ROWS_BUILT], (unsigned long long)n_build);
      atomicAdd(&pp.wstats[WS_ROWS_USED], (unsigned long long)n_used);
    }
  }
}

// Thread per query (small workgroups: while a k_daat launch owns the chip, a workgroup of another queue only
// gets the wave slots two finishing k_daat waves leave behind - a 1024-thread workgroup can wait 200 us for
// a compute unit to have room): descriptors, candidate slots, item-bucket totals, dense-row uses.
__global__ __launch_bounds__(WAVE) void k_prep_query(const PrepParams pp) {
  // The workgroup's (= the wave's) contributions to the item-bucket totals and the dense-row uses are summed in LDS and flushed
  // once, every non-empty bin by its own lane and all at once.  (Round 5 aggregated per distinct key with shuffles - one pass and
  // one atomic per key and entry position, 20-50 us of a wave's time while a scoring kernel keeps the memory system busy.)
  __shared__ uint32_t h_bucket[PREP_BUCKETS];
  __shared__ uint32_t h_row[PREP_MAX_ROWS];
  for (uint32_t k = threadIdx.x; k < PREP_BUCKETS; k += WAVE) h_bucket[k] = 0u;
  h_row[threadIdx.x & (PREP_MAX_ROWS - 1u)] = 0u;
  __syncthreads();
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  const bool have = q < pp.B;
  const uint32_t b = have ? pp.qbeg[q] : 0u, n = have ? pp.qbeg[q + 1] - b : 0u;
  uint32_t slots = 0;
  // (plans of <= 4 entries - one list per query term: C2, C4 - entirely in registers; wider ones walk their arrays in HBM.
  // An 8-entry register variant cost this kernel 145 VGPRs and 58 SGPR spills for every batch: 76 / 0 without it.)
  uint32_t groups = 0;
  double th0 = 0.0;  // the query's primed threshold
  PrepSmall<4> ps;
  if (n) slots = n <= 4 ? prep_query_small<4>(pp, q, b, n, groups, th0, ps) : prep_query_general(pp, q, b, n, th0);
  if (have && pp.gthr != nullptr) pp.gthr[q] = (unsigned long long)__double_as_longlong(th0);
  // PLAN_BIG's rule (k_plan): more than 4 lists, or several lists under one query term -> k_daat's part of the batch
  const bool big = pp.split_kinds && (n > 4u || groups != n);
  // candidate slots: query-major within the query; the wave's queries take one block of the batch's slots
  const uint32_t s0 = wave_add_by_key(&pp.ctl->total_slots, 0u, slots, have && n != 0);
  if (have) { pp.qslot[q] = n ? s0 : 0u; pp.qslot_n[q] = slots; }
  uint32_t n_max = n;
  for (int o = 32; o > 0; o >>= 1) n_max = max(n_max, (uint32_t)__shfl_xor((int)n_max, o));
  uint32_t sl = s0;
  for (uint32_t i = 0; i < n_max; ++i) {  // (wave-uniform trip count: the aggregated atomics need every lane)
    const bool on = i < n;
    uint32_t bk = 0, bs = 0, nc = 0, ns = 0, cd = NO_CAND;
    if (on) {
      // a small query's entries are in registers (prep_query_small); a wide one's are read back from its arrays
      uint32_t len_i, rk;
      bool dead;
      if (n <= 4u) {
        const uint32_t k = i & 3u;
        len_i = k == 0u ? ps.len[0] : k == 1u ? ps.len[1] : k == 2u ? ps.len[2] : ps.len[3];
        rk = k == 0u ? ps.rank[0] : k == 1u ? ps.rank[1] : k == 2u ? ps.rank[2] : ps.rank[3];
        cd = k == 0u ? ps.cand[0] : k == 1u ? ps.cand[1] : k == 2u ? ps.cand[2] : ps.cand[3];
        dead = ((ps.dead >> k) & 1u) != 0u;
      } else {
        const ps_plan_entry& en = pp.plan[b + i];
        len_i = en.len;
        rk = pp.dentry[b + i].rank;
        dead = !pp.prime_keep_items && pp.dentry[b + i].skip_thr < th0;  // non-essential before the launch: no items, no candidate slots
        if (pp.n_cand) cd = pp.cand_of_layer[en.node];
      }
      const uint32_t c = prep_chunk(pp, len_i);
      nc = dead ? 0u : (len_i + c - 1) / c;
      ns = dead ? 0u : (big || !pp.split_kinds || pp.sample_small) ? prep_sample_chunks(pp, pp.plan[b + i], c, nc) : 0u;
      pp.gen[b + i] = DItemGen{(big ? 1u : 0u) | (dead ? 2u : 0u), ns, c, sl};  // (`entry`: gen is indexed by entry - the word carries the query's kind and "no items")
      sl += nc;
      bk = prep_bucket(rk, len_i, pp.multi != 0u && (big || !pp.split_kinds), big);
      bs = (big ? PREP_SET_BUCKETS : 0u) + (rk < PREP_SAMPLE_BUCKETS ? rk : PREP_SAMPLE_BUCKETS - 1u);
    }
    if (on && nc != ns) atomicAdd(&h_bucket[bk], nc - ns);
    if (pp.sample_tile && on && ns != 0) atomicAdd(&h_bucket[bs], ns);
    if (pp.n_cand && on && cd != NO_CAND) {
      atomicAdd(&h_row[cd], 1u);
      atomicMax(&pp.ctl->row_first[cd], ~(unsigned long long)(b + i));
    }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < PREP_BUCKETS; k += WAVE) {
    const uint32_t v = h_bucket[k];
    if (v) atomicAdd(&pp.ctl->bucket_total[k], v);
  }
  if (pp.n_cand) {
    const uint32_t v = h_row[threadIdx.x & (PREP_MAX_ROWS - 1u)];
    if (v) atomicAdd(&pp.ctl->row_use[threadIdx.x & (PREP_MAX_ROWS - 1u)], v);
  }
  // the wave that finishes last closes the batch's counters (k_prep_finish's work, without its launch)
  __threadfence();
  uint32_t t = 0;
  if (threadIdx.x == 0) t = atomicAdd(&pp.ctl->ticket, 1u);
  t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
  if (t + 1u == gridDim.x) {
    __threadfence();
    prep_finish(pp);
    if (pp.fault != nullptr) {
      // the scoring launches are sized from the host's item counts and clamp to the device's: a device count ABOVE the host's
      // would silently skip items (a wrong top-k, no error) - make it loud instead
      __threadfence();
      if ((threadIdx.x & (WAVE - 1)) == 0u) {
        const uint32_t n = __hip_atomic_load(&pp.ctl->n_items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t split = pp.split_kinds ? __hip_atomic_load(&pp.ctl->bucket_start[PREP_SET_BUCKETS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : n;
        const bool bad = n > pp.host_items || (pp.split_kinds && (split > pp.host_items - pp.host_items_big || n - split > pp.host_items_big));
        if (bad && atomicCAS(&pp.fault[0], 0u, FAULT_ITEM_COUNTS) == 0u) { pp.fault[1] = n; pp.fault[2] = split; pp.fault[3] = pp.host_items; }
      }
    }
  }
}

// A thread per list: the list's place in the item order (the next free items of its bucket: one aggregated
// atomic per bucket and wave), its dense-row flag, its items.
__global__ __launch_bounds__(2 * WAVE) void k_prep_items(const PrepParams pp) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool have = i < pp.ne;
  uint32_t nc = 0, ns = 0, bk = 0, bs = 0, len_i = 0, chunk = 1, first_slot = 0, q_i = 0;
  double skip_i = 0.0;
  if (have) {
    ps_plan_entry& en = pp.plan[i];
    const DItemGen g = pp.gen[i];
    const uint32_t len = en.len, c = g.chunk;
    len_i = len; chunk = c; first_slot = g.first_slot;
    const DEntry de_i = pp.dentry[i];
    const bool dead = (g.entry & 2u) != 0u;  // (k_prep_query: the list is non-essential under the query's primed threshold - no items)
    nc = dead ? 0u : (len + c - 1) / c;
    ns = g.item_at;  // (k_prep_query: the list's chunks in the sample phase)
    const bool big = (g.entry & 1u) != 0u;  // (... and its query's kind)
    skip_i = de_i.skip_thr; q_i = de_i.q;
    bk = prep_bucket(de_i.rank, len, pp.multi != 0u && (big || !pp.split_kinds), big);
    bs = (big ? PREP_SET_BUCKETS : 0u) + (de_i.rank < PREP_SAMPLE_BUCKETS ? de_i.rank : PREP_SAMPLE_BUCKETS - 1u);
    if (pp.n_cand) {
      const uint32_t cd = pp.cand_of_layer[en.node];
      if (cd != NO_CAND) {
        const RowState rs = pp.row_state[cd];
        if (rs.use_now && rs.idf_bits == (unsigned long long)__double_as_longlong(en.idf) &&
            rs.eb_bits == (unsigned long long)__double_as_longlong(en.boost)) {
          en.shift |= DENSE_FLAG;
          en.node = cd;
        }
      }
    }
  }
  // the lists' places in their buckets: offsets within the workgroup from LDS atomics, then ONE round trip to the buckets' fill
  // counters for all of the workgroup's buckets together (a lane per bucket), instead of one round trip per distinct bucket
  __shared__ uint32_t h_cnt[PREP_BUCKETS];
  __shared__ uint32_t h_base[PREP_BUCKETS];
  for (uint32_t k = threadIdx.x; k < PREP_BUCKETS; k += 2 * WAVE) h_cnt[k] = 0u;
  __syncthreads();
  uint32_t local = 0, local_s = 0;
  if (have && nc != ns) local = atomicAdd(&h_cnt[bk], nc - ns);
  if (pp.sample_tile && have && ns != 0) local_s = atomicAdd(&h_cnt[bs], ns);
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < PREP_BUCKETS; k += 2 * WAVE) {
    const uint32_t v = h_cnt[k];
    h_base[k] = v ? pp.ctl->bucket_start[k] + atomicAdd(&pp.ctl->bucket_fill[k], v) : 0u;
  }
  __syncthreads();
  const uint32_t at = have && nc != ns ? h_base[bk] + local : 0u;
  const uint32_t at_s = pp.sample_tile && have && ns ? h_base[bs] + local_s : 0u;
  if (have) {
    // the list's items (~20 on average): stores nobody waits for; chunks [0, ns) are in the sample phase
    for (uint32_t j = 0; j < nc; ++j) {
      const uint32_t pb = j * chunk;
      const uint32_t a = j < ns ? at_s + j : at + (j - ns);
      if (a < pp.items_cap) pp.items[a] = DItem{i, pb, min(chunk, len_i - pb), first_slot + j, skip_i, q_i, 0u};
    }
  }
}

// K0b over the rows k_prep_finish listed: a fixed grid, every workgroup takes (row, tile range) units
// until none are left (the host does not know how many rows the batch builds).
__global__ __launch_bounds__(256) void k_dense_rows_dyn(const KParams p, double* rows, const PrepCtl* ctl, const uint32_t blocks_per_row) {
  const uint32_t n_units = ctl->n_rows_build * blocks_per_row;
  for (uint32_t w = blockIdx.x; w < n_units; w += gridDim.x) {
    dense_row_block(p, rows, p.row_desc[w / blocks_per_row], w % blocks_per_row, blocks_per_row);
    __syncthreads();
  }
}

}  // namespace ps
