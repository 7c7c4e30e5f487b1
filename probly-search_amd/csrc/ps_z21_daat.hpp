// ps_z21_daat.hpp — K1dz: exact top-K with dynamic pruning for zero_to_one (zero_to_one.rs:44-126), the
// document-at-a-time skeleton of k_daat_small with the bounds of this scorer, plus its device-side preparation.
//
// Scope: top-k batches whose queries are all "simple" (ps_engine.hip, classification: one version layer, every
// record of the query on its own (query term, trie node) pair; a node repeated under several query terms and
// several expansions of one query term are both fine) with at most Z_MAX_LISTS (8) lists.  Everything else stays
// on k_score<MODE_Z21S> / k_z21.
//
// What a document scores (zero_to_one.rs:84-126 for such queries): per field x the pool
//     pool_x = sum over the query's records in sorted order (score desc, stable: the entries arrive in that order)
//              of  (min(score_e / tf, 1) * tf) / max(field_length_x, query_terms_len)
//              for the records with tf_x >= need_e (the node's pool rule) whose query term is not consumed yet,
// and the document's score is the maximum of its pools (starting from the merged dummy 0.0).
//
// Bounds (every one computed with the operations of the score itself, in the same order - NO slack factor,
// because ties at the threshold are the common case of this scorer and have to prune, see below):
//   * ubnum_e (host): the largest numerator min(score_e / t, 1) * t over the term frequencies t the list holds;
//     u'_j = the non-increasing envelope of ubnum along the sorted order, so the first m of them dominate any m
//     records term by term;
//   * B(m, fl) = sum_{j < m} u'_j / max(fl, query_terms_len): no pool of a field of length fl fed by at most m
//     records can exceed it (IEEE division and addition are monotone); tabulated per item for fl < 64 in LDS,
//     and from it, per threshold, the longest field that can still beat the threshold with m records (FLMAX[m]):
//     the scan tests postings with integer compares only;
//   * zub[e][x] = ubnum_e / max(shortest field x holding the term, query_terms_len): what one list can add to
//     pool x at most; DEntry::skip_thr = max_x of the in-order sum over the lists of rank >= rank(e).
//
// Thresholds.  The result order is (score desc, key asc) and doc ids ascend with keys, so a document that only
// TIES the K-th best score still loses to K documents with lower ids.  Scores of this scorer are small
// rationals: thousands of documents tie at the threshold, and "strictly below" pruning keeps all of them alive
// (round 2's k_daat_z: 15-30 % of the postings).  Two words per query:
//   gthr[q]  K-th best score of ANY wave of the query (as k_daat_small): prunes bound <  gthr;
//   gtie[l][q]  (Z_LEVELS words per query) K-th best score of a wave that has only scanned documents below doc id D_l so
//            far (D_0 < D_1 < D_2: powers of two, D_2 near N / 4, a factor 4 apart; doc ids ascend along a chunk): K
//            documents with ids < D_l score >= gtie[l], so a trip whose postings all lie at or above D_l also prunes
//            bound == gtie[l], and so does the whole-chunk skip test of a chunk that starts at or above D_l (its level
//            rides in DItem::count, from the list's tile-offset table).
// Launch order: the chunks that lie entirely below D_0 first, then those below D_1, below D_2, then the rest (rank-major
// within a phase).  The first phase is a sample of the document space evaluated almost unpruned; each later phase is
// several times larger and pruned with what the earlier ones left.
#pragma once
#include "ps_prep_kernels.hpp"

namespace ps {

#ifndef PS_DAAT_ZU
#define PS_DAAT_ZU 4   // postings per lane in flight in the scan
#endif
#ifndef PS_DAAT_ZPF
#define PS_DAAT_ZPF 0  // 1: the next trip's own postings are requested before this trip's lookups (measured: no gain, 12-18 more VGPRs)
#endif
#ifndef PS_DAAT_Z_BTAB
#define PS_DAAT_Z_BTAB 1  // 0: the narrow instantiation recomputes its bound-table column too (A/B builds)
#endif
constexpr int Z_LEVELS = 3;
constexpr int Z_MAX_LISTS = 8;                 // most records of a query K1dz takes (k_daat_z<F, WC, 8>; <= DAAT_SMALL_MAX: the narrow instantiation)
constexpr uint32_t ZITEM_LEVEL_SHIFT = 30;     // DItem::count bits 30-31: levels l (the lowest ones) with every document of the chunk at or above D_l
constexpr uint32_t ZITEM_COUNT = 0x3FFFFFFFu;
constexpr uint32_t Z_NO_LEVEL = 0xFFFFFFFFu;   // KParams::z_dl of a level that does not exist (tiny corpora)
constexpr int Z_FLN = 64;                      // field lengths the bound table holds (entry 63 stands for >= 63)
constexpr int Z_ALL = 0x7FFFFFFF, Z_NONE = -1;

struct ZPrepParams {
  const ps_plan_entry* plan;   // [ne], per query in sorted record order
  const uint32_t* qbeg;        // [B + 1]
  const double* zub;           // [ne][F]
  const uint32_t* table;
  uint32_t B, ne, F, chunk_min, split_div;
  uint32_t dl_tile[Z_LEVELS];  // D_l >> t_log2; 0 = the level does not exist
  DEntry* dentry;
  DItemGen* gen;
  uint32_t* nbelow;            // [ne][Z_LEVELS] chunks of the list that lie entirely below D_l (non-decreasing in l)
  uint32_t* nabove_from;       // [ne][Z_LEVELS] first chunk that lies entirely at or above D_l (= chunks: none / not known)
  uint32_t* qslot;
  uint32_t* qslot_n;
  DItem* items;
  uint32_t items_cap;
  PrepCtl* ctl;
};

// Item order: phase-major (phase = the levels a chunk is NOT entirely below: 0 = below D_0 ... Z_LEVELS = the rest), rank-major
// within a phase; the last phase - most of the launch - as K1d orders it (every query's shortest list first, longest
// lists first within a rank).  Measured on C3 with one level: plain rank-major order scanned 202 M postings per batch
// (1.34 ms), the sample below D = N / 8 first 80 M (0.56 ms; N / 16: 91 M, N / 32: 121 M, N / 4: 90 M -
// profiles/r04_c3_d0_sweep.jsonl).
__device__ __forceinline__ uint32_t zprep_bucket(const uint32_t phase, const uint32_t rank, const uint32_t len) {
  if (phase < (uint32_t)Z_LEVELS) return phase * 4u + (rank < 3u ? rank : 3u);
  return 4u * Z_LEVELS + (rank <= 1u ? prep_bucket(rank, len) - PREP_SAMPLE_BUCKETS : 2u * PREP_CLASSES + (rank - 2u < 3u ? rank - 2u : 3u));
}
static_assert(4u * Z_LEVELS + 2u * PREP_CLASSES + 4u <= PREP_BUCKETS, "K1dz item buckets fit the preparation's control block");

// K1dz's batch image from a device-built plan (k_plan, zmode): a thread per query puts the query's records (<= DAAT_SMALL_MAX,
// one list each: the count pass established that every query is simple) into the record-sort order (score desc,
// stable: zero_to_one.rs:98) and derives what enqueue_daat_z_host derives on the host, in the same arithmetic:
// qterm_index = occurrence rank of the record's term among the sorted records | dense ordinal of its query term << 16;
// idf = the largest tf with min(score / tf, 1) * tf == score (the one-division arm's limit); ubnum = the largest
// numerator over the term frequencies the list holds (z_numerator_bound); zub[e][x] = ubnum / max(shortest field x
// holding the term, query_terms_len).
template <int ZN>  // most records per query: ZN (4) or Z_MAX_LISTS (8)
__global__ __launch_bounds__(64) void k_zplan_arrange(ps_plan_entry* __restrict__ entries, const uint32_t* __restrict__ qbeg,
                                                      const uint32_t* __restrict__ qtl, const uint32_t B, const uint32_t F,
                                                      const uint32_t* __restrict__ maxtf, const uint32_t* __restrict__ minfl,
                                                      double* __restrict__ ubnum, double* __restrict__ zub) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= B) return;
  const uint32_t b = qbeg[q], n = min(qbeg[q + 1] - b, (uint32_t)ZN);
  const uint32_t ql = qtl[q];
  ps_plan_entry en[ZN];
#pragma unroll
  for (int i = 0; i < ZN; ++i)
    if ((uint32_t)i < n) en[i] = entries[b + i];
  // stable insertion sort, score descending (a later record moves ahead of an earlier one only if its score is higher)
  int ord[ZN];
#pragma unroll
  for (int i = 0; i < ZN; ++i) ord[i] = i;
#pragma unroll
  for (int i = 1; i < ZN; ++i)
#pragma unroll
    for (int j = i; j > 0; --j)
      if ((uint32_t)j < n && en[ord[j]].boost > en[ord[j - 1]].boost) { const int t = ord[j]; ord[j] = ord[j - 1]; ord[j - 1] = t; }
  uint32_t s_node[ZN], s_qterm[ZN], s_qt[ZN];
#pragma unroll
  for (int i = 0; i < ZN; ++i) {
    if ((uint32_t)i >= n) break;
    ps_plan_entry e = en[0];
#pragma unroll
    for (int k = 1; k < ZN; ++k)
      if (ord[i] == k) e = en[k];
    uint32_t need = 1, qt = 0, used = 0;
    bool seen = false;
#pragma unroll
    for (int j = 0; j < ZN; ++j) {
      if (j >= i) break;
      if (s_node[j] == e.node) ++need;
      if (s_qterm[j] == e.qterm) { qt = s_qt[j]; seen = true; }
      used |= 1u << s_qt[j];
    }
    if (!seen) {
      qt = 0;
      while (used & (1u << qt)) ++qt;
    }
    s_node[i] = e.node; s_qterm[i] = e.qterm; s_qt[i] = qt;
    const double w = e.boost;
    uint64_t lim = 0;
    for (uint32_t t = 1; t <= 254u; ++t) {
      const double df = (double)t;
      if (!(fmin(w / df, 1.0) * df == w)) break;
      lim = t;
    }
    e.qterm_index = need | (qt << 16);
    e.idf = __longlong_as_double((long long)lim);
    entries[b + i] = e;
    // z_numerator_bound
    const uint32_t mt = maxtf[e.layer], hi = min(mt, 4096u);
    double best = 0.0;
    for (uint32_t t = max(need, 1u); t <= hi; ++t) {
      const double df = (double)t;
      best = fmax(best, fmin(w / df, 1.0) * df);
    }
    if (mt > hi && mt >= need) best = fmax(best, __longlong_as_double(__double_as_longlong(fmax(w, 0.0)) + 2));  // two ulps above
    ubnum[b + i] = best;
    for (uint32_t x = 0; x < F; ++x) {
      const uint32_t mn = minfl[(size_t)e.layer * F + x];
      zub[(size_t)(b + i) * F + x] = mn == 0xFFFFFFFFu ? 0.0 : best / (double)max(mn, ql);
    }
  }
}

// Thread per query: processing order (shortest list first: the long lists are the ones that become non-essential;
// any order is exact), skip thresholds, chunking, candidate slots, bucket totals.
template <int ZN>
__global__ __launch_bounds__(WAVE) void k_zprep_query(const ZPrepParams pp) {
  constexpr int NMAX = ZN;
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  const bool have = q < pp.B;
  const uint32_t b = have ? pp.qbeg[q] : 0u, n = have ? min(pp.qbeg[q + 1] - b, (uint32_t)NMAX) : 0u;
  uint32_t len[NMAX], rank[NMAX];
#pragma unroll
  for (int i = 0; i < NMAX; ++i) len[i] = (uint32_t)i < n ? pp.plan[b + i].len : 0u;
#pragma unroll
  for (int i = 0; i < NMAX; ++i) {
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < NMAX; ++j)
      if ((uint32_t)j < n && j != i && (len[j] < len[i] || (len[j] == len[i] && j < i))) ++r;
    rank[i] = r;
  }
  uint32_t slots = 0;
#pragma unroll
  for (int i = 0; i < NMAX; ++i) {
    if ((uint32_t)i < n) {
      double skip = 0.0, ub = 0.0;
      for (uint32_t x = 0; x < pp.F; ++x) {
        double sum = 0.0;  // in the sorted record order, like the pools
#pragma unroll
        for (int j = 0; j < NMAX; ++j)
          if ((uint32_t)j < n && rank[j] >= rank[i]) sum += pp.zub[(size_t)(b + j) * pp.F + x];
        skip = fmax(skip, sum);
        ub = fmax(ub, pp.zub[(size_t)(b + i) * pp.F + x]);
      }
      DEntry d;
      d.skip_thr = skip;
      d.others = 0.0;
      d.ub = ub;
      d.rank = rank[i];
      d.q = q;
      pp.dentry[b + i] = d;
      const uint32_t c = prep_chunk_of(pp.split_div, pp.chunk_min, len[i]);
      slots += (len[i] + c - 1) / c;
    }
  }
  const uint32_t s0 = wave_add_by_key(&pp.ctl->total_slots, 0u, slots, have && n != 0);
  if (have) { pp.qslot[q] = n ? s0 : 0u; pp.qslot_n[q] = slots; }
  uint32_t sl = s0;
#pragma unroll
  for (int i = 0; i < NMAX; ++i) {  // (wave-uniform trip count: the aggregated atomics need every lane)
    const bool on = (uint32_t)i < n;
    uint32_t nc = 0, nb[Z_LEVELS + 1], len_i = 0;
#pragma unroll
    for (int l = 0; l <= Z_LEVELS; ++l) nb[l] = 0;
    if (on) {
      const ps_plan_entry& en = pp.plan[b + i];
      const uint32_t c = prep_chunk_of(pp.split_div, pp.chunk_min, en.len);
      nc = (en.len + c - 1) / c;
      len_i = en.len;
      const uint32_t sh = en.shift & 0xFFu;
      uint32_t prev = 0;
#pragma unroll
      for (int l = 0; l < Z_LEVELS; ++l) {
        uint32_t na = nc, below = prev;  // (what lies below D_(l-1) lies below D_l)
        const uint32_t dt = pp.dl_tile[l];
        if (dt && (dt & ((1u << sh) - 1u)) == 0u) {
          // postings of the list with doc id < D_l: the table slot that starts at D_l (slots span T << shift documents)
          const uint32_t p0 = min(en.len, pp.table[en.tbl_off + (dt >> sh)]);
          below = max(below, p0 >= en.len ? nc : p0 / c);  // chunks [0, below) end at or before p0
          na = (p0 + c - 1) / c;                            // chunks [na, nc) start at or after p0
        }
        nb[l] = below;
        prev = below;
        pp.nbelow[(size_t)(b + i) * Z_LEVELS + l] = below;
        pp.nabove_from[(size_t)(b + i) * Z_LEVELS + l] = na;
      }
      nb[Z_LEVELS] = nc;
      pp.gen[b + i] = DItemGen{b + i, 0u, c, sl};
      sl += nc;
    }
    uint32_t lo = 0;
#pragma unroll
    for (int ph = 0; ph <= Z_LEVELS; ++ph) {  // chunks [lo, nb[ph]) are in phase ph
      const uint32_t cnt = on && nb[ph] > lo ? nb[ph] - lo : 0u;
      wave_add_by_key_noret(pp.ctl->bucket_total, zprep_bucket((uint32_t)ph, rank[i], len_i), cnt, cnt != 0);
      lo = max(lo, nb[ph]);
    }
  }
  __threadfence();
  uint32_t t = 0;
  if (threadIdx.x == 0) t = atomicAdd(&pp.ctl->ticket, 1u);
  t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
  if (t + 1u == gridDim.x) {  // the wave that finishes last closes the counters
    __threadfence();
    prep_scan_buckets(*pp.ctl);  // (the whole wave: a lane per block of buckets)
  }
}

// Thread per list: its items, each in the bucket of its phase; bits 30-31 of the count: the levels the chunk starts at or above.
__global__ __launch_bounds__(2 * WAVE) void k_zprep_items(const ZPrepParams pp) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool have = i < pp.ne;
  uint32_t nc = 0, len = 0, chunk = 1, first_slot = 0, rank = 0, q_i = 0;
  double skip_i = 0.0;
  uint32_t nb[Z_LEVELS + 1], na[Z_LEVELS];
#pragma unroll
  for (int l = 0; l <= Z_LEVELS; ++l) nb[l] = 0;
#pragma unroll
  for (int l = 0; l < Z_LEVELS; ++l) na[l] = 0;
  if (have) {
    const ps_plan_entry& en = pp.plan[i];
    const DItemGen g = pp.gen[i];
    len = en.len; chunk = g.chunk; first_slot = g.first_slot;
    nc = (len + chunk - 1) / chunk;
#pragma unroll
    for (int l = 0; l < Z_LEVELS; ++l) {
      nb[l] = pp.nbelow[(size_t)i * Z_LEVELS + l];
      na[l] = pp.nabove_from[(size_t)i * Z_LEVELS + l];
    }
    nb[Z_LEVELS] = nc;
    const DEntry de_i = pp.dentry[i];
    rank = de_i.rank; skip_i = de_i.skip_thr; q_i = de_i.q;
  }
  uint32_t at[Z_LEVELS + 1], lo = 0;
#pragma unroll
  for (int ph = 0; ph <= Z_LEVELS; ++ph) {  // chunks [lo, nb[ph]) are in phase ph
    const uint32_t cnt = have && nb[ph] > lo ? nb[ph] - lo : 0u;
    const uint32_t bk = zprep_bucket((uint32_t)ph, rank, len);
    const uint32_t off = wave_add_by_key(pp.ctl->bucket_fill, bk, cnt, cnt != 0);
    at[ph] = cnt ? pp.ctl->bucket_start[bk] + off - lo : 0u;  // (so that chunk j of the phase lands at at[ph] + j)
    lo = max(lo, nb[ph]);
  }
  if (have) {
    for (uint32_t j = 0; j < nc; ++j) {
      const uint32_t pb = j * chunk;
      uint32_t ph = 0, lv = 0;
#pragma unroll
      for (int l = 0; l < Z_LEVELS; ++l) { ph += j >= nb[l] ? 1u : 0u; lv += j >= na[l] ? 1u : 0u; }
      uint32_t a = at[0];
#pragma unroll
      for (int l = 1; l <= Z_LEVELS; ++l) a = ph == (uint32_t)l ? at[l] : a;
      a += j;
      if (a < pp.items_cap) pp.items[a] = DItem{i, pb, min(chunk, len - pb) | (lv << ZITEM_LEVEL_SHIFT), first_slot + j, skip_i, q_i, 0u};
    }
  }
}

// Does a score (or an upper bound of one) still matter?  ts: K documents score >= ts (0 = none known);
// tt: K documents with LOWER doc ids than anything this chunk holds score >= tt (0 = none / not applicable).
__device__ __forceinline__ bool z_beats(const double v, const double ts, const double tt) {
  return v >= ts && (tt == 0.0 || v > tt);
}

// The tie threshold a chunk / trip may use whose documents all lie at or above the `lv` lowest levels: the best of
// gtie[l][q], l < lv (each says: K documents with ids < D_l score at least this).
__device__ __forceinline__ double z_tie_of(const KParams& p, const uint32_t q, const uint32_t lv) {
  unsigned long long best = 0ull;  // (non-negative doubles order like their bit patterns)
#pragma unroll
  for (int l = 0; l < Z_LEVELS; ++l) {
    const unsigned long long v = __hip_atomic_load(&p.gtie[(size_t)l * p.z_tstride + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)l < lv && v > best) best = v;
  }
  return __longlong_as_double((long long)best);
}
__device__ __forceinline__ uint32_t z_level_of(const KParams& p, const uint32_t d) {  // levels l with D_l <= d (the lowest ones)
  uint32_t lv = 0;
#pragma unroll
  for (int l = 0; l < Z_LEVELS; ++l) lv += d >= p.z_dl[l] ? 1u : 0u;
  return lv;
}

// ZN: most lists per query of the batch - DAAT_SMALL_MAX (4: BASELINE config 3) or Z_MAX_LISTS (8: batches with queries of
// five to eight records; the per-list state of a document - packed words, posting, found - is ZN registers wide and the
// queues' per-list words ZN - 1 LDS planes, so the narrow instantiation stays what it was).
template <int F_, bool WC, int ZN = DAAT_SMALL_MAX>  // WC: keep the work counters (ps_work_counters); the serving instantiation carries none
__global__ __launch_bounds__(WAVE * DAAT_WGW) void k_daat_z(const KParams p) {
  static_assert(F_ >= 1 && F_ <= 4, "k_daat_z is instantiated per field count");
  static_assert(ZN == DAAT_SMALL_MAX || ZN == Z_MAX_LISTS, "k_daat_z is instantiated for 4 or 8 lists per query");
  auto cnt = [](const bool b) -> uint32_t { return WC ? (uint32_t)__popcll(__ballot(b)) : 0u; };  // wave-uniform count of lanes where b holds
  constexpr int U = PS_DAAT_ZU;
  constexpr int NE = ZN;
  constexpr int NO = NE - 1;  // other lists of a query, in sorted record order with the own one left out
  constexpr uint32_t QCAP = 128;
  constexpr uint32_t REL_NONE = 0xFFFFFFFFu, REL_MAYBE = 0xFFFFFFFEu;
  __shared__ uint32_t q_d[DAAT_WGW][QCAP];        // survivor queue: doc id
  __shared__ uint32_t q_i[DAAT_WGW][QCAP];        // ... its posting within the own list
  __shared__ uint32_t q_w[F_][DAAT_WGW][QCAP];    // ... the own posting's packed {tf, field length} words
  __shared__ uint32_t q_rel[NO][DAAT_WGW][QCAP];  // ... per other list: posting within that list (bitmap hit), REL_MAYBE (filter), REL_NONE
  // B(m, fl), m = 1..NE - tabulated per item for the narrow instantiation; the wide one (whose per-list queue planes already take
  // 15 KB per workgroup) recomputes the column of its lane whenever a threshold changes: the table's 8 KB are the difference
  // between 3 and 4 waves per SIMD there
  constexpr bool BTAB = PS_DAAT_Z_BTAB && ZN <= DAAT_SMALL_MAX;
  __shared__ double btab[DAAT_WGW][BTAB ? NE : 1][BTAB ? Z_FLN : 1];
  // Reach queue (wave-private LDS ring): the postings that passed the scan's integer test - a few percent of the lanes
  // of a trip - wait here until 64 are together; the first level of the other lists (address arithmetic, filter
  // hashes, the loads, the tightening) then runs with every lane busy instead of four times per trip for a handful
  // of lanes (the scan is VALU-issue bound: profiles/r04_C3_rocprof_summary.txt)
  __shared__ uint32_t r_d[DAAT_WGW][QCAP];
  __shared__ uint32_t r_i[DAAT_WGW][QCAP];
  __shared__ uint32_t r_w[F_][DAAT_WGW][QCAP];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t n_ditems = p.n_ditems_dev ? min(p.n_ditems, *p.n_ditems_dev) : p.n_ditems;
  const uint32_t id = blockIdx.x * (uint32_t)DAAT_WGW + (uint32_t)wave;
  if (id >= n_ditems) return;
  const DItem it = p.ditems[id];
  {
    // most waves only hold a chunk of a list that is already non-essential: they leave at once (every wave for itself -
    // the waves of a workgroup share nothing -, so none waits for its neighbour's loads)
    const double ts = __longlong_as_double((long long)__hip_atomic_load(&p.gthr[it.q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const uint32_t lv = it.count >> ZITEM_LEVEL_SHIFT;
    const double tt = lv ? z_tie_of(p, it.q, lv) : 0.0;
    if (!__builtin_amdgcn_readfirstlane((int)z_beats(it.skip_thr, ts, tt))) {
      if (lane == 0) p.cand_cnt[it.slot] = 0u;
      return;
    }
  }
  const uint32_t e_own = __builtin_amdgcn_readfirstlane(it.entry);
  const DEntry de = p.dentry[e_own];
  const uint32_t q = __builtin_amdgcn_readfirstlane(de.q);
  const uint32_t e0 = p.qbeg[q], ne = min(p.qbeg[q + 1] - e0, (uint32_t)NE);
  const uint32_t own_pos = e_own - e0;
  const uint32_t own_rank = de.rank;
  const uint32_t n_lower = ne - 1u - min(own_rank, ne - 1u);  // lists ranked below the own one
  const double skip_thr = de.skip_thr;
  const uint32_t qtl = p.qterms_len[q];
  const uint64_t own_off = p.plan[e_own].post_off;
  const uint32_t need_own = p.plan[e_own].qterm_index & 0xFFFFu;
  // what the scan needs of the other lists (wave-uniform: scalar registers); the second level reads the rest of
  // their entries when it runs
  uint32_t o_bm[NO], o_rank[NO];
  unsigned long long o_bloom[NO];
#pragma unroll
  for (int k = 0; k < NO; ++k) {
    o_bm[k] = 0xFFFFFFFFu; o_rank[k] = 0xFFFFFFFFu; o_bloom[k] = NO_BLOOM;
    if ((uint32_t)k + 1u < ne) {
      const uint32_t j = e0 + (uint32_t)k + ((uint32_t)k >= own_pos ? 1u : 0u);
      const ps_plan_entry& en = p.plan[j];
      o_bm[k] = en.bm_off;
      o_rank[k] = p.dentry[j].rank;
      if (en.bm_off == 0xFFFFFFFFu && p.layer_bloom) o_bloom[k] = p.layer_bloom[en.layer];
    }
  }
  // ---- bound table of this item: B(m, fl) for m = 1..ne records and fl = lane ----
  if constexpr (BTAB) {
    double env[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) env[j] = (uint32_t)j < ne ? p.z_ubnum[e0 + j] : 0.0;
#pragma unroll
    for (int j = NE - 2; j >= 0; --j) env[j] = fmax(env[j], env[j + 1]);  // non-increasing envelope along the sorted order
    const uint32_t den_u = (uint32_t)lane > qtl ? (uint32_t)lane : qtl;
    const double den = (double)den_u;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      if ((uint32_t)j < ne) acc += env[j] / den;
      btab[wave][j][lane] = acc;
    }
  }
  // wave-uniform: the longest field that can still matter with m contributing records - under both thresholds
  // (flmax: documents at or above tie_from) and under the strict one alone (flmax_s: queued documents below it)
  int flmax[NE + 1], flmax_s[NE + 1];
  auto set_flmax = [&](const double ts, const double tt) __attribute__((always_inline)) {
    flmax[0] = z_beats(0.0, ts, tt) ? Z_ALL : Z_NONE;
    flmax_s[0] = z_beats(0.0, ts, 0.0) ? Z_ALL : Z_NONE;
    double col[NE];  // (wide instantiation) the lane's column of the table, by the table's own operations in the same order
    if constexpr (!BTAB) {
      double env[NE];
#pragma unroll
      for (int j = 0; j < NE; ++j) env[j] = (uint32_t)j < ne ? p.z_ubnum[e0 + j] : 0.0;
#pragma unroll
      for (int j = NE - 2; j >= 0; --j) env[j] = fmax(env[j], env[j + 1]);
      const uint32_t den_u = (uint32_t)lane > qtl ? (uint32_t)lane : qtl;
      const double den = (double)den_u;
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        if ((uint32_t)j < ne) acc += env[j] / den;
        col[j] = acc;
      }
    }
#pragma unroll
    for (int m = 1; m <= NE; ++m) {
      double bv;
      if constexpr (BTAB) bv = btab[wave][m - 1][lane]; else bv = col[m - 1];
      const int n_ok = (int)__popcll(__ballot(z_beats(bv, ts, tt)));  // B is non-increasing in fl: the lanes that pass are a prefix
      flmax[m] = n_ok == WAVE ? Z_ALL : n_ok - 1;
      const int n_s = (int)__popcll(__ballot(z_beats(bv, ts, 0.0)));
      flmax_s[m] = n_s == WAVE ? Z_ALL : n_s - 1;
    }
  };
  TopK tk;
  tk.s = -1.0; tk.d = 0xFFFFFFFFu; tk.n = 0; tk.thr_s = 0.0; tk.thr_d = 0;
  double published = 0.0;
  const uint32_t end = it.begin + (it.count & ZITEM_COUNT);
  WorkStats ws;
  uint32_t q_head = 0, q_n = 0;  // wave-uniform
  uint32_t r_head = 0, r_n = 0;  // (the reach queue)
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  double theta_s = 0.0, theta_t = 0.0;  // theta_t: the tie threshold of the current trip (0 while it is not entirely at or above D_0)
  uint32_t tie_from = 0xFFFFFFFFu;      // ... which holds for the documents with ids >= tie_from (the boundary it was published below)
  uint32_t pub_level = 0;               // wave-uniform: every document scanned so far has id < D_pub_level (doc ids ascend along the chunk)

  // Second level + the pools in the sorted record order + the top-K offer for the first `count` (<= 64) queued
  // documents, one per lane.
  auto process = [&](const uint32_t count) __attribute__((always_inline)) {
    const uint32_t at = (q_head + (uint32_t)lane) & (QCAP - 1u);
    bool ok = (uint32_t)lane < count;
    const uint32_t d = ok ? q_d[wave][at] : 0u;
    const uint32_t own_i = ok ? q_i[wave][at] : 0u;
    uint32_t w[NE][F_];   // packed words of the document's posting in list j (0: not in the list)
    uint32_t rel[NE];     // ... its posting within list j
    bool found[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      found[j] = false; rel[j] = 0u;
#pragma unroll
      for (int x = 0; x < F_; ++x) w[j][x] = 0u;
      if ((uint32_t)j >= ne) continue;
      if ((uint32_t)j == own_pos) {
        found[j] = ok;
        rel[j] = own_i;
#pragma unroll
        for (int x = 0; x < F_; ++x) w[j][x] = ok ? q_w[x][wave][at] : 0u;
        continue;
      }
      const int k = j - ((uint32_t)j > own_pos ? 1 : 0);  // (own_pos is wave-uniform)
      uint32_t loc = REL_NONE;
#pragma unroll
      for (int kk = 0; kk < NO; ++kk)
        if (kk == k) loc = ok ? q_rel[kk][wave][at] : REL_NONE;
      const ps_plan_entry& en = p.plan[e0 + j];
      const uint64_t off = en.post_off;
      if (en.bm_off != 0xFFFFFFFFu) {
        found[j] = loc != REL_NONE;
        rel[j] = found[j] ? loc : 0u;
      } else {
        // a sparse list whose filter said "maybe": its table slot holds a handful of postings - up to 4 doc ids per step
        const uint32_t* docs = p.doc + off;
        bool open = loc != REL_NONE;
        uint32_t lo = 0, hi = 0;
        if (open) {
          const uint32_t slot = (d >> p.t_log2) >> (en.shift & 0xFFu);
          lo = p.table[en.tbl_off + slot];
          hi = p.table[en.tbl_off + slot + 1];
        }
        ws.probe += 2u * cnt(open);
        open = open && lo < hi;
        while (__any(open)) {
          uint32_t v[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const bool rd = open && lo + t < hi;
            v[t] = rd ? docs[lo + t] : 0xFFFFFFFFu;
            ws.probe += cnt(rd);
          }
          if (open) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (v[t] == d) { found[j] = true; rel[j] = lo + t; }
            open = !found[j] && v[3] < d && lo + 4 < hi;
            lo += 4;
          }
        }
      }
      ws.hit += cnt(found[j]);
      if (__any(found[j])) {
        uint32_t t[F_];
#pragma unroll
        for (int x = 0; x < F_; ++x) t[x] = 0u;
        if (found[j]) tfl_load<F_>(p, off + rel[j], t);
#pragma unroll
        for (int x = 0; x < F_; ++x) w[j][x] = found[j] ? t[x] : 0u;
      }
      // a document is evaluated from its highest-ranked list only
      if (found[j] && p.dentry[e0 + j].rank < own_rank) ok = false;
    }
    // pools (zero_to_one.rs:96-121), the document's score = the best pool (:122)
    double score = 0.0;
#pragma unroll
    for (int x = 0; x < F_; ++x) {
      // the field length is the document's: every posting of the document carries it
      uint32_t flu = (ok ? q_w[x][wave][at] : 0u) & TFL_FL_ESC;
      if (__any(ok && flu == TFL_FL_ESC)) {
        if (ok && flu == TFL_FL_ESC) flu = p.fl[(uint64_t)x * p.P + own_off + own_i];
      }
      const double den = (double)(flu > qtl ? flu : qtl);
      double pool = 0.0;
      uint32_t consumed = 0u;
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        if ((uint32_t)j >= ne) continue;
        const ps_plan_entry& en = p.plan[e0 + j];
        const uint32_t l_need = en.qterm_index & 0xFFFFu;          // occurrence rank of the node among the sorted records (>= 1)
        const uint32_t l_qbit = 1u << ((en.qterm_index >> 16) & 31u);  // consumed_index bit of the record's query term
        const uint32_t l_tfx = (uint32_t)__double2loint(en.idf);   // largest tf with min(score / tf, 1) * tf == score (host, same arithmetic)
        const double l_s = en.boost;                               // ScoreByTerm::score
        uint32_t tfu = w[j][x] >> 24;
        if (__any(found[j] && tfu == TFL_TF_ESC)) {
          if (found[j] && tfu == TFL_TF_ESC) tfu = p.tf[(uint64_t)x * p.P + en.post_off + rel[j]];
        }
        const bool take = ok && found[j] && tfu >= l_need && tfu > 0u && !(consumed & l_qbit);
        if (__any(take)) {
          double num = l_s;
          if (__any(take && tfu > l_tfx)) {
            const double df = (double)(tfu ? tfu : 1u);
            num = fmin(l_s / df, 1.0) * df;
          }
          const double c = num / den;
          pool += take ? c : 0.0;
          consumed |= take ? l_qbit : 0u;
        }
      }
      score = fmax(pool, score);
    }
    // (the queue may hold documents of a trip below D0 next to documents of one above it: the tie rule is per document)
    const bool offer = ok && z_beats(score, theta_s, d >= tie_from ? theta_t : 0.0);
    ws.offer += cnt(offer);
    if (__any(offer)) topk_offer(tk, p.K, lane, offer, score, d, theta_s);
    q_head = (q_head + count) & (QCAP - 1u);
    q_n -= count;
    if (tk.n == p.K && tk.thr_s > published && tk.thr_s > theta_s) {
      published = tk.thr_s;
      if (lane == 0) {
        atomicMax(&p.gthr[q], (unsigned long long)__double_as_longlong(tk.thr_s));
        if (pub_level < (uint32_t)Z_LEVELS)
          atomicMax(&p.gtie[(size_t)pub_level * p.z_tstride + q], (unsigned long long)__double_as_longlong(tk.thr_s));
      }
    }
  };

  // First level of every other list for the first `count` (<= 64) documents of the reach queue, one per lane, all
  // loads in flight together; then what it tells - how many lists can still count, exact membership of bitmap lists -
  // against the current thresholds; the survivors move on to the survivor queue.
  auto level1 = [&](const uint32_t count) __attribute__((always_inline)) {
    const uint32_t rat = (r_head + (uint32_t)lane) & (QCAP - 1u);
    const bool on = (uint32_t)lane < count;
    const uint32_t dq = on ? r_d[wave][rat] : 0u, iq = on ? r_i[wave][rat] : 0u;
    uint32_t wq[F_];
#pragma unroll
    for (int x = 0; x < F_; ++x) wq[x] = on ? r_w[x][wave][rat] : 0u;
    r_head = (r_head + count) & (QCAP - 1u);
    r_n -= count;
    ws.reached += cnt(on);
    uint2 fl[NO];
#pragma unroll
    for (int k = 0; k < NO; ++k) {
      fl[k] = make_uint2(0u, 0u);
      if ((uint32_t)k + 1u >= ne) continue;
      ws.cell += cnt(on);
      if (o_bm[k] != 0xFFFFFFFFu) {
        if (on) fl[k] = *reinterpret_cast<const uint2*>(p.bits + (uint64_t)o_bm[k] + 2 * (uint64_t)(dq >> 5));
      } else if (o_bloom[k] != NO_BLOOM) {
        uint64_t wi;
        unsigned long long mk;
        bloom_probe(dq, o_bloom[k], wi, mk);
        const unsigned long long wd = on ? p.bloom[wi] : 0ull;
        fl[k].x = (on && (wd & mk) == mk) ? 1u : 0u;  // maybe
      } else {
        fl[k].x = on ? 1u : 0u;  // no filter: ask the table
      }
    }
    bool alive = on;
    uint32_t hits = 0;
    uint32_t loc[NO];
#pragma unroll
    for (int k = 0; k < NO; ++k) {
      loc[k] = REL_NONE;
      if ((uint32_t)k + 1u >= ne) continue;
      if (o_bm[k] != 0xFFFFFFFFu) {
        const uint32_t bit = dq & 31u;
        const bool hit = (fl[k].x >> bit) & 1u;
        if (hit) {
          loc[k] = fl[k].y + (uint32_t)__popc(fl[k].x & ((1u << bit) - 1u));
          if (o_rank[k] < own_rank) alive = false;  // evaluated from its highest-ranked list only
          else ++hits;
        }
      } else if (fl[k].x != 0u) {
        loc[k] = REL_MAYBE;  // (a "maybe" of a higher-ranked list adds nothing to the bound: found there, the document is cancelled)
        if (o_rank[k] > own_rank) ++hits;
      }
    }
    // (the queue may hold documents of a trip below tie_from next to documents above it: the tie rule is per document)
    const bool tie = dq >= tie_from;
    bool any = false;
#pragma unroll
    for (int x = 0; x < F_; ++x) {
      const uint32_t tfu = wq[x] >> 24;
      const uint32_t m = hits + ((tfu >= need_own && tfu > 0u) ? 1u : 0u);
      int fm = tie ? flmax[0] : flmax_s[0];
#pragma unroll
      for (int k = 1; k <= NE; ++k) fm = m == (uint32_t)k ? (tie ? flmax[k] : flmax_s[k]) : fm;
      any = any || (int)(wq[x] & TFL_FL_ESC) <= fm;
    }
    alive = alive && any;
    // ---- survivors wait in the survivor queue until 64 are together ----
    const unsigned long long mm = __ballot(alive);
    if (mm) {
      if (alive) {
        const uint32_t at = (q_head + q_n + (uint32_t)__popcll(mm & lt)) & (QCAP - 1u);
        q_d[wave][at] = dq;
        q_i[wave][at] = iq;
#pragma unroll
        for (int x = 0; x < F_; ++x) q_w[x][wave][at] = wq[x];
#pragma unroll
        for (int k = 0; k < NO; ++k)
          if ((uint32_t)k + 1u < ne) q_rel[k][wave][at] = loc[k];
      }
      q_n += (uint32_t)__popcll(mm);
      if (q_n >= (uint32_t)WAVE) process((uint32_t)WAVE);
    }
  };

#ifdef PS_ITEM_TRACE
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  uint32_t n_trips = 0;
#endif
  bool first = true, essential = true;
  // The own postings of a trip (doc id + packed words; out-of-range lanes re-read the chunk's last posting).  With
  // PS_DAAT_ZPF the postings of trip t + 1 are requested before trip t's lookups go out: a trip is otherwise two or
  // three memory round trips one after the other (own postings -> first level -> survivors' second level), and this
  // kernel - a scan with few lookups - is bound by that chain, not by a throughput roof.
  auto load_trip = [&](const uint32_t i0, uint32_t (&dd)[U], uint32_t (&ww)[U][F_]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = i0 + u * WAVE + lane;
      const uint32_t pi = i < end ? i : end - 1;
      dd[u] = p.doc[own_off + pi];
      tfl_load<F_>(p, own_off + pi, ww[u]);
    }
  };
  uint32_t d[U], wv[U][F_];
  if (PS_DAAT_ZPF) load_trip(it.begin, d, wv);
  for (uint32_t i0 = it.begin; i0 < end && essential; i0 += WAVE * U) {
#ifdef PS_ITEM_TRACE
    ++n_trips;
#endif
    const unsigned long long sbits = __hip_atomic_load(&p.gthr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long tl[Z_LEVELS];
#pragma unroll
    for (int l = 0; l < Z_LEVELS; ++l) tl[l] = __hip_atomic_load(&p.gtie[(size_t)l * p.z_tstride + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t dn[U], wn[U][F_];  // (PS_DAAT_ZPF) the next trip's postings, in flight while this trip is worked on
    if (PS_DAAT_ZPF) {
      if (i0 + WAVE * U < end) load_trip(i0 + WAVE * U, dn, wn);
    } else {
      load_trip(i0, d, wv);
    }
    bool inr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) inr[u] = i0 + u * WAVE + lane < end;
    {
      // the trip's place among the levels: its first posting (lane 0 of slot 0) is its lowest doc id, its last valid
      // posting its highest
      const uint32_t lv_first = z_level_of(p, (uint32_t)__builtin_amdgcn_readfirstlane((int)d[0]));
      const uint32_t last = min(end - i0, (uint32_t)(WAVE * U)) - 1u;
      uint32_t d_last = 0;
#pragma unroll
      for (int u = 0; u < U; ++u)
        if ((last >> 6) == (uint32_t)u) d_last = readlane_u32(d[u], (int)(last & 63u));
      pub_level = max(pub_level, z_level_of(p, d_last));
      unsigned long long tbits = 0ull;
      uint32_t from = 0xFFFFFFFFu;
#pragma unroll
      for (int l = 0; l < Z_LEVELS; ++l) {
        // (wave-uniform: one load instruction returned one value to the whole wave; the casts through uint32_t keep the
        // int readfirstlane returns from sign-extending into the upper half)
        const unsigned long long v = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tl[l]) |
                                     ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(tl[l] >> 32)) << 32);
        if ((uint32_t)l < lv_first) { if (v > tbits) tbits = v; from = p.z_dl[l]; }
      }
      tie_from = from;
      const double ts = __hiloint2double(__builtin_amdgcn_readfirstlane((int)(sbits >> 32)), __builtin_amdgcn_readfirstlane((int)(uint32_t)sbits));
      const double tt = __longlong_as_double((long long)tbits);
      if (first || ts != theta_s || tt != theta_t) {
        theta_s = ts; theta_t = tt;
        set_flmax(ts, tt);
        first = false;
      }
    }
    essential = z_beats(skip_thr, theta_s, theta_t);  // false: the whole list has become non-essential
    const uint32_t n_in = min(end - i0, (uint32_t)(WAVE * U));
    if (!essential) {  // (its doc ids and words were requested with the thresholds: booked, then out)
      if (WC) ws.probe += n_in * (1u + (uint32_t)F_);
      break;
    }
    if (p.alive != nullptr) {  // delta removals
      uint32_t aw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) aw[u] = p.alive[d[u] >> 5];
#pragma unroll
      for (int u = 0; u < U; ++u) inr[u] = inr[u] & (bool)((aw[u] >> (d[u] & 31u)) & 1u);
    }
    // ---- first bound test, integers only: per field, could the own record (if it counts) + every other list
    // still beat the thresholds? ----
    // A document is evaluated from its highest-ranked list, so one that is evaluated HERE is in no list ranked above
    // the own one: only the n_lower lists ranked below can add to it (a document that does sit in a higher-ranked
    // list is cancelled below if it gets that far, and is evaluated there with the bound of that list's items).
    int fmw = Z_NONE, fmo = Z_NONE;  // (flmax[] indexed by a wave-uniform value, without dynamic register indexing)
#pragma unroll
    for (int m = 0; m <= NE; ++m) {
      if ((uint32_t)m == n_lower + 1u) fmw = flmax[m];
      if ((uint32_t)m == n_lower) fmo = flmax[m];
    }
    if (WC) ws.scanned += n_in;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bool any = false;
#pragma unroll
      for (int x = 0; x < F_; ++x) {
        const uint32_t tfu = wv[u][x] >> 24;
        const int flu = (int)(wv[u][x] & TFL_FL_ESC);
        const bool c = tfu >= need_own && tfu > 0u;  // the own record can count for field x
        any = any || flu <= (c ? fmw : fmo);
      }
      const bool rch = inr[u] && any;
      // ---- the postings that passed wait in the reach queue until 64 are together ----
      const unsigned long long mm = __ballot(rch);
      if (mm) {
        if (rch) {
          const uint32_t at = (r_head + r_n + (uint32_t)__popcll(mm & lt)) & (QCAP - 1u);
          r_d[wave][at] = d[u];
          r_i[wave][at] = i0 + (uint32_t)u * WAVE + (uint32_t)lane;  // (reached: within the chunk)
#pragma unroll
          for (int x = 0; x < F_; ++x) r_w[x][wave][at] = wv[u][x];
        }
        r_n += (uint32_t)__popcll(mm);
        if (r_n >= (uint32_t)WAVE) level1((uint32_t)WAVE);
      }
    }
    if (PS_DAAT_ZPF && i0 + WAVE * U < end) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        d[u] = dn[u];
#pragma unroll
        for (int x = 0; x < F_; ++x) wv[u][x] = wn[u][x];
      }
    }
  }
  while (r_n) level1(min(r_n, (uint32_t)WAVE));
  while (q_n) process(min(q_n, (uint32_t)WAVE));
  if ((uint32_t)lane < p.K) {
    const uint64_t o = (uint64_t)it.slot * p.K + lane;
    const bool ok = (uint32_t)lane < tk.n;
    p.cand_score[o] = ok ? tk.s : 0.0;
    p.cand_doc[o] = ok ? tk.d : 0xFFFFFFFFu;
    if (lane == 0) p.cand_cnt[it.slot] = tk.n;
  }
#ifdef PS_ITEM_TRACE
  if (p.item_trace != nullptr && lane == 0) {  // (profiling builds: tools/item_trace.py; "rank" = the chunk's level here)
    unsigned long long* tr = p.item_trace + (size_t)id * 4;
    tr[0] = t_start; tr[1] = __builtin_amdgcn_s_memrealtime(); tr[2] = (unsigned long long)n_trips | ((unsigned long long)(it.count >> ZITEM_LEVEL_SHIFT) << 32);
    tr[3] = ws.scanned | ((unsigned long long)ws.reached << 32);
  }
#endif
  if (WC && lane == 0) {
    unsigned long long* w = p.wstats + (size_t)(blockIdx.x & (WS_SLOTS - 1u)) * WS_WORDS;
    atomicAdd(&w[WS_ITEMS_RUN], 1ull);
    if (ws.scanned) atomicAdd(&w[WS_Z_SCANNED], (unsigned long long)ws.scanned);
    if (ws.reached) atomicAdd(&w[WS_REACHED], (unsigned long long)ws.reached);
    if (ws.cell) atomicAdd(&w[WS_CELL], (unsigned long long)ws.cell);
    if (ws.probe) atomicAdd(&w[WS_PROBE], (unsigned long long)ws.probe);
    if (ws.hit) atomicAdd(&w[WS_Z_HIT], (unsigned long long)ws.hit);
    if (ws.offer) atomicAdd(&w[WS_OFFER], (unsigned long long)ws.offer);
  }
}

}  // namespace ps
