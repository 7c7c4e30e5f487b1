// ps_kernels_daat.hpp — K1d, exact top-K with dynamic pruning for BM25: Bloom filters of the sparse lists, score planes, lookups,
// k_daat (any plan up to 64 lists, multi-expansion arm), k_daat_small (plans of <= 4 lists: C2 / C4), K3d k_merge_items.
// (zero_to_one's kernel of this family: ps_z21_daat.hpp; the device-side preparation: ps_prep_kernels.hpp.)  Part of ps_kernels.hpp.
#pragma once
#include "ps_kernels_score.hpp"

namespace ps {

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

}  // namespace ps
