// ps_kernels_plan.hpp — N2, the device-side query planner (k_plan, k_plan_scan: tokenise, trie lookup, prefix expansion, before_each)
// and the small data-movement kernels (k_upload, k_pack_tfl, k_pack_results).  Part of ps_kernels.hpp.
#pragma once
#include "ps_kernels_common.hpp"

namespace ps {

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
