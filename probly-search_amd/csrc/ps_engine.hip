// ps_engine.hip — HIP kernels (gfx950 / CDNA4) and batch execution for the query-scoring path.
//
// Replaces HOT LOOP #1 + #2 of Index::query (src/query.rs:45, 61-89 of probly-search 2.0.1), the
// per-posting ScoreCalculator::score of BM25 (src/score/default/bm25.rs:60-93) and zero_to_one
// (src/score/default/zero_to_one.rs:44-126), max_score_merger (src/query.rs:150-164) and the
// result materialisation + sort (src/query.rs:97-105) with:
//
//   K1  k_bm25      one wavefront (64 lanes) per (query, run of S doc tiles).  The wave owns an
//                   LDS tile of T f64 accumulators (+ u16 visited tags); for every plan entry, in
//                   plan order, it streams the tile's slice of the doc-sorted posting list with
//                   coalesced u32 loads (doc, tf[F], fl[F]), evaluates the BM25 term in f64 in the
//                   reference's exact association (no FMA contraction) and applies the
//                   add / max / assign merge to its own LDS slot.  A list holds a document at
//                   most once, and LDS operations of one wave execute in order, so there are no
//                   atomics and no barriers, and the result is bit-reproducible.
//                   Epilogue per tile: scan the T slots, offer present documents to a wave-wide
//                   top-K kept in registers (lane i = i-th best; ballot + popcount rank insert).
//   K2  k_z21       same traversal; records per (doc, distinct node) tf vectors in LDS, then one
//                   lane per document runs zero_to_one's greedy finalize in registers.
//   K3  k_merge     one wave per query merges the per-run top-K lists, maps doc id -> key.
//
// Everything here is memory/VALU-f64 bound sparse gather-reduce: no MFMA on purpose.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>

#include "ps_engine.hpp"
#include "ps_sort.hpp"

namespace ps {

#define PS_HIP(call)                                                                            \
  do {                                                                                          \
    hipError_t _e = (call);                                                                     \
    if (_e != hipSuccess)                                                                       \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #call); \
  } while (0)

constexpr int MAX_F = 8;
constexpr int WAVE = 64;
#ifndef PS_UNROLL
#define PS_UNROLL 2
#endif
#ifndef PS_WG_WAVES
#define PS_WG_WAVES 4
#endif
#ifndef PS_G
#define PS_G 4
#endif
#ifndef PS_FU
#define PS_FU 1
#endif
constexpr int UNROLL = PS_UNROLL;      // postings per lane per trip of the streaming loop
constexpr int WG_WAVES = PS_WG_WAVES;  // waves per workgroup of K1; each wave owns its own LDS tile
constexpr int LUT_TF = 16;   // LUT columns: term frequency 0..15
#ifndef PS_ABLATE_BUILD
#define PS_ABLATE_BUILD 0  // profiling builds only: honour KParams::ablate in the hot loops
#endif

struct RowDesc {  // one hot (list, idf, expansion_boost) combination of the batch
  uint64_t post_off;
  uint32_t len;
  uint32_t _pad;
  double idf, eb;
};

constexpr uint32_t DENSE_FLAG = 0x80000000u;  // ps_plan_entry::shift bit 31: entry reads dense row `node`

struct KParams {
  const uint32_t* doc;
  const uint32_t* tf;
  const uint32_t* fl;
  const uint32_t* table;
  const uint64_t* keys;
  const ps_plan_entry* plan;
  const uint32_t* qbeg;
  const uint32_t* qterms_len;  // zero_to_one
  const uint32_t* gen_queries; // zero_to_one: the n_general queries k_z21 has to run
  const uint32_t* qflags;      // zero_to_one: bit 0 = "simple" query (k_score<MODE_Z21S> owns it)
  uint32_t slice_bytes;        // per-wave LDS for the table slices (0 = look ranges up in global memory)
  const uint32_t* zorder;      // zero_to_one: per query, entry indices sorted by (score desc, plan order)
  uint64_t P;
  uint32_t B, n_tiles, T, S, n_super, K, n_docs, F, max_qterms, z_nodes, z_tile;
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
  uint32_t* work_counter;    // next (query, run) item for the persistent waves of k_score
  unsigned long long* gthr;  // [B] bits of the best published local K-th score per query (0 = none)
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
};

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
__global__ __launch_bounds__(256) void k_dense_rows(const KParams p, double* rows) {
  const RowDesc rd = p.row_desc[blockIdx.y];
  double* row = rows + (uint64_t)blockIdx.y * p.row_planes * p.row_stride;
  if (p.row_mode != 0) {
    // zero_to_one.rs:117-120 per field: (min(score/tf, 1)*tf) / max(field_length, all_query_terms_len)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rd.len; i += gridDim.x * blockDim.x) {
      const uint64_t pi = rd.post_off + i;
      const uint32_t d = p.doc[pi];
      const uint32_t qtl = rd._pad & 0xFFFFu, need = rd._pad >> 16;
      for (uint32_t x = 0; x < p.F; ++x) {
        const uint32_t tfu = p.tf[(uint64_t)x * p.P + pi];
        if (tfu >= need) {
          const uint32_t flu = p.fl[(uint64_t)x * p.P + pi];
          const double df = (double)tfu;
          row[(uint64_t)x * p.row_stride + d] = fmin(rd.idf / df, 1.0) * df / (double)(flu > qtl ? flu : qtl);
        }
      }
    }
    return;
  }
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rd.len; i += gridDim.x * blockDim.x) {
    const uint64_t pi = rd.post_off + i;
    double s = 0.0;
    for (uint32_t x = 0; x < p.F; ++x) {
      const uint32_t tfu = p.tf[(uint64_t)x * p.P + pi];
      if (tfu > 0) s += bm25_tfn(p, x, tfu, p.fl[(uint64_t)x * p.P + pi]) * rd.idf * p.boost[x] * rd.eb;
    }
    row[p.doc[pi]] = s;
  }
}

// zero_to_one rows: plane x of the row goes to accumulator plane x of the tile ([F][T] in LDS).
// mask_bit != 0: the query has several expansions per query term; a (doc, field) takes the row
// value only if its consumed-query-term mask does not hold the bit yet (zero_to_one.rs:101-103).
template <bool MASKS>
__device__ __forceinline__ void dense_apply_z(const KParams& p, double* acc, uint32_t* zmask, const int lane,
                                              const uint32_t row, const uint32_t tile_base, const uint32_t mask_bit) {
  for (uint32_t x = 0; x < p.F; ++x) {
    const double* r = p.rows + ((uint64_t)row * p.F + x) * p.row_stride + tile_base;
    constexpr int CH = 4;
    for (uint32_t c0 = 0; c0 < p.T; c0 += CH * 2 * WAVE) {
      double2 v[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k)
        if (c0 + k * 2 * WAVE < p.T) v[k] = *reinterpret_cast<const double2*>(r + c0 + k * 2 * WAVE + 2 * lane);
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        if (c0 + k * 2 * WAVE < p.T) {
          const uint32_t i = c0 + k * 2 * WAVE + 2 * lane;
          bool t0 = v[k].x > 0.0, t1 = v[k].y > 0.0;
          if (MASKS && mask_bit) {
            uint2* zm = reinterpret_cast<uint2*>(zmask + x * p.T + i);
            const uint2 mk = *zm;
            t0 = t0 && !(mk.x & mask_bit);
            t1 = t1 && !(mk.y & mask_bit);
            if (t0 || t1) *zm = make_uint2(mk.x | (t0 ? mask_bit : 0u), mk.y | (t1 ? mask_bit : 0u));
          }
          if (t0) __hip_atomic_fetch_add(&acc[x * p.T + i], v[k].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          if (t1) __hip_atomic_fetch_add(&acc[x * p.T + i + 1], v[k].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
      }
    }
  }
}

// Merge one dense row's slice for this tile into the wave's LDS tile (same merge rules as
// score_trip; a row value > 0 <=> the list holds that document).
template <bool TAGS>
__device__ __forceinline__ void dense_apply(const KParams& p, double* acc, uint16_t* tag, const int lane,
                                            const uint32_t row, const uint32_t tile_base, const uint16_t mytag) {
  const double* r = p.rows + (uint64_t)row * p.row_stride + tile_base;
  constexpr int CH = 4;  // 4 x 128 documents per batch of 16-byte loads (1 KiB per load instruction)
  for (uint32_t c0 = 0; c0 < p.T; c0 += CH * 2 * WAVE) {
    double2 v[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k)
      if (c0 + k * 2 * WAVE < p.T) v[k] = *reinterpret_cast<const double2*>(r + c0 + k * 2 * WAVE + 2 * lane);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (c0 + k * 2 * WAVE < p.T) {
        const uint32_t i = c0 + k * 2 * WAVE + 2 * lane;
        if (TAGS) {
          const double c0v = acc[i], c1v = acc[i + 1];
          const uint16_t t0 = tag[i], t1 = tag[i + 1];
          if (v[k].x > 0.0) {
            acc[i] = (c0v > 0.0) ? (t0 == mytag ? fmax(c0v, v[k].x) : c0v + v[k].x) : v[k].x;
            tag[i] = mytag;
          }
          if (v[k].y > 0.0) {
            acc[i + 1] = (c1v > 0.0) ? (t1 == mytag ? fmax(c1v, v[k].y) : c1v + v[k].y) : v[k].y;
            tag[i + 1] = mytag;
          }
        } else {
          if (v[k].x > 0.0) __hip_atomic_fetch_add(&acc[i], v[k].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          if (v[k].y > 0.0) __hip_atomic_fetch_add(&acc[i + 1], v[k].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
      }
    }
  }
}

enum { MODE_BM25 = 0, MODE_Z21S = 1 };

struct EntryC {      // wave-uniform per-entry constants (SGPRs)
  uint64_t post_off;
  uint32_t shift;
  uint32_t tag;      // BM25: visited tag of the entry's query term for the current tile
  double w0;         // BM25: idf              | Z21S: ScoreByTerm::score
  double w1;         // BM25: expansion_boost  | Z21S: unused
};

template <int F_, int U>
__device__ __forceinline__ void load_trip(const KParams& p, const int lane, const uint64_t post_off, const uint32_t i0,
                                          const uint32_t re, uint32_t (&dv)[U], uint32_t (&tfv)[U][F_ ? F_ : MAX_F],
                                          uint32_t (&flv)[U][F_ ? F_ : MAX_F]) {
  constexpr int FA = F_ ? F_ : MAX_F;
  const uint32_t F = F_ ? (uint32_t)F_ : p.F;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t i = i0 + u * WAVE + lane;
    const uint64_t pi = post_off + (i < re ? i : re - 1);  // clamp: always a valid posting
    dv[u] = p.doc[pi];
#pragma unroll
    for (int x = 0; x < FA; ++x) {
      if ((uint32_t)x < F) {
        tfv[u][x] = p.tf[(uint64_t)x * p.P + pi];
        flv[u][x] = p.fl[(uint64_t)x * p.P + pi];
      }
    }
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
                                           const uint32_t (&tfv)[U][F_ ? F_ : MAX_F],
                                           const uint32_t (&flv)[U][F_ ? F_ : MAX_F], const EntryC& ec,
                                           const uint32_t qtl) {
  constexpr int FA = F_ ? F_ : MAX_F;
  const uint32_t F = F_ ? (uint32_t)F_ : p.F;
  if (PS_ABLATE_BUILD && (p.ablate & 2u)) {  // profiling only: loads stay alive, no scoring
#pragma unroll
    for (int u = 0; u < U; ++u)
      if ((dv[u] ^ tfv[u][0] ^ flv[u][0]) == 0xFFFFFFF1u) acc[0] = 1.0;
    return;
  }
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
          tfn[u][x] = lut[in_lut ? tfu * p.lut_stride + p.lut_base[x] + flu : 0u];
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
            const uint32_t tfu = tfv[u][x], flu = flv[u][x];
            if (!(tfu < (uint32_t)LUT_TF && flu < p.lut_cap[x]))
              tfn[u][x] = bm25_tfn_cold(p.k1, p.k1p1, p.one_minus_b, p.b, p.avg[x], tfu, flu);
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
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int x = 0; x < FA; ++x) {
        if ((uint32_t)x < F) {
          const uint32_t tfu = tfv[u][x], flu = flv[u][x];
          const double df = (double)tfu;
          const uint32_t den = flu > qtl ? flu : qtl;
          const double c = fmin(ec.w0 / df, 1.0) * df / (double)den;
          // ec.tag = occurrence rank of the node (low 16 bits, >= 1: the pool rule) | query-term ordinal
          bool take = ok[u] && tfu >= (ec.tag & 0xFFFFu);
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
    uint32_t dv[UN], tfv[UN][FA], flv[UN][FA];
    uint32_t dn[UN], tfnx[UN][FA], flnx[UN][FA];
    load_trip<F_, UN>(p, lane, ec.post_off, i0, re, dv, tfv, flv);
    while (re - i0 >= (uint32_t)(UN * WAVE)) {
      const uint32_t nx = i0 + UN * WAVE;
      const bool more = re - nx >= (uint32_t)(UN * WAVE);
      if (more) load_trip<F_, UN>(p, lane, ec.post_off, nx, re, dn, tfnx, flnx);
      score_trip<MODE, F_, TAGS, UN>(p, lut, acc, tag, lane, tile_base, i0, re, dv, tfv, flv, ec, qtl);
      if (more) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          dv[u] = dn[u];
#pragma unroll
          for (int x = 0; x < FA; ++x)
            if ((uint32_t)x < F) { tfv[u][x] = tfnx[u][x]; flv[u][x] = flnx[u][x]; }
        }
      }
      i0 = nx;
    }
  }
  // tail (< UN*64 postings): one masked trip when it is long (all loads in flight together), one
  // 64-wide trip when it is short (no empty lane slots to pay for)
  if (i0 < re) {
    if (re - i0 > (uint32_t)WAVE) {
      uint32_t dv[UN], tfv[UN][FA], flv[UN][FA];
      load_trip<F_, UN>(p, lane, ec.post_off, i0, re, dv, tfv, flv);
      score_trip<MODE, F_, TAGS, UN>(p, lut, acc, tag, lane, tile_base, i0, re, dv, tfv, flv, ec, qtl);
    } else {
      uint32_t dv[1], tfv[1][FA], flv[1][FA];
      load_trip<F_, 1>(p, lane, ec.post_off, i0, re, dv, tfv, flv);
      score_trip<MODE, F_, TAGS, 1>(p, lut, acc, tag, lane, tile_base, i0, re, dv, tfv, flv, ec, qtl);
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
  for (;;) {
  uint32_t item = 0;
  if (lane == 0) item = atomicAdd(p.work_counter, 1u);
  item = __builtin_amdgcn_readfirstlane(item);
  if (item >= n_items) break;
  const uint32_t q = item % p.B;
  const uint32_t sup = item / p.B;
  const uint32_t e0 = p.qbeg[q], e1 = p.qbeg[q + 1];
  const uint32_t ne = e1 - e0;
  const bool mine = MODE == MODE_BM25 || (p.qflags[q] & 1u);  // Z21S: only "simple" queries
  if (MODE == MODE_Z21S && !mine) continue;                   // k_z21 owns this query's candidate slots
  const uint32_t qtl = MODE == MODE_Z21S ? p.qterms_len[q] : 0u;

  TopK tk;
  tk.s = -1.0; tk.d = 0xFFFFFFFFu; tk.n = 0; tk.thr_s = 0.0; tk.thr_d = 0;

  if (ne != 0) {
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
    uint32_t ec_qterm[G], ec_tbl[G], ec_row[G];
    uint32_t rb[G], re[G];
    uint32_t dv[G][FU], tfv[G][FU][FA], flv[G][FU][FA];
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
        ec[g].w1 = en.boost;                                                                                    \
        ec_qterm[g] = MODE == MODE_Z21S ? en.qterm_index : en.qterm;                                            \
        ec_tbl[g] = en.tbl_off;                                                                                 \
        ec_row[g] = (en.shift & DENSE_FLAG) ? en.node : 0xFFFFFFFFu;                     \
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
      if (rb[g] < re[g]) load_trip<F_, FU>(p, lane, ec[g].post_off, rb[g], re[g], dv[g], tfv[g], flv[g]);       \
    }                                                                                                           \
  }
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
          if (PS_ABLATE_BUILD && (p.ablate & 8u)) {
          } else if (MODE == MODE_BM25) dense_apply<TAGS>(p, acc, tag, lane, ec_row[g], tile_base, (uint16_t)(tagbase + ec_qterm[g]));
          else dense_apply_z<TAGS>(p, acc, reinterpret_cast<uint32_t*>(tag), lane, ec_row[g], tile_base,
                                   (ec_qterm[g] >> 31) ? (1u << ((ec_qterm[g] >> 16) & 31u)) : 0u);
        } else if (rb[g] < re[g]) {
          dirty = true;
          ec[g].tag = MODE == MODE_Z21S ? ec_qterm[g] : tagbase + ec_qterm[g];
          score_trip<MODE, F_, TAGS, FU>(p, lut, acc, tag, lane, tile_base, rb[g], re[g], dv[g], tfv[g], flv[g], ec[g], qtl);
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
      if (more) { PS_PHASE1(nt, neg, false) }
      const bool harvest = last && dirty && !(PS_ABLATE_BUILD && (p.ablate & 4u));
      if (last) dirty = false;
      t = nt; eg = neg;
      if (harvest) {
      // tile epilogue: harvest + reset (two f64 per lane per LDS access where the layout allows)
      double gt = 0.0;
      if (!FULL) gt = __longlong_as_double((long long)__hip_atomic_load(&p.gthr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (MODE == MODE_BM25) {
        for (uint32_t c = 0; c < T; c += 2 * WAVE) {
          double2* slot = reinterpret_cast<double2*>(&acc[c + 2 * lane]);
          const double2 v = *slot;
          const bool h0 = v.x > 0.0, h1 = v.y > 0.0;
          if (h0 || h1) *slot = make_double2(0.0, 0.0);
          const uint32_t d = tile_base + c + 2 * lane;
          if (FULL) {
            full_emit(p, q, lane, h0, v.x, d);
            full_emit(p, q, lane, h1, v.y, d + 1);
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
      } else {
        // accumulators are planar ([field][T]); two documents per lane per 16-byte LDS access
        for (uint32_t c = 0; c < T; c += 2 * WAVE) {
          // result.score = max(score_by_pool, result.score) over fields, from the dummy 0. (zero_to_one.rs:81,122)
          double b0 = 0.0, b1 = 0.0;
          bool h0 = false, h1 = false;
#pragma unroll
          for (int x = 0; x < FA; ++x) {
            if ((uint32_t)x < F) {
              double2* slot = reinterpret_cast<double2*>(&acc[(uint32_t)x * T + c + 2 * lane]);
              const double2 v = *slot;
              if (v.x > 0.0 || v.y > 0.0) {
                *slot = make_double2(0.0, 0.0);
                if (TAGS) *reinterpret_cast<uint2*>(reinterpret_cast<uint32_t*>(tag) + (uint32_t)x * T + c + 2 * lane) = make_uint2(0u, 0u);
              }
              h0 |= v.x > 0.0; h1 |= v.y > 0.0;
              b0 = fmax(v.x, b0); b1 = fmax(v.y, b1);
            }
          }
          const uint32_t d = tile_base + c + 2 * lane;
          if (FULL) {
            full_emit(p, q, lane, h0, b0, d);
            full_emit(p, q, lane, h1, b1, d + 1);
          } else {
            const double lo = (tk.n == p.K && tk.thr_s > gt) ? tk.thr_s : gt;
            if (__any((h0 && b0 >= lo) || (h1 && b1 >= lo))) {
              topk_offer(tk, p.K, lane, h0, b0, d, gt);
              topk_offer(tk, p.K, lane, h1, b1, d + 1, gt);
            }
          }
        }
      }
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
    const uint64_t o = (uint64_t)item * p.K + lane;
    const bool ok = (uint32_t)lane < tk.n;
    p.cand_score[o] = ok ? tk.s : 0.0;
    p.cand_doc[o] = ok ? tk.d : 0xFFFFFFFFu;
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
  const uint32_t F = p.F, ZN = p.z_nodes, ZT = p.z_tile;
  const uint32_t stride = ZN * F;
  uint32_t* rec = reinterpret_cast<uint32_t*>(smem);  // [ZT][ZN][F] term frequencies
  uint32_t* fls = rec + (size_t)ZT * stride;           // [ZT][F]     field lengths
  const int lane = threadIdx.x;
  // grid = n_general x n_super: only the queries the simple path could not take
  const uint32_t q = p.gen_queries[blockIdx.x % p.n_general];
  const uint32_t sup = blockIdx.x / p.n_general;
  const uint32_t item = sup * p.B + q;  // candidate slot, as k_merge expects it
  const uint32_t e0 = p.qbeg[q], e1 = p.qbeg[q + 1];

  TopK tk;
  tk.s = -1.0; tk.d = 0xFFFFFFFFu; tk.n = 0; tk.thr_s = 0.0; tk.thr_d = 0;

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
          bool has = false;
          double best = 0.0;  // the merged dummy Some(0.) (zero_to_one.rs:81,122)
          for (uint32_t x = 0; x < F; ++x) {
            unsigned long long consumed_q = 0ull;  // consumed_index: bit = query-term ordinal
            unsigned long long consumed_e = 0ull;  // consumed entries: bit = position in the query
            double pool = 0.0;                     // score_by_pool
            bool any = false;
            for (uint32_t z = e0; z < e1; ++z) {
              const uint32_t e = p.zorder[z];
              const uint32_t node = p.plan[e].node;
              const uint32_t tfu = rec[local * stride + node * F + x];
              if (tfu == 0) continue;  // no record for this (entry, doc, field)
              any = true;
              const uint32_t qt = p.plan[e].qterm;
              if ((consumed_q >> qt) & 1ull) continue;  // :101-103
              // df_pool_by_id (:104-113): a node may be consumed term_frequency times in total
              const unsigned long long same_node = (unsigned long long)__double_as_longlong(p.plan[e].idf);
              if ((uint32_t)__popcll(consumed_e & same_node) >= tfu) continue;
              consumed_e |= 1ull << (e - e0);
              consumed_q |= 1ull << qt;
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
}

// ------------------------------------------------------------------------------------------
// K3: merge per-run top-K lists -> final top-K per query, doc id -> key   (query.rs:97-105)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_merge(const KParams p) {
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  TopK tk;
  tk.s = -1.0; tk.d = 0xFFFFFFFFu; tk.n = 0; tk.thr_s = 0.0; tk.thr_d = 0;
  const uint32_t K = p.K;
  // candidates of (q, sup) live at item = sup * B + q
  const uint32_t per_round = WAVE / K;  // runs handled per 64-lane load
  for (uint32_t s0 = 0; s0 < p.n_super; s0 += per_round) {
    const uint32_t sup = s0 + lane / K;
    const uint32_t k = lane % K;
    bool has = false;
    double v = 0.0;
    uint32_t d = 0xFFFFFFFFu;
    if ((uint32_t)lane < per_round * K && sup < p.n_super) {
      const uint64_t o = ((uint64_t)sup * p.B + q) * K + k;
      d = p.cand_doc[o];
      v = p.cand_score[o];
      has = d != 0xFFFFFFFFu;
    }
    topk_offer(tk, K, lane, has, v, d);
  }
  if ((uint32_t)lane < K) {
    const bool ok = (uint32_t)lane < tk.n;
    const uint64_t o = (uint64_t)q * K + lane;
    p.out_keys[o] = ok ? p.keys[tk.d] : ~0ull;
    p.out_scores[o] = ok ? tk.s : 0.0;
  }
  if (lane == 0) p.out_counts[q] = tk.n;
}

// Full-result mode: the first (out_off[q+1] - out_off[q]) sorted results of run q -> {key, score}.
__global__ __launch_bounds__(256) void k_pack_results(const uint32_t* doc, const double* score, const uint64_t* run_off,
                                                      const uint64_t* out_off, const uint64_t* keys, ps_result* out) {
  const uint32_t q = blockIdx.x;
  const uint64_t src = run_off[q], dst = out_off[q], n = out_off[q + 1] - out_off[q];
  for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) out[dst + i] = ps_result{keys[doc[src + i]], score[src + i]};
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  void ensure(size_t n) {
    if (n <= cap) return;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 4 + 64;
    PS_HIP(hipMalloc((void**)&p, want * sizeof(T)));
    cap = want;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Pinned staging slot; `done` fences reuse so a caller-supplied stream may run ahead of the host.
struct Stage {
  unsigned char* p = nullptr;
  size_t cap = 0;
  hipEvent_t done = nullptr;
  bool pending = false;
  void ensure(size_t n) {
    if (pending) { PS_HIP(hipEventSynchronize(done)); pending = false; }
    if (n <= cap) return;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 4 + 256;
    PS_HIP(hipHostMalloc((void**)&p, want, hipHostMallocDefault));
    cap = want;
  }
};

constexpr int N_STAGE = 4;
constexpr int N_KTIMER = 32;

struct EngineImpl {
  const Snapshot* snap;
  int device;
  hipStream_t stream = nullptr;
  hipEvent_t ev[6] = {};
  uint32_t* d_doc = nullptr;
  uint32_t* d_tf = nullptr;
  uint32_t* d_fl = nullptr;
  uint32_t* d_table = nullptr;
  uint64_t* d_keys = nullptr;
  double* d_lut = nullptr;
  uint32_t* d_work = nullptr;
  int n_cu = 256;
  uint64_t bytes = 0;
  std::mutex mu;
  // per-batch device buffers (grow-only; reuse is ordered by the stream)
  DevBuf<unsigned char> d_stage;  // plan entries + per-query arrays, one H2D copy per batch
  DevBuf<uint32_t> d_cand_doc, d_out_counts, d_full_doc, d_full_cnt;
  DevBuf<double> d_cand_score, d_out_scores, d_full_score;
  DevBuf<uint64_t> d_out_keys, d_full_off;
  DevBuf<unsigned long long> d_gthr;
  DevBuf<double> d_rows;  // dense per-document score rows of the batch's hot lists
  DevBuf<uint32_t> d_sort_doc, d_seg;  // K4 scratch
  DevBuf<uint64_t> d_sort_score, d_pack_off;
  DevBuf<unsigned char> d_sort_tmp;
  DevBuf<ps_result> d_pack;
  Stage stage[N_STAGE];
  int next_stage = 0;
  Stage result;  // download staging (engine stream only)
  // HIP-event pairs around every launch of the scoring kernel (K1/K2), harvested lazily so a
  // caller that pipelines batches on its own stream still gets per-launch durations.
  struct KTimer { hipEvent_t a = nullptr, b = nullptr; bool pending = false; };
  KTimer kt[N_KTIMER];
  KTimer* last_kt = nullptr;
  uint64_t last_layout_bytes = 0;  // of the most recently staged batch
  uint32_t last_rows = 0;
  int next_kt = 0;
  double kt_total_ms = 0.0;
  uint64_t kt_launches = 0;
  void harvest(KTimer& t, bool wait) {
    if (!t.pending) return;
    if (!wait && hipEventQuery(t.b) != hipSuccess) return;
    if (wait) (void)hipEventSynchronize(t.b);
    float ms = 0;
    if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { kt_total_ms += ms; kt_launches++; }
    t.pending = false;
  }
};

int device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

Engine::Engine(const Snapshot& snap, int device) : impl_(new EngineImpl()) {
  EngineImpl& m = *impl_;
  m.snap = &snap;
  m.device = device;
  try {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
      throw std::runtime_error("no HIP device available (this engine has no CPU scoring fallback)");
    if (device < 0 || device >= n) throw std::runtime_error("device index out of range");
    PS_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PS_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
      throw std::runtime_error(std::string("built for gfx950 (MI355X) only; device is ") + prop.gcnArchName);
    m.n_cu = prop.multiProcessorCount;
    PS_HIP(hipMalloc((void**)&m.d_work, 64));
    PS_HIP(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
    for (auto& ev : m.ev) PS_HIP(hipEventCreate(&ev));
    for (auto& sg : m.stage) PS_HIP(hipEventCreateWithFlags(&sg.done, hipEventDisableTiming));
    PS_HIP(hipEventCreateWithFlags(&m.result.done, hipEventDisableTiming));
    for (auto& t : m.kt) { PS_HIP(hipEventCreate(&t.a)); PS_HIP(hipEventCreate(&t.b)); }
    const size_t P = snap.P, F = snap.F;
    PS_HIP(hipMalloc((void**)&m.d_doc, P * 4));
    PS_HIP(hipMalloc((void**)&m.d_tf, P * F * 4));
    PS_HIP(hipMalloc((void**)&m.d_fl, P * F * 4));
    PS_HIP(hipMalloc((void**)&m.d_table, snap.table.size() * 4));
    PS_HIP(hipMalloc((void**)&m.d_keys, std::max<size_t>(1, snap.keys.size()) * 8));
    PS_HIP(hipMalloc((void**)&m.d_lut, ((size_t)snap.lut_rows + 4) * LUT_TF * 8));
    PS_HIP(hipMemcpy(m.d_doc, snap.doc.data(), P * 4, hipMemcpyHostToDevice));
    PS_HIP(hipMemcpy(m.d_tf, snap.tf.data(), P * F * 4, hipMemcpyHostToDevice));
    PS_HIP(hipMemcpy(m.d_fl, snap.fl.data(), P * F * 4, hipMemcpyHostToDevice));
    PS_HIP(hipMemcpy(m.d_table, snap.table.data(), snap.table.size() * 4, hipMemcpyHostToDevice));
    if (!snap.keys.empty())
      PS_HIP(hipMemcpy(m.d_keys, snap.keys.data(), snap.keys.size() * 8, hipMemcpyHostToDevice));
    m.bytes = P * 4 + 2 * P * F * 4 + snap.table.size() * 4 + snap.keys.size() * 8;
  } catch (...) {
    delete impl_;
    impl_ = nullptr;
    throw;
  }
}

Engine::~Engine() {
  if (!impl_) return;
  EngineImpl& m = *impl_;
  (void)hipSetDevice(m.device);
  (void)hipDeviceSynchronize();
  for (void* p : {(void*)m.d_doc, (void*)m.d_tf, (void*)m.d_fl, (void*)m.d_table, (void*)m.d_keys, (void*)m.d_lut, (void*)m.d_work})
    if (p) (void)hipFree(p);
  m.d_stage.release(); m.d_cand_doc.release();
  m.d_out_counts.release(); m.d_full_doc.release(); m.d_full_cnt.release(); m.d_cand_score.release();
  m.d_out_scores.release(); m.d_full_score.release(); m.d_out_keys.release(); m.d_full_off.release();
  m.d_gthr.release(); m.d_rows.release();
  m.d_sort_doc.release(); m.d_seg.release(); m.d_sort_score.release(); m.d_pack_off.release();
  m.d_sort_tmp.release(); m.d_pack.release();
  for (auto& sg : m.stage) {
    if (sg.p) (void)hipHostFree(sg.p);
    if (sg.done) (void)hipEventDestroy(sg.done);
  }
  if (m.result.p) (void)hipHostFree(m.result.p);
  if (m.result.done) (void)hipEventDestroy(m.result.done);
  for (auto& ev : m.ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& t : m.kt) {
    if (t.a) (void)hipEventDestroy(t.a);
    if (t.b) (void)hipEventDestroy(t.b);
  }
  if (m.stream) (void)hipStreamDestroy(m.stream);
  delete impl_;
}

uint64_t Engine::device_bytes() const { return impl_->bytes; }
void Engine::kernel_times(double* total_ms, uint64_t* launches, bool reset) {
  EngineImpl& m = *impl_;
  std::lock_guard<std::mutex> lock(m.mu);
  (void)hipSetDevice(m.device);
  for (auto& t : m.kt) m.harvest(t, true);
  if (total_ms) *total_ms = m.kt_total_ms;
  if (launches) *launches = m.kt_launches;
  if (reset) { m.kt_total_ms = 0.0; m.kt_launches = 0; }
}
int Engine::device() const { return impl_->device; }

namespace {

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

uint32_t env_u32(const char* name, uint32_t dflt) {
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  return (uint32_t)strtoul(v, nullptr, 10);
}

void validate(const Snapshot& s, const ps_scorer_desc& sc, const Plan& plan) {
  if (s.F > (uint32_t)MAX_F) throw std::length_error("the GPU path supports at most 8 fields");
  if (sc.kind != PS_SCORER_BM25 && sc.kind != PS_SCORER_ZERO_TO_ONE) throw std::invalid_argument("unknown scorer kind");
  if (plan.max_qterms >= 0x7FFF) throw std::length_error("more than 32766 non-empty terms in one query");
}

// Uploads the plan + per-query arrays through a pinned staging slot; fills the common KParams.
void stage_plan(EngineImpl& m, const ps_scorer_desc& sc, const double* boosts, const Plan& plan, hipStream_t st,
                KParams& kp) {
  const Snapshot& s = *m.snap;
  const size_t B = plan.qbeg.size() - 1;
  const size_t ne = plan.entries.size();
  const bool z = sc.kind == PS_SCORER_ZERO_TO_ONE;
  const size_t off_e = 0;
  const size_t off_q = off_e + ne * sizeof(ps_plan_entry);
  const size_t off_l = off_q + (B + 1) * 4;
  const size_t off_z = off_l + B * 4;
  const size_t off_f = off_z + ne * 4;
  const size_t off_g = off_f + B * 4;
  const size_t off_r = (off_g + B * 4 + 15) & ~(size_t)15;
  const size_t max_rows = env_u32("PS_DENSE_MAX_ROWS", 64);
  const size_t total = off_r + max_rows * sizeof(RowDesc);
  Stage& sg = m.stage[m.next_stage];
  m.next_stage = (m.next_stage + 1) % N_STAGE;
  sg.ensure(total + 16);
  unsigned char* h = sg.p;
  ps_plan_entry* he = reinterpret_cast<ps_plan_entry*>(h + off_e);
  if (ne) memcpy(he, plan.entries.data(), ne * sizeof(ps_plan_entry));
  memcpy(h + off_q, plan.qbeg.data(), (B + 1) * 4);
  if (B) memcpy(h + off_l, plan.qterms_len.data(), B * 4);
  uint32_t n_rows = 0;
  uint64_t layout_bytes = 0;
  uint32_t n_simple = 0, n_general = 0, z_masked = 0;
  if (z) {
    // per query: entry indices stably sorted by ScoreByTerm::score desc (zero_to_one.rs:98);
    // the records' push order == plan order (query term asc, expansion order, newest version first).
    // A query whose entries all have their own trie node, their own query term and a single
    // version layer is "simple": finalize's greedy scan can never skip a record, so k_score sums
    // the contributions in that sorted order directly; its entries are uploaded pre-sorted.
    uint32_t* zo = reinterpret_cast<uint32_t*>(h + off_z);
    uint32_t* qf = reinterpret_cast<uint32_t*>(h + off_f);
    uint32_t* gq = reinterpret_cast<uint32_t*>(h + off_g);  // queries the general kernel has to run
    std::vector<ps_plan_entry> tmp;
    for (size_t q = 0; q < B; ++q) {
      const uint32_t b = plan.qbeg[q], e = plan.qbeg[q + 1];
      for (uint32_t i = b; i < e; ++i) zo[i] = i;
      std::stable_sort(zo + b, zo + e,
                       [&](uint32_t a, uint32_t c) { return plan.entries[c].boost < plan.entries[a].boost; });
      // simple: one entry per query term and a single version layer.  The same trie node may
      // appear several times ("abc abc"): the k-th record of a node in the sorted order is consumed
      // iff the node's pool still holds something, i.e. iff term_frequency >= k (zero_to_one.rs:104-113)
      bool simple = true, masked = false;
      for (uint32_t i = b; i < e && simple; ++i) {
        if (plan.entries[i].shift >> 8) simple = false;
        for (uint32_t j = b; j < i && simple; ++j) {
          const bool same_q = plan.entries[j].qterm == plan.entries[i].qterm;
          const bool same_n = plan.entries[j].node == plan.entries[i].node;
          // several expansions of one query term: fine as long as every record has its own node
          // (then the pool never blocks and only consumed_index decides) -> mask variant
          if (same_q) masked = true;
          if (same_q && same_n) simple = false;
        }
      }
      if (masked) {
        // masks are u32 per (doc, field); mixing "same node under two query terms" with
        // expansions needs the full pool bookkeeping of the general kernel
        for (uint32_t i = b; i < e && simple; ++i) {
          if (plan.entries[i].qterm >= 32) simple = false;
          for (uint32_t j = b; j < i && simple; ++j)
            if (plan.entries[j].node == plan.entries[i].node) simple = false;
        }
      }
      // the simple path keeps F f64 accumulators per document of the tile in LDS
      if ((size_t)WG_WAVES * ((size_t)s.T * s.F * 8 + 4096) > 160 * 1024) simple = false;
      if (masked && (size_t)WG_WAVES * ((size_t)s.T * s.F * 12 + 4096) > 160 * 1024) simple = false;
      if (env_u32("PS_Z21_GENERAL_ONLY", 0)) simple = false;
      qf[q] = simple ? 1u : 0u;
      if (!simple) masked = false;
      if (simple) {
        ++n_simple;
        tmp.assign(plan.entries.begin() + b, plan.entries.begin() + e);
        for (uint32_t i = b; i < e; ++i) {
          he[i] = tmp[zo[i] - b];
          uint32_t need = 1;  // occurrence rank of the node among the sorted records
          for (uint32_t j = b; j < i; ++j)
            if (he[j].node == he[i].node) ++need;
          he[i].qterm_index = need | ((he[i].qterm & 31u) << 16) | (masked ? 0x80000000u : 0u);
        }
        if (masked) { z_masked = 1; qf[q] |= 2u; }
      } else {
        gq[n_general++] = (uint32_t)q;  // empty queries too: somebody has to write their (empty) candidate slots
        if (e - b > 64) throw std::length_error("zero_to_one with repeated terms supports at most 64 expanded lists per query on the GPU");
      }
    }
  }
  // ---- hot dense lists (see k_dense_rows) -------------------------------------------------------
  // BM25: key = (list, idf, expansion_boost).  zero_to_one (simple queries only): key = (list,
  // ScoreByTerm::score, all_query_terms_len); a row then holds one plane per field.
  {
    bool sane = max_rows > 0 && s.n_docs > 0;
    if (!z) {
      sane = sane && std::isfinite(sc.bm25_k1) && sc.bm25_k1 >= 0.0 && sc.bm25_b >= 0.0 && sc.bm25_b <= 1.0;
      for (uint32_t x = 0; x < s.F && sane; ++x)
        sane = std::isfinite(boosts[x]) && boosts[x] > 0.0 && std::isfinite(s.avg[x]) && s.avg[x] > 0.0;
    }
    const uint32_t min_uses = env_u32("PS_DENSE_MIN_USES", 4);
    const double min_density = env_u32("PS_DENSE_MIN_DENSITY_PCT", 25) / 100.0;
    const uint32_t planes = z ? s.F : 1u;
    if (sane && ne) {
      struct Key { uint64_t post_off, w, k3; };
      struct Agg { uint32_t uses, len; };
      auto kless = [](const Key& a, const Key& b) {
        return a.post_off != b.post_off ? a.post_off < b.post_off : a.w != b.w ? a.w < b.w : a.k3 < b.k3;
      };
      const uint32_t* qf = reinterpret_cast<const uint32_t*>(h + off_f);
      auto key_of = [&](const ps_plan_entry& e, size_t q) {
        Key k{e.post_off, 0, 0};
        if (z) { memcpy(&k.w, &e.boost, 8); k.k3 = (uint64_t)plan.qterms_len[q] | ((uint64_t)(e.qterm_index & 0xFFFFu) << 32); }
        else { memcpy(&k.w, &e.idf, 8); memcpy(&k.k3, &e.boost, 8); }
        return k;
      };
      std::map<Key, Agg, decltype(kless)> agg(kless);
      for (size_t q = 0; q < B; ++q) {
        if (z && !(qf[q] & 1u)) continue;  // zero_to_one: queries of the fast path only
        for (uint32_t i = plan.qbeg[q]; i < plan.qbeg[q + 1]; ++i) {
          if ((double)he[i].len < min_density * (double)s.n_docs) continue;
          Agg& a = agg[key_of(he[i], q)];
          a.uses++;
          a.len = he[i].len;
        }
      }
      std::vector<std::pair<uint64_t, Key>> hot;  // (saved posting visits, key)
      for (auto& kv : agg)
        if (kv.second.uses >= min_uses) hot.emplace_back((uint64_t)kv.second.uses * kv.second.len, kv.first);
      std::sort(hot.begin(), hot.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
      const uint64_t row_bytes = (uint64_t)s.n_tiles * s.T * 8 * planes;
      const uint64_t mem_cap = (uint64_t)env_u32("PS_DENSE_MAX_MB", 4096) << 20;
      while (hot.size() > max_rows || hot.size() * row_bytes > mem_cap) hot.pop_back();
      if (!hot.empty()) {
        RowDesc* rd = reinterpret_cast<RowDesc*>(h + off_r);
        std::map<Key, uint32_t, decltype(kless)> row_of(kless);
        for (auto& hk : hot) {
          RowDesc d;
          d.post_off = hk.second.post_off;
          d.len = agg[hk.second].len;
          // zero_to_one: all_query_terms_len (low 16 bits) | required term frequency (high 16 bits)
          d._pad = z ? ((uint32_t)(hk.second.k3 & 0xFFFFu) | ((uint32_t)(hk.second.k3 >> 32) << 16)) : 0u;
          memcpy(&d.idf, &hk.second.w, 8);           // BM25: idf | zero_to_one: ScoreByTerm::score
          if (z) d.eb = 0.0; else memcpy(&d.eb, &hk.second.k3, 8);
          row_of[hk.second] = n_rows;
          rd[n_rows++] = d;
        }
        for (size_t q = 0; q < B; ++q) {
          if (z && !(qf[q] & 1u)) continue;
          for (uint32_t i = plan.qbeg[q]; i < plan.qbeg[q + 1]; ++i) {
            auto it = row_of.find(key_of(he[i], q));
            if (it != row_of.end()) { he[i].shift |= DENSE_FLAG; he[i].node = it->second; }
          }
        }
      }
    }
  }
  // one H2D copy: the device image has the staging layout (entries | qbeg | qterms_len | zorder | qflags)
  m.d_stage.ensure(total + 64);
  PS_HIP(hipMemcpyAsync(m.d_stage.p, h, n_rows ? off_r + n_rows * sizeof(RowDesc) : (z ? off_r : off_z),
                        hipMemcpyHostToDevice, st));
  PS_HIP(hipEventRecord(sg.done, st));
  sg.pending = true;

  memset(&kp, 0, sizeof(kp));
  kp.doc = m.d_doc; kp.tf = m.d_tf; kp.fl = m.d_fl; kp.table = m.d_table; kp.keys = m.d_keys;
  kp.plan = reinterpret_cast<const ps_plan_entry*>(m.d_stage.p + off_e);
  kp.qbeg = reinterpret_cast<const uint32_t*>(m.d_stage.p + off_q);
  kp.qterms_len = reinterpret_cast<const uint32_t*>(m.d_stage.p + off_l);
  kp.zorder = reinterpret_cast<const uint32_t*>(m.d_stage.p + off_z);
  kp.qflags = reinterpret_cast<const uint32_t*>(m.d_stage.p + off_f);
  kp.gen_queries = reinterpret_cast<const uint32_t*>(m.d_stage.p + off_g);
  {
    // bytes of the layout actually used (never claim the wider figure for a narrower stream)
    const uint64_t pb = 4 + 8 * (uint64_t)s.F, row_bytes = (uint64_t)s.n_tiles * s.T * 8 * (z ? s.F : 1u);
    uint64_t lb = 0;
    for (size_t i = 0; i < ne; ++i) lb += (he[i].shift & DENSE_FLAG) ? row_bytes : (uint64_t)he[i].len * pb;
    const RowDesc* rd = reinterpret_cast<const RowDesc*>(h + off_r);
    for (uint32_t r = 0; r < n_rows; ++r) lb += (uint64_t)rd[r].len * (pb + 8 * (z ? s.F : 1u)) + row_bytes;
    layout_bytes = lb;
  }
  kp.row_desc = reinterpret_cast<const RowDesc*>(m.d_stage.p + off_r);
  kp.n_rows = n_rows;
  kp.row_planes = z ? s.F : 1u;
  kp.row_mode = z ? 1u : 0u;
  kp.row_stride = (uint64_t)s.n_tiles * s.T;
  if (n_rows) {
    m.d_rows.ensure((size_t)n_rows * kp.row_planes * kp.row_stride + 16);
    kp.rows = m.d_rows.p;
  }
  // control words, zeroed by one memset per batch: gthr[0..B) + the persistent waves' item counter
  m.d_gthr.ensure(B + 2);
  kp.gthr = m.d_gthr.p;
  kp.work_counter = reinterpret_cast<uint32_t*>(m.d_gthr.p + B + 1);
  PS_HIP(hipMemsetAsync(m.d_gthr.p, 0, (B + 2) * 8, st));
  kp.n_simple = n_simple; kp.n_general = n_general; kp.z_masked = z_masked;
  kp.layout_bytes = layout_bytes;
  m.last_layout_bytes = layout_bytes;
  m.last_rows = n_rows;
  kp.P = s.P;
  kp.B = (uint32_t)B; kp.n_tiles = s.n_tiles; kp.T = s.T; kp.n_docs = (uint32_t)s.n_docs; kp.F = s.F;
  kp.max_qterms = std::max<uint32_t>(1, plan.max_qterms);
  kp.ablate = env_u32("PS_ABLATE", 0);
  kp.k1 = sc.bm25_k1; kp.b = sc.bm25_b;
  kp.k1p1 = sc.bm25_k1 + 1.0;        // (self.bm25k1 + 1_f64), bm25.rs:78 — same IEEE add on the host
  kp.one_minus_b = 1.0 - sc.bm25_b;  // (1_f64 - self.bm25b),  bm25.rs:80
  for (uint32_t x = 0; x < s.F; ++x) { kp.avg[x] = s.avg[x]; kp.boost[x] = boosts[x]; }
  if (sc.kind == PS_SCORER_BM25 && env_u32("PS_LUT", 1)) {
    kp.lut = m.d_lut;
    kp.lut_rows = s.lut_rows;
    kp.lut_stride = s.lut_rows ? ((s.lut_rows + 1) | 1u) : 0;  // odd stride; LUT bytes = stride*128, so tiles stay 16-B aligned
    for (uint32_t x = 0; x < s.F; ++x) { kp.lut_cap[x] = s.lut_cap[x]; kp.lut_base[x] = s.lut_base[x]; }
  }
  // work decomposition: one wave per (query, run of S tiles)
  const uint64_t target = env_u32("PS_TARGET_ITEMS", 65536);
  uint64_t S = ((uint64_t)s.n_tiles * std::max<size_t>(B, 1) + target - 1) / target;
  const uint32_t s_env = env_u32("PS_TILES_PER_RUN", 0);
  if (s_env) S = s_env;
  if (S < 1) S = 1;
  if (S > s.n_tiles) S = s.n_tiles;
  if (S > 32) S = 32;  // a run's table slice is fetched by one wave-wide load (lane <-> tile)
  kp.S = (uint32_t)S;
  kp.n_super = (uint32_t)((s.n_tiles + S - 1) / S);
  // per-wave LDS for the table slices: [entry][rb|re][S] u32; fall back to global lookups if large
  const size_t slice = (((size_t)plan.max_entries * 2 * S * 4) + 15) & ~(size_t)15;
  kp.slice_bytes = (slice <= 4096 && env_u32("PS_SLICES", 1)) ? (uint32_t)slice : 0u;
}

void allow_lds(const void* fn, size_t lds) {
  if (lds <= 65536) return;
  if (lds > 160 * 1024) throw std::length_error("LDS tile exceeds 160 KiB: use a smaller tile_docs");
  PS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
}

template <int MODE, bool FULL>
void launch_k_score(KParams& kp, bool tags, int n_cu, hipStream_t st) {
  const uint32_t n_items = kp.B * kp.n_super;
  const uint32_t aw = MODE == MODE_Z21S ? kp.F : 1u;
  const size_t lut_b = MODE == MODE_BM25 ? (size_t)kp.lut_stride * LUT_TF * 8 : 0;
  const size_t wave_b = (size_t)kp.T * aw * 8 + (tags ? (MODE == MODE_Z21S ? (size_t)kp.T * aw * 4 : (size_t)kp.T * 2) : 0) +
                        kp.slice_bytes;
  // Workgroups of 8 waves share one LUT copy: two of them (16 waves) fit a CU's 160 KiB when a
  // wave's tile is small enough; otherwise 4-wave workgroups pack the LDS better.
  const bool wide = !FULL && lut_b + 8 * wave_b <= 80 * 1024 && env_u32("PS_WG8", 1);
  const uint32_t wgw = wide ? 8u : (uint32_t)WG_WAVES;
  uint32_t n_wg = (n_items + wgw - 1) / wgw;
  const size_t lds = lut_b + wgw * wave_b;
#define PS_LAUNCH_W(FV, TG, W)                                                                         \
  do {                                                                                                 \
    const void* fn = reinterpret_cast<const void*>(&k_score<MODE, FV, TG, FULL, W>);                   \
    allow_lds(fn, lds);                                                                                \
    int per_cu = 0;                                                                                    \
    PS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, WAVE * W, lds));                  \
    const uint32_t resident = (uint32_t)std::max(1, per_cu) * (uint32_t)n_cu;                          \
    if (n_wg > resident) n_wg = resident;                                                              \
    hipLaunchKernelGGL((k_score<MODE, FV, TG, FULL, W>), dim3(n_wg), dim3(WAVE * W), lds, st, kp);     \
  } while (0)
#define PS_LAUNCH(FV, TG)                                                                              \
  do {                                                                                                 \
    if (!FULL && wide) PS_LAUNCH_W(FV, TG, (FULL ? WG_WAVES : 8));                                     \
    else PS_LAUNCH_W(FV, TG, WG_WAVES);                                                                \
  } while (0)
  if (tags) {
    if (kp.F == 1) PS_LAUNCH(1, true);
    else if (kp.F == 2) PS_LAUNCH(2, true);
    else PS_LAUNCH(0, true);
  } else {
    if (kp.F == 1) PS_LAUNCH(1, false);
    else if (kp.F == 2) PS_LAUNCH(2, false);
    else PS_LAUNCH(0, false);
  }
#undef PS_LAUNCH
#undef PS_LAUNCH_W
}

void launch_rows(const KParams& kp, hipStream_t st) {
  if (!kp.n_rows) return;
  PS_HIP(hipMemsetAsync(const_cast<double*>(kp.rows), 0, (size_t)kp.n_rows * kp.row_planes * kp.row_stride * 8, st));
  hipLaunchKernelGGL(k_dense_rows, dim3(256, kp.n_rows), dim3(256), 0, st, kp, const_cast<double*>(kp.rows));
}

template <bool FULL>
void launch_score(const ps_scorer_desc& sc, const Plan& plan, KParams& kp, int n_cu, hipStream_t st) {
  const uint32_t n_items = kp.B * kp.n_super;
  if (n_items == 0) return;
  if (sc.kind == PS_SCORER_BM25) {
    if (kp.lut_rows) hipLaunchKernelGGL(k_bm25_lut, dim3(4), dim3(256), 0, st, kp, const_cast<double*>(kp.lut));
    launch_rows(kp, st);
    launch_k_score<MODE_BM25, FULL>(kp, plan.multi_expansion, n_cu, st);
  } else {
    launch_rows(kp, st);
    if (kp.n_simple) launch_k_score<MODE_Z21S, FULL>(kp, kp.z_masked != 0, n_cu, st);
    if (kp.n_general) {
      // general zero_to_one: the LDS sub-tile shrinks with (distinct nodes x fields) to fit the budget
      kp.z_nodes = std::max<uint32_t>(1, plan.max_nodes);
      const uint32_t per_doc = (kp.z_nodes * kp.F + kp.F) * 4;
      uint32_t zt = kp.T;
      const uint32_t budget = env_u32("PS_Z21_LDS", 20480);
      while (zt > (uint32_t)WAVE && (size_t)zt * per_doc > budget) zt >>= 1;
      if ((size_t)zt * per_doc > 65536) throw std::length_error("zero_to_one: fields x expanded terms exceed the LDS tile");
      kp.z_tile = zt;
      hipLaunchKernelGGL((k_z21<FULL>), dim3(kp.n_general * kp.n_super), dim3(WAVE), (size_t)zt * per_doc, st, kp);
    }
  }
  PS_HIP(hipGetLastError());
}

void fill_stats(const EngineImpl& m, ps_batch_stats& st, const Snapshot& s, const Plan& plan, uint64_t emitted) {
  st.layout_bytes = m.last_layout_bytes + emitted * 16;
  st.dense_rows = m.last_rows;
  st.n_queries = plan.qbeg.size() - 1;
  st.n_plan_entries = plan.entries.size();
  st.postings_visited = plan.postings;
  st.algorithmic_bytes = plan.postings * (4 + 8 * (uint64_t)s.F) + emitted * 16;
}

// Enqueue plan upload + K1/K2 + K3 on `st`, writing the final top-k to the given device buffers.
void enqueue_topk(EngineImpl& m, const ps_scorer_desc& sc, const double* boosts, const Plan& plan, size_t top_k,
                  void* d_keys, void* d_scores, void* d_counts, hipStream_t st) {
  const size_t B = plan.qbeg.size() - 1;
  KParams kp;
  static const bool trace = env_u32("PS_TRACE", 0) != 0;
  double tt = now_ms();
  auto TT = [&](const char* what) {
    if (!trace) return;
    double n = now_ms();
    fprintf(stderr, "[ps] %-12s %.3f ms\n", what, n - tt);
    tt = n;
  };
  stage_plan(m, sc, boosts, plan, st, kp);
  TT("stage_plan");
  kp.K = (uint32_t)top_k;
  const size_t n_cand = (size_t)B * kp.n_super * top_k;
  m.d_cand_score.ensure(n_cand + 1);
  m.d_cand_doc.ensure(n_cand + 1);
  kp.cand_score = m.d_cand_score.p;
  kp.cand_doc = m.d_cand_doc.p;
  TT("ctl");
  kp.out_keys = (uint64_t*)d_keys;
  kp.out_scores = (double*)d_scores;
  kp.out_counts = (uint32_t*)d_counts;
  EngineImpl::KTimer& kt = m.kt[m.next_kt];
  m.next_kt = (m.next_kt + 1) % N_KTIMER;
  m.harvest(kt, true);
  TT("harvest");
  PS_HIP(hipEventRecord(kt.a, st));
  launch_score<false>(sc, plan, kp, m.n_cu, st);
  TT("launch");
  PS_HIP(hipEventRecord(kt.b, st));
  kt.pending = true;
  m.last_kt = &kt;
  if (B) {
    hipLaunchKernelGGL(k_merge, dim3((uint32_t)B), dim3(WAVE), 0, st, kp);
    PS_HIP(hipGetLastError());
  }
  TT("merge");
}

// Latency-oriented stream wait: poll for a short while (a blocking hipStreamSynchronize costs tens
// of microseconds of wake-up latency, comparable to a whole single-query batch), then block.
void sync_stream(hipStream_t st) {
  const double t0 = now_ms();
  while (now_ms() - t0 < 0.5) {
    hipError_t e = hipStreamQuery(st);
    if (e == hipSuccess) return;
    if (e != hipErrorNotReady) PS_HIP(e);
  }
  PS_HIP(hipStreamSynchronize(st));
}

// After a stream sync: duration of the batch's scoring kernel from its HIP-event pair.
void read_kernel_times(EngineImpl& m, ps_batch_stats& stats) {
  float b = 0;
  if (m.last_kt && hipEventElapsedTime(&b, m.last_kt->a, m.last_kt->b) != hipSuccess) b = 0;
  stats.h2d_ms = 0;
  stats.score_kernel_ms = b;
  stats.kernel_ms = b;
}

Plan sub_plan(const Plan& plan, size_t b, size_t e) {
  Plan p2;
  p2.qbeg.push_back(0);
  for (size_t q = b; q < e; ++q) {
    for (uint32_t i = plan.qbeg[q]; i < plan.qbeg[q + 1]; ++i) {
      p2.entries.push_back(plan.entries[i]);
      p2.postings += plan.entries[i].len;
    }
    p2.qbeg.push_back((uint32_t)p2.entries.size());
    p2.qterms_len.push_back(plan.qterms_len[q]);
    p2.n_nodes.push_back(plan.n_nodes[q]);
  }
  p2.max_entries = plan.max_entries;
  p2.max_qterms = plan.max_qterms;
  p2.max_nodes = plan.max_nodes;
  p2.multi_expansion = plan.multi_expansion;
  return p2;
}

}  // namespace

void Engine::run_device(const ps_scorer_desc& sc, const double* boosts, const Plan& plan, size_t top_k, void* d_keys,
                        void* d_scores, void* d_counts, void* stream, ps_batch_stats& stats) {
  EngineImpl& m = *impl_;
  const Snapshot& s = *m.snap;
  validate(s, sc, plan);
  if (top_k < 1 || top_k > PS_MAX_DEVICE_TOPK)
    throw std::invalid_argument("top_k must be in [1, 64] for the device top-k path");
  std::lock_guard<std::mutex> lock(m.mu);
  PS_HIP(hipSetDevice(m.device));
  const double t0 = now_ms();
  hipStream_t st = stream ? (hipStream_t)stream : m.stream;
  enqueue_topk(m, sc, boosts, plan, top_k, d_keys, d_scores, d_counts, st);
  memset(&stats, 0, sizeof(stats));
  fill_stats(m, stats, s, plan, (uint64_t)(plan.qbeg.size() - 1) * top_k);
  if (!stream) {
    PS_HIP(hipStreamSynchronize(st));
    read_kernel_times(m, stats);
  }
  stats.total_ms = now_ms() - t0;
}

void Engine::run_host(const ps_scorer_desc& sc, const double* boosts, const Plan& plan, size_t top_k,
                      std::vector<ps_result>& out, std::vector<size_t>& offsets, ps_batch_stats& stats) {
  EngineImpl& m = *impl_;
  const Snapshot& s = *m.snap;
  const size_t B = plan.qbeg.size() - 1;
  validate(s, sc, plan);
  out.clear();
  offsets.assign(B + 1, 0);
  memset(&stats, 0, sizeof(stats));
  const double t0 = now_ms();

  if (top_k >= 1 && top_k <= PS_MAX_DEVICE_TOPK) {
    std::lock_guard<std::mutex> lock(m.mu);
    PS_HIP(hipSetDevice(m.device));
    hipStream_t st = m.stream;
    const size_t nb = B * top_k;
    // keys | scores | counts in one device block -> one D2H copy
    const size_t res_bytes = nb * 16 + B * 4;
    m.d_out_keys.ensure(res_bytes / 8 + 2);
    unsigned char* dres = reinterpret_cast<unsigned char*>(m.d_out_keys.p);
    enqueue_topk(m, sc, boosts, plan, top_k, dres, dres + nb * 8, dres + nb * 16, st);
    m.result.ensure(res_bytes + 64);
    uint64_t* hk = reinterpret_cast<uint64_t*>(m.result.p);
    double* hs = reinterpret_cast<double*>(m.result.p + nb * 8);
    uint32_t* hc = reinterpret_cast<uint32_t*>(m.result.p + nb * 16);
    if (res_bytes) PS_HIP(hipMemcpyAsync(m.result.p, dres, res_bytes, hipMemcpyDeviceToHost, st));
    sync_stream(st);
    read_kernel_times(m, stats);
    stats.d2h_ms = 0;
    size_t total = 0;
    for (size_t q = 0; q < B; ++q) { offsets[q] = total; total += hc[q]; }
    offsets[B] = total;
    out.resize(total);
    for (size_t q = 0; q < B; ++q)
      for (uint32_t k = 0; k < hc[q]; ++k) out[offsets[q] + k] = ps_result{hk[q * top_k + k], hs[q * top_k + k]};
    fill_stats(m, stats, s, plan, total);
    stats.total_ms = now_ms() - t0;
    return;
  }

  // ---- full-result mode: top_k == 0 (every match, like the reference) or top_k > 64 ----------
  // upper bound of matches per query: min(N, sum of its list lengths)
  std::vector<uint64_t> cap(B + 1, 0);
  for (size_t q = 0; q < B; ++q) {
    uint64_t sum = 0;
    for (uint32_t e = plan.qbeg[q]; e < plan.qbeg[q + 1]; ++e) sum += plan.entries[e].len;
    cap[q + 1] = cap[q] + std::min<uint64_t>(sum, s.n_docs);
  }
  const uint64_t total_cap = cap[B];
  const uint64_t budget = (uint64_t)env_u32("PS_FULL_BUDGET_MB", 4096) << 20;
  if (total_cap * 12 > budget && B > 1) {
    // keep the result buffers bounded: run the two halves of the batch one after the other
    const size_t half = B / 2;
    std::vector<ps_result> o1, o2;
    std::vector<size_t> f1, f2;
    ps_batch_stats s1, s2;
    run_host(sc, boosts, sub_plan(plan, 0, half), top_k, o1, f1, s1);
    run_host(sc, boosts, sub_plan(plan, half, B), top_k, o2, f2, s2);
    out = std::move(o1);
    out.insert(out.end(), o2.begin(), o2.end());
    for (size_t q = 0; q <= half; ++q) offsets[q] = f1[q];
    for (size_t q = half; q <= B; ++q) offsets[q] = f1[half] + f2[q - half];
    fill_stats(m, stats, s, plan, out.size());
    stats.h2d_ms = s1.h2d_ms + s2.h2d_ms;
    stats.kernel_ms = s1.kernel_ms + s2.kernel_ms;
    stats.score_kernel_ms = s1.score_kernel_ms + s2.score_kernel_ms;
    stats.d2h_ms = s1.d2h_ms + s2.d2h_ms;
    stats.total_ms = now_ms() - t0;
    return;
  }

  std::lock_guard<std::mutex> lock(m.mu);
  PS_HIP(hipSetDevice(m.device));
  hipStream_t st = m.stream;
  KParams kp;
  stage_plan(m, sc, boosts, plan, st, kp);
  kp.K = 1;
  m.d_full_doc.ensure(total_cap + 1);
  m.d_full_score.ensure(total_cap + 1);
  m.d_full_off.ensure(B + 1);
  m.d_full_cnt.ensure(B + 1);
  kp.full_doc = m.d_full_doc.p;
  kp.full_score = m.d_full_score.p;
  kp.full_off = m.d_full_off.p;
  kp.full_cnt = m.d_full_cnt.p;
  m.result.ensure((B + 1) * 12 + 64);
  uint64_t* h_off = reinterpret_cast<uint64_t*>(m.result.p);
  uint32_t* h_cnt = reinterpret_cast<uint32_t*>(m.result.p + (B + 1) * 8);
  memcpy(h_off, cap.data(), (B + 1) * 8);
  PS_HIP(hipMemcpyAsync(m.d_full_off.p, h_off, (B + 1) * 8, hipMemcpyHostToDevice, st));
  PS_HIP(hipMemsetAsync(m.d_full_cnt.p, 0, (B + 1) * 4, st));
  EngineImpl::KTimer& kt = m.kt[m.next_kt];
  m.next_kt = (m.next_kt + 1) % N_KTIMER;
  m.harvest(kt, true);
  PS_HIP(hipEventRecord(kt.a, st));
  launch_score<true>(sc, plan, kp, m.n_cu, st);
  PS_HIP(hipEventRecord(kt.b, st));
  kt.pending = true;
  m.last_kt = &kt;
  // K4 (query.rs:97-105, "materialise + sort"): canonical order (score desc, doc id asc == key asc)
  // of every query's run, on the device (ps_sort.hip)
  if (total_cap >= 0xFFFFFFF0ull) throw std::length_error("full-result batch too large for one pass");
  m.d_sort_doc.ensure(total_cap + 1);
  m.d_sort_score.ensure(total_cap + 1);
  m.d_seg.ensure(2 * (B + 1));
  SortBuffers sb{m.d_full_doc.p, reinterpret_cast<uint64_t*>(m.d_full_score.p), m.d_sort_doc.p, m.d_sort_score.p,
                 m.d_seg.p, m.d_seg.p + B + 1};
  // few or huge runs: device-wide sort per run (needs the counts); many small runs: one segmented sort
  const bool few_runs = B <= 8 || total_cap / B > 32768;
  if (total_cap && B && !few_runs) {
    size_t tb = 0;
    PS_HIP(sort_results(sb, (unsigned)total_cap, (unsigned)B, m.d_full_off.p, m.d_full_cnt.p, nullptr, tb, st));
    m.d_sort_tmp.ensure(tb + 256);
    PS_HIP(sort_results(sb, (unsigned)total_cap, (unsigned)B, m.d_full_off.p, m.d_full_cnt.p, m.d_sort_tmp.p, tb, st));
  }
  PS_HIP(hipMemcpyAsync(h_cnt, m.d_full_cnt.p, (B + 1) * 4, hipMemcpyDeviceToHost, st));
  sync_stream(st);
  std::vector<uint32_t> cnt(h_cnt, h_cnt + B);
  read_kernel_times(m, stats);
  if (few_runs) {
    for (size_t q = 0; q < B; ++q) {
      if (cnt[q] < 2) continue;
      size_t tb = 0;
      PS_HIP(sort_run(sb, cap[q], cnt[q], nullptr, tb, st));
      m.d_sort_tmp.ensure(tb + 256);
      PS_HIP(sort_run(sb, cap[q], cnt[q], m.d_sort_tmp.p, tb, st));
    }
  }
  // pack the first `keep` results of every run as {key, score} records, one D2H copy
  size_t total = 0;
  for (size_t q = 0; q < B; ++q) {
    offsets[q] = total;
    total += (top_k ? std::min<size_t>(top_k, cnt[q]) : cnt[q]);
  }
  offsets[B] = total;
  out.resize(total);
  if (total) {
    m.d_pack_off.ensure(B + 1);
    m.d_pack.ensure(total);
    m.result.ensure(std::max<size_t>((B + 1) * 8, total * sizeof(ps_result)) + 64);
    uint64_t* h_po = reinterpret_cast<uint64_t*>(m.result.p);
    for (size_t q = 0; q <= B; ++q) h_po[q] = offsets[q];
    PS_HIP(hipMemcpyAsync(m.d_pack_off.p, h_po, (B + 1) * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_pack_results, dim3((uint32_t)B), dim3(256), 0, st, m.d_full_doc.p, m.d_full_score.p,
                       m.d_full_off.p, m.d_pack_off.p, m.d_keys, m.d_pack.p);
    PS_HIP(hipGetLastError());
    sync_stream(st);  // h_po (pinned) is reused as the download target
    PS_HIP(hipMemcpyAsync(m.result.p, m.d_pack.p, total * sizeof(ps_result), hipMemcpyDeviceToHost, st));
    sync_stream(st);
    memcpy(out.data(), m.result.p, total * sizeof(ps_result));
  }
  fill_stats(m, stats, s, plan, total);
  stats.total_ms = now_ms() - t0;
}

}  // namespace ps
