// ps_kernels_score.hpp — the streaming family: K0 k_bm25_lut, K0b k_dense_rows, K1 k_score (BM25 / simple zero_to_one: every posting of
// every list through LDS tiles), K2 k_z21 (zero_to_one, general case), K3 k_merge.  Part of ps_kernels.hpp.
#pragma once
#include "ps_kernels_common.hpp"

namespace ps {

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

}  // namespace ps
