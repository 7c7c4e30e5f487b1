// ps_sort.hip — K4: canonical ordering of full result lists on the device.
//
// Index::query ends with `result.sort_by(score desc)` over every matching document
// (src/query.rs:97-105); the canonical tie-break is key ascending (test_util::test_score,
// src/lib.rs:54-58) and documents are numbered in ascending key order, so the order wanted is
// (score desc, doc id asc) inside each query's run.  Two stable segmented radix sorts
// (rocPRIM): doc id ascending, then score descending.  Scores are >= +0.0 on this path (BM25
// emits only s > 0, zero_to_one only sums of positives), so the f64 bit pattern sorted as u64 is
// the numeric order.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <algorithm>

#include "ps_sort.hpp"

namespace ps {

__global__ void k_seg_bounds(const uint64_t* off, const uint32_t* cnt, uint32_t n, uint32_t* begin, uint32_t* end) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) {
    begin[q] = (uint32_t)off[q];
    end[q] = (uint32_t)off[q] + cnt[q];
  }
}

hipError_t sort_results(SortBuffers& b, unsigned n_items, unsigned n_segments, const uint64_t* d_off,
                        const uint32_t* d_cnt, void* temp, size_t& temp_bytes, hipStream_t st) {
  if (temp != nullptr && n_segments) {
    hipLaunchKernelGGL(k_seg_bounds, dim3((n_segments + 255) / 256), dim3(256), 0, st, d_off, d_cnt, n_segments,
                       b.seg_begin, b.seg_end);
  }
  size_t t1 = temp_bytes, t2 = temp_bytes;
  hipError_t e = rocprim::segmented_radix_sort_pairs(temp, t1, b.doc, b.doc_tmp, b.score_bits, b.score_tmp, n_items,
                                                     n_segments, b.seg_begin, b.seg_end, 0, 32, st);
  if (e != hipSuccess) return e;
  e = rocprim::segmented_radix_sort_pairs_desc(temp, t2, b.score_tmp, b.score_bits, b.doc_tmp, b.doc, n_items,
                                               n_segments, b.seg_begin, b.seg_end, 0, 64, st);
  if (e != hipSuccess) return e;
  if (temp == nullptr) temp_bytes = t1 > t2 ? t1 : t2;
  return hipSuccess;
}

// One large run: device-wide radix sorts (the segmented sort gives a segment to one workgroup,
// which is the wrong shape for a single query with ~10^6 matches).
hipError_t sort_run(SortBuffers& b, size_t offset, unsigned n, void* temp, size_t& temp_bytes, hipStream_t st) {
  size_t t1 = temp_bytes, t2 = temp_bytes;
  hipError_t e = rocprim::radix_sort_pairs(temp, t1, b.doc + offset, b.doc_tmp + offset, b.score_bits + offset,
                                           b.score_tmp + offset, n, 0, 32, st);
  if (e != hipSuccess) return e;
  e = rocprim::radix_sort_pairs_desc(temp, t2, b.score_tmp + offset, b.score_bits + offset, b.doc_tmp + offset,
                                     b.doc + offset, n, 0, 64, st);
  if (e != hipSuccess) return e;
  if (temp == nullptr) temp_bytes = t1 > t2 ? t1 : t2;
  return hipSuccess;
}

// ---- one set of device-wide sorts for all runs -------------------------------------------------------------------
namespace {
// the run a compact index belongs to: the last q with off[q] <= j
__device__ __forceinline__ uint32_t run_of(const uint64_t* off, uint32_t n_runs, uint64_t j) {
  uint32_t lo = 0, hi = n_runs;  // off[lo] <= j < off[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= j) lo = mid; else hi = mid;
  }
  return lo;
}
__global__ __launch_bounds__(256) void k_gs_compact(const uint32_t* doc, const uint64_t* score, const uint64_t* run_off,
                                                    const uint64_t* cmp_off, uint32_t run0, uint32_t n_runs,
                                                    uint32_t first, uint32_t n, uint32_t* kd, uint64_t* sc, uint32_t* idx) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const uint32_t j = first + t;
    const uint32_t q = run0 + run_of(cmp_off + run0, n_runs, j);
    const uint64_t src = run_off[q] + (j - cmp_off[q]);
    kd[j] = doc[src];
    sc[j] = score[src];
    idx[j] = j;
  }
}
__global__ __launch_bounds__(256) void k_gs_gather_score(const uint64_t* sc, const uint32_t* perm, uint32_t n, uint64_t* ks) {
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) ks[j] = sc[perm[j]];
}
__global__ __launch_bounds__(256) void k_gs_gather_run(const uint64_t* cmp_off, uint32_t n_runs, const uint32_t* perm,
                                                       uint32_t n, uint32_t* kq) {
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
    kq[j] = run_of(cmp_off, n_runs, perm[j]);
}
struct PackedResult { uint64_t key; uint64_t score_bits; };  // == ps_result {u64 key; f64 score}
__global__ __launch_bounds__(256) void k_gs_pack(const uint32_t* kd, const uint64_t* sc, const uint32_t* perm,
                                                 const uint64_t* cmp_off, const uint64_t* out_off, uint32_t run0,
                                                 uint32_t n_runs, uint64_t out_first, uint64_t n_out, const uint64_t* keys,
                                                 PackedResult* out) {
  for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < n_out; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t o = out_first + t;
    const uint32_t q = run0 + run_of(out_off + run0, n_runs, o);
    const uint32_t i = perm[cmp_off[q] + (o - out_off[q])];
    out[o] = PackedResult{keys[kd[i]], sc[i]};
  }
}
inline unsigned bits_for(uint32_t count) {  // bits needed for values < count
  unsigned b = 1;
  while (b < 32 && (1ull << b) < count) ++b;
  return b;
}
inline unsigned grid_for(uint64_t n) { return (unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 16); }
}  // namespace

hipError_t sort_runs_global(GlobalSort& g, void* temp, size_t& temp_bytes, hipStream_t st) {
  const unsigned doc_bits = bits_for(g.n_docs), run_bits = bits_for(g.n_runs);
  // the part's slices of the scratch arrays (the permutation VALUES stay indices of the whole batch)
  uint32_t *kd = g.kd + g.first, *k32 = g.k32 + g.first, *ia = g.ia + g.first, *ib = g.ib + g.first;
  uint64_t *ks = g.ks + g.first, *ks_out = g.ks_out + g.first;
  uint32_t* kq_out = reinterpret_cast<uint32_t*>(ks_out);  // free again once the score sort is done
  size_t t1 = temp_bytes, t2 = temp_bytes, t3 = temp_bytes;
  hipError_t e;
  if (temp != nullptr) {
    hipLaunchKernelGGL(k_gs_compact, dim3(grid_for(g.n)), dim3(256), 0, st, g.doc, g.score_bits, g.run_off, g.cmp_off,
                       g.run0, g.n_runs, g.first, g.n, g.kd, g.sc, g.ia);
  }
  e = rocprim::radix_sort_pairs(temp, t1, kd, k32, ia, ib, g.n, 0, doc_bits, st);
  if (e != hipSuccess) return e;
  if (temp != nullptr) hipLaunchKernelGGL(k_gs_gather_score, dim3(grid_for(g.n)), dim3(256), 0, st, g.sc, ib, g.n, ks);
  e = rocprim::radix_sort_pairs_desc(temp, t2, ks, ks_out, ib, ia, g.n, 0, 64, st);
  if (e != hipSuccess) return e;
  g.perm = g.ia;
  if (g.n_runs > 1) {
    if (temp != nullptr)
      hipLaunchKernelGGL(k_gs_gather_run, dim3(grid_for(g.n)), dim3(256), 0, st, g.cmp_off + g.run0, g.n_runs, ia, g.n, k32);
    e = rocprim::radix_sort_pairs(temp, t3, k32, kq_out, ia, ib, g.n, 0, run_bits, st);
    if (e != hipSuccess) return e;
    g.perm = g.ib;
  } else {
    t3 = 0;
  }
  if (temp == nullptr) temp_bytes = std::max(t1, std::max(t2, t3));
  return temp != nullptr ? hipGetLastError() : hipSuccess;
}

hipError_t pack_sorted(const GlobalSort& g, const uint64_t* out_off, uint64_t out_first, uint64_t n_out,
                       const uint64_t* keys, void* out, hipStream_t st) {
  if (!n_out) return hipSuccess;
  hipLaunchKernelGGL(k_gs_pack, dim3(grid_for(n_out)), dim3(256), 0, st, g.kd, g.sc, g.perm, g.cmp_off, out_off, g.run0,
                     g.n_runs, out_first, n_out, keys, static_cast<PackedResult*>(out));
  return hipGetLastError();
}

}  // namespace ps
