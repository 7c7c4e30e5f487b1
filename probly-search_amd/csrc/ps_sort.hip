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

}  // namespace ps
