// ps_build.hip — GPU bulk indexing: the per-token work of Index::add_document
// (src/index.rs:77-158: tokenise every field, find-or-create the term's trie node, count the term's
// frequency per field of the document) for a whole flat corpus at once, as
//   tokenise (one thread per byte)  ->  hash every token  ->  stable radix sort by hash
//   ->  term boundaries (distinct strings; equal hashes are verified byte by byte)
//   ->  (term, document) boundaries  ->  per-posting term-frequency counts (segmented reduce)
// i.e. "add_document as sort-by-term + segmented reduce".  What comes back to the host is already
// grouped: per distinct term its first occurrence (the trie is built by interning the terms in
// first-occurrence order, which reproduces the reference's newest-first child lists) and its
// postings in document order with their per-field term frequencies; per document its field lengths.
// Index::bulk_load (ps_index.cpp) assembles the mutable host index from that without tokenising,
// hashing or searching anything again.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <stdexcept>
#include <string>

#include "ps_build.hpp"
#include "ps_errors.hpp"

namespace ps {

namespace {

#define PB_HIP(call)                                                                              \
  do {                                                                                            \
    hipError_t _e = (call);                                                                       \
    if (_e != hipSuccess)                                                                         \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #call); \
  } while (0)

template <typename T>
struct Dev {
  T* p = nullptr;
  explicit Dev(size_t n) { PB_HIP(hipMalloc((void**)&p, (n ? n : 1) * sizeof(T))); }
  ~Dev() { if (p) (void)hipFree(p); }
  Dev(const Dev&) = delete;
  Dev& operator=(const Dev&) = delete;
};

// segment s = (document s / F, field s % F) owns text[offsets[s], offsets[s+1])
__device__ __forceinline__ uint32_t segment_of(const uint64_t* offsets, uint32_t n_seg, uint64_t pos) {
  // last segment whose start is <= pos and that is not empty at pos (empty segments share a start)
  uint32_t lo = 0, hi = n_seg;  // upper_bound(offsets[0..n_seg), pos) - 1
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (offsets[mid] <= pos) lo = mid + 1; else hi = mid;
  }
  return lo - 1;
}

// A token starts at a non-space byte that follows a space or opens its segment (`s.split(' ')`,
// src/lib.rs:42-44; empty tokens are skipped at index time, src/index.rs:101).
__global__ void k_token_starts(const char* text, uint64_t n, const uint8_t* seg_start, uint32_t* is_start) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    is_start[i] = (text[i] != ' ' && (i == 0 || text[i - 1] == ' ' || seg_start[i])) ? 1u : 0u;
}

__global__ void k_mark_segments(const uint64_t* offsets, uint32_t n_seg, uint64_t n, uint8_t* seg_start) {
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_seg; s += gridDim.x * blockDim.x)
    if (offsets[s] < n) seg_start[offsets[s]] = 1;
}

__global__ void k_token_records(const char* text, uint64_t n, const uint64_t* offsets, uint32_t n_seg, const uint32_t* is_start,
                                const uint32_t* ordinal, uint32_t* tok_pos, uint32_t* tok_len, uint32_t* tok_seg, uint64_t* tok_hash,
                                uint32_t* tok_id, uint32_t* fl) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    if (!is_start[i]) continue;
    const uint32_t t = ordinal[i];
    const uint32_t seg = segment_of(offsets, n_seg, i);
    const uint64_t end = offsets[seg + 1];
    uint64_t h = 1469598103934665603ull;  // FNV-1a, then a finalizer: only grouping quality matters (equality is verified)
    uint64_t j = i;
    for (; j < end && text[j] != ' '; ++j) { h ^= (unsigned char)text[j]; h *= 1099511628211ull; }
    h ^= h >> 32;
    h *= 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    tok_pos[t] = (uint32_t)i;
    tok_len[t] = (uint32_t)(j - i);
    tok_seg[t] = seg;
    tok_hash[t] = h;
    tok_id[t] = t;
    atomicAdd(&fl[seg], 1u);  // DocumentDetails::field_length = non-empty tokens of the field (index.rs:101-114)
  }
}

__device__ __forceinline__ bool same_string(const char* text, uint32_t pa, uint32_t la, uint32_t pb, uint32_t lb) {
  if (la != lb) return false;
  for (uint32_t k = 0; k < la; ++k)
    if (text[pa + k] != text[pb + k]) return false;
  return true;
}

// sorted position j starts a new TERM if its hash differs from its predecessor's; equal hashes of
// different strings (a 64-bit collision) are reported and the caller falls back to the host indexer.
// It starts a new POSTING if the term or the document changes (tokens of one term are in text order
// after the stable sort, i.e. grouped by document, fields ascending).
__global__ void k_boundaries(const char* text, const uint64_t* hash, const uint32_t* tok, const uint32_t* tok_pos,
                             const uint32_t* tok_len, const uint32_t* tok_seg, uint32_t n_tok, uint32_t F, uint32_t* term_flag,
                             uint32_t* post_flag, uint32_t* collision) {
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_tok; j += gridDim.x * blockDim.x) {
    bool nt = j == 0 || hash[j] != hash[j - 1];
    if (!nt && !same_string(text, tok_pos[tok[j]], tok_len[tok[j]], tok_pos[tok[j - 1]], tok_len[tok[j - 1]])) {
      atomicExch(collision, 1u);
      nt = true;
    }
    term_flag[j] = nt ? 1u : 0u;
    post_flag[j] = (nt || tok_seg[tok[j]] / F != tok_seg[tok[j - 1]] / F) ? 1u : 0u;
  }
}

__global__ void k_fill(const uint32_t* tok, const uint32_t* tok_pos, const uint32_t* tok_len, const uint32_t* tok_seg,
                       const uint32_t* term_flag, const uint32_t* post_flag, const uint32_t* term_id, const uint32_t* post_id,
                       uint32_t n_tok, uint32_t F, uint32_t* term_pos, uint32_t* term_len, uint32_t* term_first_tok,
                       uint32_t* term_post_begin, uint32_t* post_doc, uint32_t* post_tf) {
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_tok; j += gridDim.x * blockDim.x) {
    const uint32_t t = tok[j], seg = tok_seg[t];
    const uint32_t pid = post_id[j] - 1;  // inclusive scans: ids are 1-based
    atomicAdd(&post_tf[(uint64_t)pid * F + seg % F], 1u);  // DocumentPointer::term_frequency[field]
    if (post_flag[j]) post_doc[pid] = seg / F;
    if (term_flag[j]) {
      const uint32_t tid = term_id[j] - 1;
      term_pos[tid] = tok_pos[t];
      term_len[tid] = tok_len[t];
      term_first_tok[tid] = t;  // stable sort: the first token of the run is the term's first occurrence
      term_post_begin[tid] = pid;
    }
  }
}

template <typename In, typename Out>
void inclusive_scan_u32(const In* in, Out* out, size_t n, hipStream_t st) {
  size_t bytes = 0;
  PB_HIP(rocprim::inclusive_scan(nullptr, bytes, in, out, n, rocprim::plus<uint32_t>(), st));
  Dev<unsigned char> tmp(bytes + 16);
  PB_HIP(rocprim::inclusive_scan(tmp.p, bytes, in, out, n, rocprim::plus<uint32_t>(), st));
  PB_HIP(hipStreamSynchronize(st));
}

}  // namespace

bool gpu_group_corpus(int device, uint32_t F, size_t n_docs, const char* text, const uint64_t* offsets, GroupedCorpus& out) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) throw NoDeviceError("no HIP device available for GPU bulk indexing");
  if (device < 0 || device >= n_dev) throw std::invalid_argument("device index out of range");
  PB_HIP(hipSetDevice(device));
  const uint64_t n_seg64 = (uint64_t)n_docs * F;
  if (n_seg64 >= 0xFFFFFFF0ull) throw std::length_error("GPU bulk indexing: more than 2^32 (document, field) segments");
  const uint32_t n_seg = (uint32_t)n_seg64;
  // the kernels follow offsets[] without further checks: it has to start at 0 and never decrease
  if (n_seg && offsets[0] != 0) throw std::invalid_argument("GPU bulk indexing: offsets[0] must be 0");
  for (uint32_t i = 0; i < n_seg; ++i)
    if (offsets[i + 1] < offsets[i]) throw std::invalid_argument("GPU bulk indexing: offsets must be non-decreasing");
  const uint64_t n = n_seg ? offsets[n_seg] : 0;  // bytes of text
  if (n >= 0xFFFFFFF0ull) throw std::length_error("GPU bulk indexing: more than 4 GiB of text in one call (split the corpus)");
  out = GroupedCorpus{};
  out.field_length.assign(n_seg, 0);
  if (n == 0) return true;
  hipStream_t st = nullptr;
  Dev<char> d_text(n);
  Dev<uint64_t> d_off(n_seg + 1);
  PB_HIP(hipMemcpy(d_text.p, text, n, hipMemcpyHostToDevice));
  PB_HIP(hipMemcpy(d_off.p, offsets, ((size_t)n_seg + 1) * 8, hipMemcpyHostToDevice));
  uint32_t n_tok = 0;
  Dev<uint32_t> d_fl(n_seg);
  PB_HIP(hipMemset(d_fl.p, 0, (size_t)n_seg * 4));
  const dim3 blk(256), grd(4096);
  // ---- tokenise ----
  Dev<uint32_t> d_is_start(n);
  {
    Dev<uint8_t> d_seg_start(n);
    PB_HIP(hipMemset(d_seg_start.p, 0, n));
    hipLaunchKernelGGL(k_mark_segments, grd, blk, 0, st, d_off.p, n_seg, n, d_seg_start.p);
    hipLaunchKernelGGL(k_token_starts, grd, blk, 0, st, d_text.p, n, d_seg_start.p, d_is_start.p);
    PB_HIP(hipGetLastError());
    PB_HIP(hipStreamSynchronize(st));
  }
  Dev<uint32_t> d_ord(n);
  {
    size_t bytes = 0;
    PB_HIP(rocprim::exclusive_scan(nullptr, bytes, d_is_start.p, d_ord.p, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
    Dev<unsigned char> tmp(bytes + 16);
    PB_HIP(rocprim::exclusive_scan(tmp.p, bytes, d_is_start.p, d_ord.p, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
    uint32_t last_ord = 0, last_flag = 0;
    PB_HIP(hipMemcpy(&last_ord, d_ord.p + (n - 1), 4, hipMemcpyDeviceToHost));
    PB_HIP(hipMemcpy(&last_flag, d_is_start.p + (n - 1), 4, hipMemcpyDeviceToHost));
    n_tok = last_ord + last_flag;
  }
  out.n_tokens = n_tok;
  if (n_tok == 0) {
    PB_HIP(hipMemcpy(out.field_length.data(), d_fl.p, (size_t)n_seg * 4, hipMemcpyDeviceToHost));
    return true;
  }
  Dev<uint32_t> d_tok_pos(n_tok), d_tok_len(n_tok), d_tok_seg(n_tok), d_tok_id(n_tok), d_tok_sorted(n_tok);
  Dev<uint64_t> d_hash(n_tok), d_hash_sorted(n_tok);
  hipLaunchKernelGGL(k_token_records, grd, blk, 0, st, d_text.p, n, d_off.p, n_seg, d_is_start.p, d_ord.p, d_tok_pos.p, d_tok_len.p,
                     d_tok_seg.p, d_hash.p, d_tok_id.p, d_fl.p);
  PB_HIP(hipGetLastError());
  // ---- sort by term (hash), stable: tokens of a term stay in text order ----
  {
    size_t bytes = 0;
    PB_HIP(rocprim::radix_sort_pairs(nullptr, bytes, d_hash.p, d_hash_sorted.p, d_tok_id.p, d_tok_sorted.p, (size_t)n_tok, 0, 64, st));
    Dev<unsigned char> tmp(bytes + 16);
    PB_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, d_hash.p, d_hash_sorted.p, d_tok_id.p, d_tok_sorted.p, (size_t)n_tok, 0, 64, st));
    PB_HIP(hipStreamSynchronize(st));
  }
  // ---- term / posting boundaries, ids by inclusive scan ----
  Dev<uint32_t> d_term_flag(n_tok), d_post_flag(n_tok), d_term_id(n_tok), d_post_id(n_tok), d_collision(1);
  PB_HIP(hipMemset(d_collision.p, 0, 4));
  hipLaunchKernelGGL(k_boundaries, grd, blk, 0, st, d_text.p, d_hash_sorted.p, d_tok_sorted.p, d_tok_pos.p, d_tok_len.p, d_tok_seg.p,
                     n_tok, F, d_term_flag.p, d_post_flag.p, d_collision.p);
  PB_HIP(hipGetLastError());
  inclusive_scan_u32(d_term_flag.p, d_term_id.p, n_tok, st);
  inclusive_scan_u32(d_post_flag.p, d_post_id.p, n_tok, st);
  uint32_t collision = 0, n_terms = 0, n_post = 0;
  PB_HIP(hipMemcpy(&collision, d_collision.p, 4, hipMemcpyDeviceToHost));
  if (collision) return false;  // two different terms share a 64-bit hash: the caller takes the host indexer
  PB_HIP(hipMemcpy(&n_terms, d_term_id.p + (n_tok - 1), 4, hipMemcpyDeviceToHost));
  PB_HIP(hipMemcpy(&n_post, d_post_id.p + (n_tok - 1), 4, hipMemcpyDeviceToHost));
  // ---- segmented reduce: per posting its document and per-field term frequencies ----
  Dev<uint32_t> d_term_pos(n_terms), d_term_len(n_terms), d_term_first(n_terms), d_term_pb(n_terms), d_post_doc(n_post);
  Dev<uint32_t> d_post_tf((size_t)n_post * F);
  PB_HIP(hipMemset(d_post_tf.p, 0, (size_t)n_post * F * 4));
  hipLaunchKernelGGL(k_fill, grd, blk, 0, st, d_tok_sorted.p, d_tok_pos.p, d_tok_len.p, d_tok_seg.p, d_term_flag.p, d_post_flag.p,
                     d_term_id.p, d_post_id.p, n_tok, F, d_term_pos.p, d_term_len.p, d_term_first.p, d_term_pb.p, d_post_doc.p,
                     d_post_tf.p);
  PB_HIP(hipGetLastError());
  PB_HIP(hipStreamSynchronize(st));
  out.term_pos.resize(n_terms); out.term_len.resize(n_terms); out.term_first_token.resize(n_terms);
  out.term_post_begin.resize((size_t)n_terms + 1);
  out.post_doc.resize(n_post);
  out.post_tf.resize((size_t)n_post * F);
  PB_HIP(hipMemcpy(out.term_pos.data(), d_term_pos.p, (size_t)n_terms * 4, hipMemcpyDeviceToHost));
  PB_HIP(hipMemcpy(out.term_len.data(), d_term_len.p, (size_t)n_terms * 4, hipMemcpyDeviceToHost));
  PB_HIP(hipMemcpy(out.term_first_token.data(), d_term_first.p, (size_t)n_terms * 4, hipMemcpyDeviceToHost));
  PB_HIP(hipMemcpy(out.term_post_begin.data(), d_term_pb.p, (size_t)n_terms * 4, hipMemcpyDeviceToHost));
  out.term_post_begin[n_terms] = n_post;
  PB_HIP(hipMemcpy(out.post_doc.data(), d_post_doc.p, (size_t)n_post * 4, hipMemcpyDeviceToHost));
  PB_HIP(hipMemcpy(out.post_tf.data(), d_post_tf.p, (size_t)n_post * F * 4, hipMemcpyDeviceToHost));
  PB_HIP(hipMemcpy(out.field_length.data(), d_fl.p, (size_t)n_seg * 4, hipMemcpyDeviceToHost));
  return true;
}

}  // namespace ps
