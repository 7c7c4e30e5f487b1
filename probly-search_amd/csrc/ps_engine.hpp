// ps_engine.hpp — device side of a snapshot: HBM-resident CSR planes, per-batch plan upload,
// the scoring kernels and result download.  Implemented in ps_engine.hip (HIP, gfx950 only).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ps_snapshot.hpp"

namespace ps {

struct EngineImpl;

// Result records on their way to the C caller: a malloc'd block that grows without value-initialising
// (a full-result batch is hundreds of MB: every avoidable pass over it shows) and is handed over as it is.
// Large result blocks are PINNED host memory the device writes into directly (no pageable copy, whose first-touch
// page faults cost 17 ms for the 213 MB of a 24-query C2 batch); ps_free() hands such a block back to a process-wide
// pool instead of the allocator, the next batch reuses it (PS_RESULT_POOL_MB of idle blocks are kept, default 1024).
// result_block_release returns false for a pointer the pool does not know (then it is a malloc'd block).
void* result_block_acquire(size_t bytes, size_t* capacity);  // nullptr: no pinned memory to be had
bool result_block_release(void* p);

struct ResultBuf {
  ps_result* p = nullptr;
  size_t n = 0;
  size_t pinned_cap = 0;  // bytes of the pool block p points to, 0 = malloc'd
  ResultBuf() = default;
  ResultBuf(const ResultBuf&) = delete;
  ResultBuf& operator=(const ResultBuf&) = delete;
  ~ResultBuf() { drop(); }
  void drop() {
    if (pinned_cap) result_block_release(p); else free(p);
    p = nullptr;
    n = pinned_cap = 0;
  }
  void resize(size_t k) {
    const size_t bytes = (k ? k : 1) * sizeof(ps_result);
    if (pinned_cap) {
      if (bytes > pinned_cap) {  // grow: a larger pool block, or back to the allocator
        size_t cap = 0;
        void* q = result_block_acquire(bytes, &cap);
        if (!q) { q = malloc(bytes); cap = 0; }
        if (!q) throw std::bad_alloc();
        memcpy(q, p, n * sizeof(ps_result));
        result_block_release(p);
        p = static_cast<ps_result*>(q);
        pinned_cap = cap;
      }
      n = k;
      return;
    }
    void* q = realloc(p, bytes);
    if (!q) throw std::bad_alloc();
    p = static_cast<ps_result*>(q);
    n = k;
  }
  // k results in a pinned pool block when there is one to be had (contents undefined); false: malloc'd as resize()
  bool resize_pinned(size_t k) {
    drop();
    size_t cap = 0;
    void* q = result_block_acquire((k ? k : 1) * sizeof(ps_result), &cap);
    if (!q) { resize(k); return false; }
    p = static_cast<ps_result*>(q);
    pinned_cap = cap;
    n = k;
    return true;
  }
  void clear() { n = 0; }
  size_t size() const { return n; }
  ps_result* data() { return p; }
  ps_result& operator[](size_t i) { return p[i]; }
  void append(const ResultBuf& o) {
    const size_t at = n;
    resize(n + o.n);
    if (o.n) memcpy(p + at, o.p, o.n * sizeof(ps_result));
  }
  ps_result* release() {  // the caller's block now: ps_free() knows both kinds
    if (!p) resize(0);
    ps_result* r = p;
    p = nullptr;
    n = pinned_cap = 0;
    return r;
  }
};

class Engine {
 public:
  // Uploads the snapshot's planes to `device`.  Throws std::runtime_error on HIP failure.
  Engine(const Snapshot& snap, int device);
  ~Engine();
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;

  // Runs one planned batch.  top_k == 0: every match (canonical order) into out/offsets.
  // top_k  > 0: the first top_k of the canonical order per query.
  void run_host(const ps_scorer_desc& sc, const double* boosts, const Plan& plan, size_t top_k,
                ResultBuf& out, std::vector<size_t>& offsets, ps_batch_stats& stats);
  // Device-resident top-k (1..PS_MAX_DEVICE_TOPK) into caller buffers, ordered on `stream`
  // (nullptr = engine stream, synchronous).
  void run_device(const ps_scorer_desc& sc, const double* boosts, const Plan& plan, size_t top_k, void* d_keys,
                  void* d_scores, void* d_counts, void* stream, ps_batch_stats& stats);

  // SURVEY 8f N2 - device-side planner (BM25 and zero_to_one, built-in tokenizer): tokenise, term lookup in the frozen
  // trie, prefix expansion and before_each run on the GPU.  plan_device copies the plan back (tests);
  // run_device_planned scores the batch without the plan ever existing on the host.
  void plan_device(const char* text, const uint64_t* offsets, size_t n_queries, Plan& out);
  // Whether a flat BM25 top-k batch of this size (built-in tokenizer) should be planned on the device
  // (PS_DEVICE_PLAN, default on; small batches keep the host planner's latency path).
  bool wants_device_plan(size_t n_queries);
  // Announce the NEXT flat BM25 batch (ps_snapshot_plan_ahead_flat): its planner count pass starts now.
  bool plan_ahead(const ps_scorer_desc& sc, const char* text, const uint64_t* offsets, size_t n_queries);
  // false (zero_to_one only): the batch is not one K1dz takes (a query that is not simple, more than 4 lists, fewer
  // than PS_DAAT_MIN_BATCH queries ...) - nothing was enqueued, the caller plans it on the host.
  bool run_device_planned(const ps_scorer_desc& sc, const double* boosts, const char* text, const uint64_t* offsets,
                          size_t n_queries, size_t top_k, void* d_keys, void* d_scores, void* d_counts, void* stream,
                          ps_batch_stats& stats);

  // HIP-event durations summed over every batch since the last reset: the scoring kernel alone and
  // K0 / K0b in front of it (waits for outstanding launches).
  void kernel_times(ps_kernel_times& out, bool reset);
  // What the scoring kernels counted themselves (postings scanned, lookups by kind, ...) since the last reset.
  void work_counters(ps_work_counters& out, bool reset);
  // Device side of Snapshot::apply_delta: uploads exactly the ranges it changed (appended postings,
  // table entries, keys, alive words), drops what was derived from the old state and re-counts, on
  // the device, the per-layer pointer count of delta-removed documents.  Call with no batch in flight.
  void apply_delta(const DeltaRanges& r, std::vector<uint64_t>& removed_df, uint64_t* bytes_uploaded);
  uint64_t device_bytes() const;
  int device() const;

 private:
  EngineImpl* impl_;
};

int device_count();
// Tuning knobs by name (the PS_* names of DESIGN.md section 11); read when a snapshot's engine is created.
void set_option(const char* name, uint32_t value);
bool get_option(const char* name, uint32_t* value);

}  // namespace ps
