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
struct ResultBuf {
  ps_result* p = nullptr;
  size_t n = 0;
  ResultBuf() = default;
  ResultBuf(const ResultBuf&) = delete;
  ResultBuf& operator=(const ResultBuf&) = delete;
  ~ResultBuf() { free(p); }
  void resize(size_t k) {
    void* q = realloc(p, (k ? k : 1) * sizeof(ps_result));
    if (!q) throw std::bad_alloc();
    p = static_cast<ps_result*>(q);
    n = k;
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
  ps_result* release() {
    if (!p) resize(0);
    ps_result* r = p;
    p = nullptr;
    n = 0;
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

  // SURVEY 8f N2 - device-side planner (BM25, built-in tokenizer): tokenise, term lookup in the frozen
  // trie, prefix expansion and before_each run on the GPU.  plan_device copies the plan back (tests);
  // run_device_planned scores the batch without the plan ever existing on the host.
  void plan_device(const char* text, const uint64_t* offsets, size_t n_queries, Plan& out);
  // Whether a flat BM25 top-k batch of this size (built-in tokenizer) should be planned on the device
  // (PS_DEVICE_PLAN, default on; small batches keep the host planner's latency path).
  bool wants_device_plan(size_t n_queries);
  void run_device_planned(const ps_scorer_desc& sc, const double* boosts, const char* text, const uint64_t* offsets,
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
