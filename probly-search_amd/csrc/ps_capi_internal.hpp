// ps_capi_internal.hpp — the handle structs behind the opaque C ABI types, shared by the
// translation units that implement the ABI (ps_capi.cpp, ps_comm.hip).
#pragma once
#include <memory>
#include <mutex>

#include "../../include/probly_search_amd.h"
#include "ps_engine.hpp"
#include "ps_pool.hpp"
#include "ps_snapshot.hpp"

struct ps_snapshot {
  std::shared_ptr<ps::Snapshot> snap;  // shared by the replicas of ps_index_snapshot_multi
  std::unique_ptr<ps::Engine> engine;  // null for host-only snapshots
  int device = -1;
  uint32_t tile_docs = 0, headroom_pct = 0;  // as requested at creation (a full re-flatten reuses them)
  std::mutex stats_mu;
  ps_batch_stats last{};
  std::mutex pool_mu;
  std::unique_ptr<ps::Pool> pool;  // planner threads, created on the first large batch
};

namespace ps {
// Plans `n_queries` flat queries and enqueues the batch on `hip_stream` (ps_capi.cpp).
ps_status run_device_flat(ps_snapshot* snap, const ps_scorer_desc* scorer, const char* text, const uint64_t* offsets,
                          size_t n_queries, const double* fields_boost, size_t n_boost, ps_tokenizer_fn tokenizer,
                          void* user, size_t top_k, void* d_keys, void* d_scores, void* d_counts, void* hip_stream);
ps_status set_error(ps_status st, const char* msg);
}  // namespace ps
