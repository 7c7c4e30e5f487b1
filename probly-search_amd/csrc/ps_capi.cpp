// ps_capi.cpp — the extern "C" surface declared in include/probly_search_amd.h.
// Translates C++ exceptions into ps_status codes (nothing unwinds across the ABI).
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>

#include "../../include/probly_search_amd.h"
#include "ps_build.hpp"
#include "ps_capi_internal.hpp"
#include "ps_engine.hpp"
#include "ps_errors.hpp"
#include "ps_index.hpp"
#include "ps_pool.hpp"
#include "ps_snapshot.hpp"

struct ps_index {
  ps::Index idx;
  ps_snapshot* cached = nullptr;  // ps_index_query's lazily rebuilt snapshot
  explicit ps_index(size_t f, size_t a = 1000, size_t b = 10000) : idx(f, a, b) {}
};

namespace {

thread_local std::string g_err;

template <typename Fn>
ps_status guard(Fn&& fn) {
  try {
    g_err.clear();
    return fn();
  } catch (const std::bad_alloc&) {
    g_err = "out of memory";
    return PS_ENOMEM;
  } catch (const std::invalid_argument& e) {
    g_err = e.what();
    return PS_EINVAL;
  } catch (const std::length_error& e) {
    g_err = e.what();
    return PS_EUNSUPPORTED;
  } catch (const ps::NoDeviceError& e) {
    g_err = e.what();
    return PS_ENODEVICE;
  } catch (const ps::RcclError& e) {
    g_err = e.what();
    return PS_ERCCL;
  } catch (const std::exception& e) {
    g_err = e.what();
    return PS_EHIP;
  } catch (...) {
    g_err = "unknown error";
    return PS_EHIP;
  }
}

ps_status fail(ps_status s, const char* msg) {
  g_err = msg;
  return s;
}

ps_status check_query_args(const ps_snapshot* snap, const ps_scorer_desc* sc, const double* boosts, size_t n_boost) {
  if (!snap || !sc) return fail(PS_EINVAL, "null snapshot or scorer");
  if (sc->kind == PS_SCORER_HOST_CALLBACKS)
    return fail(PS_EINVAL, "PS_SCORER_HOST_CALLBACKS walks the host index in the reference's list order: use ps_index_query");
  if (sc->kind != PS_SCORER_BM25 && sc->kind != PS_SCORER_ZERO_TO_ONE) return fail(PS_EINVAL, "unknown scorer kind");
  // the reference indexes fields_boost[x] for x < fields_num and panics if it is shorter (bm25.rs:85)
  if (n_boost < snap->snap->F || (snap->snap->F && !boosts)) return fail(PS_EINVAL, "fields_boost shorter than fields_num");
  if (!snap->engine) return fail(PS_ENODEVICE, "host-only snapshot: no HIP device attached (no CPU scoring fallback)");
  return PS_OK;
}

void set_stats(ps_snapshot* s, const ps_batch_stats& st) {
  std::lock_guard<std::mutex> l(s->stats_mu);
  s->last = st;
}

double wall_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// Host planner for a whole batch.  Queries are independent (each owns its scores / visited maps
// in the reference, src/query.rs:31,37), so with the built-in tokenizer the batch is planned by a
// small persistent thread pool, each thread producing a private Plan that is then concatenated
// in query order.  A caller-supplied tokenizer callback is not assumed to be thread-safe.
void plan_batch(ps_snapshot* snap, const ps_scorer_desc& sc, const std::vector<std::string_view>& qs,
                ps_tokenizer_fn tok, void* user, ps::Plan& plan) {
  plan.qbeg.assign(1, 0);
  const size_t n = qs.size();
  unsigned want = 1;
  if (tok == nullptr && n >= 128) {
    const char* env = getenv("PS_PLAN_THREADS");
    unsigned hw = std::thread::hardware_concurrency();
    want = env && *env ? (unsigned)strtoul(env, nullptr, 10) : std::min(16u, hw ? hw : 1u);
    if (want < 1) want = 1;
  }
  if (want == 1) {
    for (size_t i = 0; i < n; ++i) snap->snap->plan_query(sc, qs[i], tok, user, plan);
    return;
  }
  static const bool trace = getenv("PS_TRACE") && *getenv("PS_TRACE") == '2';
  const double tp0 = trace ? wall_ms() : 0.0;
  std::lock_guard<std::mutex> l(snap->pool_mu);
  if (!snap->pool || snap->pool->size() != want) snap->pool.reset(new ps::Pool(want - 1));
  std::vector<ps::Plan> parts(want);
  const double tp1 = trace ? wall_ms() : 0.0;
  snap->pool->run([&](unsigned part, unsigned nparts) {
    size_t b = n * part / nparts, e = n * (part + 1) / nparts;
    ps::Plan& pl = parts[part];
    pl.qbeg.assign(1, 0);
    for (size_t i = b; i < e; ++i) snap->snap->plan_query(sc, qs[i], nullptr, nullptr, pl);
  });
  const double tp2 = trace ? wall_ms() : 0.0;
  for (ps::Plan& pl : parts) {
    const uint32_t base = (uint32_t)plan.entries.size();
    plan.entries.insert(plan.entries.end(), pl.entries.begin(), pl.entries.end());
    for (size_t i = 1; i < pl.qbeg.size(); ++i) plan.qbeg.push_back(base + pl.qbeg[i]);
    plan.qterms_len.insert(plan.qterms_len.end(), pl.qterms_len.begin(), pl.qterms_len.end());
    plan.n_nodes.insert(plan.n_nodes.end(), pl.n_nodes.begin(), pl.n_nodes.end());
    plan.postings += pl.postings;
    plan.max_entries = std::max(plan.max_entries, pl.max_entries);
    plan.max_qterms = std::max(plan.max_qterms, pl.max_qterms);
    plan.max_nodes = std::max(plan.max_nodes, pl.max_nodes);
    plan.multi_expansion = plan.multi_expansion || pl.multi_expansion;
  }
  if (trace) fprintf(stderr, "[ps] plan_batch   setup %.3f  run %.3f  merge %.3f ms (%u threads)\n", tp1 - tp0, tp2 - tp1, wall_ms() - tp2, want);
}

std::vector<std::string_view> views_of(const ps_str* queries, size_t n) {
  std::vector<std::string_view> v(n);
  for (size_t i = 0; i < n; ++i) v[i] = std::string_view(queries[i].ptr ? queries[i].ptr : "", queries[i].len);
  return v;
}


}  // namespace

static ps_status run_device_views(ps_snapshot* snap, const ps_scorer_desc* scorer,
                                  const std::vector<std::string_view>& qs, const double* fields_boost, size_t n_boost,
                                  ps_tokenizer_fn tokenizer, void* user, size_t top_k, void* d_keys, void* d_scores,
                                  void* d_counts, void* hip_stream) {
  ps_status st = check_query_args(snap, scorer, fields_boost, n_boost);
  if (st != PS_OK) return st;
  if (!d_keys || !d_scores || !d_counts) return fail(PS_EINVAL, "null argument");
  const double t0 = wall_ms();
  ps::Plan plan;
  plan_batch(snap, *scorer, qs, tokenizer, user, plan);
  const double t1 = wall_ms();
  ps_batch_stats stats;
  snap->engine->run_device(*scorer, fields_boost, plan, top_k, d_keys, d_scores, d_counts, hip_stream, stats);
  stats.plan_ms = t1 - t0;
  stats.total_ms = wall_ms() - t0;
  set_stats(snap, stats);
  return PS_OK;
}

namespace ps {
ps_status set_error(ps_status st, const char* msg) { return fail(st, msg); }

ps_status run_device_flat(ps_snapshot* snap, const ps_scorer_desc* scorer, const char* text, const uint64_t* offsets,
                          size_t n_queries, const double* fields_boost, size_t n_boost, ps_tokenizer_fn tokenizer,
                          void* user, size_t top_k, void* d_keys, void* d_scores, void* d_counts, void* hip_stream) {
  return guard([&]() -> ps_status {
    if (n_queries && (!text || !offsets)) return fail(PS_EINVAL, "null argument");
    // BM25 top-k batches with the built-in tokenizer are planned on the device too (k_plan): the host
    // hands the text over and sizes the launches from the plan's totals
    // (zero_to_one: the batches K1dz takes; run_device_planned hands the others back before anything is enqueued)
    if (!tokenizer && snap && scorer && (scorer->kind == PS_SCORER_BM25 || scorer->kind == PS_SCORER_ZERO_TO_ONE) && snap->engine &&
        top_k >= 1 && top_k <= PS_MAX_DEVICE_TOPK && offsets && snap->engine->wants_device_plan(n_queries)) {
      ps_status st = check_query_args(snap, scorer, fields_boost, n_boost);
      if (st != PS_OK) return st;
      if (!d_keys || !d_scores || !d_counts) return fail(PS_EINVAL, "null argument");
      const double t0 = wall_ms();
      ps_batch_stats stats;
      if (snap->engine->run_device_planned(*scorer, fields_boost, text, offsets, n_queries, top_k, d_keys, d_scores, d_counts, hip_stream,
                                           stats)) {
        stats.total_ms = wall_ms() - t0;
        set_stats(snap, stats);
        return PS_OK;
      }
    }
    std::vector<std::string_view> qs(n_queries);
    for (size_t i = 0; i < n_queries; ++i) qs[i] = std::string_view(text + offsets[i], (size_t)(offsets[i + 1] - offsets[i]));
    return run_device_views(snap, scorer, qs, fields_boost, n_boost, tokenizer, user, top_k, d_keys, d_scores, d_counts,
                            hip_stream);
  });
}
}  // namespace ps

extern "C" {

const char* ps_last_error(void) { return g_err.c_str(); }
void ps_free(void* p) {
  if (!ps::result_block_release(p)) free(p);  // large result blocks are pinned pool blocks (ps_engine.hpp)
}
void ps_results_split(const ps_result* r, size_t n, uint64_t* keys, double* scores) {
  if (!r || !n) return;
  auto part = [=](size_t a, size_t b) {
    if (keys) for (size_t i = a; i < b; ++i) keys[i] = r[i].key;
    if (scores) for (size_t i = a; i < b; ++i) scores[i] = r[i].score;
  };
  const unsigned nt = n >= (1u << 20) ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
  if (nt <= 1) return part(0, n);
  std::vector<std::thread> th;
  const size_t per = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t) {
    const size_t a = std::min(n, (size_t)t * per), b = std::min(n, a + per);
    if (b > a) th.emplace_back(part, a, b);
  }
  for (auto& x : th) x.join();
}
int ps_device_count(void) { return ps::device_count(); }
uint32_t ps_abi_version(void) { return PS_ABI_VERSION; }
ps_status ps_set_option(const char* name, uint32_t value) {
  if (!name || strncmp(name, "PS_", 3) != 0) return fail(PS_EINVAL, "option names are the PS_* knob names");
  // The run-time options of the ABI (include/probly_search_amd.h): what a serving process may reasonably
  // switch.  Everything else the engine reads - several dozen experiment knobs whose A/B is settled - stays reachable through the
  // environment variable of its name, and through this call only when PS_EXPERIMENT_KNOBS=1 (tools/knob_sweep.py).
  static const char* const kRuntime[] = {
      "PS_DAAT", "PS_DAAT_MIN_BATCH", "PS_DAAT_MULTI", "PS_DAAT_SMALL", "PS_DAAT_SMALL_NL", "PS_DAAT_SPLIT", "PS_DAAT_Z", "PS_DAAT_Z_SPLIT",
      "PS_DAAT_PRIME", "PS_DAAT_CHUNK", "PS_DEVICE_PLAN", "PS_ROW_CACHE_MB", "PS_WORK_COUNTERS", "PS_KERNEL_TIMERS", "PS_SCORE_ALT", "PS_DCTX",
      "PS_PLAN_AHEAD_DEPTH", "PS_RESULT_PINNED_MIN_KB", "PS_FULL_PARTS_MIN_KB", "PS_DENSE_MIN_USES", "PS_DENSE_MAX_ROWS", "PS_PLAN_THREADS",
      "PS_FLATTEN_THREADS"};
  bool known = false;
  for (const char* k : kRuntime) known = known || strcmp(k, name) == 0;
  if (!known) {
    const char* e = getenv("PS_EXPERIMENT_KNOBS");
    if (!(e && *e == '1'))
      return fail(PS_EINVAL, (std::string(name) + " is not a run-time option (see ps_set_option in probly_search_amd.h); experiment knobs are "
                              "environment variables, or set PS_EXPERIMENT_KNOBS=1").c_str());
  }
  ps::set_option(name, value);
  return PS_OK;
}
int ps_get_option(const char* name, uint32_t* value) {
  uint32_t v = 0;
  if (!name || !ps::get_option(name, &v)) return 0;
  if (value) *value = v;
  return 1;
}

ps_status ps_index_new(size_t fields_num, ps_index** out) {
  return guard([&]() -> ps_status {
    if (!out) return fail(PS_EINVAL, "null out");
    *out = new ps_index(fields_num);
    return PS_OK;
  });
}

ps_status ps_index_new_with_capacity(size_t fields_num, size_t expected_index_size, size_t expected_documents_count,
                                     ps_index** out) {
  return guard([&]() -> ps_status {
    if (!out) return fail(PS_EINVAL, "null out");
    *out = new ps_index(fields_num, expected_index_size, expected_documents_count);
    return PS_OK;
  });
}

void ps_index_free(ps_index* idx) {
  if (!idx) return;
  if (idx->cached) ps_snapshot_free(idx->cached);
  delete idx;
}

ps_status ps_index_add_document(ps_index* idx, uint64_t key, const ps_str* values, const size_t* n_values,
                                ps_tokenizer_fn tokenizer, void* user) {
  return guard([&]() -> ps_status {
    if (!idx || !n_values) return fail(PS_EINVAL, "null argument");
    idx->idx.add_document(key, values, n_values, tokenizer, user);
    return PS_OK;
  });
}

ps_status ps_index_add_documents_flat(ps_index* idx, size_t n_docs, const uint64_t* keys, const char* text,
                                      const uint64_t* offsets) {
  return guard([&]() -> ps_status {
    if (!idx || (n_docs && (!keys || !text || !offsets))) return fail(PS_EINVAL, "null argument");
    const size_t F = idx->idx.fields_len();
    std::vector<ps_str> vals(F);
    std::vector<size_t> ones(F, 1);
    for (size_t d = 0; d < n_docs; ++d) {
      for (size_t f = 0; f < F; ++f) {
        vals[f].ptr = text + offsets[d * F + f];
        vals[f].len = (size_t)(offsets[d * F + f + 1] - offsets[d * F + f]);
      }
      idx->idx.add_document(keys[d], vals.data(), ones.data(), nullptr, nullptr);
    }
    return PS_OK;
  });
}

ps_status ps_index_add_documents_flat_gpu(ps_index* idx, size_t n_docs, const uint64_t* keys, const char* text,
                                          const uint64_t* offsets, int device, int* used_gpu) {
  if (used_gpu) *used_gpu = 0;
  ps_status st = guard([&]() -> ps_status {
    if (!idx || (n_docs && (!keys || !text || !offsets))) return fail(PS_EINVAL, "null argument");
    if (!idx->idx.pristine()) return fail(PS_EINVAL, "GPU bulk indexing fills an empty index (use ps_index_add_documents_flat to append)");
    ps::GroupedCorpus g;
    if (!ps::gpu_group_corpus(device, (uint32_t)idx->idx.fields_len(), n_docs, text, offsets, g)) return PS_EUNSUPPORTED;  // hash collision
    idx->idx.bulk_load(g, n_docs, keys, text);
    if (used_gpu) *used_gpu = 1;
    return PS_OK;
  });
  if (st == PS_EUNSUPPORTED && idx && idx->idx.pristine())  // a detected 64-bit hash collision: same result, on the host
    return ps_index_add_documents_flat(idx, n_docs, keys, text, offsets);
  return st;
}

ps_status ps_index_remove_document(ps_index* idx, uint64_t key) {
  return guard([&]() -> ps_status {
    if (!idx) return fail(PS_EINVAL, "null index");
    idx->idx.remove_document(key);
    return PS_OK;
  });
}

ps_status ps_index_vacuum(ps_index* idx) {
  return guard([&]() -> ps_status {
    if (!idx) return fail(PS_EINVAL, "null index");
    idx->idx.vacuum();
    return PS_OK;
  });
}

size_t ps_index_fields_len(const ps_index* idx) { return idx ? idx->idx.fields_len() : 0; }
size_t ps_index_docs_len(const ps_index* idx) { return idx ? idx->idx.docs_len() : 0; }

ps_status ps_index_field_details(const ps_index* idx, size_t field, uint64_t* sum, double* avg) {
  if (!idx || field >= idx->idx.fields_len()) return fail(PS_EINVAL, "bad field index");
  if (sum) *sum = idx->idx.field(field).sum;
  if (avg) *avg = idx->idx.field(field).avg;
  return PS_OK;
}

int ps_index_doc_field_length(const ps_index* idx, uint64_t key, uint64_t* out) {
  if (!idx) return 0;
  const ps::DocDetails* d = idx->idx.doc(key);
  if (!d) return 0;
  if (out)
    for (size_t i = 0; i < d->field_length.size(); ++i) out[i] = d->field_length[i];
  return 1;
}

size_t ps_index_count_nodes(const ps_index* idx) { return idx ? idx->idx.count_nodes() : 0; }
size_t ps_index_live_pointers(const ps_index* idx) { return idx ? idx->idx.live_pointers() : 0; }

long ps_index_children(const ps_index* idx, const char* term, size_t len, uint32_t* out, size_t cap) {
  if (!idx) return -1;
  int32_t n = idx->idx.find_node(std::string_view(term ? term : "", len));
  if (n == ps::NIL) return -1;
  std::vector<uint32_t> c = idx->idx.children(n);
  for (size_t i = 0; i < c.size() && i < cap; ++i) out[i] = c[i];
  return (long)c.size();
}

long ps_index_count_documents(const ps_index* idx, const char* term, size_t len) {
  if (!idx) return -1;
  int32_t n = idx->idx.find_node(std::string_view(term ? term : "", len));
  if (n == ps::NIL) return -1;
  return idx->idx.count_documents(n);
}

size_t ps_index_expand_term(const ps_index* idx, const char* term, size_t len, char* buf, size_t cap,
                            size_t* bytes_needed) {
  if (!idx) return 0;
  std::vector<std::string> r = idx->idx.expand_term(std::string_view(term ? term : "", len));
  size_t off = 0;
  for (const std::string& s : r) {
    if (buf && off + s.size() + 1 <= cap) {
      memcpy(buf + off, s.data(), s.size());
      buf[off + s.size()] = 0;
    }
    off += s.size() + 1;
  }
  if (bytes_needed) *bytes_needed = off;
  return r.size();
}

ps_status ps_index_snapshot(const ps_index* idx, int device, uint32_t tile_docs, ps_snapshot** out) {
  return ps_index_snapshot_ex(idx, device, tile_docs, 0, out);
}

ps_status ps_snapshot_update(ps_snapshot* snap, const ps_index* idx, ps_update_stats* out) {
  return guard([&]() -> ps_status {
    if (!snap || !idx) return fail(PS_EINVAL, "null argument");
    if (snap->snap.use_count() > 1) return fail(PS_EUNSUPPORTED, "replicas of ps_index_snapshot_multi share their host copy: re-create them");
    ps_update_stats st;
    memset(&st, 0, sizeof(st));
    const double t0 = wall_ms();
    if (snap->snap->src_uid == idx->idx.uid() && snap->snap->src_epoch == idx->idx.epoch()) {
      if (out) *out = st;
      return PS_OK;
    }
    ps::DeltaRanges r;
    if (snap->snap->apply_delta(idx->idx, r)) {
      const double t1 = wall_ms();
      std::vector<uint64_t> removed;
      uint64_t bytes = 0;
      if (snap->engine) snap->engine->apply_delta(r, removed, &bytes);
      else removed = snap->snap->count_removed_df_host();
      snap->snap->set_removed_df(removed);
      st.mode = 1;
      st.trie_refrozen = r.trie_refrozen ? 1 : 0;
      st.docs_added = r.docs_added;
      st.docs_removed = r.docs_removed;
      st.postings_uploaded = r.plane_end - r.plane_begin;
      st.bytes_uploaded = bytes;
      st.host_ms = t1 - t0;
      st.device_ms = wall_ms() - t1;
    } else {
      // not expressible as a delta: flatten again (same tile size and headroom) and swap
      std::shared_ptr<ps::Snapshot> fresh(new ps::Snapshot(idx->idx, snap->tile_docs, snap->headroom_pct));
      const double t1 = wall_ms();
      // The new engine is built while the old one still serves the handle: if that fails for lack of HBM the
      // old planes are released and it is tried once more; a second failure leaves a consistent host-only
      // handle (the fresh flatten, no engine) - never the old engine over a host copy a half-applied
      // delta already changed.
      std::unique_ptr<ps::Engine> eng;
      if (snap->device >= 0) {
        try {
          eng.reset(new ps::Engine(*fresh, snap->device));
        } catch (const ps::DeviceOom&) {
          // (only for lack of HBM: any other failure - wrong architecture, invalid device - propagates and leaves the
          // old engine and snapshot in place)
          snap->engine.reset();
          snap->snap = fresh;
          try {
            eng.reset(new ps::Engine(*fresh, snap->device));
          } catch (const std::exception& e) {
            throw std::runtime_error(std::string(e.what()) + " (ps_snapshot_update: the re-flattened snapshot does not fit the device even "
                                     "with the old planes released; the handle now holds the fresh host copy WITHOUT an engine - queries "
                                     "return PS_ENODEVICE until ps_snapshot_update succeeds)");
          }
        }
      }
      snap->engine = std::move(eng);  // (the old engine still reads the old host arrays while it is torn down)
      snap->snap = fresh;
      st.mode = 2;
      st.postings_uploaded = fresh->n_postings;
      st.bytes_uploaded = snap->engine ? snap->engine->device_bytes() : 0;
      st.host_ms = t1 - t0;
      st.device_ms = wall_ms() - t1;
    }
    st.delta_layers = snap->snap->n_delta_layers;
    st.delta_postings = snap->snap->n_delta_postings;
    if (out) *out = st;
    return PS_OK;
  });
}

ps_status ps_index_snapshot_ex(const ps_index* idx, int device, uint32_t tile_docs, uint32_t headroom_pct,
                               ps_snapshot** out) {
  return guard([&]() -> ps_status {
    if (!idx || !out) return fail(PS_EINVAL, "null argument");
    std::unique_ptr<ps_snapshot> s(new ps_snapshot());
    s->snap.reset(new ps::Snapshot(idx->idx, tile_docs, headroom_pct));
    s->tile_docs = tile_docs;
    s->headroom_pct = headroom_pct;
    s->device = device;
    if (device >= 0) s->engine.reset(new ps::Engine(*s->snap, device));
    *out = s.release();
    return PS_OK;
  });
}

ps_status ps_index_snapshot_multi(const ps_index* idx, const int* devices, size_t n_devices, uint32_t tile_docs,
                                  ps_snapshot** out) {
  return guard([&]() -> ps_status {
    if (!idx || !out || (n_devices && !devices)) return fail(PS_EINVAL, "null argument");
    std::shared_ptr<ps::Snapshot> host(new ps::Snapshot(idx->idx, tile_docs));  // flattened once
    std::vector<std::unique_ptr<ps_snapshot>> reps;
    for (size_t i = 0; i < n_devices; ++i) {
      std::unique_ptr<ps_snapshot> s(new ps_snapshot());
      s->snap = host;
      s->tile_docs = tile_docs;
      s->device = devices[i];
      if (devices[i] >= 0) s->engine.reset(new ps::Engine(*host, devices[i]));
      reps.push_back(std::move(s));
    }
    for (size_t i = 0; i < n_devices; ++i) out[i] = reps[i].release();
    return PS_OK;
  });
}

void ps_snapshot_free(ps_snapshot* snap) { delete snap; }

ps_status ps_snapshot_save(const ps_snapshot* snap, const char* path) {
  return guard([&]() -> ps_status {
    if (!snap || !path) return fail(PS_EINVAL, "null argument");
    snap->snap->save(path);
    return PS_OK;
  });
}

ps_status ps_snapshot_load(const char* path, int device, ps_snapshot** out) {
  return guard([&]() -> ps_status {
    if (!path || !out) return fail(PS_EINVAL, "null argument");
    std::unique_ptr<ps_snapshot> s(new ps_snapshot());
    s->snap.reset(new ps::Snapshot(std::string(path)));
    s->tile_docs = s->snap->T;  // (a later ps_snapshot_update re-flattens with the same tile size)
    s->device = device;
    if (device >= 0) s->engine.reset(new ps::Engine(*s->snap, device));
    *out = s.release();
    return PS_OK;
  });
}

ps_status ps_snapshot_get_info(const ps_snapshot* snap, ps_snapshot_info* out) {
  if (!snap || !out) return fail(PS_EINVAL, "null argument");
  const ps::Snapshot& s = *snap->snap;
  memset(out, 0, sizeof(*out));
  out->fields_num = s.F;
  out->tile_docs = s.T;
  out->n_docs = s.n_docs;
  out->n_terms = s.n_live_terms;
  out->n_postings = s.n_postings;
  out->n_pointers = s.n_pointers;
  out->n_table_entries = s.table.size();
  out->device_bytes = snap->engine ? snap->engine->device_bytes() : 0;
  out->device = snap->device;
  out->max_layers = (int32_t)s.max_layers;
  out->n_ids = s.n_ids;
  out->tiles_cap = s.tiles_cap;
  out->delta_layers = s.n_delta_layers;
  out->delta_postings = s.n_delta_postings;
  return PS_OK;
}

static ps_status run_batch(ps_snapshot* snap, const ps_scorer_desc* scorer, const ps_str* queries, size_t n,
                           const double* fields_boost, size_t n_boost, ps_tokenizer_fn tokenizer, void* user,
                           size_t top_k, ps_result** out, size_t** out_offsets) {
  return guard([&]() -> ps_status {
    ps_status st = check_query_args(snap, scorer, fields_boost, n_boost);
    if (st != PS_OK) return st;
    if (!out || !out_offsets || (n && !queries)) return fail(PS_EINVAL, "null argument");
    const double t0 = wall_ms();
    ps::Plan plan;
    plan_batch(snap, *scorer, views_of(queries, n), tokenizer, user, plan);
    const double t1 = wall_ms();
    ps::ResultBuf res;  // (a malloc'd block, handed to the caller as it is)
    std::vector<size_t> offs;
    ps_batch_stats stats;
    snap->engine->run_host(*scorer, fields_boost, plan, top_k, res, offs, stats);
    stats.plan_ms = t1 - t0;
    stats.total_ms = wall_ms() - t0;
    set_stats(snap, stats);
    size_t* o = (size_t*)malloc(sizeof(size_t) * (n + 1));
    if (!o) return fail(PS_ENOMEM, "out of memory");
    memcpy(o, offs.data(), sizeof(size_t) * (n + 1));
    *out = res.release();
    *out_offsets = o;
    return PS_OK;
  });
}

ps_status ps_snapshot_query_batch(ps_snapshot* snap, const ps_scorer_desc* scorer, const ps_str* queries,
                                  size_t n_queries, const double* fields_boost, size_t n_boost,
                                  ps_tokenizer_fn tokenizer, void* user, size_t top_k, ps_result** out,
                                  size_t** out_offsets) {
  return run_batch(snap, scorer, queries, n_queries, fields_boost, n_boost, tokenizer, user, top_k, out, out_offsets);
}

ps_status ps_snapshot_query(ps_snapshot* snap, const ps_scorer_desc* scorer, const char* query, size_t query_len,
                            const double* fields_boost, size_t n_boost, ps_tokenizer_fn tokenizer, void* user,
                            size_t top_k, ps_result** out, size_t* out_len) {
  if (!out || !out_len) return fail(PS_EINVAL, "null argument");
  ps_str q{query, query_len};
  size_t* offs = nullptr;
  ps_status st = run_batch(snap, scorer, &q, 1, fields_boost, n_boost, tokenizer, user, top_k, out, &offs);
  if (st != PS_OK) return st;
  *out_len = offs[1];
  free(offs);
  return PS_OK;
}

ps_status ps_index_query(ps_index* idx, const ps_scorer_desc* scorer, const char* query, size_t query_len,
                         const double* fields_boost, size_t n_boost, ps_tokenizer_fn tokenizer, void* user,
                         size_t top_k, ps_result** out, size_t* out_len) {
  if (!idx) return fail(PS_EINVAL, "null index");
  if (scorer && scorer->kind == PS_SCORER_HOST_CALLBACKS) {
    // a custom ScoreCalculator: the reference's driver loop on the host, the callbacks do the scoring
    return guard([&]() -> ps_status {
      if (!out || !out_len) return fail(PS_EINVAL, "null argument");
      if (!scorer->callbacks) return fail(PS_EINVAL, "PS_SCORER_HOST_CALLBACKS without callbacks");
      if (n_boost < idx->idx.fields_len() || (idx->idx.fields_len() && !fields_boost))
        return fail(PS_EINVAL, "fields_boost shorter than fields_num");
      std::vector<ps_result> res;
      idx->idx.query_callbacks(*scorer->callbacks, std::string_view(query ? query : "", query_len), tokenizer, user,
                               fields_boost, n_boost, idx, res);
      if (top_k && res.size() > top_k) res.resize(top_k);
      ps_result* r = (ps_result*)malloc(sizeof(ps_result) * (res.size() ? res.size() : 1));
      if (!r) return fail(PS_ENOMEM, "out of memory");
      if (!res.empty()) memcpy(r, res.data(), sizeof(ps_result) * res.size());
      *out = r;
      *out_len = res.size();
      return PS_OK;
    });
  }
  if (!idx->cached) {
    // the lazily kept snapshot tracks the live index: room for 25 % more documents before a re-flatten
    ps_status st = ps_index_snapshot_ex(idx, 0, 0, 25, &idx->cached);
    if (st != PS_OK) return st;
  } else if (idx->cached->snap->src_epoch != idx->idx.epoch() || idx->cached->snap->src_uid != idx->idx.uid()) {
    ps_status st = ps_snapshot_update(idx->cached, idx, nullptr);  // delta if expressible, else full
    if (st != PS_OK) return st;
  }
  return ps_snapshot_query(idx->cached, scorer, query, query_len, fields_boost, n_boost, tokenizer, user, top_k, out,
                           out_len);
}

ps_status ps_snapshot_query_batch_device(ps_snapshot* snap, const ps_scorer_desc* scorer, const ps_str* queries,
                                         size_t n_queries, const double* fields_boost, size_t n_boost,
                                         ps_tokenizer_fn tokenizer, void* user, size_t top_k, void* d_keys,
                                         void* d_scores, void* d_counts, void* hip_stream) {
  return guard([&]() -> ps_status {
    if (n_queries && !queries) return fail(PS_EINVAL, "null argument");
    return run_device_views(snap, scorer, views_of(queries, n_queries), fields_boost, n_boost, tokenizer, user, top_k,
                            d_keys, d_scores, d_counts, hip_stream);
  });
}

ps_status ps_snapshot_plan_ahead_flat(ps_snapshot* snap, const ps_scorer_desc* scorer, const char* text, const uint64_t* offsets,
                                      size_t n_queries, int* accepted) {
  return guard([&]() -> ps_status {
    if (!snap || !scorer || (n_queries && (!text || !offsets))) return fail(PS_EINVAL, "null argument");
    if (!snap->engine) return fail(PS_ENODEVICE, "host-only snapshot");
    const bool ok = snap->engine->plan_ahead(*scorer, text, offsets, n_queries);
    if (accepted) *accepted = ok ? 1 : 0;
    return PS_OK;
  });
}

ps_status ps_snapshot_query_batch_device_flat(ps_snapshot* snap, const ps_scorer_desc* scorer, const char* text,
                                              const uint64_t* offsets, size_t n_queries, const double* fields_boost,
                                              size_t n_boost, ps_tokenizer_fn tokenizer, void* user, size_t top_k,
                                              void* d_keys, void* d_scores, void* d_counts, void* hip_stream) {
  return ps::run_device_flat(snap, scorer, text, offsets, n_queries, fields_boost, n_boost, tokenizer, user, top_k, d_keys,
                             d_scores, d_counts, hip_stream);
}

ps_status ps_snapshot_query_batch_device_planned_flat(ps_snapshot* snap, const ps_scorer_desc* scorer, const char* text,
                                                      const uint64_t* offsets, size_t n_queries, const double* fields_boost,
                                                      size_t n_boost, size_t top_k, void* d_keys, void* d_scores,
                                                      void* d_counts, void* hip_stream) {
  return guard([&]() -> ps_status {
    ps_status st = check_query_args(snap, scorer, fields_boost, n_boost);
    if (st != PS_OK) return st;
    if (!d_keys || !d_scores || !d_counts || !offsets || (n_queries && !text)) return fail(PS_EINVAL, "null argument");
    const double t0 = wall_ms();
    ps_batch_stats stats;
    if (!snap->engine->run_device_planned(*scorer, fields_boost, text, offsets, n_queries, top_k, d_keys, d_scores, d_counts,
                                          hip_stream, stats)) {
      // (a zero_to_one batch K1dz does not take whole: straight to the host planner - run_device_flat would ask the device
      // planner a second time: another count pass and another host wait before the same answer)
      std::vector<std::string_view> qs(n_queries);
      for (size_t i = 0; i < n_queries; ++i) qs[i] = std::string_view(text + offsets[i], (size_t)(offsets[i + 1] - offsets[i]));
      return run_device_views(snap, scorer, qs, fields_boost, n_boost, nullptr, nullptr, top_k, d_keys, d_scores, d_counts, hip_stream);
    }
    stats.total_ms = wall_ms() - t0;
    set_stats(snap, stats);
    return PS_OK;
  });
}

ps_status ps_snapshot_plan_device(ps_snapshot* snap, const ps_scorer_desc* scorer, const char* text, const uint64_t* offsets,
                                  size_t n_queries, ps_plan_entry** entries, size_t* n_entries, uint32_t** qbeg,
                                  uint32_t** query_terms_len) {
  return guard([&]() -> ps_status {
    if (!snap || !scorer || !offsets || !entries || !n_entries || !qbeg || !query_terms_len) return fail(PS_EINVAL, "null argument");
    if (scorer->kind != PS_SCORER_BM25) return fail(PS_EINVAL, "the device planner handles BM25");
    if (!snap->engine) return fail(PS_ENODEVICE, "host-only snapshot");
    ps::Plan plan;
    snap->engine->plan_device(text, offsets, n_queries, plan);
    ps_plan_entry* e = (ps_plan_entry*)malloc(sizeof(ps_plan_entry) * (plan.entries.size() ? plan.entries.size() : 1));
    uint32_t* qb = (uint32_t*)malloc(4 * (n_queries + 1));
    uint32_t* ql = (uint32_t*)malloc(4 * (n_queries ? n_queries : 1));
    if (!e || !qb || !ql) { free(e); free(qb); free(ql); return fail(PS_ENOMEM, "out of memory"); }
    if (!plan.entries.empty()) memcpy(e, plan.entries.data(), sizeof(ps_plan_entry) * plan.entries.size());
    memcpy(qb, plan.qbeg.data(), 4 * (n_queries + 1));
    if (n_queries) memcpy(ql, plan.qterms_len.data(), 4 * n_queries);
    *entries = e; *n_entries = plan.entries.size(); *qbeg = qb; *query_terms_len = ql;
    return PS_OK;
  });
}

ps_status ps_snapshot_last_stats(const ps_snapshot* snap, ps_batch_stats* out) {
  if (!snap || !out) return fail(PS_EINVAL, "null argument");
  ps_snapshot* s = const_cast<ps_snapshot*>(snap);
  std::lock_guard<std::mutex> l(s->stats_mu);
  *out = s->last;
  return PS_OK;
}

ps_status ps_snapshot_kernel_times(ps_snapshot* snap, double* total_ms, uint64_t* launches, int reset) {
  return guard([&]() -> ps_status {
    if (!snap) return fail(PS_EINVAL, "null snapshot");
    if (!snap->engine) return fail(PS_ENODEVICE, "host-only snapshot");
    ps_kernel_times kt;
    snap->engine->kernel_times(kt, reset != 0);
    if (total_ms) *total_ms = kt.score_ms;
    if (launches) *launches = kt.launches;
    return PS_OK;
  });
}

ps_status ps_snapshot_kernel_breakdown(ps_snapshot* snap, ps_kernel_times* out, int reset) {
  return guard([&]() -> ps_status {
    if (!snap || !out) return fail(PS_EINVAL, "null argument");
    if (!snap->engine) return fail(PS_ENODEVICE, "host-only snapshot");
    snap->engine->kernel_times(*out, reset != 0);
    return PS_OK;
  });
}

ps_status ps_snapshot_work_counters(ps_snapshot* snap, ps_work_counters* out, int reset) {
  return guard([&]() -> ps_status {
    if (!snap || !out) return fail(PS_EINVAL, "null argument");
    if (!snap->engine) return fail(PS_ENODEVICE, "host-only snapshot");
    snap->engine->work_counters(*out, reset != 0);
    return PS_OK;
  });
}

ps_status ps_snapshot_plan(const ps_snapshot* snap, const ps_scorer_desc* scorer, const char* query, size_t query_len,
                           ps_tokenizer_fn tokenizer, void* user, ps_plan_entry** out, size_t* out_len,
                           size_t* query_terms_len) {
  return guard([&]() -> ps_status {
    if (!snap || !scorer || !out || !out_len) return fail(PS_EINVAL, "null argument");
    ps::Plan plan;
    plan.qbeg.push_back(0);
    snap->snap->plan_query(*scorer, std::string_view(query ? query : "", query_len), tokenizer, user, plan);
    ps_plan_entry* r = (ps_plan_entry*)malloc(sizeof(ps_plan_entry) * (plan.entries.size() ? plan.entries.size() : 1));
    if (!r) return fail(PS_ENOMEM, "out of memory");
    if (!plan.entries.empty()) memcpy(r, plan.entries.data(), sizeof(ps_plan_entry) * plan.entries.size());
    *out = r;
    *out_len = plan.entries.size();
    if (query_terms_len) *query_terms_len = plan.qterms_len[0];
    return PS_OK;
  });
}

ps_status ps_snapshot_host_csr(const ps_snapshot* snap, ps_host_csr* out) {
  if (!snap || !out) return fail(PS_EINVAL, "null argument");
  const ps::Snapshot& s = *snap->snap;
  out->doc = s.doc.data();
  out->tf = s.tf.data();
  out->fl = s.fl.data();
  out->table = s.table.data();
  out->keys = s.keys.data();
  out->avg = s.avg.data();
  out->plane_stride = s.P;
  out->alive = s.alive.data();
  return PS_OK;
}

}  // extern "C"
