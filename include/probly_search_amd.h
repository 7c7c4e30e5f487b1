/*
 * probly_search_amd.h — C ABI of the MI355X-native BM25 / zero-to-one query-scoring engine.
 *
 * This is the drop-in boundary for probly-search's `Index::query` -> `ScoreCalculator` hot path
 * (reference: quantleaf/probly-search 2.0.1; citations are `file:line` relative to its root).
 * The reference is a pure-Rust library with no FFI of its own, so the entry points below are
 * what a `probly-search-amd-sys` crate would bind (see INTEGRATION.md for the Rust side):
 * plain pointers and sizes, no C++ or torch types, never unwinds.
 *
 * Ownership: every handle is created and freed by the library.  Result arrays returned through
 * `ps_result**` / `size_t**` are allocated by the library and released with ps_free() - and only with it: large
 * result blocks (PS_RESULT_PINNED_MIN_KB, default 4 MiB) are pinned host memory the device wrote into directly;
 * ps_free() hands them back to a pool for the next batch (PS_RESULT_POOL_MB of idle blocks are kept), libc free()
 * on them is undefined.
 * Threading: a `ps_index` needs external exclusion for mutation (it is `&mut self` in the
 * reference); a `ps_snapshot` is immutable and its query entry points are thread-safe
 * (`query(&self)`, src/query.rs:21-27).
 * Errors: every fallible call returns a ps_status; ps_last_error() gives the calling thread's
 * message.  Where the reference would panic (short `fields_boost`, bm25.rs:85) the call returns
 * PS_EINVAL instead.  There is NO CPU scoring fallback: without a HIP device the query entry
 * points return PS_ENODEVICE.
 */
#ifndef PROBLY_SEARCH_AMD_H
#define PROBLY_SEARCH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ps_status {
  PS_OK = 0,
  PS_EINVAL = 1,       /* bad argument (incl. n_boost < fields_num) */
  PS_ENOMEM = 2,
  PS_EHIP = 3,         /* a HIP runtime call failed */
  PS_EUNSUPPORTED = 4, /* input exceeds a documented engine limit */
  PS_ENODEVICE = 5,    /* no usable HIP device / snapshot is host-only */
  PS_ERCCL = 6         /* an RCCL call failed (ps_comm_*, ps_snapshot_query_batch_allgather_flat) */
} ps_status;

typedef struct ps_index ps_index;       /* mutable host index  == Index<u64>  (src/index.rs:19-33) */
typedef struct ps_snapshot ps_snapshot; /* immutable flattened CSR postings resident in HBM       */
typedef struct ps_comm ps_comm;         /* RCCL communicator of the ranks that share a sharded batch */

/* &str */
typedef struct ps_str {
  const char* ptr;
  size_t len;
} ps_str;

/* QueryResult<u64> (src/query.rs:10-15) */
typedef struct ps_result {
  uint64_t key;
  double score;
} ps_result;

/* Tokenizer = fn(&str) -> Vec<Cow<str>> (src/lib.rs:14).  Writes up to `cap` tokens into
 * tok_ptr/tok_len and returns the total token count.  Tokens must stay valid until the next call
 * on the same thread.  NULL selects the whitespace tokenizer every reference test uses,
 * `s.split(' ')` (src/lib.rs:42-44): empty tokens are produced, skipped, but still counted in
 * query_terms_len (src/query.rs:32-35). */
typedef size_t (*ps_tokenizer_fn)(const char* s, size_t len, const char** tok_ptr, size_t* tok_len, size_t cap,
                                  void* user);

/* ---- ScoreCalculator plugin surface (src/score/calculator.rs:9-70) ---------------------------
 * The trait's arguments are host references into the index (HashMap, arena indices), so a custom
 * implementation cannot run on the device.  The two calculators the reference ships
 * (src/score/default/) are built in and run on the GPU; any other implementation is handed over as
 * three C callbacks (PS_SCORER_HOST_CALLBACKS) and runs a host walk inside the library with the
 * reference's exact call sequence (src/query.rs:29-105): per expansion `before_each`, per
 * non-removed DocumentPointer in list order (newest first, one call per occurrence) `score` +
 * max_score_merger, once `finalize`, then the stable sort by score desc. */
/* TermData (src/score/calculator.rs:9-19) */
typedef struct ps_term_data {
  size_t query_term_index;
  ps_str query_term;
  ps_str query_term_expanded;
  size_t query_terms_len;     /* counts every token the tokenizer returned, empty ones included (src/query.rs:32) */
} ps_term_data;
/* FieldDetails (src/index.rs:391-396) */
typedef struct ps_field_details {
  uint64_t sum;
  double avg;
} ps_field_details;
/* FieldData (src/score/calculator.rs:21-26) */
typedef struct ps_field_data {
  const double* fields_boost;
  size_t n_boost;
  const ps_field_details* fields;
  size_t n_fields;
} ps_field_data;
/* DocumentPointer (src/index.rs:354-361) without the list link */
typedef struct ps_document_pointer {
  uint64_t details_key;
  const uint32_t* term_frequency; /* [fields_num] */
} ps_document_pointer;
/* DocumentDetails (src/index.rs:342-349) */
typedef struct ps_document_details {
  uint64_t key;
  const uint32_t* field_length;   /* [fields_num] */
} ps_document_details;
typedef struct ps_score_callbacks {
  /* before_each(&mut self, &TermData, document_frequency, &documents) -> Option<M>
   * (calculator.rs:43-50).  Return 1 and set *memory for Some(M), 0 for None.  `n_documents` is
   * documents.len(); `idx` may be passed to the ps_index_* read functions for anything else the
   * `documents` map would have answered.  NULL = the trait's default (None). */
  int (*before_each)(void* user, const ps_term_data* term_expansion, size_t document_frequency, size_t n_documents,
                     const ps_index* idx, void** memory);
  /* score(&mut self, Option<&M>, &DocumentPointer, &DocumentDetails, &index_node, &FieldData,
   * &TermData) -> Option<f64> (calculator.rs:58-66).  Return 1 and set *out for Some(score), 0 for
   * None.  `memory` is before_each's M (NULL for None); `index_node` is a unique id of the expanded
   * term's trie node.  Required. */
  int (*score)(void* user, const void* memory, const ps_document_pointer* document_pointer,
               const ps_document_details* document_details, uint64_t index_node, const ps_field_data* field_data,
               const ps_term_data* term_expansion, double* out);
  /* finalize(&mut self, &mut Vec<QueryResult>) (calculator.rs:69): may rewrite scores in place and
   * drop results (return the new length <= n).  NULL = the trait's default (no-op). */
  size_t (*finalize)(void* user, ps_result* results, size_t n);
  /* drop(M) once the expansion's posting walk is over.  NULL = nothing to free. */
  void (*drop_memory)(void* user, void* memory);
  void* user;
} ps_score_callbacks;

enum { PS_SCORER_BM25 = 1, PS_SCORER_ZERO_TO_ONE = 2, PS_SCORER_HOST_CALLBACKS = 3 };
typedef struct ps_scorer_desc {
  int32_t kind;   /* PS_SCORER_*                                                        */
  int32_t _pad;
  double bm25_k1; /* BM25::bm25k1, default 1.2  (src/score/default/bm25.rs:14-26)       */
  double bm25_b;  /* BM25::bm25b,  default 0.75                                         */
  const ps_score_callbacks* callbacks; /* PS_SCORER_HOST_CALLBACKS only (else NULL).  Accepted by
                     ps_index_query (the entry that owns the posting lists in the reference's
                     order); the snapshot entry points return PS_EINVAL for it.             */
} ps_scorer_desc;

const char* ps_last_error(void);
/* Releases a block a query / plan entry point returned (NULL is fine).  Thread-safe. */
void ps_free(void* p);
/* Convenience for bindings that want columns (numpy, Arrow): keys[i] = results[i].key, scores[i] = results[i].score
 * for i < n; a few host threads for large blocks.  Either output may be NULL. */
void ps_results_split(const ps_result* results, size_t n, uint64_t* keys, double* scores);
/* Number of visible HIP devices (0 if none / HIP unusable). */
int ps_device_count(void);
/* The version of THIS header's struct layouts and entry points.  The output structs (ps_kernel_times, ps_work_counters,
 * ps_update_stats, ps_query_stats ...) are written whole by the library - memset + fill, sizeof as the library was built - and grow
 * at the end between versions: a caller compiled against an older header would be written past its struct.  Compare
 * ps_abi_version() with PS_ABI_VERSION once after loading and refuse a mismatch (the Python binding does).
 * 6: ps_kernel_times.score_busy_ms (round 5), ps_comm_all_gather (round 6). */
#define PS_ABI_VERSION 6u
uint32_t ps_abi_version(void);
/* Run-time options by name.  A value set here wins over the environment variable of the same name; engines read their options when
 * they are created and again at the next batch after any ps_set_option.  The list (PS_EINVAL for any other name):
 *   PS_DAAT (1)  PS_DAAT_MIN_BATCH (8)  PS_DAAT_MULTI (1)  PS_DAAT_SMALL (1)  PS_DAAT_SMALL_NL (1)  PS_DAAT_SPLIT (1)  PS_DAAT_Z (1)
 *   PS_DAAT_Z_SPLIT (1)  PS_DAAT_PRIME (1)  PS_DAAT_CHUNK (4096): which exact-pruning kernels take which batches, threshold priming;
 *   PS_DEVICE_PLAN (1): the device-side planner;  PS_ROW_CACHE_MB (4096): resident dense rows;  PS_DENSE_MIN_USES (4), PS_DENSE_MAX_ROWS (64);
 *   PS_WORK_COUNTERS (1), PS_KERNEL_TIMERS (1): ps_snapshot_work_counters / ps_snapshot_kernel_breakdown instrumentation;
 *   PS_SCORE_ALT (1), PS_DCTX (5), PS_PLAN_AHEAD_DEPTH (3): scoring queues, batch contexts, announced batches;
 *   PS_RESULT_PINNED_MIN_KB (4096), PS_FULL_PARTS_MIN_KB (32768): full-result mode;  PS_PLAN_THREADS, PS_FLATTEN_THREADS: host pools.
 * The engine's experiment knobs are environment variables; this call takes them only under
 * PS_EXPERIMENT_KNOBS=1.  ps_get_option returns 1 and the effective override / environment value, 0 if the option is at its default. */
ps_status ps_set_option(const char* name, uint32_t value);
int ps_get_option(const char* name, uint32_t* value);

/* ------------------------------------------------------------------ index build side -------- */
/* Index::new(fields_num)  (src/index.rs:37-39) */
ps_status ps_index_new(size_t fields_num, ps_index** out);
/* Index::new_with_capacity (src/index.rs:42-60); capacities are reservation hints only. */
ps_status ps_index_new_with_capacity(size_t fields_num, size_t expected_index_size,
                                     size_t expected_documents_count, ps_index** out);
void ps_index_free(ps_index* idx);

/* Index::add_document(field_accessors, tokenizer, key, doc) (src/index.rs:77-158).  The accessors
 * have already been applied by the binding: `values` is the concatenation, field by field, of the
 * strings accessor i returned, n_values[i] of them (multi-valued fields keep the reference's
 * "sum accumulates, field_length = last value" rule, src/index.rs:112-114). */
ps_status ps_index_add_document(ps_index* idx, uint64_t key, const ps_str* values, const size_t* n_values,
                                ps_tokenizer_fn tokenizer, void* user);
/* Bulk form for single-valued fields and the default tokenizer: value (d, f) is
 * text[offsets[d*F+f] .. offsets[d*F+f+1]).  Equivalent to n_docs add_document calls in order. */
ps_status ps_index_add_documents_flat(ps_index* idx, size_t n_docs, const uint64_t* keys, const char* text,
                                      const uint64_t* offsets);
/* GPU bulk indexing (the reference's own benchmark is exactly this loop,
 * benches/test_benchmark.rs:37-63): the same result as ps_index_add_documents_flat on an EMPTY index
 * with distinct keys, but the per-token work of add_document - tokenise, find the term, count its
 * frequency per field - runs on `device` as tokenise -> hash -> stable radix sort by term ->
 * segmented reduce; the host only interns the distinct terms (first-occurrence order, so the trie
 * and its newest-first child lists come out identical) and copies the grouped postings into place.
 * Falls back to the host indexer by itself if two terms collide in the 64-bit hash (checked byte by
 * byte on the device).  *used_gpu (may be NULL) says which path ran. */
ps_status ps_index_add_documents_flat_gpu(ps_index* idx, size_t n_docs, const uint64_t* keys, const char* text,
                                          const uint64_t* offsets, int device, int* used_gpu);
/* Index::remove_document (src/index.rs:161-191) — lazy delete, fixes field sums/averages. */
ps_status ps_index_remove_document(ps_index* idx, uint64_t key);
/* Index::vacuum (src/index.rs:194-241) — unlinks removed postings, prunes empty trie subtrees. */
ps_status ps_index_vacuum(ps_index* idx);

/* ---- Key table: document keys that are not u64 -------------------------------------------------
 * The reference's index is generic over its key, `Index<T: Eq + Hash + Copy + Debug>` (src/index.rs:19-33), and
 * hands `T` back in `QueryResult<T>` (src/query.rs:10-17); this ABI carries uint64_t.  A binding for another `T`
 * (uuid, (shard, row) pair, short string) gives the key's bytes to a key table and uses the dense id it gets back as
 * the ABI key; after a query it turns the result ids back into key bytes.  Ids count up from 0 in first-seen order
 * and are never reused: a removed and re-added key meets the index under its old id, as `remove_document(key)` /
 * `add_document(.., key, ..)` require (src/index.rs:77-83,161-191).  Equal bytes == equal keys (the binding's
 * serialisation must agree with `T: Eq`).  Ties in a result come out id-ascending == first-seen order.
 * Threading: intern needs external exclusion (`&mut self` of add_document); find / key / resolve may run
 * concurrently with each other.  Pointers returned by key / resolve point into the table and stay valid until the
 * next intern.  Host code only. */
typedef struct ps_keytable ps_keytable;
ps_status ps_keytable_new(ps_keytable** out);
void ps_keytable_free(ps_keytable* kt);
size_t ps_keytable_len(const ps_keytable* kt);
/* Id of `key` (len bytes; len 0 is a valid key), inserted if new.  *inserted (may be NULL) = 1 if it was new. */
ps_status ps_keytable_intern(ps_keytable* kt, const void* key, size_t len, uint64_t* id, int* inserted);
/* Bulk form beside ps_index_add_documents_flat: key i is bytes[offsets[i] .. offsets[i+1]); ids[i] receives its id
 * (duplicates inside the batch get one id, as n_keys ps_keytable_intern calls in order would give). */
ps_status ps_keytable_intern_flat(ps_keytable* kt, size_t n_keys, const void* bytes, const uint64_t* offsets,
                                  uint64_t* ids);
/* 1 and *id (may be NULL) if `key` has an id, 0 otherwise (remove_document of a key never added is a no-op in the
 * reference, src/index.rs:161-164: the binding skips the call). */
int ps_keytable_find(const ps_keytable* kt, const void* key, size_t len, uint64_t* id);
/* Key bytes of one id / of results[i].key for i < n (keys[i] receives them).  PS_EINVAL for an id this table never
 * handed out. */
ps_status ps_keytable_key(const ps_keytable* kt, uint64_t id, ps_str* out);
ps_status ps_keytable_resolve(const ps_keytable* kt, const ps_result* results, size_t n, ps_str* keys);

/* Read-side introspection (the pub(crate) state the reference's unit tests look at). */
size_t ps_index_fields_len(const ps_index* idx);
size_t ps_index_docs_len(const ps_index* idx);                                   /* docs.len()          */
ps_status ps_index_field_details(const ps_index* idx, size_t field, uint64_t* sum, double* avg); /* FieldDetails */
int ps_index_doc_field_length(const ps_index* idx, uint64_t key, uint64_t* out); /* 1 if present        */
size_t ps_index_count_nodes(const ps_index* idx);          /* trie nodes reachable from root, root incl. */
size_t ps_index_live_pointers(const ps_index* idx);        /* == live DocumentPointer count              */
/* children chars (list order, newest first) of the node `term` leads to; -1 if no such path */
long ps_index_children(const ps_index* idx, const char* term, size_t len, uint32_t* out, size_t cap);
/* Index::count_documents of the node `term` leads to (src/index.rs:282-297); -1 if no such path */
long ps_index_count_documents(const ps_index* idx, const char* term, size_t len);
/* Index::expand_term (src/query.rs:109-147): NUL-separated terms into buf, returns the count. */
size_t ps_index_expand_term(const ps_index* idx, const char* term, size_t len, char* buf, size_t cap,
                            size_t* bytes_needed);

/* ------------------------------------------------------------------ snapshot ---------------- */
/* Flatten the trie/posting lists into CSR planes (doc id u32, per-field tf u32, per-field
 * field-length u32) plus per-list tile-offset tables, and upload them to `device`.
 * device = -1 builds a host-only snapshot (no HIP call is made; queries return PS_ENODEVICE);
 * it exists so the flattener and planner can be inspected on machines without a GPU.
 * tile_docs = documents per LDS accumulator tile (power of two, 256..4096; 0 = default 1024). */
ps_status ps_index_snapshot(const ps_index* idx, int device, uint32_t tile_docs, ps_snapshot** out);
/* The same with room to grow: headroom_pct > 0 reserves that share of extra documents, postings and
 * table entries, so that ps_snapshot_update can append documents in place (0 = exact fit: removals
 * can still be applied as a delta, additions re-flatten). */
ps_status ps_index_snapshot_ex(const ps_index* idx, int device, uint32_t tile_docs, uint32_t headroom_pct,
                               ps_snapshot** out);
/* Incremental re-flatten: brings `snap` - made from `idx` by ps_index_snapshot* - up to
 * the index's current state.  Documents removed since (Index::remove_document, src/index.rs:161-191)
 * get their alive bit cleared and df is re-counted on the device; documents added since with keys
 * above every key the snapshot holds are appended as delta lists of their terms (the frozen trie is
 * rebuilt when they brought new terms).  Host work and uploads are O(changes), not O(postings).  What
 * cannot be expressed that way (vacuum, re-added or out-of-order keys, headroom exhausted) is handled
 * by a full re-flatten inside the same call.  Not to be called while queries run on `snap`; replicas
 * made by ps_index_snapshot_multi share their host copy and are refused (PS_EUNSUPPORTED). */
typedef struct ps_update_stats {
  int32_t mode;              /* 0 = already current, 1 = delta applied, 2 = full re-flatten */
  int32_t trie_refrozen;     /* delta: the added documents brought new terms                 */
  uint64_t docs_added, docs_removed;
  uint64_t postings_uploaded; /* delta: postings appended to the planes (mode 2: all of them) */
  uint64_t bytes_uploaded;
  uint64_t delta_layers, delta_postings; /* accumulated since the last full flatten          */
  double host_ms, device_ms;
} ps_update_stats;
ps_status ps_snapshot_update(ps_snapshot* snap, const ps_index* idx, ps_update_stats* out);
/* Multi-device form (SURVEY 8b "device_mask"): flatten ONCE, upload the same planes to every
 * device of `devices[0..n_devices)`; out[i] is the replica on devices[i].  The replicas share the
 * host copy (planner, frozen trie); each is freed with ps_snapshot_free. */
ps_status ps_index_snapshot_multi(const ps_index* idx, const int* devices, size_t n_devices, uint32_t tile_docs,
                                  ps_snapshot** out);
void ps_snapshot_free(ps_snapshot* snap);
/* On-disk form of a snapshot (the reference has no persistence): a versioned dump of
 * the flattened arrays.  ps_snapshot_load needs no ps_index; device = -1 loads host-only. */
ps_status ps_snapshot_save(const ps_snapshot* snap, const char* path);
ps_status ps_snapshot_load(const char* path, int device, ps_snapshot** out);

typedef struct ps_snapshot_info {
  uint32_t fields_num;
  uint32_t tile_docs;
  uint64_t n_docs;         /* documents.len() as BM25 sees it (src/score/default/bm25.rs:41)   */
  uint64_t n_terms;        /* indexed terms with >=1 live posting                              */
  uint64_t n_postings;     /* unique (term, doc[, version]) postings in the CSR planes         */
  uint64_t n_pointers;     /* sum of df_raw == live DocumentPointer count                      */
  uint64_t n_table_entries;
  uint64_t device_bytes;
  int32_t device;
  int32_t max_layers;      /* >1 only if some key was re-added without removal                 */
  uint64_t n_ids;          /* doc id space: n_docs + documents removed by delta updates         */
  uint32_t tiles_cap;      /* tiles the tables / bitmaps are laid out for (headroom)            */
  uint32_t delta_layers;   /* delta lists appended by ps_snapshot_update since the last flatten */
  uint64_t delta_postings;
} ps_snapshot_info;
ps_status ps_snapshot_get_info(const ps_snapshot* snap, ps_snapshot_info* out);

/* ------------------------------------------------------------------ query ------------------- */
/* Index::query(query, score_calculator, tokenizer, fields_boost) -> Vec<QueryResult>
 * (src/query.rs:21-106) on the GPU.  Results are in the canonical order of
 * test_util::test_score, score desc then key asc (src/lib.rs:54-58; the reference's own tie
 * order is hashbrown iteration order, i.e. unspecified).  top_k = 0 returns every match like the
 * reference; top_k > 0 returns the first top_k of that same ordering. */
ps_status ps_snapshot_query(ps_snapshot* snap, const ps_scorer_desc* scorer, const char* query, size_t query_len,
                            const double* fields_boost, size_t n_boost, ps_tokenizer_fn tokenizer, void* user,
                            size_t top_k, ps_result** out, size_t* out_len);
/* Convenience: snapshots lazily (re-flattening after any mutation) on device 0 and queries. */
ps_status ps_index_query(ps_index* idx, const ps_scorer_desc* scorer, const char* query, size_t query_len,
                         const double* fields_boost, size_t n_boost, ps_tokenizer_fn tokenizer, void* user,
                         size_t top_k, ps_result** out, size_t* out_len);

/* Batched form: B independent queries, one kernel pass.  out[out_offsets[i] .. out_offsets[i+1])
 * are query i's results (out_offsets has B+1 entries). */
ps_status ps_snapshot_query_batch(ps_snapshot* snap, const ps_scorer_desc* scorer, const ps_str* queries,
                                  size_t n_queries, const double* fields_boost, size_t n_boost,
                                  ps_tokenizer_fn tokenizer, void* user, size_t top_k, ps_result** out,
                                  size_t** out_offsets);

/* Device-resident batched top-k for multi-GPU plumbing: results stay in HBM so the caller can
 * all-gather them over RCCL.  d_keys: u64[B*top_k], d_scores: f64[B*top_k], d_counts: u32[B]
 * (device pointers on the snapshot's device; unused slots are key=~0, score=0).  `hip_stream`
 * is a hipStream_t (NULL = the snapshot's own stream - a NON-BLOCKING stream of the library: the call is then synchronous, and it is
 * NOT ordered with work the caller enqueued on the legacy default stream (handle 0, which is what e.g.
 * torch.cuda.current_stream().cuda_stream returns unless a stream was made current): zero-fills of the output block issued there must
 * have completed - synchronise, or pass a real stream handle); the call returns after enqueueing when a
 * stream is given and all work is ordered on it.  Top-k batches of the pruning kernels (>= 8 queries) run in the snapshot's batch
 * contexts (5 in the rotation): several are in flight at once - the preparation of the next beside the scoring of the
 * current ones, consecutive scoring kernels sharing the chip on two hardware queues - whatever streams they were
 * enqueued on; what a caller can rely on is that a call's output block is complete once its stream has passed the
 * point of the call.  The other batches (streaming kernels, full-result mode) execute one after the other (they share
 * one set of per-batch device buffers; the library orders them with an event).  1 <= top_k <= PS_MAX_DEVICE_TOPK. */
#define PS_MAX_DEVICE_TOPK 64
ps_status ps_snapshot_query_batch_device(ps_snapshot* snap, const ps_scorer_desc* scorer, const ps_str* queries,
                                         size_t n_queries, const double* fields_boost, size_t n_boost,
                                         ps_tokenizer_fn tokenizer, void* user, size_t top_k, void* d_keys,
                                         void* d_scores, void* d_counts, void* hip_stream);

/* Pipelined submission (optional): announce the NEXT flat top-k batch (BM25, or zero_to_one) this snapshot will be asked to score.  The
 * library copies the text and starts the device planner's count pass at once, beside the batches still being scored;
 * the flat query call that follows with byte-identical (text, offsets) finds the plan's totals ready instead of waiting
 * ~0.25 ms for them.  Up to PS_PLAN_AHEAD_DEPTH (3) batches may be announced, and they are asked for in the order they were
 * announced (*accepted = 0 when the queue is full: nothing is dropped then); any other query call in between - another text, a
 * host-planned or full-result batch, a knob changed through ps_set_option - simply drops them (nothing is ever scored from an
 * announced batch that was not asked for).  *accepted = 0 when the batch would not be planned on
 * the device anyway (custom scorers, small batches).  src/query.rs:21-27 is still what the
 * query call mirrors; this is the async half of a server's double-buffered submission loop. */
ps_status ps_snapshot_plan_ahead_flat(ps_snapshot* snap, const ps_scorer_desc* scorer, const char* text, const uint64_t* offsets,
                                      size_t n_queries, int* accepted);
/* Same, with the batch given as one contiguous UTF-8 buffer: query i is
 * text[offsets[i] .. offsets[i+1]) (offsets has n_queries + 1 entries).  This is the zero-copy
 * form a serving loop would use; it avoids per-query pointer marshalling in language bindings. */
ps_status ps_snapshot_query_batch_device_flat(ps_snapshot* snap, const ps_scorer_desc* scorer, const char* text,
                                              const uint64_t* offsets, size_t n_queries, const double* fields_boost,
                                              size_t n_boost, ps_tokenizer_fn tokenizer, void* user, size_t top_k,
                                              void* d_keys, void* d_scores, void* d_counts, void* hip_stream);

/* ---- multi-GPU: replicated corpus, query batch sharded across ranks, RCCL all-gather of top-k ----
 * Queries are independent (each Index::query owns its scores / visited maps, src/query.rs:31,37),
 * so no collective runs while scoring; the only exchange is one ncclAllGather of every rank's
 * top-k block, and only when the batch spans more than one rank.  One process per GPU.
 *   rank 0: ps_comm_get_unique_id(id) -> ship the 128 bytes to the other ranks (any channel)
 *   every rank: ps_comm_init_rank(id, world, rank, device, &comm)
 * RCCL is resolved on first use (a process that never creates a communicator never loads it): the instance the
 * process has already mapped (e.g. the one PyTorch ships, when the host application is a torch program), else
 * librccl.so.1 by name - never a second instance beside the host application's.
 * PS_COMM_TRANSPORT=hostshm selects a debugging transport through POSIX shared memory for ranks
 * that share one GPU (RCCL refuses two ranks on one device); it is not a product path. */
#define PS_COMM_ID_BYTES 128
ps_status ps_comm_get_unique_id(void* id_out /* PS_COMM_ID_BYTES */);
ps_status ps_comm_init_rank(const void* id, int world_size, int rank, int device, ps_comm** out);
void ps_comm_free(ps_comm* comm);
int ps_comm_world_size(const ps_comm* comm);
/* Path of the RCCL library the communicators use (resolves it if that has not happened yet); NULL + ps_last_error()
 * if none could be loaded.  The string lives until the next call. */
const char* ps_comm_rccl_path(void);
int ps_comm_rank(const ps_comm* comm);
/* The communicator's one collective, for the caller's own small exchanges (a barrier, the max of a timing over the ranks): every
 * rank contributes `bytes` from d_send, d_recv receives world_size x bytes in rank order.  Device pointers; ordered on
 * `hip_stream` (NULL: blocking).  ncclAllGather over xGMI (or the debugging transport); one rank: a copy.  With it a host
 * program needs no second communicator (e.g. a torch.distributed "nccl" group) beside the library's. */
ps_status ps_comm_all_gather(ps_comm* comm, const void* d_send, void* d_recv, size_t bytes, void* hip_stream);
/* A rank's top-k block: [n_queries*top_k u64 keys | n_queries*top_k f64 scores | n_queries u32
 * counts, padded to a multiple of 16 bytes]; unused slots key = ~0, score = 0. */
size_t ps_topk_block_bytes(size_t n_queries, size_t top_k);
/* Scores this rank's shard (text/offsets as in ps_snapshot_query_batch_device_flat; every rank
 * passes the same n_queries, padding its shard with empty queries if needed) into d_local_block
 * and all-gathers the blocks into d_all_blocks (world_size blocks, rank order) with ncclAllGather
 * on `hip_stream` (NULL = the snapshot's own stream, synchronous).  comm == NULL or world_size 1:
 * no collective; d_all_blocks receives the local block. */
ps_status ps_snapshot_query_batch_allgather_flat(ps_snapshot* snap, ps_comm* comm, const ps_scorer_desc* scorer,
                                                 const char* text, const uint64_t* offsets, size_t n_queries,
                                                 const double* fields_boost, size_t n_boost, ps_tokenizer_fn tokenizer,
                                                 void* user, size_t top_k, void* d_local_block, void* d_all_blocks,
                                                 void* hip_stream);

/* SURVEY 8f N2 - the same batch call with the PLANNER on the device too: tokenising (split on ' '),
 * term lookup in the frozen trie, prefix expansion in the reference's newest-first DFS order
 * (src/query.rs:109-147, src/index.rs:300-337) and BM25's before_each (src/score/default/bm25.rs:35-58)
 * run in a kernel over the trie kept in HBM; the plan never exists on the host (the host learns the
 * plan's totals to size the launches).  BM25 with the built-in tokenizer, and the zero_to_one batches the
 * pruning kernel takes (>= 8 queries, every query "simple" - zero_to_one.rs:98-113's pool rule in closed
 * form - with at most 4 lists: the planner's count pass classifies them, the record-sort order and the
 * per-record bounds are derived on the device; any other zero_to_one batch is planned on the host, with the
 * same results).  The K1d work descriptors (per-list bounds, ranks, skip thresholds, items, dense-row
 * choice) are built on the device as well, so such a batch runs the same kernels as a host-planned one.
 * The flat batch entry points above take this path by themselves for top-k batches with the built-in
 * tokenizer (knob PS_DEVICE_PLAN, default 1); this entry asks for it explicitly.  Same results as
 * ps_snapshot_query_batch_device_flat. */
ps_status ps_snapshot_query_batch_device_planned_flat(ps_snapshot* snap, const ps_scorer_desc* scorer, const char* text,
                                                      const uint64_t* offsets, size_t n_queries, const double* fields_boost,
                                                      size_t n_boost, size_t top_k, void* d_keys, void* d_scores,
                                                      void* d_counts, void* hip_stream);
/* Timing / roofline accounting of the most recent batch executed on this snapshot. */
typedef struct ps_batch_stats {
  uint64_t n_queries;
  uint64_t n_plan_entries;     /* expanded (term[, version]) lists streamed                      */
  uint64_t postings_visited;   /* sum over plan entries of list length == unique postings visited */
  uint64_t algorithmic_bytes;  /* postings_visited*(4+8F) + emitted results*16   */
  double plan_ms;              /* host: tokenise + expand + before_each                           */
  double h2d_ms, kernel_ms, d2h_ms; /* HIP-event timed on the engine stream                       */
  double score_kernel_ms;      /* the dominant posting-accumulate kernel alone (HIP events)       */
  double total_ms;             /* host wall clock of the whole call                               */
  uint64_t layout_bytes;       /* bytes of the layout the kernels actually streamed this batch:
                                  20-byte postings for lists read as postings, 8 bytes per document
                                  for lists served from dense score rows, plus building those rows
                                  (postings read + row zero-fill + row writes) and emitted results  */
  uint32_t dense_rows;         /* hot (list, idf, boost) combinations the batch read as dense rows */
  uint32_t dense_rows_built;   /* ... of which had to be scored for this batch (the rest were resident
                                  in the snapshot's row slab from earlier batches)                  */
  uint32_t device_planned;     /* 1: the batch was planned by k_plan on the device (plan_ms is then the
                                  host's wait for the planner's totals), 0: by the host planner      */
  uint32_t bounds_recomputed;  /* 1: the batch had to (re)compute the per-list score bounds K1d prunes with
                                  (k_list_bounds, on the device: new k1 / b / averages, or a fields_boost vector
                                  not among the few most recent ones)                                  */
} ps_batch_stats;
ps_status ps_snapshot_last_stats(const ps_snapshot* snap, ps_batch_stats* out);
/* HIP-event time (ms) summed over every launch of the posting-accumulate kernel on this
 * snapshot since the last reset, and the number of launches; waits for outstanding launches.
 * Works for caller-stream (pipelined) batches too: the events are recorded on that stream. */
ps_status ps_snapshot_kernel_times(ps_snapshot* snap, double* total_ms, uint64_t* launches, int reset);
/* The same with the kernels in front of it separated and the kernel named: score_ms covers the
 * posting-accumulate kernel ALONE (K1 k_score / K1d k_daat / K2 k_z21), rows_ms K0 k_bm25_lut + K0b
 * k_dense_rows, summed over `launches` batches.  score_kernel is the demangled symbol of the most
 * recent batch's scoring kernel, as rocprofv3 --kernel-trace prints it. */
typedef struct ps_kernel_times {
  double score_ms;
  double rows_ms;
  uint64_t launches;
  char score_kernel[96];
  double score_busy_ms; /* wall-clock during which at least one of those scoring launches was executing (the union of their
                         * [start, end] intervals): equals score_ms while launches run one after the other, less when consecutive
                         * batches' kernels overlap on two hardware queues (PS_SCORE_ALT) - then THIS is the time the chip spent
                         * scoring, and score_ms / launches overstates a batch's share of it */
} ps_kernel_times;
ps_status ps_snapshot_kernel_breakdown(ps_snapshot* snap, ps_kernel_times* out, int reset);
/* Work the scoring kernels really did, counted BY THE KERNELS (always on): every wave adds its counts
 * to device words when it finishes a work item, so the figures belong to the launches that ran, not to
 * a model of them.  Summed over every batch of this snapshot since the last reset; the call waits for
 * outstanding work.  This is what bench.py's roofline (`bytes_touched`) is computed from: the exact
 * pruning kernel K1d k_daat skips most of the postings the reference walks (src/query.rs:61-89), so
 * the bytes it moves have to be counted, not derived from the plan.
 *   K1d k_daat / k_daat_small: a posting of the item's own list is "scanned" (doc id + its score-plane
 *   values: 4 + 8F bytes); one that passes the first bound test "reaches the lookups"; a lookup into
 *   another list is one 8-byte dense-row read, one 8-byte {bits, rank} bitmap-cell read, or 4-byte probes
 *   of a sparse list's table slot (two table words, the probed doc ids); a lookup that finds the document
 *   reads its score-plane values (8F bytes).
 *   K1 k_score: every posting of every (query, list) is streamed (4 + 4F bytes), a dense-row use reads
 *   8 bytes per document and plane of the tile slice. */
typedef struct ps_work_counters {
  uint64_t launches;            /* scoring-kernel launches counted                                   */
  uint64_t items;               /* K1d: work items (chunks of lists) handed to the launches          */
  uint64_t items_run;           /* K1d: items that scanned at least one trip (the rest were skipped whole) */
  uint64_t postings_scanned;    /* K1d: postings of the items' own lists read                        */
  uint64_t postings_reached_lookups; /* K1d: ... that survived the first bound test                  */
  uint64_t lookups_row;         /* K1d: 8-byte dense-row reads                                       */
  uint64_t lookups_cell;        /* K1d: 8-byte bitmap-cell reads                                     */
  uint64_t lookups_probe;       /* K1d: 4-byte binary-search probes                                  */
  uint64_t lookup_hits;         /* K1d: lookups that found the document and scored its posting       */
  uint64_t offers;              /* K1d: documents offered to a wave's top-K                          */
  uint64_t k1_items;            /* K1: (query, run) items processed                                  */
  uint64_t k1_postings;         /* K1: postings streamed through the LDS tiles                       */
  uint64_t k1_row_slices;       /* K1: tile slices of dense rows read (tile_docs x 8 bytes x planes) */
  uint64_t results;             /* results written (top-k slots filled)                              */
  uint64_t rows_built;          /* K1d: dense score rows scored by K0b (k_prep_finish's choice)      */
  uint64_t rows_used;           /* K1d: dense score rows read by the batches (resident ones included) */
  uint64_t bytes_touched;       /* the formula above applied to these counts, + 12 bytes per candidate
                                   slot written and 16 bytes per result                              */
  uint64_t z_postings_scanned;  /* K1dz (zero_to_one): the part of postings_scanned read as doc id + packed words
                                   (4 + 4F bytes instead of 4 + 8F)                                   */
  uint64_t z_lookup_hits;       /* K1dz: the part of lookup_hits that fetched packed words (4F bytes) */
} ps_work_counters;
ps_status ps_snapshot_work_counters(ps_snapshot* snap, ps_work_counters* out, int reset);

/* ------------------------------------------------------------------ host-side inspection ---- */
/* The query plan the host hands to the kernels (tokenise -> expand_term -> before_each), one
 * entry per expanded list.  Host-only; works on device = -1 snapshots. */
typedef struct ps_plan_entry {
  uint64_t post_off;   /* first posting of the list in the CSR planes                            */
  uint32_t len;        /* postings in the list                                                    */
  uint32_t tbl_off;    /* first entry of the list's tile-offset table                             */
  uint32_t shift;      /* bits 0-7: table slot of tile t is t >> shift; bits 8+: version layer    */
  uint32_t qterm;      /* ordinal of the query term (visited-set scope, src/query.rs:37)          */
  double idf;          /* BM25TermCalculations::idf | zero_to_one: unused                          */
  double boost;        /* BM25TermCalculations::expansion_boost | zero_to_one: ScoreByTerm::score */
  uint32_t node;       /* zero_to_one: ordinal of the distinct trie node within the query | BM25: ordinal of the list (layer) in the snapshot */
  uint32_t qterm_index;/* TermData::query_term_index (position in the token list)                 */
  uint32_t bm_off;     /* first word of the list's membership bitmap, 0xFFFFFFFF = none (K1d lookups) */
  uint32_t layer;      /* ordinal of the list (version / delta layer of a term) in the snapshot   */
} ps_plan_entry;
ps_status ps_snapshot_plan(const ps_snapshot* snap, const ps_scorer_desc* scorer, const char* query,
                           size_t query_len, ps_tokenizer_fn tokenizer, void* user, ps_plan_entry** out,
                           size_t* out_len, size_t* query_terms_len);
/* The plan the device planner builds, copied back for inspection (entries of all queries, qbeg with
 * n_queries + 1 entries, per-query query_terms_len; release each with ps_free).  Must equal what
 * ps_snapshot_plan returns query by query. */
ps_status ps_snapshot_plan_device(ps_snapshot* snap, const ps_scorer_desc* scorer, const char* text, const uint64_t* offsets,
                                  size_t n_queries, ps_plan_entry** entries, size_t* n_entries, uint32_t** qbeg,
                                  uint32_t** query_terms_len);

/* Borrowed pointers into the host copy of the CSR planes (valid while the snapshot lives). */
typedef struct ps_host_csr {
  const uint32_t* doc;       /* [n_postings_padded]                     */
  const uint32_t* tf;        /* [F][n_postings_padded]                  */
  const uint32_t* fl;        /* [F][n_postings_padded]                  */
  const uint32_t* table;     /* [n_table_entries]                       */
  const uint64_t* keys;      /* [n_docs] doc id -> key (ascending keys) */
  const double* avg;         /* [F] FieldDetails::avg                   */
  uint64_t plane_stride;     /* n_postings_padded                       */
  const uint32_t* alive;     /* one bit per doc id (cleared: removed by a delta update) */
} ps_host_csr;
ps_status ps_snapshot_host_csr(const ps_snapshot* snap, ps_host_csr* out);

#ifdef __cplusplus
}
#endif
#endif /* PROBLY_SEARCH_AMD_H */
