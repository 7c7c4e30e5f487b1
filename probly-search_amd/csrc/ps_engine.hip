// ps_engine.hip — batch execution of the query-scoring path on one MI355X (gfx950 / CDNA4):
// snapshot upload, per-batch plan staging, kernel launches (device code: ps_kernels.hpp, K4 sort:
// ps_sort.hip), result download, HIP-event kernel timing.
//
// Replaces HOT LOOP #1 + #2 of Index::query (src/query.rs:45, 61-89 of probly-search 2.0.1), the
// per-posting ScoreCalculator::score of BM25 (src/score/default/bm25.rs:60-93) and zero_to_one
// (src/score/default/zero_to_one.rs:44-126), max_score_merger (src/query.rs:150-164) and the
// result materialisation + sort (src/query.rs:97-105) with:
//
//       k_pack_tfl   per snapshot (and per delta): the packed {tf, field length} posting words the hot kernels stream
//       k_upload     per batch: the staged plan, read from the device-mapped pinned slot
//       k_list_bounds / k_prep_query / k_prep_items  per K1d batch (ps_prep_kernels.hpp): per-list score
//                    bounds, work descriptors, item order, candidate slots and dense-row choice, all on the device
//   K0  k_bm25_lut   per (k1, b): saturated-tf table tfn(field, tf < 16, field length), same f64 expression
//   K0b k_dense_rows per-document score rows of the hot (list, idf, boost) combinations not yet resident
//   K1  k_score      persistent waves, one (query, run of S doc tiles) item at a time; wave-private
//                    LDS tile of f64 accumulators; plan entries in plan order: first trips of up
//                    to 3 lists in flight together, double-buffered streaming for long slices,
//                    dense-row slices for hot lists; branch-free scoring (LUT gather, exact
//                    association, no FMA contraction); add / max / assign merge with u16 visited
//                    tags; per tile harvest into a register-resident wave top-K with a per-query
//                    threshold shared through one device-scope word.  No barriers after the LUT
//                    load, no inter-wave atomics on scores: bit-reproducible.
//   K1d k_daat_small / k_daat   BM25 top-k batches: exact dynamic pruning (per-list upper bounds, essential lists,
//                    document-at-a-time lookups), one wave per chunk of a list; K3d k_merge_items behind it.  A batch
//                    that holds queries of both kinds is scored by both kernels; consecutive batches' kernels share
//                    the chip on two scoring streams (normal / low priority); the score plane they read is boost-free
//   K1dz k_daat_z    the same for zero_to_one (ps_z21_daat.hpp): queries of <= 4 lists, or <= 8 (wide instantiation)
//   K2  k_z21        zero_to_one general case (same node under two query terms, version layers)
//   K3  k_merge      per query: merge of the per-run top-K lists, doc id -> key
//   K4  ps_sort.hip  full-result mode: canonical (score desc, key asc) order on the device
//
// Everything here is latency / VALU-f64 / LDS bound sparse gather-reduce: no MFMA on purpose.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <deque>
#include <map>
#include <memory>
#include <thread>
#include <unordered_map>
#include <mutex>
#include <stdexcept>
#include <string>

#include "ps_engine.hpp"
#include "ps_bounds.hpp"
#include "ps_errors.hpp"
#include "ps_kernels.hpp"
#include "ps_prep_kernels.hpp"
#include "ps_z21_daat.hpp"
#include "ps_pool.hpp"
#include "ps_sort.hpp"

namespace ps {

#define PS_HIP(call)                                                                            \
  do {                                                                                          \
    hipError_t _e = (call);                                                                     \
    if (_e == hipErrorOutOfMemory) {                                                            \
      (void)hipGetLastError();                                                                  \
      throw ps::DeviceOom(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #call);   \
    }                                                                                           \
    if (_e != hipSuccess)                                                                       \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #call); \
  } while (0)

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static std::atomic<uint64_t> g_opt_gen{1};  // bumped by every ps_set_option: engines re-read their knobs at the next batch
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  bool ensure(size_t n, bool zero = false) {  // true: (re)allocated
    if (n <= cap) return false;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 4 + 64;
    PS_HIP(hipMalloc((void**)&p, want * sizeof(T)));
    cap = want;
    if (zero) PS_HIP(hipMemset(p, 0, want * sizeof(T)));
    return true;
  }
  // grows the buffer keeping its first `keep` elements
  void ensure_keep(size_t n, size_t keep) {
    if (n <= cap) return;
    T* old = p;
    const size_t want = n + n / 4 + 64;
    T* np = nullptr;
    PS_HIP(hipMalloc((void**)&np, want * sizeof(T)));
    if (old && keep) PS_HIP(hipMemcpy(np, old, keep * sizeof(T), hipMemcpyDeviceToDevice));
    if (old) (void)hipFree(old);
    p = np;
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
  unsigned char* dp = nullptr;  // the same bytes as the device sees them (zero-copy access for tiny batches)
  size_t cap = 0;
  hipEvent_t done = nullptr;
  bool pending = false;
  void ensure(size_t n) {
    if (pending) { PS_HIP(hipEventSynchronize(done)); pending = false; }
    if (n <= cap) return;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    dp = nullptr;
    cap = 0;
    size_t want = n + n / 4 + 256;
    // fine-grained (coherent) so that kernels may read / write it in place without a copy
    PS_HIP(hipHostMalloc((void**)&p, want, hipHostMallocMapped | hipHostMallocCoherent));
    PS_HIP(hipHostGetDevicePointer((void**)&dp, p, 0));
    cap = want;
  }
};

// Tuning knobs, read from the environment once per engine (a getenv per knob per batch is a
// measurable share of a single query's round trip).
struct Tuning {
  uint32_t dense_max_rows = 64;  // PS_DENSE_MAX_ROWS
  uint32_t row_cache_mb = 4096;  // PS_ROW_CACHE_MB: slab of row slots kept across batches (0: rebuild every batch)
  uint32_t lpt = 1;  // PS_LPT
  uint32_t kernel_upload = 1;  // PS_KERNEL_UPLOAD: plan upload by a copy kernel instead of the copy engine
  uint32_t dense_fuse = 3;  // PS_DENSE_FUSE: bit 0 = last entry added while harvesting, bit 1 = first entry written
  uint32_t z21_general_only = 0;  // PS_Z21_GENERAL_ONLY
  uint32_t dense_min_uses = 4;  // PS_DENSE_MIN_USES
  uint32_t dense_min_density_pct = 25;  // PS_DENSE_MIN_DENSITY_PCT
  uint32_t dense_max_mb = 4096;  // PS_DENSE_MAX_MB
  uint32_t zero_copy = 1;  // PS_ZERO_COPY
  uint32_t ablate = 0;  // PS_ABLATE
  uint32_t lut = 1;  // PS_LUT
  uint32_t target_items = 45056;  // PS_TARGET_ITEMS
  uint32_t tiles_per_run = 0;  // PS_TILES_PER_RUN
  uint32_t slices = 1;  // PS_SLICES
  uint32_t wg8 = 1;  // PS_WG8
  uint32_t lut_cache = 1;  // PS_LUT_CACHE
  uint32_t z21_lds = 20480;  // PS_Z21_LDS
  uint32_t full_budget_mb = 4096;  // PS_FULL_BUDGET_MB
  uint32_t full_parts_min_kb = 32768;  // PS_FULL_PARTS_MIN_KB: pinned result blocks from this size on are sorted and downloaded in up to 4 parts (download of a part beside the sorts of the next)
  uint32_t result_pinned_min_kb = 4096;  // PS_RESULT_PINNED_MIN_KB: result blocks from this size on are pinned pool blocks (ps_free recycles them)
  uint32_t daat = 1;             // PS_DAAT: BM25 top-k batches take K1d k_daat (exact dynamic pruning)
  uint32_t daat_min_batch = 8;   // PS_DAAT_MIN_BATCH: smaller batches keep the k_score latency path
  uint32_t daat_chunk = 4096;    // PS_DAAT_CHUNK: smallest chunk of a list one item covers
  uint32_t daat_dense_min_density_pct = 60;  // PS_DAAT_DENSE_MIN_DENSITY_PCT (step level, profiles/r04_daat_row_density_step_level.txt: 40 / 55 / 60 / 65 -> C2 0.346 / 0.328 / 0.329 / 0.380 ms, C4 1.29 / 1.20 / 1.19 / 1.54, C5 unchanged)
  uint32_t daat_merge_waves = 4;   // PS_DAAT_MERGE_WAVES: most waves per query in K3d (16 / 4 / 2 measured 30 / 24 / 31 us on C2)
  uint32_t daat_split_div = 64;  // PS_DAAT_SPLIT_DIV: a list is cut into at most this many chunks
  uint32_t daat_rows = 1;        // PS_DAAT_ROWS: hot dense lists are looked up through dense score rows
  uint32_t daat_persistent = 0;  // PS_DAAT_PERSISTENT: persistent waves + item counter instead of one wave per item
  uint32_t z21_exact_numerator = 1;  // PS_Z21_EXACT_NUMERATOR: k_score<MODE_Z21S> one-division arm for small term frequencies (score_trip)
  uint32_t z21_field_prune = 1;  // PS_Z21_FIELD_PRUNE: k_score<MODE_Z21S> drops fields whose pool bound fell below the query's threshold
  uint32_t daat_multi = 1;       // PS_DAAT_MULTI: also take batches with several expansions per query term (0: they stay on K1)
  uint32_t daat_small = 1;       // PS_DAAT_SMALL: plans of <= 4 lists, one per query term, take k_daat_small (all lookups of a trip in flight together)
  uint32_t kernel_timers = 1;    // PS_KERNEL_TIMERS: HIP timing events around the K1d scoring launches (ps_snapshot_kernel_breakdown); 0 in a serving setup
  uint32_t work_counters = 1;    // PS_WORK_COUNTERS: the headline kernels (k_daat_small, k_daat_z) keep the work counters of ps_snapshot_work_counters (0: the serving instantiations, which carry none; the other kernels always count)
  uint32_t daat_z_d0_div = 4;    // PS_DAAT_Z_D0_DIV: K1dz's top tie-threshold level is the doc id ~ N / this (C3, three levels a factor 4 apart: 4 -> 0.495 ms, 8 -> 0.531; one level: 8 -> 0.587, 4 -> 0.667; profiles/r04_c3_levels_sweep.jsonl)
  uint32_t daat_z_level_shift = 2;  // PS_DAAT_Z_LEVEL_SHIFT: the levels below it are 2^shift apart
  uint32_t daat_z_levels = 3;    // PS_DAAT_Z_LEVELS: how many of them (1..3)
  uint32_t daat_z_split = 1;     // PS_DAAT_Z_SPLIT: a zero_to_one batch with queries K1dz does not take is split (those to the streaming kernels) instead of taking the streaming kernels whole
  uint32_t daat_z = 1;           // PS_DAAT_Z: zero_to_one top-k batches of simple queries with <= 4 lists take K1dz k_daat_z (ps_z21_daat.hpp)
  uint32_t daat_prime = 1;       // PS_DAAT_PRIME: threshold priming - a query's threshold starts at the K-th best posting score of its best single list (k_list_kth), lists that are non-essential under it get no work items (0: thresholds start at 0)
  uint32_t daat_sample_div = 24; // PS_DAAT_SAMPLE_DIV: multi-expansion K1d launches (k_daat<F, true>: C5) start with the chunks below doc id ~ N / this, of every rank (0: plain rank-major order)
  uint32_t daat_small_nl = 1;    // PS_DAAT_SMALL_NL: batches whose queries have <= 3 lists take k_daat_small<F, WC, 3> (0: always the four-list instantiation)
  uint32_t daat_split = 1;       // PS_DAAT_SPLIT: a BM25 K1d batch that holds queries k_daat_small takes AND others (more than 4 lists, several expansions of a term) is scored by both kernels, each over its part of the item array (0: one such query sends the whole batch to k_daat)
  uint32_t daat_sample_all = 0;  // PS_DAAT_SAMPLE_ALL: ... every K1d BM25 launch does (C2 / C4: slower)
  uint32_t dctx = 5;             // PS_DCTX: K1d batch contexts in the rotation (<= N_DCTX)
  uint32_t plan_ahead_depth = 3; // PS_PLAN_AHEAD_DEPTH: batches that may be announced at a time (ps_snapshot_plan_ahead_flat)
  uint32_t score_alt = 1;        // PS_SCORE_ALT: consecutive K1d batches alternate between the scoring stream and a second one at the LOWEST stream priority (1) - a hardware queue of its own, whose kernel fills what the other's tail leaves free (two streams of one priority share a queue and serialise: 3); 0: one scoring stream; 2: everything on the low-priority stream; 4: three-way rotation normal / low / high.  Round 5, same box: C2 0.3025 -> 0.289 ms per step, C3 0.363 -> 0.312, C4 1.118 -> 0.940, C5 1.585 -> 1.213
  uint32_t device_plan = 1;      // PS_DEVICE_PLAN: flat BM25 top-k batches (built-in tokenizer) are planned by k_plan on the device
  void load();
};

constexpr int N_STAGE = 9;  // (>= N_DCTX: host-planned batches take a pinned staging slot each)
constexpr int N_KTIMER = 32;
constexpr int N_DCTX = 8;  // K1d batch contexts (most batches of one snapshot in flight; PS_DCTX of them are in the rotation)

struct EngineImpl {
  const Snapshot* snap;
  int device;
  hipStream_t stream = nullptr;
  hipEvent_t ev[6] = {};
  uint32_t* d_doc = nullptr;
  uint32_t* d_tf = nullptr;
  uint32_t* d_fl = nullptr;
  uint32_t* d_tfl = nullptr;  // packed {tf, field length} words, [P][F] (k_pack_tfl): what K1 / K1d stream
  uint32_t* d_table = nullptr;
  uint32_t* d_bits = nullptr;
  uint32_t* d_alive = nullptr;  // one bit per doc id (delta removals)
  DevBuf<unsigned long long> d_removed_df;  // per layer, scratch of count_removed_df
  uint64_t* d_keys = nullptr;
  double* d_lut = nullptr;
  uint32_t* d_work = nullptr;
  unsigned long long* d_wstats = nullptr;  // work counters the scoring kernels add to (ps_work_counters)
  uint32_t* h_fault = nullptr;             // [4] pinned, device-mapped: a preparation kernel that finds the device's and the host's item counts at odds says so here
  uint32_t* d_fault = nullptr;
  uint64_t wc_launches = 0, wc_items = 0, wc_results = 0, wc_cand_slots = 0, wc_k = 0;  // host-side part of the same accounting
  int n_cu = 256;
  Tuning tune;
  uint64_t tune_gen = 0;  // g_opt_gen the knobs were read at
  uint64_t bytes = 0;
  std::mutex mu;
  // per-batch device buffers (grow-only; reuse is ordered by the stream)
  DevBuf<unsigned char> d_stage;  // plan entries + per-query arrays, uploaded once per batch (k_upload)
  DevBuf<uint32_t> d_cand_doc, d_out_counts, d_full_doc, d_full_cnt;
  DevBuf<double> d_cand_score, d_out_scores, d_full_score;
  DevBuf<uint64_t> d_out_keys, d_full_off;
  DevBuf<unsigned long long> d_gthr;
  DevBuf<double> d_rows;  // dense per-document score rows of the batch's hot lists
  // K1d: per list (layer) upper bounds of the saturated term frequency, exact for the current
  // (k1, b, avg): M[l*F+x] = max tfn_x over the list's postings, J[l] = max over postings of
  // sum_x boost_x * tfn_x (k_list_bounds).  M does not depend on the boosts; J does, and a few J
  // arrays stay resident (LRU by boost vector): a caller that alternates fields_boost between batches
  // (src/query.rs:26 takes it per call) recomputes nothing.
  struct ListBounds {
    bool m_valid = false;
    double k1 = 0, b = 0;
    std::vector<double> avg;
    size_t n_layers = 0;
    DevBuf<unsigned long long> M;
    // (three fields and more: the joint maximum of a boost vector is a pass over the packed words - no plane is written -, and the
    // arrays of the most recent vectors stay resident; one or two fields need none: M itself / the direction supports H)
    struct JSet { std::vector<double> boosts; DevBuf<unsigned long long> J; uint64_t last_use = 0; bool valid = false; hipEvent_t ready = nullptr; };
    hipEvent_t m_ready = nullptr;  // behind the kernel that last wrote M / H / the plane (batches on other streams wait for it)
    JSet j[3];
    DevBuf<double> plane;          // tfn * idf per (posting, field): boost-free, one per (k1, b, averages)
    DevBuf<unsigned long long> H;  // [n_layers][PREP_NDIR] (F == 2)
    DevBuf<unsigned long long> kth;  // [n_layers][F + 1][KTH_RANKS] r-th best plane values per list (F <= 2; threshold priming)
    DevBuf<double2> dirs;          // the PREP_NDIR directions
    uint64_t epoch = 0;
    DevBuf<BoundUnit> units;
    uint32_t n_units = 0;
    uint64_t units_sig = 0;        // what the work list was built for (lengths of its first units_layers layers)
    size_t units_layers = 0;
    uint32_t units_n = 0;          // units in the buffer (n_units is reset by forget_rows: the buffer outlives it)
    double last_ms = 0.0;  // host wall time of the most recent (re)computation's enqueue
    uint64_t recomputed = 0;
  } bounds;
  // K1d dense-row candidates: the densest lists of the snapshot (static slots), chosen on the host per
  // snapshot state / knobs; which of them a batch reads is decided on the device (k_prep_finish)
  struct RowCands {
    bool valid = false;
    uint32_t n = 0;
    std::vector<double> sig;  // knobs the choice was made with
    DevBuf<uint8_t> of_layer;
    uint64_t gen = 0;  // bumped whenever the choice changes (the contexts' resident rows follow)
  } cands;
  std::vector<uint32_t> z_minfl;  // zero_to_one field pruning: [layer][field] shortest field length holding the term (compute_z_bounds)
  std::vector<uint32_t> z_maxtf;  // [layer] largest term frequency of the list in any field (K1dz: the numerators a list can produce)
  DevBuf<uint32_t> d_z_minfl, d_z_maxtf;  // the two on the device (k_zplan_arrange: device-planned zero_to_one batches)
  size_t z_dev_layers = 0;
  std::map<std::pair<uint64_t, uint64_t>, double> z_ubnum_cache;  // (score bits, need | maxtf << 32) -> largest record numerator
  std::unordered_map<uint64_t, uint32_t> z_layer_of;  // post_off -> layer (zero_to_one plan entries do not carry it)
  hipStream_t copy_stream = nullptr;  // full-result mode: downloads of sorted parts beside the sorts of the next
  hipEvent_t part_done[4] = {nullptr, nullptr, nullptr, nullptr};
  DevBuf<uint32_t> d_sort_doc, d_seg, d_gs_u32;  // K4 scratch
  DevBuf<uint64_t> d_sort_score, d_pack_off, d_gs_u64;
  DevBuf<unsigned char> d_sort_tmp;
  DevBuf<ps_result> d_pack;
  Stage stage[N_STAGE];
  int next_stage = 0;
  Stage result;  // download staging (engine stream only)
  // HIP-event pairs around every launch of the scoring kernel (K1/K2), harvested lazily so a
  // caller that pipelines batches on its own stream still gets per-launch durations.
  // a: before K0/K0b, m: before the scoring kernel (K1 / K2 / K1d), b: after it
  struct KTimer { hipEvent_t a = nullptr, m = nullptr, b = nullptr, r = nullptr; bool pending = false, split = false; };
  KTimer kt[N_KTIMER];
  KTimer* last_kt = nullptr;
  // N2 device-side planner: the frozen trie + per-term / per-layer tables in HBM (uploaded on first
  // use, again after a delta changed them), per-batch scratch
  bool dev_trie_valid = false;
  bool dev_trie_struct_valid = false;  // the frozen trie itself (nodes, child characters): survives deltas that add no term
  DevBuf<uint4> d_fnodes, d_layer_a, d_layer_b, d_fbits;
  DevBuf<uint32_t> d_fchar, d_fchild, d_term_meta, d_term_delta;
  DevBuf<uint64_t> d_term_df;
  DevBuf<double> d_term_idf, d_eb_table, d_layer_idf;
  // Bloom filters of the lists without a bitmap (K1d lookups): built on the device per snapshot state
  DevBuf<unsigned long long> d_bloom, d_layer_bloom;
  bool bloom_valid = false;
  uint64_t bloom_words = 0;
  uint32_t eb_n = 0;
  // A device-built plan (k_plan): the batch's text, the per-query counts of the count pass, the entries.
  struct PlanSet {
    DevBuf<char> qtext;  // offsets | text
    DevBuf<uint32_t> cnt, qtl, nterms, multi, qbeg, qorder, items;
    DevBuf<unsigned long long> post;
    DevBuf<ps_plan_entry> entries;
    DevBuf<int32_t> tok_node;  // [B][64] trie node of every token (count pass -> fill pass)
    DevBuf<double> z_ubnum, z_zub;  // zero_to_one: K1dz's per-record bounds (k_zplan_arrange)
    Stage h;  // pinned copy of the batch's offsets | text (the caller's buffer may be pageable)
    void release() {
      qtext.release(); cnt.release(); qtl.release(); nterms.release(); multi.release(); qbeg.release();
      qorder.release(); items.release(); post.release(); entries.release(); tok_node.release(); z_ubnum.release(); z_zub.release();
      if (h.p) (void)hipHostFree(h.p);
      if (h.done) (void)hipEventDestroy(h.done);
    }
  };
  // Everything one K1d batch owns on the device, twice.  Batch s + 1 is uploaded / planned, prepared and
  // has its dense rows scored on the engine's PREPARATION stream (high priority: its own hardware queue,
  // and its small kernels are dispatched between the workgroups of a running k_daat) while batch s is
  // still being scored on the SCORING stream; the scoring stream then only carries k_daat and the merge
  // of consecutive batches, back to back.  Only the merge - the kernel that writes the caller's output
  // buffers - waits for the caller's stream; the caller's stream in turn waits for the batch's `done`.
  struct DaatCtx {
    hipEvent_t done = nullptr;     // behind the batch's merge (scoring stream)
    hipEvent_t entry = nullptr;    // the caller stream's position when the batch was submitted
    hipEvent_t prepared = nullptr; // behind the batch's preparation (preparation stream)
    hipEvent_t counted = nullptr;  // behind the planner's count pass (planning stream)
    hipEvent_t scored = nullptr;   // behind the batch's k_daat (scoring stream)
    hipEvent_t scored2 = nullptr;  // a split batch: behind the second part's kernel on the other scoring stream (the first stream waits for it)
    bool busy = false;
    PlanSet plan;                  // device-planned batches
    DevBuf<unsigned char> stage;   // host-planned batches: entries | qbeg | qterms_len as uploaded
    DevBuf<DEntry> dentry;
    DevBuf<DGroup> dgroup;
    DevBuf<DItemGen> gen;
    DevBuf<uint32_t> rorder, qslot, qslot_n, cand_cnt, cand_doc;
    DevBuf<uint8_t> gord;
    DevBuf<DItem> ditems;
    DevBuf<double> cand_score, rows;
    DevBuf<unsigned long long> gthr, gtie;
    DevBuf<uint32_t> z_nbelow, z_nabove;  // K1dz preparation scratch
    PrepCtl* ctl = nullptr;
    uint32_t* work = nullptr;
    RowState* row_state = nullptr;
    RowDesc* row_desc = nullptr;
    std::vector<double> row_sig;   // scorer parameters + boosts the context's resident rows were scored with
    uint64_t cands_gen = 0;        // RowCands::gen its rows belong to
    bool ctl_clean = false;        // k_merge_items left the control words zeroed
  };
  // Five: batch s + 1 reuses the context of batch s + 1 - N_DCTX, whose merge must be through first.  With three, the chain
  // behind k_daat(s - 2) - its merge (100-200 us beside a running k_daat), then text upload, count pass, scan, the host's
  // read of the totals, fill pass, k_prep_query, k_prep_items, K0b of batch s + 1 - was longer than one k_daat, so the
  // scoring stream idled 25-30 us per batch (kernel timeline, profiles/r04_c2_timeline_3ctx.txt).
  DaatCtx dctx[N_DCTX];
  int next_dctx = 0;
  hipStream_t prep_stream = nullptr, score_stream = nullptr, plan_stream = nullptr, merge_stream = nullptr;
  hipStream_t score_stream_lo = nullptr, score_stream_b = nullptr, score_stream_hi = nullptr;  // PS_SCORE_ALT
  uint32_t score_flip = 0;
  hipEvent_t lut_ready = nullptr;  // behind the most recent k_bm25_lut
  PlanTotals* h_totals = nullptr;  // [N_DCTX] pinned, device-mapped, one per batch context: k_plan_scan writes the totals where the host reads them
  PlanTotals* d_totals_mapped = nullptr;
  // A batch announced ahead of its query call (Engine::plan_ahead): its text is in the context's pinned slot, its
  // count pass is in flight or done.  The next device-planned call with the same text picks it up.
  // Several may be announced (oldest first; PS_PLAN_AHEAD_DEPTH): the host's loop - wait for a batch's totals, enqueue it, announce
  // the next - is a chain of latencies (count pass under a running k_daat 30-160 us, the wake-up, ~100 us of enqueueing), and with one
  // batch announced that chain, not the GPU, paced C2 (streams idle between kernels; kernel timeline, round 5).
  struct Ahead { int ctx = -1; size_t B = 0, n_bytes = 0; uint64_t tune_gen = 0; };  // tune_gen: the knobs its count pass chunked the lists with
  std::deque<Ahead> aheads;
  KTimer* last_kt_pending = nullptr;  // full-result path: the timer of the batch being enqueued
  uint64_t last_layout_bytes = 0;  // of the most recently staged batch
  uint32_t last_rows = 0, last_rows_built = 0;
  bool last_bounds_recomputed = false;
  // Row slab: the dense score row of a hot (list, weight) combination only depends on the
  // snapshot, the scorer parameters and the boosts, so rows stay resident across batches (288 GB
  // of HBM: a few GB of slots is nothing) and K0b only scores the combinations it has not seen.
  // LRU over the slots; everything is dropped when the scorer parameters or boosts change.
  struct RowKey {
    uint64_t post_off, w, k3;
    bool operator<(const RowKey& o) const {
      return post_off != o.post_off ? post_off < o.post_off : w != o.w ? w < o.w : k3 < o.k3;
    }
  };
  struct RowSlot { RowKey key; uint64_t last_use = 0; bool valid = false; };
  std::map<RowKey, uint32_t> row_slot_of;
  std::vector<RowSlot> row_slots;
  std::vector<double> row_sig;  // scorer kind, k1, b, boosts the resident rows were scored with
  uint64_t row_epoch = 0;
  std::vector<uint32_t> build_slots;  // row slots of the batch being enqueued that need a host-side zero fill
  int next_kt = 0;
  // control words (item counter + per-query thresholds) are left zeroed by k_merge: no memset per batch
  bool ctl_clean = false;
  // the saturated-tf table only depends on (k1, b) and the snapshot: built once, not per batch
  bool lut_valid = false;
  double lut_k1 = 0.0, lut_b = 0.0;
  hipStream_t lut_stream = nullptr;
  // Batches share the per-batch device buffers, so they execute one after the other.  On one
  // stream that is stream order; a batch enqueued on another stream first waits for `tail`, the
  // event behind the previous asynchronous batch.
  hipStream_t tail_stream = nullptr;
  bool tail_pending = false;
  std::map<std::pair<const void*, size_t>, size_t> occ_cache;  // (K1 instantiation, LDS bytes) -> waves per CU
  Stage* cur_stage = nullptr;   // slot of the batch being enqueued
  bool cur_zero_copy = false;   // its plan is read in place from pinned host memory
  double kt_total_ms = 0.0;  // scoring kernel alone (m -> b)
  double kt_rows_ms = 0.0;   // K0 + K0b in front of it (a -> m)
  uint64_t kt_launches = 0;
  // With PS_SCORE_ALT consecutive scoring kernels overlap (two hardware queues): their individual durations no longer add up
  // to the time the chip spent scoring.  kt_busy_ms = length of the union of the launches' [start, end] intervals, measured
  // against a reference event (re-recorded at every reset: hipEventElapsedTime is a float).
  hipEvent_t kt_ref = nullptr;
  double kt_busy_ms = 0.0, kt_cur_end = -1.0;
  std::string score_kernel_name;  // demangled symbol of the scoring kernel of the most recent batch
  void harvest(KTimer& t, bool wait) {
    if (!t.pending) return;
    if (!wait && hipEventQuery(t.b) != hipSuccess) return;
    if (wait) (void)hipEventSynchronize(t.b);
    float ms = 0, ms0 = 0;
    if (hipEventElapsedTime(&ms, t.m, t.b) == hipSuccess) { kt_total_ms += ms; kt_launches++; }
    float s0 = 0, e0 = 0;
    if (kt_ref && hipEventElapsedTime(&s0, kt_ref, t.m) == hipSuccess && hipEventElapsedTime(&e0, kt_ref, t.b) == hipSuccess) {
      const double from = std::max((double)s0, kt_cur_end);
      if ((double)e0 > from) kt_busy_ms += (double)e0 - from;
      kt_cur_end = std::max(kt_cur_end, (double)e0);
    } else {
      kt_busy_ms += ms;
    }
    // (K1d batches score their rows on the preparation stream: a -> r there, m -> b on the scoring stream)
    if (hipEventElapsedTime(&ms0, t.a, t.split ? t.r : t.m) == hipSuccess) kt_rows_ms += ms0;
    t.pending = false;
  }
  // hipEventElapsedTime is a float: ten seconds behind the reference an offset still resolves a microsecond, an hour behind it a
  // quarter of a millisecond - and the union of intervals would be noise.  Before a timer is taken for a new launch:
  // once the offsets pass 10 s, everything outstanding is harvested (oldest first) and the reference re-recorded.  The accumulated
  // sums carry over; only the running interval end starts again (nothing timed is in flight at that moment).
  void rebase_timers_if_stale(hipStream_t st) {
    if (kt_cur_end < 10000.0 || !kt_ref) return;
    for (int i = 0; i < N_KTIMER; ++i) harvest(kt[(next_kt + i) % N_KTIMER], true);
    if (hipEventRecord(kt_ref, st) == hipSuccess) (void)hipEventSynchronize(kt_ref);
    kt_cur_end = -1.0;
  }
};

namespace { void forget_rows(EngineImpl& m); void ensure_bound_units(EngineImpl& m); }

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
      throw NoDeviceError("no HIP device available (this engine has no CPU scoring fallback)");
    if (device < 0 || device >= n) throw std::invalid_argument("device index out of range");
    PS_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PS_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
      throw std::runtime_error(std::string("built for gfx950 (MI355X) only; device is ") + prop.gcnArchName);
    m.n_cu = prop.multiProcessorCount;
    m.tune.load();
    m.tune_gen = g_opt_gen.load();
    PS_HIP(hipMalloc((void**)&m.d_work, 256));
    PS_HIP(hipMemset(m.d_work, 0, 256));
    PS_HIP(hipEventCreateWithFlags(&m.lut_ready, hipEventDisableTiming));
    PS_HIP(hipEventCreateWithFlags(&m.bounds.m_ready, hipEventDisableTiming));
    for (auto& js : m.bounds.j) PS_HIP(hipEventCreateWithFlags(&js.ready, hipEventDisableTiming));
    PS_HIP(hipMalloc((void**)&m.d_wstats, (size_t)WS_SLOTS * WS_WORDS * 8));
    PS_HIP(hipMemset(m.d_wstats, 0, (size_t)WS_SLOTS * WS_WORDS * 8));
    PS_HIP(hipHostMalloc((void**)&m.h_fault, 4 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
    memset(m.h_fault, 0, 4 * sizeof(uint32_t));
    PS_HIP(hipHostGetDevicePointer((void**)&m.d_fault, m.h_fault, 0));
    PS_HIP(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
    for (auto& ev : m.ev) PS_HIP(hipEventCreate(&ev));
    for (auto& sg : m.stage) PS_HIP(hipEventCreateWithFlags(&sg.done, hipEventDisableTiming));
    PS_HIP(hipEventCreateWithFlags(&m.result.done, hipEventDisableTiming));
    for (auto& t : m.kt) { PS_HIP(hipEventCreate(&t.a)); PS_HIP(hipEventCreate(&t.m)); PS_HIP(hipEventCreate(&t.b)); PS_HIP(hipEventCreate(&t.r)); }
    const size_t P = snap.P, F = snap.F;
    PS_HIP(hipMalloc((void**)&m.d_doc, P * 4));
    PS_HIP(hipMalloc((void**)&m.d_tf, P * F * 4));
    PS_HIP(hipMalloc((void**)&m.d_fl, P * F * 4));
    PS_HIP(hipMalloc((void**)&m.d_tfl, std::max<size_t>(1, P * F) * 4));
    PS_HIP(hipMalloc((void**)&m.d_table, snap.table.size() * 4));
    PS_HIP(hipMalloc((void**)&m.d_bits, snap.bits.size() * 4));
    PS_HIP(hipMemcpy(m.d_bits, snap.bits.data(), snap.bits.size() * 4, hipMemcpyHostToDevice));
    PS_HIP(hipMalloc((void**)&m.d_keys, std::max<size_t>(1, (size_t)snap.tiles_cap * snap.T) * 8));
    PS_HIP(hipMalloc((void**)&m.d_alive, std::max<size_t>(1, snap.alive.size()) * 4));
    if (!snap.alive.empty()) PS_HIP(hipMemcpy(m.d_alive, snap.alive.data(), snap.alive.size() * 4, hipMemcpyHostToDevice));
    PS_HIP(hipMalloc((void**)&m.d_lut, ((size_t)snap.lut_rows + 4) * LUT_TF * 8));
    PS_HIP(hipMemcpy(m.d_doc, snap.doc.data(), P * 4, hipMemcpyHostToDevice));
    PS_HIP(hipMemcpy(m.d_tf, snap.tf.data(), P * F * 4, hipMemcpyHostToDevice));
    PS_HIP(hipMemcpy(m.d_fl, snap.fl.data(), P * F * 4, hipMemcpyHostToDevice));
    if (P * F) {
      hipLaunchKernelGGL(k_pack_tfl, dim3(2048), dim3(256), 0, m.stream, m.d_tf, m.d_fl, m.d_tfl, (uint64_t)P, (uint32_t)F, (uint64_t)0, (uint64_t)P);
      PS_HIP(hipGetLastError());
      PS_HIP(hipStreamSynchronize(m.stream));
    }
    PS_HIP(hipMemcpy(m.d_table, snap.table.data(), snap.table.size() * 4, hipMemcpyHostToDevice));
    if (!snap.keys.empty())
      PS_HIP(hipMemcpy(m.d_keys, snap.keys.data(), snap.keys.size() * 8, hipMemcpyHostToDevice));
    m.bytes = P * 4 + 3 * P * F * 4 + snap.table.size() * 4 + snap.keys.size() * 8 + snap.bits.size() * 4;
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
  for (void* p : {(void*)m.d_doc, (void*)m.d_tf, (void*)m.d_fl, (void*)m.d_tfl, (void*)m.d_table, (void*)m.d_bits, (void*)m.d_alive, (void*)m.d_keys, (void*)m.d_lut, (void*)m.d_work, (void*)m.d_wstats})
    if (p) (void)hipFree(p);
  m.d_stage.release(); m.d_cand_doc.release();
  m.d_out_counts.release(); m.d_full_doc.release(); m.d_full_cnt.release(); m.d_cand_score.release();
  m.d_out_scores.release(); m.d_full_score.release(); m.d_out_keys.release(); m.d_full_off.release();
  m.d_gthr.release(); m.d_rows.release(); m.d_removed_df.release();
  m.bounds.M.release(); m.bounds.units.release();
  if (m.bounds.m_ready) (void)hipEventDestroy(m.bounds.m_ready);
  for (auto& js : m.bounds.j) { js.J.release(); if (js.ready) (void)hipEventDestroy(js.ready); }
  m.bounds.plane.release(); m.bounds.H.release(); m.bounds.dirs.release();
  if (m.lut_ready) (void)hipEventDestroy(m.lut_ready);
  m.cands.of_layer.release();
  for (auto& c : m.dctx) {
    c.plan.release(); c.stage.release(); c.dentry.release(); c.dgroup.release(); c.gen.release(); c.rorder.release();
    c.qslot.release(); c.qslot_n.release(); c.cand_cnt.release(); c.cand_doc.release(); c.gord.release(); c.ditems.release();
    c.cand_score.release(); c.rows.release(); c.gthr.release();
    for (void* p : {(void*)c.ctl, (void*)c.work, (void*)c.row_state, (void*)c.row_desc})
      if (p) (void)hipFree(p);
    for (hipEvent_t e : {c.done, c.entry, c.prepared, c.counted, c.scored, c.scored2})
      if (e) (void)hipEventDestroy(e);
  }
  if (m.prep_stream) (void)hipStreamDestroy(m.prep_stream);
  if (m.plan_stream) (void)hipStreamDestroy(m.plan_stream);
  if (m.merge_stream) (void)hipStreamDestroy(m.merge_stream);
  if (m.score_stream) (void)hipStreamDestroy(m.score_stream);
  if (m.score_stream_lo) (void)hipStreamDestroy(m.score_stream_lo);
  if (m.score_stream_b) (void)hipStreamDestroy(m.score_stream_b);
  if (m.score_stream_hi) (void)hipStreamDestroy(m.score_stream_hi);
  if (m.kt_ref) (void)hipEventDestroy(m.kt_ref);
  if (m.copy_stream) (void)hipStreamDestroy(m.copy_stream);
  for (auto& e : m.part_done) if (e) (void)hipEventDestroy(e);
  m.d_fnodes.release(); m.d_layer_a.release(); m.d_layer_b.release(); m.d_fbits.release(); m.d_fchar.release(); m.d_fchild.release();
  m.d_term_meta.release(); m.d_term_delta.release(); m.d_term_df.release(); m.d_term_idf.release(); m.d_eb_table.release(); m.d_layer_idf.release(); m.d_bloom.release(); m.d_layer_bloom.release();
  if (m.h_totals) (void)hipHostFree(m.h_totals);
  if (m.h_fault) (void)hipHostFree(m.h_fault);
  m.d_sort_doc.release(); m.d_seg.release(); m.d_sort_score.release(); m.d_pack_off.release();
  m.d_sort_tmp.release(); m.d_pack.release(); m.d_gs_u32.release(); m.d_gs_u64.release();
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
    if (t.m) (void)hipEventDestroy(t.m);
    if (t.b) (void)hipEventDestroy(t.b);
    if (t.r) (void)hipEventDestroy(t.r);
  }
  if (m.stream) (void)hipStreamDestroy(m.stream);
  delete impl_;
}

// Per layer: sum of tf over the postings whose document a delta removed (Index::count_documents skips
// removed documents, index.rs:287-293).  One workgroup-strided pass over the doc / tf planes.
// One wave per unit (a list, or a 16 Ki-posting segment of a long one - the units of k_list_bounds): a block per LIST left the
// million-posting lists to 256 threads each (4.7 ms for C2's planes); segments keep every compute unit busy.
__global__ __launch_bounds__(256) void k_removed_df(const uint32_t* doc, const uint32_t* tf, const uint32_t* alive, uint64_t P,
                                                     uint32_t F, const BoundUnit* units, uint32_t n_units, const uint4* layer_a,
                                                     unsigned long long* out) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  if (wave >= n_units) return;
  const BoundUnit u = units[wave];
  const uint4 la = layer_a[u.layer];
  const uint64_t off = ((uint64_t)la.x | ((uint64_t)la.y << 32)) + u.begin;
  unsigned long long acc = 0;
  for (uint32_t i = lane; i < u.count; i += WAVE) {
    const uint64_t pi = off + i;
    const uint32_t d = doc[pi];
    if (!((alive[d >> 5] >> (d & 31u)) & 1u))
      for (uint32_t x = 0; x < F; ++x) acc += tf[(uint64_t)x * P + pi];
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0 && acc) atomicAdd(&out[u.layer], acc);
}

void Engine::apply_delta(const DeltaRanges& r, std::vector<uint64_t>& removed_df, uint64_t* bytes_uploaded) {
  EngineImpl& m = *impl_;
  const Snapshot& s = *m.snap;
  std::lock_guard<std::mutex> lock(m.mu);
  PS_HIP(hipSetDevice(m.device));
  PS_HIP(hipDeviceSynchronize());  // no batch of this snapshot is in flight while its planes change
  m.aheads.clear();                // (batches announced ahead were counted against the old trie: their query calls plan again)
  uint64_t up = 0;
  auto put = [&](void* dst, const void* src, size_t bytes) {
    if (!bytes) return;
    PS_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    up += bytes;
  };
  const size_t F = s.F, P = s.P;
  if (r.plane_end > r.plane_begin) {
    const size_t b = r.plane_begin, n = r.plane_end - r.plane_begin;
    put(m.d_doc + b, s.doc.data() + b, n * 4);
    for (size_t x = 0; x < F; ++x) {
      put(m.d_tf + x * P + b, s.tf.data() + x * P + b, n * 4);
      put(m.d_fl + x * P + b, s.fl.data() + x * P + b, n * 4);
    }
    hipLaunchKernelGGL(k_pack_tfl, dim3((uint32_t)std::min<size_t>(2048, (n * F + 255) / 256)), dim3(256), 0, m.stream, m.d_tf, m.d_fl,
                       m.d_tfl, (uint64_t)P, (uint32_t)F, (uint64_t)b, (uint64_t)(b + n));
    PS_HIP(hipGetLastError());
    PS_HIP(hipStreamSynchronize(m.stream));
  }
  if (r.table_end > r.table_begin) put(m.d_table + r.table_begin, s.table.data() + r.table_begin, (r.table_end - r.table_begin) * 4);
  if (r.key_end > r.key_begin) put(m.d_keys + r.key_begin, s.keys.data() + r.key_begin, (r.key_end - r.key_begin) * 8);
  if (r.alive_words.size() > 16) {  // (one blocking 4-byte copy per word cost ~10 us each: 1000 removals = 10 ms; the whole bitmap of a million documents is 125 KB)
    put(m.d_alive, s.alive.data(), s.alive.size() * 4);
  } else {
    for (uint32_t w : r.alive_words) put(m.d_alive + w, s.alive.data() + w, 4);
  }
  // what depended on the old state: resident dense rows (doc range, avg), the saturated-tf LUT (avg),
  // the device copy of the trie / term tables
  const bool filters_still_good = m.bloom_valid && r.plane_end == r.plane_begin;  // removals only: every list holds the documents it held (tombstones answer "maybe", which is right)
  forget_rows(m);
  m.bloom_valid = filters_still_good;
  m.dev_trie_valid = false;
  if (r.trie_refrozen) m.dev_trie_struct_valid = false;
  removed_df.assign(s.layers.size(), 0);
  if (s.any_dead && !s.layers.empty()) {
    const size_t nl = s.layers.size();
    // the per-layer {post_off, len} table of the new state.  (NOT ensure_dev_trie: its idf tables need the document frequencies this
    // very pass is about to correct - they are rebuilt at the next batch, dev_trie_valid stays false)
    {
      std::vector<uint4> la(nl);
      for (size_t l = 0; l < nl; ++l) la[l] = make_uint4((uint32_t)s.layers[l].post_off, (uint32_t)(s.layers[l].post_off >> 32), s.layers[l].len, s.layers[l].tbl_off);
      m.d_layer_a.ensure(nl + 1);
      PS_HIP(hipMemcpy(m.d_layer_a.p, la.data(), nl * sizeof(uint4), hipMemcpyHostToDevice));
    }
    ensure_bound_units(m);
    m.d_removed_df.ensure(nl + 8);
    unsigned long long* d_out = m.d_removed_df.p;
    PS_HIP(hipMemset(d_out, 0, nl * 8));
    if (m.bounds.n_units)
      hipLaunchKernelGGL(k_removed_df, dim3((m.bounds.n_units + 3) / 4), dim3(256), 0, m.stream, m.d_doc, m.d_tf, m.d_alive, (uint64_t)P,
                         (uint32_t)F, m.bounds.units.p, m.bounds.n_units, m.d_layer_a.p, d_out);
    PS_HIP(hipGetLastError());
    PS_HIP(hipStreamSynchronize(m.stream));
    PS_HIP(hipMemcpy(removed_df.data(), d_out, nl * 8, hipMemcpyDeviceToHost));
  }
  if (bytes_uploaded) *bytes_uploaded = up;
}

uint64_t Engine::device_bytes() const { return impl_->bytes; }
void Engine::kernel_times(ps_kernel_times& out, bool reset) {
  EngineImpl& m = *impl_;
  std::lock_guard<std::mutex> lock(m.mu);
  (void)hipSetDevice(m.device);
  for (int i = 0; i < N_KTIMER; ++i) m.harvest(m.kt[(m.next_kt + i) % N_KTIMER], true);  // (oldest first: the busy-interval bookkeeping wants launch order)
  memset(&out, 0, sizeof(out));
  out.score_ms = m.kt_total_ms;
  out.rows_ms = m.kt_rows_ms;
  out.launches = m.kt_launches;
  out.score_busy_ms = m.kt_busy_ms;
  snprintf(out.score_kernel, sizeof(out.score_kernel), "%s", m.score_kernel_name.c_str());
  if (reset) {
    m.kt_total_ms = 0.0; m.kt_rows_ms = 0.0; m.kt_launches = 0; m.kt_busy_ms = 0.0; m.kt_cur_end = -1.0;
    // a fresh reference for the busy-interval bookkeeping (every timer was harvested above: nothing refers to the old one)
    if (!m.kt_ref) (void)hipEventCreate(&m.kt_ref);
    if (m.kt_ref && hipEventRecord(m.kt_ref, m.stream) == hipSuccess) (void)hipEventSynchronize(m.kt_ref);
  }
}
int Engine::device() const { return impl_->device; }

// The kernels' own work counts (see ps_work_counters) since the last reset; waits for outstanding work.
void Engine::work_counters(ps_work_counters& out, bool reset) {
  EngineImpl& m = *impl_;
  std::lock_guard<std::mutex> lock(m.mu);
  PS_HIP(hipSetDevice(m.device));
  PS_HIP(hipDeviceSynchronize());
  std::vector<unsigned long long> w((size_t)WS_SLOTS * WS_WORDS);
  PS_HIP(hipMemcpy(w.data(), m.d_wstats, w.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long sum[WS_WORDS] = {};
  for (uint32_t sl = 0; sl < WS_SLOTS; ++sl)
    for (uint32_t k = 0; k < WS_WORDS; ++k) sum[k] += w[(size_t)sl * WS_WORDS + k];
  memset(&out, 0, sizeof(out));
  out.launches = m.wc_launches;
  out.items = m.wc_items;
  out.items_run = sum[WS_ITEMS_RUN];
  out.postings_scanned = sum[WS_SCANNED];
  out.postings_reached_lookups = sum[WS_REACHED];
  out.lookups_row = sum[WS_ROW];
  out.lookups_cell = sum[WS_CELL];
  out.lookups_probe = sum[WS_PROBE];
  out.lookup_hits = sum[WS_HIT];
  out.offers = sum[WS_OFFER];
  out.k1_items = sum[WS_K1_ITEMS];
  out.k1_postings = sum[WS_K1_POSTINGS];
  out.k1_row_slices = sum[WS_K1_ROWSLICES];
  out.results = m.wc_results;
  out.rows_built = sum[WS_ROWS_BUILT];
  out.rows_used = sum[WS_ROWS_USED];
  const uint64_t F = m.snap->F, pw = 4 + 4 * F;  // K1: doc id + packed words; K1d: doc id + score plane (8 bytes per field)
  // (K1dz reads the packed words where K1d reads the score plane: 4 bytes per field instead of 8)
  const uint64_t z_scanned = sum[WS_Z_SCANNED], z_hits = sum[WS_Z_HIT];
  out.postings_scanned += z_scanned;
  out.lookup_hits += z_hits;
  out.z_postings_scanned = z_scanned;
  out.z_lookup_hits = z_hits;
  out.bytes_touched = (out.postings_scanned - z_scanned) * (4 + 8 * F) + z_scanned * pw + out.lookups_row * 8 + out.lookups_cell * 8 +
                      out.lookups_probe * 4 + (out.lookup_hits - z_hits) * 8 * F + z_hits * 4 * F + out.k1_postings * pw +
                      out.k1_row_slices * (uint64_t)m.snap->T * 8 +
                      (m.wc_cand_slots + out.items_run * m.wc_k) * 12 + out.results * 16;
  if (reset) {
    PS_HIP(hipMemset(m.d_wstats, 0, w.size() * 8));
    m.wc_launches = m.wc_items = m.wc_results = m.wc_cand_slots = 0;
  }
}

namespace {

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

std::mutex g_opt_mu;
std::map<std::string, uint32_t>& option_overrides() {
  static std::map<std::string, uint32_t> m;
  return m;
}

// A tuning knob: ps_set_option (the ABI) wins over the environment variable of the same name.
uint32_t env_u32(const char* name, uint32_t dflt) {
  {
    std::lock_guard<std::mutex> l(g_opt_mu);
    auto it = option_overrides().find(name);
    if (it != option_overrides().end()) return it->second;
  }
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  return (uint32_t)strtoul(v, nullptr, 10);
}
}  // namespace

void set_option(const char* name, uint32_t value) {
  std::lock_guard<std::mutex> l(g_opt_mu);
  option_overrides()[name] = value;
  g_opt_gen.fetch_add(1);
}
bool get_option(const char* name, uint32_t* value) {
  std::lock_guard<std::mutex> l(g_opt_mu);
  auto it = option_overrides().find(name);
  if (it != option_overrides().end()) { *value = it->second; return true; }
  const char* v = getenv(name);
  if (!v || !*v) return false;
  *value = (uint32_t)strtoul(v, nullptr, 10);
  return true;
}

// ---- pinned result blocks (ps_engine.hpp: ResultBuf) -------------------------------------------------------------
namespace {
struct ResultPool {
  std::mutex mu;
  std::unordered_map<void*, size_t> all;            // every live pool block -> bytes
  std::vector<std::pair<void*, size_t>> idle;       // handed back, ready for reuse (oldest first)
  size_t idle_bytes = 0;
};
ResultPool& result_pool() {
  static ResultPool* p = new ResultPool();  // never destroyed: blocks may be freed after static destruction began
  return *p;
}
}  // namespace

void* result_block_acquire(size_t bytes, size_t* capacity) {
  ResultPool& rp = result_pool();
  {
    std::lock_guard<std::mutex> l(rp.mu);
    size_t best = SIZE_MAX;
    for (size_t i = 0; i < rp.idle.size(); ++i)
      if (rp.idle[i].second >= bytes && (best == SIZE_MAX || rp.idle[i].second < rp.idle[best].second)) best = i;
    // (a block several times too large stays where it is for the batch it fits)
    if (best != SIZE_MAX && rp.idle[best].second <= 2 * bytes + (64u << 20)) {
      void* q = rp.idle[best].first;
      *capacity = rp.idle[best].second;
      rp.idle_bytes -= rp.idle[best].second;
      rp.idle.erase(rp.idle.begin() + (ptrdiff_t)best);
      return q;
    }
  }
  const size_t cap = ((bytes + bytes / 8) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
  void* q = nullptr;
  if (hipHostMalloc(&q, cap, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  std::lock_guard<std::mutex> l(rp.mu);
  rp.all[q] = cap;
  *capacity = cap;
  return q;
}

bool result_block_release(void* p) {
  if (!p) return false;
  ResultPool& rp = result_pool();
  std::vector<void*> drop;
  const size_t keep = (size_t)env_u32("PS_RESULT_POOL_MB", 1024) << 20;
  {
    std::lock_guard<std::mutex> l(rp.mu);
    auto it = rp.all.find(p);
    if (it == rp.all.end()) return false;
    rp.idle.emplace_back(p, it->second);
    rp.idle_bytes += it->second;
    while (rp.idle_bytes > keep && !rp.idle.empty()) {  // oldest first
      drop.push_back(rp.idle.front().first);
      rp.idle_bytes -= rp.idle.front().second;
      rp.all.erase(rp.idle.front().first);
      rp.idle.erase(rp.idle.begin());
    }
  }
  for (void* q : drop) (void)hipHostFree(q);
  return true;
}

void Tuning::load() {
    dense_max_rows = env_u32("PS_DENSE_MAX_ROWS", dense_max_rows);
    row_cache_mb = env_u32("PS_ROW_CACHE_MB", row_cache_mb);
    lpt = env_u32("PS_LPT", lpt);
    kernel_upload = env_u32("PS_KERNEL_UPLOAD", kernel_upload);
    dense_fuse = env_u32("PS_DENSE_FUSE", dense_fuse);
    z21_general_only = env_u32("PS_Z21_GENERAL_ONLY", z21_general_only);
    dense_min_uses = env_u32("PS_DENSE_MIN_USES", dense_min_uses);
    dense_min_density_pct = env_u32("PS_DENSE_MIN_DENSITY_PCT", dense_min_density_pct);
    dense_max_mb = env_u32("PS_DENSE_MAX_MB", dense_max_mb);
    zero_copy = env_u32("PS_ZERO_COPY", zero_copy);
    ablate = env_u32("PS_ABLATE", ablate);
    lut = env_u32("PS_LUT", lut);
    target_items = env_u32("PS_TARGET_ITEMS", target_items);
    tiles_per_run = env_u32("PS_TILES_PER_RUN", tiles_per_run);
    slices = env_u32("PS_SLICES", slices);
    wg8 = env_u32("PS_WG8", wg8);
    lut_cache = env_u32("PS_LUT_CACHE", lut_cache);
    z21_lds = env_u32("PS_Z21_LDS", z21_lds);
    full_budget_mb = env_u32("PS_FULL_BUDGET_MB", full_budget_mb);
    result_pinned_min_kb = env_u32("PS_RESULT_PINNED_MIN_KB", result_pinned_min_kb);
    full_parts_min_kb = env_u32("PS_FULL_PARTS_MIN_KB", full_parts_min_kb);
    daat = env_u32("PS_DAAT", daat);
    daat_min_batch = env_u32("PS_DAAT_MIN_BATCH", daat_min_batch);
    daat_chunk = std::max(256u, env_u32("PS_DAAT_CHUNK", daat_chunk));
    daat_rows = env_u32("PS_DAAT_ROWS", daat_rows);
    daat_merge_waves = std::max(1u, std::min((uint32_t)MERGE_WAVES, env_u32("PS_DAAT_MERGE_WAVES", daat_merge_waves)));
    daat_dense_min_density_pct = env_u32("PS_DAAT_DENSE_MIN_DENSITY_PCT", daat_dense_min_density_pct);
    daat_split_div = std::max(1u, env_u32("PS_DAAT_SPLIT_DIV", daat_split_div));
    daat_persistent = env_u32("PS_DAAT_PERSISTENT", daat_persistent);
    device_plan = env_u32("PS_DEVICE_PLAN", device_plan);
    daat_small = env_u32("PS_DAAT_SMALL", daat_small);
    daat_multi = env_u32("PS_DAAT_MULTI", daat_multi);
    daat_z = env_u32("PS_DAAT_Z", daat_z);
    daat_z_split = env_u32("PS_DAAT_Z_SPLIT", daat_z_split);
    work_counters = env_u32("PS_WORK_COUNTERS", work_counters);
    kernel_timers = env_u32("PS_KERNEL_TIMERS", kernel_timers);
    score_alt = env_u32("PS_SCORE_ALT", score_alt);
    plan_ahead_depth = std::max(1u, env_u32("PS_PLAN_AHEAD_DEPTH", plan_ahead_depth));
    daat_sample_div = env_u32("PS_DAAT_SAMPLE_DIV", daat_sample_div);
    daat_prime = env_u32("PS_DAAT_PRIME", daat_prime);
    daat_sample_all = env_u32("PS_DAAT_SAMPLE_ALL", daat_sample_all);
    daat_split = env_u32("PS_DAAT_SPLIT", daat_split);
    daat_small_nl = env_u32("PS_DAAT_SMALL_NL", daat_small_nl);
    dctx = std::max(2u, std::min((uint32_t)N_DCTX, env_u32("PS_DCTX", dctx)));
    daat_z_d0_div = env_u32("PS_DAAT_Z_D0_DIV", daat_z_d0_div);
    daat_z_level_shift = env_u32("PS_DAAT_Z_LEVEL_SHIFT", daat_z_level_shift);
    daat_z_levels = env_u32("PS_DAAT_Z_LEVELS", daat_z_levels);
    z21_field_prune = env_u32("PS_Z21_FIELD_PRUNE", z21_field_prune);
    z21_exact_numerator = env_u32("PS_Z21_EXACT_NUMERATOR", z21_exact_numerator);
}

namespace {

// Knobs changed through ps_set_option since this engine last read them take effect at the next batch.
void refresh_tuning(EngineImpl& m) {
  const uint64_t g = g_opt_gen.load(std::memory_order_relaxed);
  if (m.tune_gen == g) return;
  m.tune.load();
  m.tune_gen = g;
}

void validate(const Snapshot& s, const ps_scorer_desc& sc, const Plan& plan) {
  if (s.F > (uint32_t)MAX_F) throw std::length_error("the GPU path supports at most 8 fields");
  if (sc.kind != PS_SCORER_BM25 && sc.kind != PS_SCORER_ZERO_TO_ONE) throw std::invalid_argument("unknown scorer kind");
  if (plan.max_qterms >= 0x7FFF) throw std::length_error("more than 32766 non-empty terms in one query");
}

// Uploads the plan + per-query arrays through a pinned staging slot; fills the common KParams.
// The K1 instantiation a batch will run (same choice as launch_k_score), for occupancy queries.
const void* k_score_fn(bool bm25, uint32_t F, bool tags, bool full, bool wide) {
#define PS_FN4(M, FV, TG)                                                                             \
  (full ? reinterpret_cast<const void*>(&k_score<M, FV, TG, true, WG_WAVES>)                          \
        : wide ? reinterpret_cast<const void*>(&k_score<M, FV, TG, false, 8>)                         \
               : reinterpret_cast<const void*>(&k_score<M, FV, TG, false, WG_WAVES>))
#define PS_FN3(M, FV) (tags ? PS_FN4(M, FV, true) : PS_FN4(M, FV, false))
#define PS_FN2(M) (F == 1 ? PS_FN3(M, 1) : F == 2 ? PS_FN3(M, 2) : PS_FN3(M, 0))
  return bm25 ? PS_FN2(MODE_BM25) : PS_FN2(MODE_Z21S);
#undef PS_FN2
#undef PS_FN3
#undef PS_FN4
}

// Resident K1 waves per CU for a launch geometry (cached: the occupancy query is not free).
size_t k_score_waves_per_cu(EngineImpl& m, const void* fn, uint32_t wgw, size_t lds) {
  if (lds > 160 * 1024) return 0;
  auto key = std::make_pair(fn, lds);
  auto it = m.occ_cache.find(key);
  if (it != m.occ_cache.end()) return it->second;
  if (lds > 65536) PS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 0;
  PS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, WAVE * wgw, lds));
  // The runtime's answer covers registers and waves; for LDS it divides 160 KiB by the request,
  // but gfx950 hands LDS out in 1280-byte granules (160 KiB / 128): measured on C5, three 4-wave
  // workgroups of 53 632 B are resident per CU, three of 53 888 B are not.
  const size_t granule = 1280, rounded = (lds + granule - 1) / granule * granule;
  const size_t by_lds = rounded ? (160 * 1024) / rounded : (size_t)per_cu;
  const size_t w = std::min<size_t>((size_t)std::max(0, per_cu), by_lds) * wgw;
  m.occ_cache.emplace(key, w);
  return w;
}

// Workgroups of 8 waves share one LUT copy: two of them (16 waves) fit a CU's 160 KiB when a
// wave's tile is small enough; otherwise - or when the instantiation's registers allow fewer
// resident waves that way - 4-wave workgroups pack the CU better.
bool wide_workgroups(EngineImpl& m, bool bm25, uint32_t F, bool tags, bool full, size_t lut_b, size_t wave_b) {
  if (full || !m.tune.wg8 || lut_b + 8 * wave_b > 80 * 1024) return false;
  const size_t w8 = k_score_waves_per_cu(m, k_score_fn(bm25, F, tags, false, true), 8, lut_b + 8 * wave_b);
  const size_t w4 = k_score_waves_per_cu(m, k_score_fn(bm25, F, tags, false, false), WG_WAVES, lut_b + WG_WAVES * wave_b);
  return w8 >= w4;
}

// The staged batch: the pinned slot holds exactly what the device reads
// (entries | qbeg | qterms_len | qorder | zorder | qflags | gen_queries | row descriptors).
struct BatchImage {
  size_t B = 0, ne = 0;
  size_t off_e = 0, off_q = 0, off_l = 0, off_o = 0, off_z = 0, off_f = 0, off_g = 0, off_r = 0, total = 0;
  Stage* slot = nullptr;
  unsigned char* h = nullptr;    // the slot's host pointer
  ps_plan_entry* he = nullptr;   // entries as uploaded (zero_to_one simple queries: pre-sorted; dense flags set)
  bool z = false;                // zero_to_one
  uint32_t n_rows = 0;           // rows K0b has to score for this batch
  uint32_t n_used = 0;           // rows the batch reads (resident ones included)
  uint32_t n_simple = 0, n_general = 0, z_masked = 0;
  uint32_t z_qterms = 0;  // most query terms with entries in one general zero_to_one query
  size_t off_zf = 0;  // zero_to_one: per-query per-field pool bounds (k_score's field pruning)
};

}  // namespace
namespace {

// Claims the next pinned slot and copies the plan's arrays into it.
BatchImage lay_out_batch(EngineImpl& m, const ps_scorer_desc& sc, const Plan& plan) {
  BatchImage img;
  const size_t B = img.B = plan.qbeg.size() - 1;
  const size_t ne = img.ne = plan.entries.size();
  img.z = sc.kind == PS_SCORER_ZERO_TO_ONE;
  img.off_e = 0;
  img.off_q = img.off_e + ne * sizeof(ps_plan_entry);
  img.off_l = img.off_q + (B + 1) * 4;
  img.off_o = img.off_l + B * 4;
  img.off_z = img.off_o + B * 4;
  img.off_f = img.off_z + ne * 4;
  img.off_g = img.off_f + B * 4;
  img.off_r = (img.off_g + B * 4 + 15) & ~(size_t)15;
  img.total = img.off_r + (size_t)m.tune.dense_max_rows * sizeof(RowDesc);
  if (img.z) {
    img.off_zf = (img.total + 15) & ~(size_t)15;
    img.total = img.off_zf + B * m.snap->F * sizeof(double);
  }
  Stage& sg = m.stage[m.next_stage];
  m.next_stage = (m.next_stage + 1) % N_STAGE;
  sg.ensure(img.total + 16);
  img.slot = &sg;
  img.h = sg.p;
  img.he = reinterpret_cast<ps_plan_entry*>(img.h + img.off_e);
  if (ne) memcpy(img.he, plan.entries.data(), ne * sizeof(ps_plan_entry));
  memcpy(img.h + img.off_q, plan.qbeg.data(), (B + 1) * 4);
  if (B) memcpy(img.h + img.off_l, plan.qterms_len.data(), B * 4);
  return img;
}

// ---- K1d: bounds and work descriptors ------------------------------------------------------------
bool bm25_params_sane(const Snapshot& s, const ps_scorer_desc& sc, const double* boosts) {
  bool sane = s.n_docs > 0 && std::isfinite(sc.bm25_k1) && sc.bm25_k1 >= 0.0 && sc.bm25_b >= 0.0 && sc.bm25_b <= 1.0;
  for (uint32_t x = 0; x < s.F && sane; ++x)
    sane = std::isfinite(boosts[x]) && boosts[x] > 0.0 && std::isfinite(s.avg[x]) && s.avg[x] > 0.0;
  return sane;
}

void ensure_dev_trie(EngineImpl& m);  // (defined with the device planner below)

// Per-list score bounds on the device (k_list_bounds) for the current (k1, b, avg) and boosts; see
// EngineImpl::ListBounds.  Enqueued on `st` in front of the batch that needs them; nothing is recomputed
// while the parameters stay what they were, and a boost vector seen recently finds its J array resident.
struct BoundsRef { const double* M; const double* J; const double* plane; const double* H; uint32_t h_lo; double h_a, h_b; const double* kth; };

// The work list of the per-list passes (k_list_bounds, k_list_kth, k_removed_df): one wave per list, long lists in 16 Ki-posting
// segments; rebuilt when the snapshot's layers change (a delta appended some).
void ensure_bound_units(EngineImpl& m) {
  const Snapshot& s = *m.snap;
  EngineImpl::ListBounds& lb = m.bounds;
  const size_t nl = s.layers.size();
  auto sig_of = [&](size_t n) {
    uint64_t sig = n;
    for (size_t l = 0; l < n; ++l) sig = sig * 0x9E3779B97F4A7C15ull + s.layers[l].len;
    return sig;
  };
  auto add_layer = [&](std::vector<BoundUnit>& units, size_t l) {
    for (uint32_t b0 = 0; b0 < s.layers[l].len; b0 += 16384u)
      units.push_back(BoundUnit{(uint32_t)l, b0, std::min<uint32_t>(16384u, s.layers[l].len - b0)});
  };
  if (lb.units_n != 0 && lb.units_layers <= nl && sig_of(lb.units_layers) == lb.units_sig) {
    if (lb.units_layers == nl) { lb.n_units = lb.units_n; return; }
    // a delta appended layers (the lists it touched): their units go behind the existing ones - no re-sort, no re-upload of the rest
    std::vector<BoundUnit> extra;
    for (size_t l = lb.units_layers; l < nl; ++l) add_layer(extra, l);
    lb.units.ensure_keep(lb.units_n + extra.size() + 1, lb.units_n);
    if (!extra.empty()) PS_HIP(hipMemcpy(lb.units.p + lb.units_n, extra.data(), extra.size() * sizeof(BoundUnit), hipMemcpyHostToDevice));
    lb.units_n += (uint32_t)extra.size();
  } else {
    std::vector<BoundUnit> units;
    units.reserve(nl + s.n_postings / 16384 + 1);
    for (size_t l = 0; l < nl; ++l) add_layer(units, l);
    // longest segments first, so a launch ends on its cheapest waves
    std::stable_sort(units.begin(), units.end(), [](const BoundUnit& a, const BoundUnit& b) { return a.count > b.count; });
    lb.units.ensure(units.size() + 1);
    if (!units.empty()) PS_HIP(hipMemcpy(lb.units.p, units.data(), units.size() * sizeof(BoundUnit), hipMemcpyHostToDevice));
    lb.units_n = (uint32_t)units.size();
  }
  lb.n_units = lb.units_n;
  lb.units_layers = nl;
  lb.units_sig = sig_of(nl);
}

static_assert(BOUND_NDIR == PREP_NDIR, "host and device agree on the directions of the two-field joint bound");
BoundsRef ensure_list_bounds(EngineImpl& m, const ps_scorer_desc& sc, const double* boosts, const KParams& kp, hipStream_t st) {
  const Snapshot& s = *m.snap;
  EngineImpl::ListBounds& lb = m.bounds;
  ensure_dev_trie(m);  // the per-layer table (post_off, len) lives with the device planner's tables
  const size_t nl = s.layers.size(), F = s.F;
  const std::vector<double> avg(s.avg.begin(), s.avg.end()), bv(boosts, boosts + F);
  const bool m_ok = lb.m_valid && lb.k1 == sc.bm25_k1 && lb.b == sc.bm25_b && lb.avg == avg && lb.n_layers == nl;
  if (!m_ok)
    for (auto& js : lb.j) js.valid = false;
  ++lb.epoch;
  const bool need_j = F >= 3;  // (one field: M is the joint maximum; two: the direction supports H give it for any boosts)
  EngineImpl::ListBounds::JSet* tgt = nullptr;
  if (need_j)
    for (auto& js : lb.j)
      if (js.valid && js.boosts == bv) tgt = &js;
  BoundsRef ref{nullptr, nullptr, nullptr, nullptr, 0u, 0.0, 0.0, nullptr};
  if (F == 2) boost_cone(boosts, ref.h_lo, ref.h_a, ref.h_b);
  auto fill = [&]() {
    ref.M = reinterpret_cast<const double*>(lb.M.p);
    ref.J = need_j ? reinterpret_cast<const double*>(tgt->J.p) : nullptr;
    ref.plane = lb.plane.p;
    ref.H = F == 2 ? reinterpret_cast<const double*>(lb.H.p) : nullptr;
    ref.kth = F <= 2 ? reinterpret_cast<const double*>(lb.kth.p) : nullptr;
    return ref;
  };
  if (m_ok && (!need_j || tgt)) {
    if (tgt) tgt->last_use = lb.epoch;
    // (they may have been computed on another stream a moment ago)
    PS_HIP(hipStreamWaitEvent(st, lb.m_ready, 0));
    if (tgt) PS_HIP(hipStreamWaitEvent(st, tgt->ready, 0));
    return fill();
  }
  // M, H and a recycled J array are only read by the preparation kernels, which run on this same stream, in order.  The
  // score PLANE is read by k_daat / k_daat_small on the scoring streams, by batches that may still be in flight: when the
  // scorer parameters, the averages or the lists changed (never for a new fields_boost) nothing is overwritten before every
  // batch enqueued so far has left its scoring kernel.
  if (!m_ok)
    for (auto& c : m.dctx)
      if (c.busy && c.scored) PS_HIP(hipStreamWaitEvent(st, c.scored, 0));
  const double t0 = now_ms();
  ensure_bound_units(m);
  if (need_j && !tgt) {  // an invalid slot, else the least recently used one
    for (auto& js : lb.j)
      if (!tgt || (!js.valid && tgt->valid) || (js.valid == tgt->valid && js.last_use < tgt->last_use)) tgt = &js;
    if (!tgt->ready) PS_HIP(hipEventCreateWithFlags(&tgt->ready, hipEventDisableTiming));
  }
  if (!lb.dirs.p) {
    std::vector<double2> dv(PREP_NDIR);
    for (int d = 0; d < PREP_NDIR; ++d) bound_dir(d, dv[d].x, dv[d].y);
    lb.dirs.ensure(PREP_NDIR);
    PS_HIP(hipMemcpy(lb.dirs.p, dv.data(), sizeof(double2) * PREP_NDIR, hipMemcpyHostToDevice));
  }
  lb.M.ensure(nl * F + 1);
  if (F == 2) lb.H.ensure(nl * PREP_NDIR + 1);
  if (tgt) tgt->J.ensure(nl + 1);
  lb.plane.ensure((size_t)s.P * F + 2);
  if (!m_ok) {
    PS_HIP(hipMemsetAsync(lb.M.p, 0, (nl * F + 1) * 8, st));
    if (F == 2) PS_HIP(hipMemsetAsync(lb.H.p, 0, (nl * PREP_NDIR + 1) * 8, st));
  }
  if (tgt) PS_HIP(hipMemsetAsync(tgt->J.p, 0, (nl + 1) * 8, st));
  if (lb.n_units) {
    hipLaunchKernelGGL(k_list_bounds, dim3((lb.n_units + 3) / 4), dim3(256), 0, st, kp, lb.units.p, lb.n_units, m.d_layer_a.p, lb.M.p,
                       tgt ? tgt->J.p : nullptr, (!m_ok && F == 2) ? lb.H.p : nullptr, lb.dirs.p, m_ok ? 0 : 1,
                       m_ok ? nullptr : lb.plane.p, m.d_layer_idf.p);
    PS_HIP(hipGetLastError());
  }
  if (!m_ok && F <= 2) {  // threshold priming tables from the plane just written (same stream, in order)
    const size_t nk = nl * (F == 1 ? 1 : 3) * KTH_RANKS + 1;
    lb.kth.ensure(nk);
    PS_HIP(hipMemsetAsync(lb.kth.p, 0, nk * 8, st));
    if (lb.n_units) {
      if (F == 1) hipLaunchKernelGGL(k_list_kth<1>, dim3((lb.n_units + 3) / 4), dim3(256), 0, st, lb.units.p, lb.n_units, m.d_layer_a.p, lb.plane.p, m.d_doc, kp.alive, lb.kth.p);
      else hipLaunchKernelGGL(k_list_kth<2>, dim3((lb.n_units + 3) / 4), dim3(256), 0, st, lb.units.p, lb.n_units, m.d_layer_a.p, lb.plane.p, m.d_doc, kp.alive, lb.kth.p);
      PS_HIP(hipGetLastError());
    }
  }
  if (!m_ok) PS_HIP(hipEventRecord(lb.m_ready, st)); else PS_HIP(hipStreamWaitEvent(st, lb.m_ready, 0));
  if (tgt) {
    PS_HIP(hipEventRecord(tgt->ready, st));
    tgt->valid = true; tgt->boosts = bv; tgt->last_use = lb.epoch;
  }
  lb.m_valid = true; lb.k1 = sc.bm25_k1; lb.b = sc.bm25_b; lb.avg = avg; lb.n_layers = nl;
  lb.last_ms = now_ms() - t0;
  ++lb.recomputed;
  return fill();
}

// Bloom filters of the lists that have no membership bitmap (fewer than N / 128 postings): 16 bits per posting,
// three bits per document in one 64-bit word, a power-of-two number of words per list.  K1d asks them before it
// touches a sparse list's table: a filter of a few KB stays in L2, and nearly every answer is "no".
void ensure_bloom(EngineImpl& m, hipStream_t st) {
  if (m.bloom_valid) return;
  const Snapshot& s = *m.snap;
  ensure_dev_trie(m);
  const size_t nl = s.layers.size();
  std::vector<unsigned long long> desc(std::max<size_t>(nl, 1), NO_BLOOM);
  uint64_t words = 0;
  uint32_t id_bits = 0;  // bits of the largest doc id a delta can still append
  while (id_bits < 32 && ((uint64_t)s.tiles_cap * s.T - 1) >> id_bits) ++id_bits;
  for (size_t l = 0; l < nl; ++l) {
    const LayerInfo& L = s.layers[l];
    if (L.bm_off != NO_BITMAP || L.len == 0) continue;
    uint64_t nw = 1;
    uint32_t lg = 0;
    while (nw * 64 < (uint64_t)L.len * BLOOM_BITS_PER_KEY) { nw <<= 1; ++lg; }
    // doc-ordered layout: word = doc id >> shift, every id of the snapshot's id space (headroom included) below nw
    const uint32_t shift = id_bits > lg ? id_bits - lg : 0u;
    desc[l] = words | ((unsigned long long)shift << 40) | ((unsigned long long)lg << 58);
    words += nw;
  }
  if (words >= (1ull << 40)) throw std::length_error("Bloom filters: more than 2^40 words");
  m.d_layer_bloom.ensure(desc.size() + 1);
  m.d_bloom.ensure(words + 2);
  PS_HIP(hipMemcpy(m.d_layer_bloom.p, desc.data(), desc.size() * 8, hipMemcpyHostToDevice));
  PS_HIP(hipMemsetAsync(m.d_bloom.p, 0, (words + 1) * 8, st));
  if (nl) {
    hipLaunchKernelGGL(k_build_bloom, dim3((uint32_t)((nl + 3) / 4)), dim3(256), 0, st, m.d_doc, m.d_layer_a.p, m.d_layer_bloom.p, (uint32_t)nl,
                       m.d_bloom.p);
    PS_HIP(hipGetLastError());
  }
  m.bloom_words = words;
  m.bloom_valid = true;
}

// K1d dense-row candidates: the snapshot's densest lists with one table slot per tile, longest first,
// each with a fixed row slot.  Chosen on the host when the snapshot's layers or the knobs change; which
// candidates a batch reads, and with which weights, is decided on the device (k_prep_finish).
void ensure_row_candidates(EngineImpl& m, hipStream_t st) {
  const Snapshot& s = *m.snap;
  EngineImpl::RowCands& rc = m.cands;
  const uint32_t pct = std::max(m.tune.dense_min_density_pct, m.tune.daat_dense_min_density_pct);
  const std::vector<double> sig{(double)m.tune.daat_rows, (double)pct, (double)m.tune.dense_max_rows, (double)m.tune.dense_max_mb,
                                (double)s.layers.size(), (double)s.n_ids, (double)s.tiles_cap, (double)s.n_postings};
  if (rc.valid && rc.sig == sig) return;
  const size_t nl = s.layers.size();
  std::vector<std::pair<uint32_t, uint32_t>> c;  // (len, layer)
  if (m.tune.daat_rows && m.tune.dense_max_rows && s.n_ids > 0)
    for (size_t l = 0; l < nl; ++l) {
      const LayerInfo& L = s.layers[l];
      if (L.shift == 0 && L.len && (double)L.len >= (pct / 100.0) * (double)s.n_ids) c.emplace_back(L.len, (uint32_t)l);
    }
  std::stable_sort(c.begin(), c.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
  const uint64_t row_bytes = (uint64_t)s.tiles_cap * s.T * 8;
  size_t cap = std::min<size_t>(std::min<uint32_t>(m.tune.dense_max_rows, PREP_MAX_ROWS), c.size());
  while (cap && cap * row_bytes > ((uint64_t)m.tune.dense_max_mb << 20)) --cap;
  c.resize(cap);
  std::vector<uint8_t> of(std::max<size_t>(nl, 1), (uint8_t)NO_CAND);
  for (size_t k = 0; k < c.size(); ++k) of[c[k].second] = (uint8_t)k;
  rc.of_layer.ensure(of.size() + 1);
  PS_HIP(hipMemcpyAsync(rc.of_layer.p, of.data(), of.size(), hipMemcpyHostToDevice, st));
  PS_HIP(hipStreamSynchronize(st));  // (`of` is pageable and leaves scope; this happens once per snapshot state)
  rc.n = (uint32_t)c.size();
  rc.sig = sig;
  ++rc.gen;  // every context drops its resident rows
  rc.valid = true;
}

// A preparation kernel found the device's item counts above the ones the host sized a scoring launch from (the "which kernel takes
// this query" rule lives in the planner's count pass, in k_plan and in k_prep_query): some batch since the last check may have
// skipped items.  Reported at the next submission / result fetch as an internal error instead of a silently wrong top-k.
void check_device_fault(EngineImpl& m) {
  if (!m.h_fault) return;
  const volatile uint32_t* f = m.h_fault;
  if (f[0] == 0u) return;
  const uint32_t code = f[0], n = f[1], split = f[2], host = f[3];
  m.h_fault[0] = 0u;
  throw std::logic_error("internal: K1d preparation and host disagree on a batch's work items (code " + std::to_string(code) + ": device " +
                         std::to_string(n) + " items, split at " + std::to_string(split) + ", host " + std::to_string(host) +
                         "); an earlier batch's results may be incomplete");
}

// The device-side preparation of a K1d batch (ps_prep_kernels.hpp) over a plan that already lives in
// HBM - uploaded by the host planner or written by the device planner: bounds -> descriptors / order /
// candidate slots -> items and dense-row flags.  `items_bound` >= the batch's item count (exact when the
// host knows the list lengths).  Fills the K1d members of `kp`.
void launch_prep(EngineImpl& m, EngineImpl::DaatCtx& c, const ps_scorer_desc& sc, const double* boosts, KParams& kp, ps_plan_entry* d_plan,
                 const uint32_t* d_qbeg, size_t B, size_t ne, bool multi, size_t items_bound, bool split_kinds, size_t items_big = 0) {
  const Snapshot& s = *m.snap;
  hipStream_t st = m.prep_stream;
  const uint64_t rc0 = m.bounds.recomputed;
  const BoundsRef br = ensure_list_bounds(m, sc, boosts, kp, st);
  m.last_bounds_recomputed = m.bounds.recomputed != rc0;
  ensure_row_candidates(m, st);
  ensure_bloom(m, st);
  kp.bloom = m.d_bloom.p;
  kp.layer_bloom = m.d_layer_bloom.p;
  // the context's resident rows are only valid for the candidates and parameters they were scored with
  {
    std::vector<double> sig{sc.bm25_k1, sc.bm25_b};
    for (uint32_t x = 0; x < s.F; ++x) { sig.push_back(boosts[x]); sig.push_back(s.avg[x]); }
    if (sig != c.row_sig || c.cands_gen != m.cands.gen) {
      PS_HIP(hipMemsetAsync(c.row_state, 0, sizeof(RowState) * PREP_MAX_ROWS, st));
      c.row_sig = sig;
      c.cands_gen = m.cands.gen;
    }
    c.rows.ensure(std::max<size_t>(1, m.cands.n) * (size_t)s.tiles_cap * s.T + 16);
  }
  c.dentry.ensure(ne + 1); c.rorder.ensure(ne + 1); c.gord.ensure(ne + 16); c.gen.ensure(ne + 1);
  if (multi) c.dgroup.ensure(ne + 1);
  c.qslot.ensure(B + 1); c.qslot_n.ensure(B + 1);
  c.ditems.ensure(items_bound + 1);
  c.cand_cnt.ensure(items_bound + 1);
  PrepParams pp;
  memset(&pp, 0, sizeof(pp));
  pp.plan = d_plan; pp.qbeg = d_qbeg;
  pp.B = (uint32_t)B; pp.ne = (uint32_t)ne; pp.F = s.F; pp.multi = multi ? 1u : 0u;
  pp.chunk_min = m.tune.daat_chunk; pp.split_div = m.tune.daat_split_div;
  pp.table = m.d_table;
  pp.split_kinds = split_kinds ? 1u : 0u;
  pp.sample_small = m.tune.daat_sample_all;
  if (m.tune.daat_sample_div > 1 && (multi || m.tune.daat_sample_all) && s.n_ids >= 4096) {  // the sample boundary D0 ~ N / div, on a tile boundary that every table of this snapshot resolves as finely as it can
    const uint32_t tiles = std::max<uint32_t>(1u, (uint32_t)((s.n_ids / m.tune.daat_sample_div) / s.T));
    // lists of >= one chunk have one table slot per tile (shift 0); a multiple of 8 tiles also serves the tables up to 8 x coarser
    uint32_t al = 1;
    while (al * 2 <= tiles) al *= 2;
    pp.sample_tile = tiles >= 16 ? (tiles & ~7u) : al;
  }
  for (uint32_t x = 0; x < s.F; ++x) pp.boost[x] = boosts[x];
  pp.bound_m = br.M; pp.bound_j = br.J; pp.bound_h = br.H; pp.h_lo = br.h_lo; pp.h_a = br.h_a; pp.h_b = br.h_b;
  kp.splane = br.plane;
  pp.dentry = c.dentry.p; pp.rorder = c.rorder.p; pp.dgroup = multi ? c.dgroup.p : nullptr; pp.gord = c.gord.p;
  pp.gen = c.gen.p; pp.qslot = c.qslot.p; pp.qslot_n = c.qslot_n.p;
  pp.items = c.ditems.p; pp.items_cap = (uint32_t)items_bound;
  pp.ctl = c.ctl;
  pp.cand_of_layer = m.cands.of_layer.p; pp.n_cand = m.cands.n; pp.min_uses = std::max(1u, m.tune.dense_min_uses);
  pp.rows_resident = m.tune.row_cache_mb != 0;
  pp.layer_a = m.d_layer_a.p; pp.row_state = c.row_state; pp.row_desc = c.row_desc; pp.wstats = m.d_wstats;
  pp.host_items = (uint32_t)items_bound; pp.host_items_big = (uint32_t)items_big; pp.fault = m.d_fault;
  // threshold priming: off for K beyond the stored ranks and for three fields and more (no tables).  Tombstones: the tables skip
  // removed documents (k_list_kth reads the alive bitmap; every delta recomputes them with the bounds)
  pp.gthr = kp.gthr;
  pp.kth = nullptr;
  if (m.tune.daat_prime && br.kth != nullptr && kp.K >= 1 && kp.K <= 64) {
    static const uint32_t ranks[KTH_RANKS] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 40, 50, 64};
    uint32_t j = 0;
    while (ranks[j] < kp.K) ++j;
    pp.kth = br.kth;
    pp.kth_rank = j;
    pp.prime_keep_items = m.tune.daat_prime == 2 ? 1u : 0u;
  }
  if (B) {
    hipLaunchKernelGGL(k_prep_query, dim3((uint32_t)((B + WAVE - 1) / WAVE)), dim3(WAVE), 0, st, pp);
    if (ne) hipLaunchKernelGGL(k_prep_items, dim3((uint32_t)((ne + 2 * WAVE - 1) / (2 * WAVE))), dim3(2 * WAVE), 0, st, pp);
    PS_HIP(hipGetLastError());
  }
  kp.dentry = c.dentry.p; kp.ditems = c.ditems.p; kp.qslot = c.qslot.p; kp.qslot_n = c.qslot_n.p;
  kp.rorder = c.rorder.p; kp.dgroup = multi ? c.dgroup.p : nullptr;
  kp.n_ditems = (uint32_t)items_bound;
  kp.n_ditems_dev = &c.ctl->n_items;
  kp.prep_ctl = reinterpret_cast<uint32_t*>(c.ctl);
  kp.prep_ctl_words = (uint32_t)(sizeof(PrepCtl) / 4);
  kp.cand_cnt = c.cand_cnt.p;
  kp.rows = c.rows.p; kp.row_desc = c.row_desc; kp.n_rows = 0;
  kp.row_planes = 1; kp.row_mode = 0; kp.row_stride = (uint64_t)s.tiles_cap * s.T;
}

// Drops a batch that was announced ahead (Engine::plan_ahead) and never asked for: its context goes back into the
// rotation behind its count pass.
void drop_ahead(EngineImpl& m) {
  for (const auto& a : m.aheads) {
    EngineImpl::DaatCtx& c = m.dctx[a.ctx];
    PS_HIP(hipEventRecord(c.done, m.plan_stream));  // (whoever gets the context next waits for the abandoned count pass)
    c.busy = true;
  }
  m.aheads.clear();
}

// The next K1d batch context (they alternate); the two streams, its events and control blocks exist from first use.
EngineImpl::DaatCtx& acquire_ctx(EngineImpl& m) {
  for (const auto& a : m.aheads)
    if (a.ctx == m.next_dctx) { drop_ahead(m); break; }  // (the rotation came round to an announced batch nobody asked for)
  if (!m.prep_stream) {
    // the preparation stream at the highest priority: a queue of its own (queues are pooled per priority),
    // and its small kernels do not wait behind the tens of thousands of workgroups of a running k_daat
    int lo = 0, hi = 0;
    PS_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    PS_HIP(hipStreamCreateWithPriority(&m.prep_stream, hipStreamNonBlocking, hi));
    // the planner's count pass (text upload, k_plan count, k_plan_scan) has a stream of its own: the host waits
    // for its totals, and that wait should not sit behind the previous batch's preparation kernels
    PS_HIP(hipStreamCreateWithPriority(&m.plan_stream, hipStreamNonBlocking, hi));
    PS_HIP(hipStreamCreateWithFlags(&m.score_stream, hipStreamNonBlocking));
    PS_HIP(hipStreamCreateWithPriority(&m.score_stream_lo, hipStreamNonBlocking, lo));
    PS_HIP(hipStreamCreateWithFlags(&m.score_stream_b, hipStreamNonBlocking));
    PS_HIP(hipStreamCreateWithPriority(&m.score_stream_hi, hipStreamNonBlocking, hi));
    // the merge of batch s (the kernel that waits for the caller's stream) off the scoring stream: k_daat of
    // batch s + 1 starts the moment k_daat of batch s ends
    PS_HIP(hipStreamCreateWithPriority(&m.merge_stream, hipStreamNonBlocking, hi));
  }
  EngineImpl::DaatCtx& c = m.dctx[m.next_dctx];
  m.next_dctx = (m.next_dctx + 1) % (int)m.tune.dctx;
  if (!c.done) {
    PS_HIP(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
    PS_HIP(hipEventCreateWithFlags(&c.entry, hipEventDisableTiming));
    PS_HIP(hipEventCreateWithFlags(&c.prepared, hipEventDisableTiming));
    PS_HIP(hipEventCreateWithFlags(&c.counted, hipEventDisableTiming));
    PS_HIP(hipEventCreateWithFlags(&c.scored, hipEventDisableTiming));
    PS_HIP(hipEventCreateWithFlags(&c.scored2, hipEventDisableTiming));
    PS_HIP(hipEventCreateWithFlags(&c.plan.h.done, hipEventDisableTiming));
    PS_HIP(hipMalloc((void**)&c.ctl, sizeof(PrepCtl)));
    PS_HIP(hipMalloc((void**)&c.work, 256));
    PS_HIP(hipMalloc((void**)&c.row_state, sizeof(RowState) * PREP_MAX_ROWS));
    PS_HIP(hipMalloc((void**)&c.row_desc, sizeof(RowDesc) * PREP_MAX_ROWS));
    PS_HIP(hipMemset(c.row_state, 0, sizeof(RowState) * PREP_MAX_ROWS));
    c.ctl_clean = false;
  }
  // the batch that used this context last must be through before its buffers are written again
  if (c.busy) { PS_HIP(hipStreamWaitEvent(m.prep_stream, c.done, 0)); PS_HIP(hipStreamWaitEvent(m.plan_stream, c.done, 0)); }
  if (m.tail_pending) {  // (a k_score / full-result batch still in flight)
    PS_HIP(hipStreamWaitEvent(m.prep_stream, m.ev[0], 0));
    PS_HIP(hipStreamWaitEvent(m.plan_stream, m.ev[0], 0));
  }
  return c;
}

// Batches that do not run in a K1d context (k_score, zero_to_one, full-result mode) share the engine's
// single set of per-batch buffers and the saturated-tf table with nothing else in flight.
void wait_daat_contexts(EngineImpl& m, hipStream_t st) {
  for (auto& c : m.dctx)
    if (c.busy) PS_HIP(hipStreamWaitEvent(st, c.done, 0));
}

// Items a batch has under the chunking rule of k_prep_query (host-planned batches: exact).
size_t count_daat_items(const EngineImpl& m, const Plan& plan, uint32_t* max_slots, size_t* n_big = nullptr) {
  size_t n = 0, nb = 0;
  uint32_t mx = 0;
  const size_t B = plan.qbeg.size() - 1;
  auto chunks = [&](uint32_t len, uint32_t div) {
    const uint32_t c = std::max<uint32_t>(m.tune.daat_chunk, ((len + div - 1) / div + 255) & ~255u);
    return (len + c - 1) / c;
  };
  for (size_t q = 0; q < B; ++q) {
    uint32_t sl = 0;
    bool big = plan.qbeg[q + 1] - plan.qbeg[q] > (uint32_t)DAAT_SMALL_MAX;  // PLAN_BIG's rule (k_plan, k_prep_query)
    for (uint32_t i = plan.qbeg[q]; i < plan.qbeg[q + 1]; ++i) {
      const uint32_t len = plan.entries[i].len;
      sl += chunks(len, m.tune.daat_split_div);
      if (i > plan.qbeg[q] && plan.entries[i].qterm == plan.entries[i - 1].qterm) big = true;
    }
    n += sl;
    if (big) nb += sl;
    mx = std::max(mx, sl);
  }
  if (max_slots) *max_slots = mx;
  if (n_big) *n_big = nb;
  return n;
}

// zero_to_one: per (layer, field) the shortest field length among the postings that hold the term in
// that field (0xFFFFFFFF = none does).  A record contributes (min(score/tf, 1) * tf) / max(field_length,
// query_terms_len) <= score * (1 + 1e-12) / max(that minimum, query_terms_len).  Scorer-independent:
// computed once per snapshot (incrementally for the layers a delta appended).
void compute_z_bounds(EngineImpl& m) {
  const Snapshot& s = *m.snap;
  const size_t nl = s.layers.size(), F = s.F;
  const size_t first = m.z_minfl.size() / std::max<size_t>(F, 1);
  if (first == nl) return;
  m.z_minfl.resize(nl * F, 0xFFFFFFFFu);
  m.z_maxtf.resize(nl, 0u);
  auto body = [&](size_t l) {
    const LayerInfo& L = s.layers[l];
    uint32_t mt = 0;
    for (size_t x = 0; x < F; ++x) {
      uint32_t mn = 0xFFFFFFFFu;
      const uint32_t* tf = s.tf.data() + x * s.P + L.post_off;
      const uint32_t* fl = s.fl.data() + x * s.P + L.post_off;
      for (uint32_t i = 0; i < L.len; ++i) {
        if (tf[i] && fl[i] < mn) mn = fl[i];
        if (tf[i] > mt) mt = tf[i];
      }
      m.z_minfl[l * F + x] = mn;
    }
    m.z_maxtf[l] = mt;
  };
  const unsigned n_thr = s.n_postings > (1u << 20) ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
  if (n_thr <= 1 || nl - first < 1024) {
    for (size_t l = first; l < nl; ++l) body(l);
  } else {
    std::atomic<size_t> next{first};
    auto worker = [&]() {
      for (;;) {
        const size_t b0 = next.fetch_add(64);
        if (b0 >= nl) break;
        for (size_t l = b0; l < std::min(nl, b0 + 64); ++l) body(l);
      }
    };
    std::vector<std::thread> th;
    for (unsigned i = 1; i < n_thr; ++i) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
  }
}

// qorder: K1 hands out the items of a run heaviest query first (longest-processing-time order).
// The last items a launch hands out are then its cheapest ones, which shortens the tail where
// most waves have run dry while a few still chew on a head-term query.
void order_queries(const EngineImpl& m, const Plan& plan, BatchImage& img) {
  const size_t B = img.B;
  uint32_t* qo = reinterpret_cast<uint32_t*>(img.h + img.off_o);
  std::vector<uint64_t> cost(B, 0);
  for (size_t q = 0; q < B; ++q) {
    uint64_t c = 0;
    for (uint32_t i = plan.qbeg[q]; i < plan.qbeg[q + 1]; ++i) c += plan.entries[i].len;
    cost[q] = c;
    qo[q] = (uint32_t)q;
  }
  if (m.tune.lpt) std::stable_sort(qo, qo + B, [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
}

// zero_to_one: classifies every query (simple / masked / general) and pre-sorts the entries of the
// simple ones into the record-sort order.
void classify_zero_to_one(const EngineImpl& m, const Plan& plan, BatchImage& img) {
  const Snapshot& s = *m.snap;
  const size_t B = img.B;
  unsigned char* h = img.h;
  ps_plan_entry* he = img.he;
  const size_t off_z = img.off_z, off_f = img.off_f, off_g = img.off_g;
  uint32_t& n_simple = img.n_simple;
  uint32_t& n_general = img.n_general;
  uint32_t& z_masked = img.z_masked;
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
    if (m.tune.z21_general_only) simple = false;
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
      // k_z21 keeps consumed_index as one bit per query term THAT HAS ENTRIES: renumber the terms densely
      // (plan.qterm counts every non-empty token, also those that matched nothing)
      uint32_t dense = 0, prev = 0xFFFFFFFFu;
      for (uint32_t i = b; i < e; ++i) {
        if (plan.entries[i].qterm != prev) { prev = plan.entries[i].qterm; ++dense; }
        he[i].qterm = dense - 1;
      }
      img.z_qterms = std::max(img.z_qterms, dense);
    }
  }
}

// ---- hot dense lists (see k_dense_rows) -------------------------------------------------------
// BM25: key = (list, idf, expansion_boost).  zero_to_one (simple queries only): key = (list,
// ScoreByTerm::score, all_query_terms_len); a row then holds one plane per field.
void select_dense_rows(EngineImpl& m, const ps_scorer_desc& sc, const double* boosts, const Plan& plan,
                       BatchImage& img) {
  const Snapshot& s = *m.snap;
  const size_t B = img.B, ne = img.ne;
  unsigned char* h = img.h;
  ps_plan_entry* he = img.he;
  const bool z = img.z;
  const size_t off_f = img.off_f, off_r = img.off_r;
  const size_t max_rows = m.tune.dense_max_rows;
  uint32_t& n_rows = img.n_rows;
  uint32_t& n_used = img.n_used;
  const uint32_t z_masked = img.z_masked;
  {
    bool sane = max_rows > 0 && s.n_docs > 0;
    if (!z) {
      sane = sane && std::isfinite(sc.bm25_k1) && sc.bm25_k1 >= 0.0 && sc.bm25_b >= 0.0 && sc.bm25_b <= 1.0;
      for (uint32_t x = 0; x < s.F && sane; ++x)
        sane = std::isfinite(boosts[x]) && boosts[x] > 0.0 && std::isfinite(s.avg[x]) && s.avg[x] > 0.0;
    }
    const uint32_t min_uses = m.tune.dense_min_uses;
    // (K1d only looks rows up: below ~40 % density the bitmap cell + posting is as good as the row costs to build)
    const double min_density = m.tune.dense_min_density_pct / 100.0;
    const uint32_t planes = z ? s.F : 1u;
    if (sane && ne) {
      struct Key { uint64_t post_off, w, k3; };
      struct Agg { uint32_t uses, len, tbl_off, shift; };
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
          if ((double)he[i].len < min_density * (double)s.n_ids) continue;
          Agg& a = agg[key_of(he[i], q)];
          a.uses++;
          a.len = he[i].len;
          a.tbl_off = he[i].tbl_off;
          a.shift = he[i].shift & 0xFFu;
        }
      }
      std::vector<std::pair<uint64_t, Key>> hot;  // (saved posting visits, key)
      for (auto& kv : agg)
        if (kv.second.uses >= min_uses) hot.emplace_back((uint64_t)kv.second.uses * kv.second.len, kv.first);
      std::sort(hot.begin(), hot.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
      const uint64_t row_bytes = (uint64_t)s.tiles_cap * s.T * 8 * planes;
      const uint64_t mem_cap = (uint64_t)m.tune.dense_max_mb << 20;
      while (hot.size() > max_rows || hot.size() * row_bytes > mem_cap) hot.pop_back();
      // slab geometry: as many slots as the cache budget holds (at least one batch's worth)
      const uint64_t row_elems = (uint64_t)s.tiles_cap * s.T * planes;
      size_t n_slots = std::max<size_t>(max_rows, std::min<size_t>(4096, ((uint64_t)m.tune.row_cache_mb << 20) / row_bytes));
      if (!hot.empty()) {
        // resident rows are only valid for the parameters they were scored with
        std::vector<double> sig{(double)sc.kind, z ? 0.0 : sc.bm25_k1, z ? 0.0 : sc.bm25_b, (double)n_slots};
        for (uint32_t x = 0; x < s.F; ++x) sig.push_back(z ? 0.0 : boosts[x]);
        if (sig != m.row_sig || m.row_slots.size() != n_slots || m.tune.row_cache_mb == 0) {
          m.row_sig = sig;
          m.row_slot_of.clear();
          m.row_slots.assign(n_slots, EngineImpl::RowSlot{});
        }
        try {
          m.d_rows.ensure((size_t)n_slots * row_elems + 16);
        } catch (const std::runtime_error&) {
          // not enough free HBM for the whole slab: keep one batch's worth of slots
          (void)hipGetLastError();
          m.tune.row_cache_mb = 1;  // and stop asking for the big slab
          n_slots = std::max<size_t>(max_rows, std::min<size_t>(4096, ((uint64_t)1 << 20) / row_bytes));
          m.row_sig[3] = (double)n_slots;
          m.row_slot_of.clear();
          m.row_slots.assign(n_slots, EngineImpl::RowSlot{});
          m.d_rows.ensure((size_t)n_slots * row_elems + 16);
        }
        const uint64_t epoch = ++m.row_epoch;
        RowDesc* rd = reinterpret_cast<RowDesc*>(h + off_r);
        std::map<Key, uint32_t, decltype(kless)> row_of(kless);
        // first pass: pin the resident rows this batch uses, so the misses cannot evict them
        for (auto& hk : hot) {
          auto it = m.row_slot_of.find(EngineImpl::RowKey{hk.second.post_off, hk.second.w, hk.second.k3});
          if (it != m.row_slot_of.end()) m.row_slots[it->second].last_use = epoch;
        }
        size_t victim = 0;
        for (auto& hk : hot) {
          const EngineImpl::RowKey ck{hk.second.post_off, hk.second.w, hk.second.k3};
          auto hit = m.row_slot_of.find(ck);
          if (hit != m.row_slot_of.end()) {
            row_of[hk.second] = hit->second;
            ++n_used;
            continue;
          }
          // miss: the least recently used slot no row of this batch lives in
          uint32_t slot = 0xFFFFFFFFu;
          uint64_t oldest = ~0ull;
          for (size_t k = 0; k < n_slots; ++k) {
            const size_t j = (victim + k) % n_slots;
            if (!m.row_slots[j].valid) { slot = (uint32_t)j; break; }
            if (m.row_slots[j].last_use < epoch && m.row_slots[j].last_use < oldest) { oldest = m.row_slots[j].last_use; slot = (uint32_t)j; }
          }
          if (slot == 0xFFFFFFFFu) continue;  // cannot happen: n_slots >= rows per batch
          victim = slot + 1;
          if (m.row_slots[slot].valid) m.row_slot_of.erase(m.row_slots[slot].key);
          m.row_slots[slot].key = ck; m.row_slots[slot].valid = true; m.row_slots[slot].last_use = epoch;
          m.row_slot_of[ck] = slot;
          ++n_used;
          RowDesc d;
          d.slot = slot;
          d.tbl_off = agg[hk.second].shift == 0 ? agg[hk.second].tbl_off : NO_TABLE;  // one table slot per tile
          d.post_off = hk.second.post_off;
          d.len = agg[hk.second].len;
          // zero_to_one: all_query_terms_len (low 16 bits) | required term frequency (high 16 bits)
          d._pad = z ? ((uint32_t)(hk.second.k3 & 0xFFFFu) | ((uint32_t)(hk.second.k3 >> 32) << 16)) : 0u;
          memcpy(&d.idf, &hk.second.w, 8);           // BM25: idf | zero_to_one: ScoreByTerm::score
          if (z) d.eb = 0.0; else memcpy(&d.eb, &hk.second.k3, 8);
          row_of[hk.second] = slot;
          rd[n_rows++] = d;
        }
        for (size_t q = 0; q < B; ++q) {
          if (z && !(qf[q] & 1u)) continue;
          const uint32_t b = plan.qbeg[q], e = plan.qbeg[q + 1];
          for (uint32_t i = b; i < e; ++i) {
            if ((double)he[i].len < min_density * (double)s.n_ids) continue;  // never a row candidate: no map lookup
            auto it = row_of.find(key_of(he[i], q));
            if (it != row_of.end()) { he[i].shift |= DENSE_FLAG; he[i].node = it->second; }
          }
          // BM25 with one list per query term (plain sum in plan order): two row uses need no
          // read-modify-write of the LDS tile.  The query's LAST entry, if dense, is added in
          // registers while the tile is harvested.  A dense entry in position 0 or 1 goes first
          // (IEEE addition commutes, so (e0 + e1) + ... keeps its bits) and is WRITTEN into the
          // freshly zeroed tile instead of added to it.
          // (zero_to_one simple queries without consumed-term masks sum their records the same way,
          // in sorted order: same two tricks, for 1 or 2 fields)
          const bool plain_sum = z ? (z_masked == 0 && s.F <= 2) : !plan.multi_expansion;
          if (plain_sum && m.tune.dense_fuse && e > b) {
            if ((m.tune.dense_fuse & 1u) && (he[e - 1].shift & DENSE_FLAG)) he[e - 1].shift |= DENSE_FUSE_FLAG;
            if ((m.tune.dense_fuse & 2u) && e - b >= 2) {
              const bool d0 = (he[b].shift & DENSE_FLAG) != 0;
              const bool d1 = (he[b + 1].shift & DENSE_FLAG) && !(he[b + 1].shift & DENSE_FUSE_FLAG);
              if (!d0 && d1) std::swap(he[b], he[b + 1]);
              if (d0 || d1) he[b].shift |= DENSE_ASSIGN_FLAG;
            }
          }
        }
      }
    }
  }
}

// Bytes of the layout the batch actually streams (never claim the wider figure for a
// narrower stream): 4+8F per sparse posting, 8 per document and field plane of a dense-row use,
// plus scoring the rows that were not resident.
uint64_t layout_bytes_of(const EngineImpl& m, const BatchImage& img) {
  const Snapshot& s = *m.snap;
  const uint32_t planes = img.z ? s.F : 1u;
  const uint64_t pb = 4 + 4 * (uint64_t)s.F, row_bytes = (uint64_t)s.tiles_cap * s.T * 8 * planes;
  uint64_t lb = 0;
  for (size_t i = 0; i < img.ne; ++i) lb += (img.he[i].shift & DENSE_FLAG) ? row_bytes : (uint64_t)img.he[i].len * pb;
  const RowDesc* rd = reinterpret_cast<const RowDesc*>(img.h + img.off_r);
  for (uint32_t r = 0; r < img.n_rows; ++r) lb += (uint64_t)rd[r].len * (pb + 8 * planes) + row_bytes;
  return lb;
}

// Work decomposition: one wave per (query, run of S tiles).  Sets kp.S / n_super / slice_bytes.
void choose_run_length(EngineImpl& m, const ps_scorer_desc& sc, const Plan& plan, const BatchImage& img, bool topk_path,
                       KParams& kp) {
  const Snapshot& s = *m.snap;
  const size_t B = img.B;
  const uint32_t z_masked = img.z_masked;
  const uint64_t target = m.tune.target_items;
  uint64_t S = ((uint64_t)s.n_tiles * std::max<size_t>(B, 1) + target - 1) / target;
  // an item's cost grows with the lists per query: keep items comparable to the 3-list case the
  // target was tuned on (C5, 8 lists per query: 9 tiles per run measured 3 % faster than 19)
  if (plan.max_entries > 3) S = std::max<uint64_t>(1, (S * 3 + plan.max_entries - 1) / plan.max_entries);
  const uint32_t s_env = m.tune.tiles_per_run;
  if (s_env) S = s_env;
  if (S < 1) S = 1;
  if (S > s.n_tiles) S = s.n_tiles;
  if (S > 32) S = 32;  // a run's table slice is fetched by one wave-wide load (lane <-> tile)
  if (!s_env && m.tune.slices) {
    // The table slices of a run live in the wave's LDS next to its tile: a longer run must not
    // cost resident waves (C5, 8 lists per query: 20 tiles per run instead of 16 dropped a 4-wave
    // workgroup per CU and 35 % of the speed).  Within [S/4, S] take the longest run that keeps
    // the waves per CU of the shortest one.
    const bool bm25 = sc.kind == PS_SCORER_BM25;
    const size_t aw = bm25 ? 1 : s.F;
    const bool tags = bm25 ? plan.multi_expansion : z_masked != 0;
    const size_t tile_b = (size_t)s.T * aw * 8 + (tags ? (bm25 ? (size_t)s.T * 2 : (size_t)s.T * aw * 4) : 0);
    const size_t lut_b = bm25 ? (size_t)kp.lut_stride * LUT_TF * 8 : 0;
    auto waves_per_cu = [&](uint64_t runs) -> size_t {
      size_t slice = (((size_t)plan.max_entries * 2 * runs * 4) + 15) & ~(size_t)15;
      if (slice > 4096) slice = 0;
      const size_t wave_b = tile_b + slice;
      const bool wide = wide_workgroups(m, bm25, s.F, tags, !topk_path, lut_b, wave_b);
      const uint32_t wgw = wide ? 8u : (uint32_t)WG_WAVES;
      return k_score_waves_per_cu(m, k_score_fn(bm25, s.F, tags, !topk_path, wide), wgw, lut_b + wgw * wave_b);
    };
    const uint64_t s_lo = std::max<uint64_t>(1, S / 4);
    const size_t best = waves_per_cu(s_lo);
    while (S > s_lo && waves_per_cu(S) < best) --S;
  }
  kp.S = (uint32_t)S;
  kp.n_super = (uint32_t)((s.n_tiles + S - 1) / S);
  // per-wave LDS for the table slices: [entry][rb|re][S] u32; fall back to global lookups if large
  const size_t slice = (((size_t)plan.max_entries * 2 * S * 4) + 15) & ~(size_t)15;
  kp.slice_bytes = (slice <= 4096 && m.tune.slices) ? (uint32_t)slice : 0u;
}

//   topk_path: k_merge follows and leaves the control words zeroed again (no memset next time)
//   sync_path: the caller waits for the stream before returning, so the slot needs no reuse fence
void stage_plan(EngineImpl& m, const ps_scorer_desc& sc, const double* boosts, const Plan& plan, hipStream_t st,
                KParams& kp, bool topk_path, bool sync_path) {
  const Snapshot& s = *m.snap;
  refresh_tuning(m);
  m.last_bounds_recomputed = false;
  // (BM25 top-k batches that qualify for K1d never get here: enqueue_daat_host / run_device_planned)
  static const bool trace_sp = env_u32("PS_TRACE", 0) != 0;
  double tsp = now_ms();
  auto SP = [&](const char* what) {
    if (!trace_sp) return;
    const double n = now_ms();
    fprintf(stderr, "[ps]   stage %-10s %.3f ms\n", what, n - tsp);
    tsp = now_ms();
  };
  wait_daat_contexts(m, st);
  BatchImage img = lay_out_batch(m, sc, plan);
  SP("layout");
  const size_t B = img.B;
  const bool z = img.z;
  order_queries(m, plan, img);
  if (z) classify_zero_to_one(m, plan, img);
  if (z && m.tune.z21_exact_numerator) {
    // k_score's one-division arm (score_trip): per entry the largest L with fmin(score / t, 1.) * t == score
    // for every term frequency t <= L, evaluated here in the same IEEE f64 arithmetic; it rides in the
    // entry's idf word, which zero_to_one does not use
    double last = -1.0;
    uint64_t last_l = 0;
    for (size_t i = 0; i < img.ne; ++i) {
      const double w = img.he[i].boost;
      if (!(w == last)) {
        last = w;
        last_l = 0;
        for (uint32_t t = 1; t <= 254; ++t) {  // (255 is the packed words' "see the tf plane" value)
          const double df = (double)t;
          if (!(std::fmin(w / df, 1.0) * df == w)) break;
          last_l = t;
        }
      }
      memcpy(&img.he[i].idf, &last_l, 8);
    }
  }
  bool z_field_bounds = false;
  if (z && topk_path && m.tune.z21_field_prune && img.n_simple && s.n_docs > 0 && !plan.entries.empty()) {
    // per query and field: no document's pool of that field can exceed the sum over the query's lists of
    // score / max(shortest such field holding the term, query terms)
    compute_z_bounds(m);
    if (m.z_layer_of.size() != s.layers.size()) {
      m.z_layer_of.clear();
      for (size_t l = 0; l < s.layers.size(); ++l) m.z_layer_of.emplace(s.layers[l].post_off, (uint32_t)l);
    }
    double* zf = reinterpret_cast<double*>(img.h + img.off_zf);
    const uint32_t F = s.F;
    for (size_t q = 0; q < img.B; ++q) {
      for (uint32_t x = 0; x < F; ++x) {
        double sum = 0.0;
        for (uint32_t i = plan.qbeg[q]; i < plan.qbeg[q + 1]; ++i) {
          const ps_plan_entry& e = img.he[i];
          const uint32_t mn = m.z_minfl[(size_t)m.z_layer_of.at(e.post_off) * F + x];
          if (mn != 0xFFFFFFFFu) sum += e.boost * (1.0 + 1e-12) / (double)std::max(mn, plan.qterms_len[q]);
        }
        zf[q * F + x] = sum * (1.0 + 1e-9);
      }
    }
    z_field_bounds = true;
  }
  SP("order");
  select_dense_rows(m, sc, boosts, plan, img);
  SP("rows");
  const uint32_t n_rows = img.n_rows, n_used = img.n_used, n_general = img.n_general;
  Stage& sg = *img.slot;
  unsigned char* h = img.h;
  const size_t off_e = img.off_e, off_q = img.off_q, off_l = img.off_l, off_o = img.off_o, off_z = img.off_z,
               off_f = img.off_f, off_g = img.off_g, off_r = img.off_r, total = img.total;
  // A tiny batch whose plans stay in SGPRs for a whole run (<= G entries per query) is read in
  // place from the pinned slot: a few hundred bytes over PCIe per wave, in parallel, instead of a
  // copy-engine hand-over in front of the kernel (latency path of a single query).
  const uint32_t g_regs = (s.F == 1 || s.F == 2) ? (uint32_t)PS_G : 1u;
  const bool zero_copy = B <= 4 && plan.max_entries <= g_regs && n_used == 0 && n_general == 0 && m.tune.zero_copy;
  const unsigned char* dbase;
  m.cur_stage = &sg;
  m.cur_zero_copy = zero_copy;
  if (zero_copy) {
    dbase = sg.dp;
  } else {
    // one upload: the device image has the staging layout (entries | qbeg | qterms_len | qorder | zorder | qflags)
    m.d_stage.ensure(total + 64);
    const size_t up_bytes = z_field_bounds ? total : n_rows ? off_r + n_rows * sizeof(RowDesc) : (z ? off_r : off_z);
    if (m.tune.kernel_upload && up_bytes <= ((size_t)4 << 20)) {
      const size_t n16 = (up_bytes + 15) / 16;  // slot and device buffer are both padded past `total`
      const uint32_t blocks = (uint32_t)std::min<size_t>(256, (n16 + 255) / 256);
      hipLaunchKernelGGL(k_upload, dim3(std::max(1u, blocks)), dim3(256), 0, st, reinterpret_cast<const uint4*>(sg.dp),
                         reinterpret_cast<uint4*>(m.d_stage.p), n16);
      PS_HIP(hipGetLastError());
    } else {
      PS_HIP(hipMemcpyAsync(m.d_stage.p, h, up_bytes, hipMemcpyHostToDevice, st));
    }
    if (!sync_path) {
      PS_HIP(hipEventRecord(sg.done, st));
      sg.pending = true;
    }
    dbase = m.d_stage.p;
  }

  memset(&kp, 0, sizeof(kp));
  kp.doc = m.d_doc; kp.tf = m.d_tf; kp.fl = m.d_fl; kp.tfl = m.d_tfl; kp.table = m.d_table; kp.keys = m.d_keys; kp.bits = m.d_bits;
  kp.alive = s.any_dead ? m.d_alive : nullptr;
  kp.plan = reinterpret_cast<const ps_plan_entry*>(dbase + off_e);
  kp.qbeg = reinterpret_cast<const uint32_t*>(dbase + off_q);
  kp.qterms_len = reinterpret_cast<const uint32_t*>(dbase + off_l);
  kp.qorder = reinterpret_cast<const uint32_t*>(dbase + off_o);
  kp.zorder = reinterpret_cast<const uint32_t*>(dbase + off_z);
  kp.qflags = reinterpret_cast<const uint32_t*>(dbase + off_f);
  kp.gen_queries = reinterpret_cast<const uint32_t*>(dbase + off_g);
  const uint64_t layout_bytes = layout_bytes_of(m, img);
  kp.row_desc = reinterpret_cast<const RowDesc*>(dbase + off_r);
  kp.n_rows = n_rows;
  kp.zfub = z_field_bounds ? reinterpret_cast<const double*>(dbase + img.off_zf) : nullptr;
  m.build_slots.clear();  // rows the host has to zero-fill for K0b
  for (uint32_t r = 0; r < n_rows; ++r) {
    const RowDesc& d = reinterpret_cast<const RowDesc*>(h + off_r)[r];
    if (d.tbl_off == NO_TABLE) m.build_slots.push_back(d.slot);
  }
  kp.row_planes = z ? s.F : 1u;
  kp.row_mode = z ? 1u : 0u;
  kp.row_stride = (uint64_t)s.tiles_cap * s.T;
  if (n_used) {
    kp.rows = m.d_rows.p;
  }
  // control words: the persistent waves' item counter and the per-query thresholds.  k_merge
  // zeroes them again behind itself, so a memset is only needed after a (re)allocation, a
  // full-result batch or an error.  The counter has a cache line of its own: sharing one with
  // threshold words cost 70 % of K1's speed (L2 atomics on the line stall the epilogues' loads of
  // the neighbouring thresholds, and the other way round).
  const size_t n_thr = B + 2;
  const bool fresh = m.d_gthr.ensure(n_thr, true);
  kp.work_counter = m.d_work;
  kp.wstats = m.d_wstats;
  kp.gthr = m.d_gthr.p;
  if (!(m.ctl_clean && topk_path && !fresh)) {
    PS_HIP(hipMemsetAsync(m.d_gthr.p, 0, n_thr * 8, st));
    PS_HIP(hipMemsetAsync(m.d_work, 0, 256, st));
  }
  m.ctl_clean = false;  // enqueue_topk sets it once k_merge is in the stream
  kp.n_simple = img.n_simple; kp.n_general = n_general; kp.z_masked = img.z_masked;
  kp.z_qwords = std::max<uint32_t>(1, (img.z_qterms + 31) / 32);
  kp.layout_bytes = layout_bytes;
  m.last_layout_bytes = layout_bytes;
  m.last_rows = n_used;
  m.last_rows_built = n_rows;
  kp.P = s.P;
  kp.t_log2 = 0;
  while ((1u << kp.t_log2) < s.T) ++kp.t_log2;
  kp.B = (uint32_t)B; kp.n_tiles = s.n_tiles; kp.T = s.T; kp.n_docs = (uint32_t)s.n_ids; kp.F = s.F;  // (id space: live + delta-removed documents)
  kp.max_qterms = std::max<uint32_t>(1, plan.max_qterms);
  kp.ablate = m.tune.ablate;
  kp.k1 = sc.bm25_k1; kp.b = sc.bm25_b;
  kp.k1p1 = sc.bm25_k1 + 1.0;        // (self.bm25k1 + 1_f64), bm25.rs:78 — same IEEE add on the host
  kp.one_minus_b = 1.0 - sc.bm25_b;  // (1_f64 - self.bm25b),  bm25.rs:80
  for (uint32_t x = 0; x < s.F; ++x) { kp.avg[x] = s.avg[x]; kp.boost[x] = boosts[x]; }
  if (sc.kind == PS_SCORER_BM25 && m.tune.lut) {
    kp.lut = m.d_lut;
    kp.lut_rows = s.lut_rows;
    kp.lut_stride = s.lut_rows ? ((s.lut_rows + 1) | 1u) : 0;  // odd stride; LUT bytes = stride*128, so tiles stay 16-B aligned
    for (uint32_t x = 0; x < s.F; ++x) { kp.lut_cap[x] = s.lut_cap[x]; kp.lut_base[x] = s.lut_base[x]; }
  }
  choose_run_length(m, sc, plan, img, topk_path, kp);
  SP("rest");
  static const bool trace = env_u32("PS_TRACE", 0) != 0;
  if (trace && B > 1)
    fprintf(stderr, "[ps] geometry     B=%zu tiles=%u S=%u runs=%u items=%zu slice=%u B/wave max_entries=%u rows=%u\n", B,
            s.n_tiles, kp.S, kp.n_super, B * kp.n_super, kp.slice_bytes, plan.max_entries, n_rows);
}

void allow_lds(const void* fn, size_t lds) {
  if (lds <= 65536) return;
  if (lds > 160 * 1024) throw std::length_error("LDS tile exceeds 160 KiB: use a smaller tile_docs");
  PS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
}

template <int MODE, bool FULL>
void launch_k_score(EngineImpl& m, KParams& kp, bool tags, int n_cu, hipStream_t st) {
  const uint32_t n_items = kp.B * kp.n_super;
  const uint32_t aw = MODE == MODE_Z21S ? kp.F : 1u;
  const size_t lut_b = MODE == MODE_BM25 ? (size_t)kp.lut_stride * LUT_TF * 8 : 0;
  const size_t wave_b = (size_t)kp.T * aw * 8 + (tags ? (MODE == MODE_Z21S ? (size_t)kp.T * aw * 4 : (size_t)kp.T * 2) : 0) +
                        kp.slice_bytes;
  // Workgroups of 8 waves share one LUT copy: two of them (16 waves) fit a CU's 160 KiB when a
  // wave's tile is small enough; otherwise 4-wave workgroups pack the LDS better.
  const bool wide = wide_workgroups(m, MODE == MODE_BM25, kp.F, tags, FULL, lut_b, wave_b);
  const uint32_t wgw = wide ? 8u : (uint32_t)WG_WAVES;
  uint32_t n_wg = (n_items + wgw - 1) / wgw;
  const size_t lds = lut_b + wgw * wave_b;
#define PS_LAUNCH_W(FV, TG, W)                                                                         \
  do {                                                                                                 \
    const void* fn = reinterpret_cast<const void*>(&k_score<MODE, FV, TG, FULL, W>);                   \
    allow_lds(fn, lds);                                                                                \
    const uint32_t per_cu = (uint32_t)(k_score_waves_per_cu(m, fn, W, lds) / (W));                     \
    const uint32_t resident = std::max(1u, per_cu) * (uint32_t)n_cu;                                   \
    if (n_wg > resident) n_wg = resident;                                                              \
    char nm[96];                                                                                       \
    snprintf(nm, sizeof(nm), "ps::k_score<%d, %d, %s, %s, %d>", (int)MODE, (int)(FV), (TG) ? "true" : "false",          \
             FULL ? "true" : "false", (int)(W));                                                       \
    m.score_kernel_name = nm;                                                                          \
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

// K1d: persistent 8-wave workgroups (the LUT is the only LDS), items from the device-scope counter
// `small_lists`: 0 = not for k_daat_small; else the most lists any query of the launch has (<= DAAT_SMALL_MAX)
void launch_daat(EngineImpl& m, KParams& kp, bool multi, uint32_t small_lists, int n_cu, hipStream_t st) {
  const bool small = small_lists != 0;
  // (PS_DAAT_PAD_LDS: extra dynamic LDS per workgroup - an occupancy cap for experiments; 24000 = 3 waves per SIMD)
  static const size_t pad_lds = env_u32("PS_DAAT_PAD_LDS", 0);
  const size_t lds = pad_lds;  // (K1d reads score planes: no table to stage)
#define PS_DAAT(FV, MU)                                                                                  \
  do {                                                                                                   \
    const void* fn = reinterpret_cast<const void*>(&k_daat<FV, MU>);                                     \
    const uint32_t per_cu = (uint32_t)(k_score_waves_per_cu(m, fn, DAAT_WGW, lds) / DAAT_WGW);                         \
    const uint32_t n_wg = m.tune.daat_persistent                                                         \
        ? std::min<uint32_t>((kp.n_ditems + DAAT_WGW - 1) / DAAT_WGW, std::max(1u, per_cu) * (uint32_t)n_cu)                 \
        : (kp.n_ditems + DAAT_WGW - 1) / DAAT_WGW;                                                                         \
    char nm[96];                                                                                         \
    snprintf(nm, sizeof(nm), "ps::k_daat<%d, %s>", (int)(FV), (MU) ? "true" : "false");                  \
    m.score_kernel_name = nm;                                                                            \
    hipLaunchKernelGGL((k_daat<FV, MU>), dim3(n_wg), dim3(WAVE * DAAT_WGW), lds, st, kp);                       \
  } while (0)
  if (small && !multi && !m.tune.daat_persistent && m.tune.daat_small) {
    // plans of <= 4 lists, one per query term: the short-chain kernel
    const uint32_t n_wg = (kp.n_ditems + DAAT_WGW - 1) / DAAT_WGW;
    char nm[96];
    const bool three = small_lists <= 3 && m.tune.daat_small_nl;  // (queries of <= 3 lists: the instantiation with two other lists of state)
    snprintf(nm, sizeof(nm), "ps::k_daat_small<%d, %s, %d>", kp.F <= 2 ? (int)kp.F : 0, m.tune.work_counters ? "true" : "false", three ? 3 : 4);
    m.score_kernel_name = nm;
#define PS_SMALL(FV)                                                                                                    \
  do {                                                                                                                  \
    if (three) {                                                                                                        \
      if (m.tune.work_counters) hipLaunchKernelGGL((k_daat_small<FV, true, 3>), dim3(n_wg), dim3(WAVE * DAAT_WGW), lds, st, kp); \
      else hipLaunchKernelGGL((k_daat_small<FV, false, 3>), dim3(n_wg), dim3(WAVE * DAAT_WGW), lds, st, kp);             \
    } else if (m.tune.work_counters) hipLaunchKernelGGL((k_daat_small<FV, true>), dim3(n_wg), dim3(WAVE * DAAT_WGW), lds, st, kp); \
    else hipLaunchKernelGGL((k_daat_small<FV, false>), dim3(n_wg), dim3(WAVE * DAAT_WGW), lds, st, kp);                  \
  } while (0)
    if (kp.F == 1) PS_SMALL(1); else if (kp.F == 2) PS_SMALL(2); else PS_SMALL(0);
#undef PS_SMALL
  } else if (multi) {
    if (kp.F == 1) PS_DAAT(1, true); else if (kp.F == 2) PS_DAAT(2, true); else PS_DAAT(0, true);
  } else {
    if (kp.F == 1) PS_DAAT(1, false); else if (kp.F == 2) PS_DAAT(2, false); else PS_DAAT(0, false);
  }
#undef PS_DAAT
}

void launch_rows(const KParams& kp, const std::vector<uint32_t>& zero_slots, hipStream_t st) {
  if (!kp.n_rows) return;  // every row this batch reads is resident
  const size_t row_b = (size_t)kp.row_planes * kp.row_stride * 8;
  for (uint32_t slot : zero_slots)  // lists without a per-tile table (never the dense ones in practice)
    PS_HIP(hipMemsetAsync(const_cast<double*>(kp.rows) + (size_t)slot * kp.row_planes * kp.row_stride, 0, row_b, st));
  // workgroups per row, each owning a range of tiles (PS_ROW_BLOCKS; 256 / 512 / 1024 / 2048 measured 46 / 41 / 37 / 38 us
  // per batch on C2 and 211 / 177 / 158 / 154 us on C4)
  static const uint32_t row_blocks_env = env_u32("PS_ROW_BLOCKS", 0);
  const uint32_t row_blocks = row_blocks_env ? row_blocks_env : std::min(2048u, std::max(256u, kp.n_tiles));
  hipLaunchKernelGGL(k_dense_rows, dim3(row_blocks, kp.n_rows), dim3(256), 0, st, kp, const_cast<double*>(kp.rows));
  PS_HIP(hipGetLastError());
}

// `mid` (may be null): recorded between K0 / K0b and the scoring kernel, so the latter is timed alone
template <bool FULL>
void launch_score(EngineImpl& m, const ps_scorer_desc& sc, const Plan& plan, KParams& kp, int n_cu, hipStream_t st,
                  hipEvent_t mid) {
  const uint32_t n_items = kp.B * kp.n_super;
  if (n_items == 0) {
    if (mid) PS_HIP(hipEventRecord(mid, st));
    return;
  }
  // host-side part of ps_work_counters (the rest is counted by the kernels)
  m.wc_launches++;
  if (!FULL) {
    m.wc_k = kp.K;
    m.wc_results += (uint64_t)kp.B * kp.K;
    m.wc_cand_slots += (uint64_t)n_items * kp.K;
  }
  if (sc.kind == PS_SCORER_BM25) {
    // K0 runs when (k1, b) change.  The table is shared by every stream: a rebuild first waits for the K1d
    // contexts still reading it, and launches on other streams wait for the rebuild's event.
    if (kp.lut_rows && !(m.lut_valid && m.lut_k1 == sc.bm25_k1 && m.lut_b == sc.bm25_b && m.tune.lut_cache)) {
      for (auto& c : m.dctx)
        if (c.busy) PS_HIP(hipStreamWaitEvent(st, c.done, 0));
      if (m.tail_pending && m.tail_stream != st) PS_HIP(hipStreamWaitEvent(st, m.ev[0], 0));
      hipLaunchKernelGGL(k_bm25_lut, dim3(4), dim3(256), 0, st, kp, const_cast<double*>(kp.lut));
      PS_HIP(hipEventRecord(m.lut_ready, st));
      m.lut_valid = true; m.lut_k1 = sc.bm25_k1; m.lut_b = sc.bm25_b; m.lut_stream = st;
    } else if (kp.lut_rows && m.lut_stream != st) {
      PS_HIP(hipStreamWaitEvent(st, m.lut_ready, 0));
    }
    launch_rows(kp, m.build_slots, st);
    if (mid) PS_HIP(hipEventRecord(mid, st));
    launch_k_score<MODE_BM25, FULL>(m, kp, plan.multi_expansion, n_cu, st);
  } else {
    launch_rows(kp, m.build_slots, st);
    if (mid) PS_HIP(hipEventRecord(mid, st));
    if (kp.n_simple) launch_k_score<MODE_Z21S, FULL>(m, kp, kp.z_masked != 0, n_cu, st);
    if (kp.n_general) {
      // general zero_to_one: the LDS sub-tile shrinks with (distinct nodes x fields) to fit the budget
      // (any number of expanded lists per query: the per-node pool lives in the record words, the
      // consumed query terms in an LDS bit array; the sub-tile shrinks until a wave's state fits)
      kp.z_nodes = std::max<uint32_t>(1, plan.max_nodes);
      const size_t per_doc = ((size_t)kp.z_nodes * kp.F + kp.F + kp.z_qwords) * 4;
      uint32_t zt = kp.T;
      const uint32_t budget = m.tune.z21_lds;
      while (zt > (uint32_t)WAVE && (size_t)zt * per_doc > budget) zt >>= 1;
      while (zt > 8u && (size_t)zt * per_doc > 160 * 1024) zt >>= 1;  // very wide plans: fewer documents per pass
      if ((size_t)zt * per_doc > 160 * 1024) throw std::length_error("zero_to_one: fields x distinct expanded terms exceed 160 KiB of LDS even for 8 documents");
      if (m.snap->max_fl.size() && *std::max_element(m.snap->max_fl.begin(), m.snap->max_fl.end()) >= 0xFFFFu)
        throw std::length_error("zero_to_one: field lengths >= 65535 tokens are not supported by the general kernel");
      kp.z_tile = zt;
      if (!kp.n_simple) m.score_kernel_name = FULL ? "ps::k_z21<true>" : "ps::k_z21<false>";
      {
        const void* fn = FULL ? reinterpret_cast<const void*>(&k_z21<true>) : reinterpret_cast<const void*>(&k_z21<false>);
        allow_lds(fn, (size_t)zt * per_doc);
      }
      hipLaunchKernelGGL((k_z21<FULL>), dim3(kp.n_general * kp.n_super), dim3(WAVE), (size_t)zt * per_doc, st, kp);
    }
  }
  PS_HIP(hipGetLastError());
}

// A throw between select_dense_rows (which books row slots) and the launch of K0b would leave
// slots marked resident whose rows were never scored: drop the whole slab's bookkeeping instead.
void forget_rows(EngineImpl& m) {
  m.row_slot_of.clear();
  for (auto& sl : m.row_slots) sl = EngineImpl::RowSlot{};
  m.ctl_clean = false;
  m.lut_valid = false;
  // K1d: bounds, row candidates and resident rows are re-derived from the snapshot's current state
  m.bounds.m_valid = false;
  m.bounds.n_units = 0;
  for (auto& js : m.bounds.j) js.valid = false;
  m.cands.valid = false;
  m.bloom_valid = false;
  for (auto& c : m.dctx) { c.row_sig.clear(); c.ctl_clean = false; }
}

void fill_stats(const EngineImpl& m, ps_batch_stats& st, const Snapshot& s, const Plan& plan, uint64_t emitted) {
  st.layout_bytes = m.last_layout_bytes + emitted * 16;
  st.dense_rows = m.last_rows;
  st.dense_rows_built = m.last_rows_built;
  st.bounds_recomputed = m.last_bounds_recomputed ? 1u : 0u;
  st.n_queries = plan.qbeg.size() - 1;
  st.n_plan_entries = plan.entries.size();
  st.postings_visited = plan.postings;
  st.algorithmic_bytes = plan.postings * (4 + 8 * (uint64_t)s.F) + emitted * 16;
}

void fill_common_kparams(EngineImpl& m, const ps_scorer_desc& sc, const double* boosts, size_t B, uint32_t max_qterms, KParams& kp);

// K1d takes BM25 top-k batches whose parameters make every score a positive, monotone function of the
// saturated term frequency and whose plans have at most 64 entries per query; everything else stays on K1.
bool daat_eligible(const EngineImpl& m, const ps_scorer_desc& sc, const double* boosts, size_t B, size_t n_entries, uint32_t max_entries,
                   bool multi) {
  return m.tune.daat && sc.kind == PS_SCORER_BM25 && B >= m.tune.daat_min_batch && n_entries != 0 && m.tune.lut &&
         bm25_params_sane(*m.snap, sc, boosts) && max_entries <= 64 && (!multi || m.tune.daat_multi) && m.snap->lut_rows != 0;
}

// The part every K1d batch shares.  Preparation stream: control words, device-side preparation, K0b over
// the rows the preparation listed.  Scoring stream: k_daat, then - once the caller's stream has reached the
// point of the call, because the merge is the kernel that writes the caller's buffers - k_merge_items.
// The caller's stream is made to wait for the batch, so work enqueued on it afterwards sees the results.
// K1dz (zero_to_one, ps_z21_daat.hpp): what the batch's image carries beside the plan
struct ZBatch {
  const double* d_ubnum;  // [ne]
  const double* d_zub;    // [ne][F]
  uint32_t dl[Z_LEVELS];  // doc ids D_0 < D_1 < ... of the tie-threshold levels, Z_NO_LEVEL for a level that does not exist
  int zn = DAAT_SMALL_MAX;  // most records of a query of the batch, rounded up to an instantiation (DAAT_SMALL_MAX or Z_MAX_LISTS)
};
inline int z_width(uint32_t max_entries) { return max_entries <= (uint32_t)DAAT_SMALL_MAX ? DAAT_SMALL_MAX : Z_MAX_LISTS; }
void launch_prep_z(EngineImpl& m, EngineImpl::DaatCtx& c, const ZBatch& zb, KParams& kp, ps_plan_entry* d_plan, const uint32_t* d_qbeg,
                   size_t B, size_t ne, size_t items_bound);
void launch_daat_z(EngineImpl& m, KParams& kp, hipStream_t st, int zn);

void enqueue_daat(EngineImpl& m, EngineImpl::DaatCtx& c, const ps_scorer_desc& sc, const double* boosts, ps_plan_entry* d_plan,
                  const uint32_t* d_qbeg, const uint32_t* d_qtl, size_t B, size_t ne, uint32_t max_qterms, uint32_t max_entries, bool multi,
                  size_t n_items, uint32_t max_slots, size_t top_k, void* d_keys, void* d_scores, void* d_counts, hipStream_t caller,
                  const ZBatch* zb = nullptr, const uint32_t* d_out_row = nullptr, size_t n_items_big = 0) {
  const Snapshot& s = *m.snap;
  check_device_fault(m);
  hipStream_t P = m.prep_stream, S = m.score_stream;
  if (m.tune.score_alt == 4) {  // three-way rotation: normal, low, high
    const uint32_t r = m.score_flip++ % 3u;
    S = r == 0 ? m.score_stream : r == 1 ? m.score_stream_lo : m.score_stream_hi;
  } else if (m.tune.score_alt) {
    const bool odd = (m.score_flip++ & 1u) != 0;
    S = m.tune.score_alt == 2 ? m.score_stream_lo : !odd ? m.score_stream : m.tune.score_alt == 3 ? m.score_stream_b : m.score_stream_lo;
  }
  KParams kp;
  EngineImpl::KTimer* kt = nullptr;
  try {
    fill_common_kparams(m, sc, boosts, B, max_qterms, kp);
    kp.plan = d_plan; kp.qbeg = d_qbeg; kp.qterms_len = d_qtl;
    kp.work_counter = c.work;
    const size_t n_thr = B + 2;
    bool fresh = c.gthr.ensure(n_thr, true);
    kp.gthr = c.gthr.p;
    if (zb) {  // (Z_LEVELS more threshold words per query; zeroed behind the batch by k_merge_items like the first)
      fresh = c.gtie.ensure(n_thr * Z_LEVELS, true) || fresh;
      kp.gtie = c.gtie.p;
      kp.z_tstride = (uint32_t)n_thr;
    }
    if (!(c.ctl_clean && !fresh)) {
      PS_HIP(hipMemsetAsync(c.gthr.p, 0, c.gthr.cap * 8, P));
      if (c.gtie.p) PS_HIP(hipMemsetAsync(c.gtie.p, 0, c.gtie.cap * 8, P));
      PS_HIP(hipMemsetAsync(c.work, 0, 256, P));
      PS_HIP(hipMemsetAsync(c.ctl, 0, sizeof(PrepCtl), P));
    }
    c.ctl_clean = false;
    kp.S = 1; kp.n_super = s.n_tiles; kp.slice_bytes = 0;
    // a BM25 batch with both kinds of queries - those k_daat_small takes (<= 4 lists, one per query term) and others - is scored
    // by both kernels, each over its part of one item array (the preparation puts the second kind's items behind the first's)
    // (... unless most of the batch is of the second kind: after a delta with additions nearly every query has a delta layer under some
    // term, and two launches - a near-empty k_daat_small beside k_daat - measured 1.64 ms per batch against 1.01 for k_daat alone)
    const bool split_kinds = !zb && m.tune.daat_split && m.tune.daat_small && !m.tune.daat_persistent && n_items_big > 0 && n_items_big < n_items &&
                             n_items_big * 2 < n_items;
    kp.K = (uint32_t)top_k;  // (the preparation primes the thresholds for this K)
    if (zb) launch_prep_z(m, c, *zb, kp, d_plan, d_qbeg, B, ne, n_items);
    else launch_prep(m, c, sc, boosts, kp, d_plan, d_qbeg, B, ne, multi, n_items, split_kinds, split_kinds ? n_items_big : 0);
    const size_t n_cand = n_items * top_k;
    c.cand_score.ensure(n_cand + 1);
    c.cand_doc.ensure(n_cand + 1);
    kp.cand_score = c.cand_score.p;
    kp.cand_doc = c.cand_doc.p;
    kp.out_keys = (uint64_t*)d_keys; kp.out_scores = (double*)d_scores; kp.out_counts = (uint32_t*)d_counts;
    kp.out_row = d_out_row;
    // (PS_KERNEL_TIMERS=0: no HIP timing events around the launches - four stream packets less per batch between two
    // consecutive scoring kernels; ps_snapshot_kernel_breakdown then has nothing to report for these batches)
    const bool timers = m.tune.kernel_timers != 0;
    if (timers) {
      m.rebase_timers_if_stale(m.stream);
      kt = &m.kt[m.next_kt];
      m.next_kt = (m.next_kt + 1) % N_KTIMER;
      m.harvest(*kt, true);
      kt->split = true;
    }
    // K0b on the preparation stream too: the rows to score were listed on the device (k_prep_finish); a fixed
    // grid takes (row, tile range) units.  (It evaluates the BM25 expression itself: no table involved.)
    if (timers) PS_HIP(hipEventRecord(kt->a, P));
    if (m.cands.n && !zb) {
      const uint32_t per_row = std::min(2048u, std::max(256u, kp.n_tiles));
      const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)per_row * m.cands.n, 4096);
      hipLaunchKernelGGL(k_dense_rows_dyn, dim3(grid), dim3(256), 0, P, kp, c.rows.p, c.ctl, per_row);
    }
    if (timers) PS_HIP(hipEventRecord(kt->r, P));
    PS_HIP(hipEventRecord(c.prepared, P));
    // ---- scoring stream ----
    PS_HIP(hipStreamWaitEvent(S, c.prepared, 0));
    m.wc_launches++;
    m.wc_k = kp.K;
    m.wc_results += (uint64_t)kp.B * kp.K;
    m.wc_items += kp.n_ditems;
#ifdef PS_ITEM_TRACE
    static DevBuf<unsigned long long> trace_buf;
    trace_buf.ensure((size_t)kp.n_ditems * 4 + 8);
    PS_HIP(hipMemsetAsync(trace_buf.p, 0, (size_t)kp.n_ditems * 32, S));
    kp.item_trace = trace_buf.p;
#endif
    if (timers) PS_HIP(hipEventRecord(kt->m, S));
    if (zb) launch_daat_z(m, kp, S, zb->zn);
    else if (split_kinds) {
      uint32_t* split_at = &c.ctl->bucket_start[PREP_SET_BUCKETS];  // first item of the second kind, as k_prep_query's last wave counted it
      // the two parts are independent (disjoint queries): the second one goes to the OTHER scoring stream - a hardware queue of
      // its own -, so its few long items start with the launch instead of behind the first part's tail (one stream, one after
      // the other, 32 five-term queries among 1024: 0.91 ms per synchronous batch against 0.77 for the whole batch on k_daat)
      hipStream_t S2 = S == m.score_stream ? m.score_stream_lo : m.score_stream;
      PS_HIP(hipStreamWaitEvent(S2, c.prepared, 0));
      KParams kb = kp;
      kb.n_ditems = (uint32_t)n_items_big;
      kb.item_split_dev = split_at;
      launch_daat(m, kb, multi, 0u, m.n_cu, S2);
      PS_HIP(hipEventRecord(c.scored2, S2));
      const std::string big_name = m.score_kernel_name;
      KParams ks = kp;
      ks.n_ditems = (uint32_t)(n_items - n_items_big);
      ks.n_ditems_dev = split_at;
      launch_daat(m, ks, false, (uint32_t)DAAT_SMALL_MAX, m.n_cu, S);  // (the first part's longest plan is not known apart: the four-list instantiation)
      PS_HIP(hipStreamWaitEvent(S, c.scored2, 0));
      m.score_kernel_name += " + " + big_name;
    } else launch_daat(m, kp, multi, max_entries <= (uint32_t)DAAT_SMALL_MAX ? std::max(1u, max_entries) : 0u, m.n_cu, S);
#ifdef PS_ITEM_TRACE
    {  // profiling builds: the items' start / end times of this launch -> $PS_ITEM_TRACE_FILE (last batch wins)
      PS_HIP(hipStreamSynchronize(S));
      std::vector<unsigned long long> h((size_t)kp.n_ditems * 4);
      PS_HIP(hipMemcpy(h.data(), trace_buf.p, h.size() * 8, hipMemcpyDeviceToHost));
      const char* f = getenv("PS_ITEM_TRACE_FILE");
      if (f) { FILE* fp = fopen(f, "wb"); if (fp) { fwrite(h.data(), 8, h.size(), fp); fclose(fp); } }
    }
#endif
    PS_HIP(hipGetLastError());
    if (timers) {
      PS_HIP(hipEventRecord(kt->b, S));
      kt->pending = true;
    }
    m.last_kt = kt;
    PS_HIP(hipEventRecord(c.scored, S));
    S = m.merge_stream;
    PS_HIP(hipStreamWaitEvent(S, c.scored, 0));
    if (caller && caller != S) {  // the merge overwrites the caller's output buffers: not before the caller's earlier work is through
      PS_HIP(hipEventRecord(c.entry, caller));
      PS_HIP(hipStreamWaitEvent(S, c.entry, 0));
    }
    const uint32_t mw = max_slots ? std::min<uint32_t>(m.tune.daat_merge_waves, std::max<uint32_t>(1, (max_slots + 7) / 8)) : m.tune.daat_merge_waves;
    hipLaunchKernelGGL(k_merge_items, dim3((uint32_t)B), dim3(WAVE * mw), 0, S, kp);
    PS_HIP(hipGetLastError());
    c.ctl_clean = true;  // k_merge_items zeroes the context's control words behind itself
  } catch (...) {
    c.ctl_clean = false;
    c.row_sig.clear();
    forget_rows(m);
    throw;
  }
  PS_HIP(hipEventRecord(c.done, S));
  c.busy = true;
  if (caller && caller != S) PS_HIP(hipStreamWaitEvent(caller, c.done, 0));
  m.last_layout_bytes = 0;
  m.last_rows = 0;
  m.last_rows_built = 0;
}

// A host-planned K1d batch: the plan image (entries | qbeg | qterms_len) goes through a pinned slot into
// the context's device buffer (k_upload on the context's stream), everything else as above.
void enqueue_daat_host(EngineImpl& m, const ps_scorer_desc& sc, const double* boosts, const Plan& plan, size_t n_items, uint32_t max_slots,
                       size_t top_k, void* d_keys, void* d_scores, void* d_counts, hipStream_t caller, size_t n_items_big) {
  EngineImpl::DaatCtx& c = acquire_ctx(m);
  hipStream_t st = m.prep_stream;
  const size_t B = plan.qbeg.size() - 1, ne = plan.entries.size();
  const size_t off_q = ne * sizeof(ps_plan_entry), off_l = off_q + (B + 1) * 4, total = (off_l + B * 4 + 15) & ~(size_t)15;
  Stage& sg = m.stage[m.next_stage];
  m.next_stage = (m.next_stage + 1) % N_STAGE;
  sg.ensure(total + 16);
  memcpy(sg.p, plan.entries.data(), ne * sizeof(ps_plan_entry));
  memcpy(sg.p + off_q, plan.qbeg.data(), (B + 1) * 4);
  if (B) memcpy(sg.p + off_l, plan.qterms_len.data(), B * 4);
  c.stage.ensure(total + 64);
  const size_t n16 = (total + 15) / 16;
  hipLaunchKernelGGL(k_upload, dim3((uint32_t)std::max<size_t>(1, std::min<size_t>(256, (n16 + 255) / 256))), dim3(256), 0, st,
                     reinterpret_cast<const uint4*>(sg.dp), reinterpret_cast<uint4*>(c.stage.p), n16);
  PS_HIP(hipGetLastError());
  PS_HIP(hipEventRecord(sg.done, st));
  sg.pending = true;
  enqueue_daat(m, c, sc, boosts, reinterpret_cast<ps_plan_entry*>(c.stage.p), reinterpret_cast<const uint32_t*>(c.stage.p + off_q),
               reinterpret_cast<const uint32_t*>(c.stage.p + off_l), B, ne, plan.max_qterms, plan.max_entries, plan.multi_expansion, n_items, max_slots, top_k,
               d_keys, d_scores, d_counts, caller, nullptr, nullptr, n_items_big);
}

// ---- K1dz: zero_to_one top-k batches through the K1d pipeline (ps_z21_daat.hpp) ------------------------------------
void launch_prep_z(EngineImpl& m, EngineImpl::DaatCtx& c, const ZBatch& zb, KParams& kp, ps_plan_entry* d_plan, const uint32_t* d_qbeg,
                   size_t B, size_t ne, size_t items_bound) {
  const Snapshot& s = *m.snap;
  hipStream_t st = m.prep_stream;
  ensure_bloom(m, st);
  kp.bloom = m.d_bloom.p;
  kp.layer_bloom = m.d_layer_bloom.p;
  c.dentry.ensure(ne + 1); c.gen.ensure(ne + 1); c.z_nbelow.ensure(ne * Z_LEVELS + 1); c.z_nabove.ensure(ne * Z_LEVELS + 1);
  c.qslot.ensure(B + 1); c.qslot_n.ensure(B + 1);
  c.ditems.ensure(items_bound + 1);
  c.cand_cnt.ensure(items_bound + 1);
  ZPrepParams pp;
  memset(&pp, 0, sizeof(pp));
  pp.plan = d_plan; pp.qbeg = d_qbeg; pp.zub = zb.d_zub; pp.table = m.d_table;
  pp.B = (uint32_t)B; pp.ne = (uint32_t)ne; pp.F = s.F;
  pp.chunk_min = m.tune.daat_chunk; pp.split_div = m.tune.daat_split_div;
  for (int l = 0; l < Z_LEVELS; ++l) pp.dl_tile[l] = zb.dl[l] == Z_NO_LEVEL ? 0u : zb.dl[l] >> kp.t_log2;
  pp.dentry = c.dentry.p; pp.gen = c.gen.p; pp.nbelow = c.z_nbelow.p; pp.nabove_from = c.z_nabove.p;
  pp.qslot = c.qslot.p; pp.qslot_n = c.qslot_n.p;
  pp.items = c.ditems.p; pp.items_cap = (uint32_t)items_bound;
  pp.ctl = c.ctl;
  if (B) {
    if (zb.zn <= DAAT_SMALL_MAX) hipLaunchKernelGGL(k_zprep_query<DAAT_SMALL_MAX>, dim3((uint32_t)((B + WAVE - 1) / WAVE)), dim3(WAVE), 0, st, pp);
    else hipLaunchKernelGGL(k_zprep_query<Z_MAX_LISTS>, dim3((uint32_t)((B + WAVE - 1) / WAVE)), dim3(WAVE), 0, st, pp);
    if (ne) hipLaunchKernelGGL(k_zprep_items, dim3((uint32_t)((ne + 2 * WAVE - 1) / (2 * WAVE))), dim3(2 * WAVE), 0, st, pp);
    PS_HIP(hipGetLastError());
  }
  kp.dentry = c.dentry.p; kp.ditems = c.ditems.p; kp.qslot = c.qslot.p; kp.qslot_n = c.qslot_n.p;
  kp.n_ditems = (uint32_t)items_bound;
  kp.n_ditems_dev = &c.ctl->n_items;
  kp.prep_ctl = reinterpret_cast<uint32_t*>(c.ctl);
  kp.prep_ctl_words = (uint32_t)(sizeof(PrepCtl) / 4);
  kp.cand_cnt = c.cand_cnt.p;
  kp.z_ubnum = zb.d_ubnum;
  for (int l = 0; l < Z_LEVELS; ++l) kp.z_dl[l] = zb.dl[l];
}

void launch_daat_z(EngineImpl& m, KParams& kp, hipStream_t st, int zn) {
  const uint32_t n_wg = (kp.n_ditems + DAAT_WGW - 1) / DAAT_WGW;
  char nm[64];
  const bool wide = zn > DAAT_SMALL_MAX;
  snprintf(nm, sizeof(nm), wide ? "ps::k_daat_z<%d, %s, 8>" : "ps::k_daat_z<%d, %s, 4>", (int)std::min(kp.F, 4u), m.tune.work_counters ? "true" : "false");  // (the demangled symbols: what a profiler prints)
  m.score_kernel_name = nm;
#define PS_Z(FV)                                                                                                  \
  do {                                                                                                            \
    if (wide) {                                                                                                   \
      if (m.tune.work_counters) hipLaunchKernelGGL((k_daat_z<FV, true, Z_MAX_LISTS>), dim3(n_wg), dim3(WAVE * DAAT_WGW), 0, st, kp); \
      else hipLaunchKernelGGL((k_daat_z<FV, false, Z_MAX_LISTS>), dim3(n_wg), dim3(WAVE * DAAT_WGW), 0, st, kp);  \
    } else if (m.tune.work_counters) hipLaunchKernelGGL((k_daat_z<FV, true>), dim3(n_wg), dim3(WAVE * DAAT_WGW), 0, st, kp); \
    else hipLaunchKernelGGL((k_daat_z<FV, false>), dim3(n_wg), dim3(WAVE * DAAT_WGW), 0, st, kp);                  \
  } while (0)
  switch (kp.F) {
    case 1: PS_Z(1); break;
    case 2: PS_Z(2); break;
    case 3: PS_Z(3); break;
    default: PS_Z(4); break;
  }
#undef PS_Z
}

// K1dz's tie-threshold levels: powers of two (slot boundaries of every list's table up to that coarseness), the top one near
// N / PS_DAAT_Z_D0_DIV, the others 2^PS_DAAT_Z_LEVEL_SHIFT apart below it; a level of less than one tile does not exist
void z_levels(const EngineImpl& m, ZBatch& zb) {
  const Snapshot& s = *m.snap;
  const uint64_t div = std::max(2u, m.tune.daat_z_d0_div);
  uint64_t top = 0;
  if (s.n_ids >= div * s.T) {
    top = s.T;
    while (top * 2 <= s.n_ids / div) top *= 2;
  }
  const uint32_t want = std::min<uint32_t>(std::max(1u, m.tune.daat_z_levels), Z_LEVELS), shift = std::min(m.tune.daat_z_level_shift, 8u);
  int n = 0;
  for (uint32_t k = 0; k < want && top; ++k) {
    const uint64_t d = top >> (shift * k);
    if (d >= s.T && (n == 0 || d < zb.dl[n - 1])) zb.dl[n++] = (uint32_t)d;
  }
  std::sort(zb.dl, zb.dl + n);
  for (; n < Z_LEVELS; ++n) zb.dl[n] = Z_NO_LEVEL;
}

// zero_to_one's per-list bounds where k_zplan_arrange reads them (grown with the layers; rare: a blocking copy)
void ensure_dev_z_bounds(EngineImpl& m) {
  compute_z_bounds(m);
  const size_t nl = m.z_maxtf.size(), F = m.snap->F;
  if (m.z_dev_layers == nl) return;
  m.d_z_maxtf.ensure(nl + 1);
  m.d_z_minfl.ensure(nl * F + 1);
  if (nl) {
    PS_HIP(hipMemcpy(m.d_z_maxtf.p, m.z_maxtf.data(), nl * 4, hipMemcpyHostToDevice));
    PS_HIP(hipMemcpy(m.d_z_minfl.p, m.z_minfl.data(), nl * F * 4, hipMemcpyHostToDevice));
  }
  m.z_dev_layers = nl;
}

// Largest numerator min(score / t, 1) * t (zero_to_one.rs:117-118) over the term frequencies need <= t <= maxtf, in
// the scorer's own f64 arithmetic; 0 when no record of the list can be consumed.  A saturated maximum (the exact
// value is not known here) is bounded two ulps above the real-number value.
double z_numerator_bound(EngineImpl& m, double score, uint32_t need, uint32_t maxtf) {
  uint64_t sb;
  memcpy(&sb, &score, 8);
  const auto key = std::make_pair(sb, (uint64_t)need | ((uint64_t)maxtf << 32));
  auto it = m.z_ubnum_cache.find(key);
  if (it != m.z_ubnum_cache.end()) return it->second;
  double best = 0.0;
  const uint32_t hi = std::min(maxtf, 4096u);
  for (uint32_t t = std::max(need, 1u); t <= hi; ++t) {
    const double df = (double)t;
    best = std::max(best, std::fmin(score / df, 1.0) * df);
  }
  if (maxtf > hi && maxtf >= need) best = std::max(best, std::nextafter(std::nextafter(std::max(score, 0.0), INFINITY), INFINITY));
  if (m.z_ubnum_cache.size() > (1u << 16)) m.z_ubnum_cache.clear();
  m.z_ubnum_cache.emplace(key, best);
  return best;
}

// A zero_to_one top-k batch for K1dz, or false when the batch does not qualify (it then takes k_score / k_z21).
// Qualifies: every query "simple" in the sense of classify_zero_to_one - one version layer per entry, no two
// records with the same (query term, node) - with at most Z_MAX_LISTS entries.  The image: entries per query in
// the record-sort order (score desc, stable; zero_to_one.rs:98) | qbeg | query_terms_len | ubnum | zub.
// Whether K1dz can take query q: at most Z_MAX_LISTS records, one version layer each, and "simple" (the rule of
// classify_zero_to_one).
bool z_query_fits_k1dz(const Plan& plan, size_t q) {
  if (plan.qbeg[q + 1] - plan.qbeg[q] > (uint32_t)Z_MAX_LISTS) return false;
  bool same_q = false, same_n = false;
  for (uint32_t i = plan.qbeg[q]; i < plan.qbeg[q + 1]; ++i) {
    if (plan.entries[i].shift >> 8) return false;
    for (uint32_t j = plan.qbeg[q]; j < i; ++j) {
      same_q = same_q || plan.entries[j].qterm == plan.entries[i].qterm;
      same_n = same_n || plan.entries[j].node == plan.entries[i].node;
    }
  }
  // several expansions of a query term AND a node hit by two records: the pool rule's closed form (need) no longer holds
  return !(same_q && same_n);
}

// `rows` (a batch split between K1dz and the streaming kernels): the output row of each of the plan's queries.
bool enqueue_daat_z_host(EngineImpl& m, const ps_scorer_desc& sc, const double* boosts, const Plan& plan, size_t top_k, void* d_keys,
                         void* d_scores, void* d_counts, hipStream_t caller, const std::vector<uint32_t>* rows = nullptr) {
  const Snapshot& s = *m.snap;
  const size_t B = plan.qbeg.size() - 1, ne = plan.entries.size(), F = s.F;
  if (!(m.tune.daat && m.tune.daat_z && sc.kind == PS_SCORER_ZERO_TO_ONE && B >= m.tune.daat_min_batch && ne != 0 &&
        plan.max_entries <= (uint32_t)Z_MAX_LISTS && F <= 4 && s.n_ids > 0))
    return false;
  for (size_t q = 0; q < B; ++q)
    if (!z_query_fits_k1dz(plan, q)) return false;
  uint32_t max_slots = 0;
  const size_t n_items = count_daat_items(m, plan, &max_slots);
  if (!n_items || n_items >= 0x3FFFFFF0ull) return false;
  compute_z_bounds(m);
  EngineImpl::DaatCtx& c = acquire_ctx(m);
  hipStream_t st = m.prep_stream;
  const size_t off_q = ne * sizeof(ps_plan_entry), off_l = off_q + (B + 1) * 4, off_u = (off_l + B * 4 + 15) & ~(size_t)15,
               off_z = off_u + ne * 8, off_m = (off_z + ne * F * 8 + 15) & ~(size_t)15, total = (off_m + (rows ? B * 4 : 0) + 15) & ~(size_t)15;
  Stage& sg = m.stage[m.next_stage];
  m.next_stage = (m.next_stage + 1) % N_STAGE;
  sg.ensure(total + 16);
  ps_plan_entry* he = reinterpret_cast<ps_plan_entry*>(sg.p);
  double* hu = reinterpret_cast<double*>(sg.p + off_u);
  double* hz = reinterpret_cast<double*>(sg.p + off_z);
  memcpy(sg.p + off_q, plan.qbeg.data(), (B + 1) * 4);
  memcpy(sg.p + off_l, plan.qterms_len.data(), B * 4);
  if (rows) memcpy(sg.p + off_m, rows->data(), B * 4);
  double last_w = -1.0;
  uint64_t last_l = 0;
  for (size_t q = 0; q < B; ++q) {
    const uint32_t b = plan.qbeg[q], e = plan.qbeg[q + 1], qtl = plan.qterms_len[q];
    uint32_t zo[Z_MAX_LISTS];
    for (uint32_t i = b; i < e; ++i) zo[i - b] = i;
    std::stable_sort(zo, zo + (e - b), [&](uint32_t a, uint32_t d) { return plan.entries[d].boost < plan.entries[a].boost; });
    for (uint32_t i = b; i < e; ++i) {
      ps_plan_entry en = plan.entries[zo[i - b]];
      uint32_t need = 1, qt = 0;  // occurrence rank of the node among the sorted records; dense ordinal of the query term
      bool seen_qt = false;
      for (uint32_t j = b; j < i; ++j) {
        if (he[j].node == en.node) ++need;
        if (plan.entries[zo[j - b]].qterm == en.qterm) { qt = (he[j].qterm_index >> 16) & 31u; seen_qt = true; }
      }
      if (!seen_qt) {  // a new query term: the next free ordinal (<= 3)
        uint32_t used = 0;
        for (uint32_t j = b; j < i; ++j) used |= 1u << ((he[j].qterm_index >> 16) & 31u);
        while (used & (1u << qt)) ++qt;
      }
      en.qterm_index = need | (qt << 16);
      if (!(en.boost == last_w)) {  // the one-division arm's limit, as stage_plan computes it
        last_w = en.boost;
        last_l = 0;
        for (uint32_t t = 1; t <= 254; ++t) {
          const double df = (double)t;
          if (!(std::fmin(en.boost / df, 1.0) * df == en.boost)) break;
          last_l = t;
        }
      }
      memcpy(&en.idf, &last_l, 8);
      he[i] = en;
      const double un = z_numerator_bound(m, en.boost, need, m.z_maxtf[en.layer]);
      hu[i] = un;
      for (size_t x = 0; x < F; ++x) {
        const uint32_t mn = m.z_minfl[(size_t)en.layer * F + x];
        hz[i * F + x] = mn == 0xFFFFFFFFu ? 0.0 : un / (double)std::max(mn, qtl);
      }
    }
  }
  c.stage.ensure(total + 64);
  const size_t n16 = (total + 15) / 16;
  hipLaunchKernelGGL(k_upload, dim3((uint32_t)std::max<size_t>(1, std::min<size_t>(256, (n16 + 255) / 256))), dim3(256), 0, st,
                     reinterpret_cast<const uint4*>(sg.dp), reinterpret_cast<uint4*>(c.stage.p), n16);
  PS_HIP(hipGetLastError());
  PS_HIP(hipEventRecord(sg.done, st));
  sg.pending = true;
  ZBatch zb;
  zb.zn = z_width(plan.max_entries);
  zb.d_ubnum = reinterpret_cast<const double*>(c.stage.p + off_u);
  zb.d_zub = reinterpret_cast<const double*>(c.stage.p + off_z);
  z_levels(m, zb);
  enqueue_daat(m, c, sc, boosts, reinterpret_cast<ps_plan_entry*>(c.stage.p), reinterpret_cast<const uint32_t*>(c.stage.p + off_q),
               reinterpret_cast<const uint32_t*>(c.stage.p + off_l), B, ne, plan.max_qterms, plan.max_entries, false, n_items, max_slots, top_k,
               d_keys, d_scores, d_counts, caller, &zb, rows ? reinterpret_cast<const uint32_t*>(c.stage.p + off_m) : nullptr);
  return true;
}

// Enqueue plan upload + K1/K2 + K3 on `st`, writing the final top-k to the given buffers (device
// memory, or device-mapped pinned host memory).  `sync_path`: the caller waits for `st` before it
// returns — the latency path: no staging-slot fence, and HIP timing events only for batches of
// >= 8 queries (two extra stream packets are a visible share of a single query's round trip).
// The queries `qs` of a plan as a plan of their own.
Plan sub_plan_of(const Plan& plan, const std::vector<uint32_t>& qs) {
  Plan p2;
  p2.qbeg.push_back(0);
  for (uint32_t q : qs) {
    uint32_t qt_max = 0;
    for (uint32_t i = plan.qbeg[q]; i < plan.qbeg[q + 1]; ++i) {
      p2.entries.push_back(plan.entries[i]);
      p2.postings += plan.entries[i].len;
      qt_max = std::max(qt_max, plan.entries[i].qterm + 1);
      if (i > plan.qbeg[q] && plan.entries[i].qterm == plan.entries[i - 1].qterm) p2.multi_expansion = true;
    }
    p2.qbeg.push_back((uint32_t)p2.entries.size());
    p2.qterms_len.push_back(plan.qterms_len[q]);
    p2.n_nodes.push_back(plan.n_nodes[q]);
    p2.max_entries = std::max(p2.max_entries, plan.qbeg[q + 1] - plan.qbeg[q]);
    p2.max_nodes = std::max(p2.max_nodes, plan.n_nodes[q]);
    p2.max_qterms = std::max(p2.max_qterms, qt_max);
  }
  return p2;
}

// `rows`: the plan is one part of a split batch - query q's results go to row rows[q] of the output block.
void enqueue_topk(EngineImpl& m, const ps_scorer_desc& sc, const double* boosts, const Plan& plan, size_t top_k,
                  void* d_keys, void* d_scores, void* d_counts, hipStream_t st, bool sync_path, const std::vector<uint32_t>* rows = nullptr) {
  const size_t B = plan.qbeg.size() - 1;
  KParams kp;
  static const bool trace = env_u32("PS_TRACE", 0) != 0;
  static const bool time_all = env_u32("PS_TIME_ALL", 0) != 0;
  double tt = now_ms();
  auto TT = [&](const char* what) {
    if (!trace) return;
    double n = now_ms();
    fprintf(stderr, "[ps] %-12s %.3f ms\n", what, n - tt);
    tt = now_ms();
  };
  refresh_tuning(m);
  if (rows == nullptr) drop_ahead(m);  // (a host-planned call between an announcement and its query call: the announced batch is dropped, as the header says)
  m.last_bounds_recomputed = false;
  if (rows == nullptr && daat_eligible(m, sc, boosts, B, plan.entries.size(), plan.max_entries, plan.multi_expansion)) {
    uint32_t max_slots = 0;
    size_t n_big = 0;
    const size_t n_items = count_daat_items(m, plan, &max_slots, &n_big);
    if (n_items && n_items < 0xFFFFFFF0ull) {
      enqueue_daat_host(m, sc, boosts, plan, n_items, max_slots, top_k, d_keys, d_scores, d_counts, st, n_big);
      TT("k1d batch");
      return;
    }
  }
  if (sc.kind == PS_SCORER_ZERO_TO_ONE && rows == nullptr) {
    if (enqueue_daat_z_host(m, sc, boosts, plan, top_k, d_keys, d_scores, d_counts, st)) {
      TT("k1dz batch");
      return;
    }
    // A mixed batch: the queries K1dz takes (simple, <= 8 lists) go there, the others through the streaming kernels -
    // two sub-batches, each kernel's merge writing its queries' rows of the caller's block (KParams::out_row).  (Whole, the
    // batch would take the streaming kernels: 2.1 ms against 0.36 ms on C3's shape.)
    if (m.tune.daat && m.tune.daat_z && m.tune.daat_z_split && B >= m.tune.daat_min_batch && m.snap->F <= 4 && m.snap->n_ids > 0) {
      std::vector<uint32_t> rows_a, rows_b;
      for (size_t q = 0; q < B; ++q) (z_query_fits_k1dz(plan, q) ? rows_a : rows_b).push_back((uint32_t)q);
      if (rows_a.size() >= m.tune.daat_min_batch && !rows_b.empty()) {
        const Plan plan_a = sub_plan_of(plan, rows_a), plan_b = sub_plan_of(plan, rows_b);
        if (enqueue_daat_z_host(m, sc, boosts, plan_a, top_k, d_keys, d_scores, d_counts, st, &rows_a)) {
          bool second_ok = true;
          try {
            enqueue_topk(m, sc, boosts, plan_b, top_k, d_keys, d_scores, d_counts, st, sync_path, &rows_b);
          } catch (const std::exception&) {
            // (staging the second half failed - an allocation, a length limit - with the first half already in flight: the
            // caller's block must not stay half-written, so the WHOLE batch goes through the streaming kernels below; they
            // wait for the K1dz context and write every row, the first half's with the same bits)
            second_ok = false;
          }
          if (second_ok) {
            TT("split batch");
            return;
          }
        }
      }
    }
  }
  if (m.tail_pending && m.tail_stream != st) PS_HIP(hipStreamWaitEvent(st, m.ev[0], 0));
  m.tail_pending = false;
  const bool timed = !sync_path || B >= 8 || time_all;
  EngineImpl::KTimer* kt = nullptr;
  Stage* row_slot = nullptr;
  try {
  stage_plan(m, sc, boosts, plan, st, kp, true, sync_path);
  TT("stage_plan");
  if (rows) {  // the merge kernel reads the rows from a pinned, device-mapped slot (fenced behind it below)
    row_slot = &m.stage[m.next_stage];
    m.next_stage = (m.next_stage + 1) % N_STAGE;
    if (row_slot == m.cur_stage) {  // (the slot stage_plan has just filled: the next one)
      row_slot = &m.stage[m.next_stage];
      m.next_stage = (m.next_stage + 1) % N_STAGE;
    }
    row_slot->ensure(B * 4 + 16);
    memcpy(row_slot->p, rows->data(), B * 4);
    kp.out_row = reinterpret_cast<const uint32_t*>(row_slot->dp);
  }
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
  m.last_kt = nullptr;
  if (timed) {
    m.rebase_timers_if_stale(m.stream);
    kt = &m.kt[m.next_kt];
    m.next_kt = (m.next_kt + 1) % N_KTIMER;
    m.harvest(*kt, true);
    kt->split = false;
    TT("harvest");
    PS_HIP(hipEventRecord(kt->a, st));
  }
  launch_score<false>(m, sc, plan, kp, m.n_cu, st, timed ? kt->m : nullptr);
  } catch (...) {
    forget_rows(m);
    throw;
  }
  TT("launch");
  if (timed) {
    PS_HIP(hipEventRecord(kt->b, st));
    kt->pending = true;
    m.last_kt = kt;
  }
  if (B) {
    // one wave per ~4 wave-wide candidate loads, at most MERGE_WAVES
    const uint32_t mw = (uint32_t)std::min<size_t>(MERGE_WAVES, std::max<size_t>(1, ((size_t)kp.n_super * top_k + 255) / 256));
    hipLaunchKernelGGL(k_merge, dim3((uint32_t)B), dim3(WAVE * mw), 0, st, kp);
    PS_HIP(hipGetLastError());
    m.ctl_clean = true;  // k_merge zeroes the control words behind itself
  }
  if (row_slot) {
    PS_HIP(hipEventRecord(row_slot->done, st));
    row_slot->pending = true;
  }
  if (m.cur_zero_copy && !sync_path) {  // the kernels read the slot in place: fence it behind them
    PS_HIP(hipEventRecord(m.cur_stage->done, st));
    m.cur_stage->pending = true;
  }
  if (!sync_path) {  // a synchronous caller leaves nothing in flight
    PS_HIP(hipEventRecord(m.ev[0], st));
    m.tail_stream = st;
    m.tail_pending = true;
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
  float b = 0, r = 0;
  if (m.last_kt && hipEventElapsedTime(&b, m.last_kt->m, m.last_kt->b) != hipSuccess) b = 0;
  if (m.last_kt && hipEventElapsedTime(&r, m.last_kt->a, m.last_kt->split ? m.last_kt->r : m.last_kt->m) != hipSuccess) r = 0;
  stats.h2d_ms = 0;
  stats.score_kernel_ms = b;  // the posting-accumulate kernel alone
  stats.kernel_ms = b + r;    // ... plus K0 / K0b in front of it
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

namespace {

// Uploads the frozen trie, the per-term tables (df, idf, lengths, layer ranges) and the per-layer
// tables for the device-side planner.  idf and the expansion-boost table are computed HERE with the
// host's libm, by the very expressions of Snapshot::plan_query - the device never calls log.
void ensure_dev_trie(EngineImpl& m) {
  if (m.dev_trie_valid) return;
  const Snapshot& s = *m.snap;
  const size_t nn = s.fnodes.size(), nt = s.terms.size(), nl = s.layers.size();
  // (a delta that brought no new term leaves the frozen trie as it was: only the per-term / per-layer tables follow the new state)
  const bool structure = !m.dev_trie_struct_valid;
  std::vector<uint4> fn(structure ? nn : 0), la(std::max<size_t>(nl, 1)), lb(std::max<size_t>(nl, 1));
  for (size_t i = 0; i < fn.size(); ++i) fn[i] = make_uint4(s.fnodes[i].child_begin, s.fnodes[i].child_count, s.fnodes[i].term_begin, s.fnodes[i].term_end);
  for (size_t l = 0; l < nl; ++l) {
    const LayerInfo& L = s.layers[l];
    la[l] = make_uint4((uint32_t)L.post_off, (uint32_t)(L.post_off >> 32), L.len, L.tbl_off);
    lb[l] = make_uint4(L.shift, L.bm_off, L.next, 0u);
  }
  std::vector<uint32_t> meta(std::max<size_t>(nt, 1) * 4), delta(std::max<size_t>(nt, 1));
  std::vector<uint64_t> df(std::max<size_t>(nt, 1));
  std::vector<double> idf(std::max<size_t>(nt, 1));
  uint32_t max_len = 0;
  for (size_t o = 0; o < nt; ++o) {
    const TermInfo& t = s.terms[o];
    meta[4 * o] = t.byte_len; meta[4 * o + 1] = t.first_layer; meta[4 * o + 2] = t.n_layers; meta[4 * o + 3] = t.fnode;
    delta[o] = t.delta_head;
    df[o] = t.df_raw;
    // BM25::before_each, bm25.rs:41-56 (as in Snapshot::plan_query)
    const uint64_t frequency = std::min<uint64_t>(s.n_docs, t.df_raw);
    const uint64_t diff = s.n_docs - frequency;
    idf[o] = std::log(1.0 + ((double)diff + 0.5) / ((double)frequency + 0.5));
    max_len = std::max(max_len, t.byte_len);
  }
  m.eb_n = max_len + 2;
  std::vector<double> eb(m.eb_n);
  for (uint32_t d = 0; d < m.eb_n; ++d) eb[d] = std::log(1.0 + (1.0 / (1.0 + (double)d)));  // bm25.rs:48-53, (1 + len_exp) - len_q == 1 + d
  auto up = [&](auto& buf, const auto& v) {
    buf.ensure(v.size() + 1);
    PS_HIP(hipMemcpy(buf.p, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice));
  };
  if (structure) up(m.d_fnodes, fn);
  up(m.d_layer_a, la); up(m.d_layer_b, lb); up(m.d_term_meta, meta); up(m.d_term_delta, delta);
  up(m.d_term_df, df); up(m.d_term_idf, idf); up(m.d_eb_table, eb);
  {  // per list (layer) the idf of its term: what the score planes are built with
    std::vector<double> lidf(std::max<size_t>(nl, 1), 0.0);
    for (size_t o = 0; o < nt; ++o) {
      const TermInfo& t = s.terms[o];
      for (uint32_t l = 0; l < t.n_layers; ++l) lidf[t.first_layer + l] = idf[o];
      for (uint32_t l = t.delta_head; l != 0xFFFFFFFFu && l < nl; l = s.layers[l].next) lidf[l] = idf[o];
    }
    up(m.d_layer_idf, lidf);
  }
  if (structure) {
  std::vector<uint32_t> fc(s.fchar.begin(), s.fchar.end()), fd(s.fchild.begin(), s.fchild.end());
  if (fc.empty()) { fc.push_back(0); fd.push_back(0); }
  up(m.d_fchar, fc); up(m.d_fchild, fd);
  {  // per node the set of its child characters below U+0100 (children are sorted by character: position = bits below)
    std::vector<uint4> fb(std::max<size_t>(nn, 1) * 2, make_uint4(0, 0, 0, 0));
    for (size_t i = 0; i < nn; ++i) {
      uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (uint32_t k = 0; k < s.fnodes[i].child_count; ++k) {
        const uint32_t ch = s.fchar[s.fnodes[i].child_begin + k];
        if (ch < 256u) w[ch >> 5] |= 1u << (ch & 31u);
      }
      fb[2 * i] = make_uint4(w[0], w[1], w[2], w[3]);
      fb[2 * i + 1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
    up(m.d_fbits, fb);
  }
  m.dev_trie_struct_valid = true;
  }
  if (!m.h_totals) {
    PS_HIP(hipHostMalloc((void**)&m.h_totals, N_DCTX * sizeof(PlanTotals), hipHostMallocMapped | hipHostMallocCoherent));
    PS_HIP(hipHostGetDevicePointer((void**)&m.d_totals_mapped, m.h_totals, 0));
  }
  m.dev_trie_valid = true;
}

// Plans a flat BM25 batch on the device into context `c`, in two halves.  device_plan_begin: text -> k_plan count
// pass -> k_plan_scan (planning stream; nothing waits).  device_plan_finish: the host reads the totals - one short
// wait for the context's `counted` event unless the batch was announced ahead (Engine::plan_ahead), in which case the
// count pass finished long ago - and enqueues the fill pass on the preparation stream.
void device_plan_begin(EngineImpl& m, EngineImpl::DaatCtx& c, const char* text, const uint64_t* offsets, size_t B) {
  ensure_dev_trie(m);
  EngineImpl::PlanSet& ps_ = c.plan;
  hipStream_t st = m.plan_stream;
  const size_t n_bytes = B ? (size_t)offsets[B] : 0;
  if (n_bytes >= 0xFFFFFFF0ull) throw std::length_error("device planner: more than 4 GiB of query text");
  // k_plan follows the offsets without further checks: they must ascend and stay inside the text
  for (size_t q = 0; q < B; ++q)
    if (offsets[q] > offsets[q + 1] || offsets[q + 1] > offsets[B]) throw std::invalid_argument("query offsets must be non-decreasing and end at offsets[n_queries]");
  const size_t off_bytes = (B + 1) * 8, text_at = (off_bytes + 15) & ~(size_t)15;
  ps_.h.ensure(text_at + n_bytes + 16);  // (its previous copy finished before the previous totals were read)
  memcpy(ps_.h.p, offsets, off_bytes);
  if (n_bytes) memcpy(ps_.h.p + text_at, text, n_bytes);
  // offsets | text in one image, copied by a kernel that reads the device-mapped pinned slot (the copy
  // engine's hand-over costs tens of microseconds per transfer, and the host waits for this stream below)
  const size_t n16 = (text_at + n_bytes + 15) / 16;
  ps_.qtext.ensure(n16 * 16 + 64);
  hipLaunchKernelGGL(k_upload, dim3((uint32_t)std::max<size_t>(1, std::min<size_t>(256, (n16 + 255) / 256))), dim3(256), 0, st,
                     reinterpret_cast<const uint4*>(ps_.h.dp), reinterpret_cast<uint4*>(ps_.qtext.p), n16);
  const uint64_t* d_qoff = reinterpret_cast<const uint64_t*>(ps_.qtext.p);
  const char* d_qtext = ps_.qtext.p + text_at;
  ps_.cnt.ensure(B + 1); ps_.qtl.ensure(B + 1); ps_.nterms.ensure(B + 1); ps_.multi.ensure(B + 1); ps_.items.ensure(B + 1);
  ps_.post.ensure(B + 1); ps_.qbeg.ensure(B + 2); ps_.qorder.ensure(B + 1);
  DevTrie t{m.d_fnodes.p, m.d_fchar.p, m.d_fchild.p, m.d_term_df.p, m.d_term_meta.p, m.d_term_delta.p, m.d_term_idf.p,
            m.d_layer_a.p, m.d_layer_b.p, m.d_eb_table.p, m.eb_n, m.d_fbits.p};
  const uint32_t blocks = (uint32_t)((B + PLAN_WAVES - 1) / PLAN_WAVES);
  ps_.tok_node.ensure(B * (size_t)WAVE + 1);
  hipLaunchKernelGGL((k_plan<false>), dim3(std::max(1u, blocks)), dim3(WAVE * PLAN_WAVES), 0, st, t, d_qtext, d_qoff, (uint32_t)B, nullptr,
                     nullptr, ps_.cnt.p, ps_.qtl.p, ps_.nterms.p, ps_.multi.p, ps_.post.p, nullptr, ps_.items.p, m.tune.daat_chunk,
                     m.tune.daat_split_div, ps_.tok_node.p, 0u);
  // (the totals are written straight into pinned, device-mapped host memory: no copy-engine transfer to wait for)
  const int ci = (int)(&c - m.dctx);
  hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(WAVE), 0, st, ps_.cnt.p, ps_.nterms.p, ps_.multi.p, ps_.post.p, ps_.items.p, (uint32_t)B,
                     ps_.qbeg.p, m.d_totals_mapped + ci);
  PS_HIP(hipGetLastError());
  PS_HIP(hipEventRecord(c.counted, st));
}

PlanTotals device_plan_totals(EngineImpl& m, EngineImpl::DaatCtx& c) {
  {  // latency-oriented wait for the count pass (poll briefly, then block)
    const double t0 = now_ms();
    hipError_t e = hipErrorNotReady;
    while (now_ms() - t0 < 0.5 && (e = hipEventQuery(c.counted)) == hipErrorNotReady) {}
    if (e != hipSuccess) PS_HIP(hipEventSynchronize(c.counted));
  }
  return m.h_totals[&c - m.dctx];
}

// The fill pass; zmode: zero_to_one's record score and the expanded term's node in every entry (K1dz; plan_token)
void device_plan_fill(EngineImpl& m, EngineImpl::DaatCtx& c, size_t B, const PlanTotals& tot, uint32_t zmode) {
  EngineImpl::PlanSet& ps_ = c.plan;
  ps_.entries.ensure((size_t)tot.n_entries + 1);
  hipStream_t st = m.prep_stream;  // the fill pass and everything behind it: preparation stream
  PS_HIP(hipStreamWaitEvent(st, c.counted, 0));
  const size_t off_bytes = (B + 1) * 8, text_at = (off_bytes + 15) & ~(size_t)15;
  const uint64_t* d_qoff = reinterpret_cast<const uint64_t*>(ps_.qtext.p);
  const char* d_qtext = ps_.qtext.p + text_at;
  DevTrie t{m.d_fnodes.p, m.d_fchar.p, m.d_fchild.p, m.d_term_df.p, m.d_term_meta.p, m.d_term_delta.p, m.d_term_idf.p,
            m.d_layer_a.p, m.d_layer_b.p, m.d_eb_table.p, m.eb_n, m.d_fbits.p};
  const uint32_t blocks = (uint32_t)((B + PLAN_WAVES - 1) / PLAN_WAVES);
  hipLaunchKernelGGL((k_plan<true>), dim3(std::max(1u, blocks)), dim3(WAVE * PLAN_WAVES), 0, st, t, d_qtext, d_qoff, (uint32_t)B, ps_.qbeg.p,
                     ps_.entries.p, nullptr, nullptr, nullptr, nullptr, nullptr, ps_.qorder.p, nullptr, m.tune.daat_chunk,
                     m.tune.daat_split_div, ps_.tok_node.p, zmode);
  PS_HIP(hipGetLastError());
}

PlanTotals device_plan_finish(EngineImpl& m, EngineImpl::DaatCtx& c, size_t B) {
  const PlanTotals tot = device_plan_totals(m, c);
  device_plan_fill(m, c, B, tot, 0u);
  return tot;
}

PlanTotals device_plan(EngineImpl& m, EngineImpl::DaatCtx& c, const char* text, const uint64_t* offsets, size_t B) {
  device_plan_begin(m, c, text, offsets, B);
  return device_plan_finish(m, c, B);
}

// The context of a batch announced ahead, if this is that batch (same query count, same offsets, same text);
// otherwise the announced batch is dropped (its context goes back into the rotation behind its count pass).
EngineImpl::DaatCtx* take_ahead(EngineImpl& m, const char* text, const uint64_t* offsets, size_t B) {
  if (m.aheads.empty()) return nullptr;
  const EngineImpl::Ahead ah = m.aheads.front();  // (announcements are taken in the order they were made)
  EngineImpl::DaatCtx& c = m.dctx[ah.ctx];
  if (offsets != nullptr && ah.B == B) {
    const size_t n_bytes = B ? (size_t)offsets[B] : 0, off_bytes = (B + 1) * 8, text_at = (off_bytes + 15) & ~(size_t)15;
    // (a knob changed since the announcement - ps_set_option; refresh_tuning ran just before this - means the count
    // pass's item total was computed under another chunking rule than the preparation kernels will apply: plan again)
    if (ah.tune_gen == m.tune_gen && ah.n_bytes == n_bytes && memcmp(c.plan.h.p, offsets, off_bytes) == 0 &&
        (n_bytes == 0 || memcmp(c.plan.h.p + text_at, text, n_bytes) == 0)) {
      m.aheads.pop_front();
      // what acquire_ctx orders for a fresh context, for whatever was enqueued since the announcement: a k_score /
      // full-result batch still in flight shares the engine's per-batch buffers and tables
      if (m.tail_pending) {
        PS_HIP(hipStreamWaitEvent(m.prep_stream, m.ev[0], 0));
        PS_HIP(hipStreamWaitEvent(m.plan_stream, m.ev[0], 0));
      }
      return &c;
    }
  }
  drop_ahead(m);
  return nullptr;
}

// The parts of KParams every top-k launch over this snapshot shares.
void fill_common_kparams(EngineImpl& m, const ps_scorer_desc& sc, const double* boosts, size_t B, uint32_t max_qterms, KParams& kp) {
  const Snapshot& s = *m.snap;
  memset(&kp, 0, sizeof(kp));
  kp.doc = m.d_doc; kp.tf = m.d_tf; kp.fl = m.d_fl; kp.tfl = m.d_tfl; kp.table = m.d_table; kp.keys = m.d_keys; kp.bits = m.d_bits;
  kp.alive = s.any_dead ? m.d_alive : nullptr;
  kp.work_counter = m.d_work;
  kp.wstats = m.d_wstats;
  kp.P = s.P;
  while ((1u << kp.t_log2) < s.T) ++kp.t_log2;
  kp.B = (uint32_t)B; kp.n_tiles = s.n_tiles; kp.T = s.T; kp.n_docs = (uint32_t)s.n_ids; kp.F = s.F;
  kp.max_qterms = std::max<uint32_t>(1, max_qterms);
  kp.ablate = m.tune.ablate;
  kp.k1 = sc.bm25_k1; kp.b = sc.bm25_b; kp.k1p1 = sc.bm25_k1 + 1.0; kp.one_minus_b = 1.0 - sc.bm25_b;
  for (uint32_t x = 0; x < s.F; ++x) { kp.avg[x] = s.avg[x]; kp.boost[x] = boosts[x]; }
  if (sc.kind == PS_SCORER_BM25 && m.tune.lut) {
    kp.lut = m.d_lut;
    kp.lut_rows = s.lut_rows;
    kp.lut_stride = s.lut_rows ? ((s.lut_rows + 1) | 1u) : 0;
    for (uint32_t x = 0; x < s.F; ++x) { kp.lut_cap[x] = s.lut_cap[x]; kp.lut_base[x] = s.lut_base[x]; }
  }
  kp.row_stride = (uint64_t)s.tiles_cap * s.T;
}

}  // namespace

// N2: the plan of a flat BM25 batch built on the device, copied back (inspection / parity tests).
void Engine::plan_device(const char* text, const uint64_t* offsets, size_t B, Plan& out) {
  EngineImpl& m = *impl_;
  std::lock_guard<std::mutex> lock(m.mu);
  PS_HIP(hipSetDevice(m.device));
  refresh_tuning(m);
  (void)take_ahead(m, nullptr, nullptr, (size_t)-1);  // (an announced batch is dropped: this call plans its own)
  EngineImpl::DaatCtx& c = acquire_ctx(m);
  const PlanTotals tot = device_plan(m, c, text, offsets, B);
  out = Plan{};
  out.entries.resize(tot.n_entries);
  out.qbeg.resize(B + 1);
  out.qterms_len.resize(B);
  out.n_nodes.assign(B, 0);
  PS_HIP(hipStreamSynchronize(m.prep_stream));
  PS_HIP(hipStreamSynchronize(m.plan_stream));
  if (tot.n_entries) PS_HIP(hipMemcpy(out.entries.data(), c.plan.entries.p, (size_t)tot.n_entries * sizeof(ps_plan_entry), hipMemcpyDeviceToHost));
  PS_HIP(hipMemcpy(out.qbeg.data(), c.plan.qbeg.p, (B + 1) * 4, hipMemcpyDeviceToHost));
  if (B) PS_HIP(hipMemcpy(out.qterms_len.data(), c.plan.qtl.p, B * 4, hipMemcpyDeviceToHost));
  out.postings = tot.postings;
  out.max_entries = tot.max_entries;
  out.max_qterms = tot.max_qterms;
  out.multi_expansion = (tot.multi & PLAN_MULTI) != 0;
}

// Announces the next flat BM25 batch: text copy + planner count pass start now, on the planning stream, while the
// batches before it are still being scored; the query call that follows with the same text finds the totals ready
// instead of waiting for them.  Returns false when the batch would not be planned on the device anyway.
bool Engine::plan_ahead(const ps_scorer_desc& sc, const char* text, const uint64_t* offsets, size_t B) {
  EngineImpl& m = *impl_;
  const bool z = sc.kind == PS_SCORER_ZERO_TO_ONE;
  if ((sc.kind != PS_SCORER_BM25 && !z) || m.snap->F > (uint32_t)MAX_F) return false;
  std::lock_guard<std::mutex> lock(m.mu);
  PS_HIP(hipSetDevice(m.device));
  refresh_tuning(m);
  if (!(m.tune.device_plan && m.tune.daat && B >= m.tune.daat_min_batch)) return false;
  if (z && !(m.tune.daat_z && m.snap->F <= 4 && m.snap->n_ids > 0)) return false;  // (the count pass is the same for both scorers)
  // (announced batches each hold a context: at most plan_ahead_depth of them, and never so many that the rotation comes round)
  if (m.aheads.size() >= std::min<size_t>(m.tune.plan_ahead_depth, m.tune.dctx > 2 ? m.tune.dctx - 2 : 1)) return false;
  static const bool trace_host = env_u32("PS_TRACE_HOST", 0) != 0;
  const double t0 = now_ms();
  EngineImpl::DaatCtx& c = acquire_ctx(m);
  device_plan_begin(m, c, text, offsets, B);
  if (trace_host) fprintf(stderr, "[ps host] announce ctx %d: %.3f ms\n", (int)(&c - m.dctx), now_ms() - t0);
  EngineImpl::Ahead a;
  a.ctx = (int)(&c - m.dctx);
  a.B = B;
  a.n_bytes = B ? (size_t)offsets[B] : 0;
  a.tune_gen = m.tune_gen;
  m.aheads.push_back(a);
  return true;
}

bool Engine::wants_device_plan(size_t n_queries) {
  EngineImpl& m = *impl_;
  std::lock_guard<std::mutex> lock(m.mu);
  refresh_tuning(m);
  return m.tune.device_plan && m.tune.daat && n_queries >= m.tune.daat_min_batch;
}

// N2: a BM25 top-k batch whose plan never exists on the host: query text -> k_plan (count, scan, fill) ->
// device-side preparation -> K1d k_daat -> K3d k_merge_items (K1 k_score / K3 k_merge for the batches K1d
// does not take).  The host only learns the plan's totals (entries, largest plan, most query terms,
// whether any term has several expansions, work items) to size the launches.
bool Engine::run_device_planned(const ps_scorer_desc& sc, const double* boosts, const char* text, const uint64_t* offsets,
                                size_t B, size_t top_k, void* d_keys, void* d_scores, void* d_counts, void* stream,
                                ps_batch_stats& stats) {
  EngineImpl& m = *impl_;
  const Snapshot& s = *m.snap;
  const bool z = sc.kind == PS_SCORER_ZERO_TO_ONE;
  if (sc.kind != PS_SCORER_BM25 && !z) throw std::invalid_argument("the device planner handles the built-in scorers");
  if (s.F > (uint32_t)MAX_F) throw std::length_error("the GPU path supports at most 8 fields");
  if (top_k < 1 || top_k > PS_MAX_DEVICE_TOPK) throw std::invalid_argument("top_k must be in [1, 64] for the device top-k path");
  std::lock_guard<std::mutex> lock(m.mu);
  PS_HIP(hipSetDevice(m.device));
  refresh_tuning(m);
  const double t0 = now_ms();
  m.last_bounds_recomputed = false;
  hipStream_t st = stream ? (hipStream_t)stream : m.stream;
  // zero_to_one: the batches K1dz takes (ps_z21_daat.hpp) are planned here; the others go back to the caller for the
  // host planner (false), before or after the count pass has classified the queries
  if (z && !(m.tune.daat && m.tune.daat_z && m.tune.device_plan && B >= m.tune.daat_min_batch && s.F <= 4 && s.n_ids > 0)) return false;
  if (z) ensure_dev_z_bounds(m);
  static const bool trace_host = env_u32("PS_TRACE_HOST", 0) != 0;  // (stderr: where the submitting thread's time goes, per call)
  EngineImpl::DaatCtx* announced = take_ahead(m, text, offsets, B);
  const double ta = now_ms();
  EngineImpl::DaatCtx& c = announced ? *announced : acquire_ctx(m);
  if (!announced) device_plan_begin(m, c, text, offsets, B);
  const double tb = now_ms();
  const bool was_ready = hipEventQuery(c.counted) == hipSuccess;
  const PlanTotals tot = device_plan_totals(m, c);
  if (trace_host) fprintf(stderr, "[ps host] ctx %d announced %d ready %d: take %.3f begin %.3f totals %.3f ms\n", (int)(&c - m.dctx), announced ? 1 : 0,
                          was_ready ? 1 : 0, ta - t0, tb - ta, now_ms() - tb);
  if (z && !(tot.n_entries && tot.max_entries <= (uint32_t)Z_MAX_LISTS && !(tot.multi & PLAN_Z_NOT_SIMPLE) && tot.n_items &&
             tot.n_items < 0x3FFFFFF0ull)) {
    PS_HIP(hipEventRecord(c.done, m.plan_stream));  // (the context goes back into the rotation behind its count pass)
    c.busy = true;
    return false;
  }
  device_plan_fill(m, c, B, tot, z ? 1u : 0u);
  if (tot.max_qterms >= 0x7FFF) throw std::length_error("more than 32766 non-empty terms in one query");
  const double t1 = now_ms();
  EngineImpl::PlanSet& ps_ = c.plan;
  if (z) {
    // records into the record-sort order + their bounds (k_zplan_arrange), then K1dz as for a host-planned batch
    const size_t ne = tot.n_entries;
    ps_.z_ubnum.ensure(ne + 1);
    ps_.z_zub.ensure(ne * s.F + 1);
    if (z_width(tot.max_entries) <= DAAT_SMALL_MAX)
      hipLaunchKernelGGL(k_zplan_arrange<DAAT_SMALL_MAX>, dim3((uint32_t)((B + 63) / 64)), dim3(64), 0, m.prep_stream, ps_.entries.p, ps_.qbeg.p,
                         ps_.qtl.p, (uint32_t)B, s.F, m.d_z_maxtf.p, m.d_z_minfl.p, ps_.z_ubnum.p, ps_.z_zub.p);
    else
      hipLaunchKernelGGL(k_zplan_arrange<Z_MAX_LISTS>, dim3((uint32_t)((B + 63) / 64)), dim3(64), 0, m.prep_stream, ps_.entries.p, ps_.qbeg.p,
                         ps_.qtl.p, (uint32_t)B, s.F, m.d_z_maxtf.p, m.d_z_minfl.p, ps_.z_ubnum.p, ps_.z_zub.p);
    PS_HIP(hipGetLastError());
    ZBatch zb;
    zb.zn = z_width(tot.max_entries);
    zb.d_ubnum = ps_.z_ubnum.p;
    zb.d_zub = ps_.z_zub.p;
    z_levels(m, zb);
    enqueue_daat(m, c, sc, boosts, ps_.entries.p, ps_.qbeg.p, ps_.qtl.p, B, ne, tot.max_qterms, tot.max_entries, false, (size_t)tot.n_items, 0u,
                 top_k, d_keys, d_scores, d_counts, st, &zb);
  } else {
  const bool multi = (tot.multi & PLAN_MULTI) != 0;
  if (daat_eligible(m, sc, boosts, B, tot.n_entries, tot.max_entries, multi) && tot.n_items && tot.n_items < 0xFFFFFFF0ull) {
    enqueue_daat(m, c, sc, boosts, ps_.entries.p, ps_.qbeg.p, ps_.qtl.p, B, tot.n_entries, tot.max_qterms, tot.max_entries, multi,
                 (size_t)tot.n_items, 0u, top_k, d_keys, d_scores, d_counts, st, nullptr, nullptr, (size_t)tot.n_items_big);
  } else {
    // the batches K1d does not take: K1 k_score / K3 k_merge from the device-built plan, in the engine's
    // single set of per-batch buffers, once nothing else is in flight
    PS_HIP(hipEventRecord(c.prepared, m.prep_stream));
    wait_daat_contexts(m, st);
    if (m.tail_pending && m.tail_stream != st) PS_HIP(hipStreamWaitEvent(st, m.ev[0], 0));
    m.tail_pending = false;
    PS_HIP(hipStreamWaitEvent(st, c.prepared, 0));
    Plan shape;  // the scalars the launch geometry needs; the entries stay on the device
    shape.max_entries = tot.max_entries;
    shape.max_qterms = tot.max_qterms;
    shape.multi_expansion = multi;
    shape.postings = tot.postings;
    KParams kp;
    try {
      fill_common_kparams(m, sc, boosts, B, shape.max_qterms, kp);
      kp.plan = ps_.entries.p;
      kp.qbeg = ps_.qbeg.p;
      kp.qterms_len = ps_.qtl.p;
      kp.qorder = ps_.qorder.p;
      const size_t n_thr = B + 2;
      const bool fresh = m.d_gthr.ensure(n_thr, true);
      kp.gthr = m.d_gthr.p;
      if (!(m.ctl_clean && !fresh)) {
        PS_HIP(hipMemsetAsync(m.d_gthr.p, 0, n_thr * 8, st));
        PS_HIP(hipMemsetAsync(m.d_work, 0, 256, st));
      }
      m.ctl_clean = false;
      BatchImage img;
      img.B = B;
      choose_run_length(m, sc, shape, img, true, kp);
      kp.K = (uint32_t)top_k;
      const size_t n_cand = (size_t)B * kp.n_super * top_k;
      m.d_cand_score.ensure(n_cand + 1);
      m.d_cand_doc.ensure(n_cand + 1);
      kp.cand_score = m.d_cand_score.p;
      kp.cand_doc = m.d_cand_doc.p;
      kp.out_keys = (uint64_t*)d_keys; kp.out_scores = (double*)d_scores; kp.out_counts = (uint32_t*)d_counts;
      m.rebase_timers_if_stale(m.stream);
      EngineImpl::KTimer* kt = &m.kt[m.next_kt];
      m.next_kt = (m.next_kt + 1) % N_KTIMER;
      m.harvest(*kt, true);
      kt->split = false;
      PS_HIP(hipEventRecord(kt->a, st));
      m.build_slots.clear();
      launch_score<false>(m, sc, shape, kp, m.n_cu, st, kt->m);
      PS_HIP(hipEventRecord(kt->b, st));
      kt->pending = true;
      m.last_kt = kt;
      if (B) {
        const uint32_t mw = (uint32_t)std::min<size_t>(MERGE_WAVES, std::max<size_t>(1, ((size_t)kp.n_super * top_k + 255) / 256));
        hipLaunchKernelGGL(k_merge, dim3((uint32_t)B), dim3(WAVE * mw), 0, st, kp);
        PS_HIP(hipGetLastError());
        m.ctl_clean = true;
      }
    } catch (...) {
      forget_rows(m);
      throw;
    }
    PS_HIP(hipEventRecord(c.done, st));  // the context's plan buffers are free again behind this batch
    c.busy = true;
    PS_HIP(hipEventRecord(m.ev[0], st));
    m.tail_stream = st;
    m.tail_pending = true;
  }
  }
  memset(&stats, 0, sizeof(stats));
  stats.n_queries = B;
  stats.n_plan_entries = tot.n_entries;
  stats.postings_visited = tot.postings;
  stats.algorithmic_bytes = tot.postings * (4 + 8 * (uint64_t)s.F) + (uint64_t)B * top_k * 16;
  stats.plan_ms = t1 - t0;  // device planner incl. its one synchronisation (of the context's stream)
  if (trace_host) fprintf(stderr, "[ps host] call %.3f ms (plan part %.3f, enqueue %.3f)\n", now_ms() - t0, t1 - t0, now_ms() - t1);
  stats.device_planned = 1;
  stats.bounds_recomputed = m.last_bounds_recomputed ? 1u : 0u;
  if (!stream) {
    PS_HIP(hipStreamSynchronize(st));
    read_kernel_times(m, stats);
  }
  stats.total_ms = now_ms() - t0;
  return true;
}

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
  enqueue_topk(m, sc, boosts, plan, top_k, d_keys, d_scores, d_counts, st, false);
  memset(&stats, 0, sizeof(stats));
  fill_stats(m, stats, s, plan, (uint64_t)(plan.qbeg.size() - 1) * top_k);
  if (!stream) {
    PS_HIP(hipStreamSynchronize(st));
    read_kernel_times(m, stats);
  }
  stats.total_ms = now_ms() - t0;
}

void Engine::run_host(const ps_scorer_desc& sc, const double* boosts, const Plan& plan, size_t top_k,
                      ResultBuf& out, std::vector<size_t>& offsets, ps_batch_stats& stats) {
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
    // keys | scores | counts in one block.  Small result sets are written by k_merge straight
    // into the pinned (device-mapped, coherent) download buffer; larger ones take one D2H copy.
    const size_t res_bytes = nb * 16 + B * 4;
    m.result.ensure(res_bytes + 64);
    const bool direct = res_bytes <= 16384 && m.tune.zero_copy;
    unsigned char* dres;
    if (direct) {
      dres = m.result.dp;
    } else {
      m.d_out_keys.ensure(res_bytes / 8 + 2);
      dres = reinterpret_cast<unsigned char*>(m.d_out_keys.p);
    }
    try {
      enqueue_topk(m, sc, boosts, plan, top_k, dres, dres + nb * 8, dres + nb * 16, st, true);
      if (!direct && res_bytes) PS_HIP(hipMemcpyAsync(m.result.p, dres, res_bytes, hipMemcpyDeviceToHost, st));
      sync_stream(st);
    } catch (...) {
      (void)hipStreamSynchronize(st);  // nothing may still be reading the staging slot
      throw;
    }
    uint64_t* hk = reinterpret_cast<uint64_t*>(m.result.p);
    double* hs = reinterpret_cast<double*>(m.result.p + nb * 8);
    uint32_t* hc = reinterpret_cast<uint32_t*>(m.result.p + nb * 16);
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
    cap[q + 1] = cap[q] + std::min<uint64_t>(sum, s.n_ids);
  }
  const uint64_t total_cap = cap[B];
  const uint64_t budget = (uint64_t)m.tune.full_budget_mb << 20;
  if (B <= 1 && total_cap >= 0xFFFFFFF0ull) throw std::length_error("full-result batch too large for one pass");
  if ((total_cap * 12 > budget || total_cap >= 0xFFFFFFF0ull) && B > 1) {
    // keep the result buffers bounded: run the two halves of the batch one after the other
    const size_t half = B / 2;
    ResultBuf o2;
    std::vector<size_t> f1, f2;
    ps_batch_stats s1, s2;
    run_host(sc, boosts, sub_plan(plan, 0, half), top_k, out, f1, s1);
    run_host(sc, boosts, sub_plan(plan, half, B), top_k, o2, f2, s2);
    out.append(o2);
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
  if (m.tail_pending && m.tail_stream != st) PS_HIP(hipStreamWaitEvent(st, m.ev[0], 0));
  m.tail_pending = false;
  drop_ahead(m);
  try {
  stage_plan(m, sc, boosts, plan, st, kp, false, false);
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
  memcpy(h_off, cap.data(), (B + 1) * 8);
  PS_HIP(hipMemcpyAsync(m.d_full_off.p, h_off, (B + 1) * 8, hipMemcpyHostToDevice, st));
  PS_HIP(hipMemsetAsync(m.d_full_cnt.p, 0, (B + 1) * 4, st));
  m.rebase_timers_if_stale(m.stream);
  EngineImpl::KTimer& kt = m.kt[m.next_kt];
  m.last_kt_pending = &kt;
  m.next_kt = (m.next_kt + 1) % N_KTIMER;
  m.harvest(kt, true);
  kt.split = false;
  PS_HIP(hipEventRecord(kt.a, st));
  launch_score<true>(m, sc, plan, kp, m.n_cu, st, kt.m);
  } catch (...) {
    forget_rows(m);
    throw;
  }
  EngineImpl::KTimer& kt = *m.last_kt_pending;
  uint32_t* h_cnt = reinterpret_cast<uint32_t*>(m.result.p + (B + 1) * 8);
  PS_HIP(hipEventRecord(kt.b, st));
  kt.pending = true;
  m.last_kt = &kt;
  if (m.cur_zero_copy) {  // the kernel reads the staging slot in place: fence it
    PS_HIP(hipEventRecord(m.cur_stage->done, st));
    m.cur_stage->pending = true;
  }
  // K4 (query.rs:97-105, "materialise + sort"): canonical order (score desc, doc id asc == key asc)
  // of every query's run, on the device (ps_sort.hip)
  // few or huge runs: one set of device-wide sorts over all runs (needs the counts); many small runs: one segmented sort
  const bool few_runs = B <= 8 || total_cap / B > 32768;
  if (total_cap && B && !few_runs) {
    m.d_sort_doc.ensure(total_cap + 1);
    m.d_sort_score.ensure(total_cap + 1);
    m.d_seg.ensure(2 * (B + 1));
    SortBuffers sb{m.d_full_doc.p, reinterpret_cast<uint64_t*>(m.d_full_score.p), m.d_sort_doc.p, m.d_sort_score.p,
                   m.d_seg.p, m.d_seg.p + B + 1};
    size_t tb = 0;
    PS_HIP(sort_results(sb, (unsigned)total_cap, (unsigned)B, m.d_full_off.p, m.d_full_cnt.p, nullptr, tb, st));
    m.d_sort_tmp.ensure(tb + 256);
    PS_HIP(sort_results(sb, (unsigned)total_cap, (unsigned)B, m.d_full_off.p, m.d_full_cnt.p, m.d_sort_tmp.p, tb, st));
  }
  PS_HIP(hipMemcpyAsync(h_cnt, m.d_full_cnt.p, (B + 1) * 4, hipMemcpyDeviceToHost, st));
  sync_stream(st);
  std::vector<uint32_t> cnt(h_cnt, h_cnt + B);
  read_kernel_times(m, stats);
  static const bool ftrace = getenv("PS_FULL_TRACE") != nullptr;
  const double t_scored = now_ms();
  // the first `keep` results of every run as {key, score} records
  size_t total = 0, found = 0;
  std::vector<uint64_t> cmp_off(B + 1, 0);
  for (size_t q = 0; q < B; ++q) {
    offsets[q] = total;
    total += (top_k ? std::min<size_t>(top_k, cnt[q]) : cnt[q]);
    cmp_off[q] = found;
    found += cnt[q];
  }
  offsets[B] = total;
  cmp_off[B] = found;
  if (found >= 0xFFFFFFF0ull) throw std::length_error("full-result batch too large for one pass");
  const size_t bytes = total * sizeof(ps_result);
  // large blocks: pinned, written by the device, handed to the caller as they are (ps_free recycles them)
  const bool pinned = bytes >= ((size_t)m.tune.result_pinned_min_kb << 10) && out.resize_pinned(total);
  if (!pinned) out.resize(total);
  double t_packed = t_scored;
  bool downloaded = false;
  if (total) {
    m.d_pack_off.ensure(2 * (B + 1));
    m.d_pack.ensure(total);
    m.result.ensure(std::max<size_t>(2 * (B + 1) * 8, pinned ? 0 : bytes) + 64);
    uint64_t* h_po = reinterpret_cast<uint64_t*>(m.result.p);
    for (size_t q = 0; q <= B; ++q) { h_po[q] = offsets[q]; h_po[B + 1 + q] = cmp_off[q]; }
    PS_HIP(hipMemcpyAsync(m.d_pack_off.p, h_po, 2 * (B + 1) * 8, hipMemcpyHostToDevice, st));
    if (few_runs) {
      m.d_gs_u32.ensure(4 * found + 4);
      m.d_gs_u64.ensure(3 * found + 3);
      GlobalSort gs{};
      gs.doc = m.d_full_doc.p;
      gs.score_bits = reinterpret_cast<const uint64_t*>(m.d_full_score.p);
      gs.run_off = m.d_full_off.p;
      gs.cmp_off = m.d_pack_off.p + B + 1;
      gs.n_docs = s.n_ids;
      gs.kd = m.d_gs_u32.p;
      gs.k32 = m.d_gs_u32.p + found;
      gs.ia = m.d_gs_u32.p + 2 * found;
      gs.ib = m.d_gs_u32.p + 3 * found;
      gs.sc = m.d_gs_u64.p;
      gs.ks = m.d_gs_u64.p + found;
      gs.ks_out = m.d_gs_u64.p + 2 * found;
      // Large batches in up to 4 parts of whole runs: the download of a part (the link: 57 GB/s, more than half of
      // a large batch's time) runs on the copy stream while the next part is sorted.
      const size_t n_parts = (pinned && bytes >= ((size_t)m.tune.full_parts_min_kb << 10)) ? std::min<size_t>(4, B) : 1;
      if (n_parts > 1 && !m.copy_stream) {
        PS_HIP(hipStreamCreateWithFlags(&m.copy_stream, hipStreamNonBlocking));
        for (auto& e : m.part_done) PS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      }
      {  // the sorts' temporary storage once, for the largest part there can be (no reallocation between parts)
        gs.run0 = 0; gs.n_runs = (uint32_t)B; gs.first = 0; gs.n = (uint32_t)found;
        size_t tb = 0;
        PS_HIP(sort_runs_global(gs, nullptr, tb, st));
        m.d_sort_tmp.ensure(tb + 256);
      }
      size_t qa = 0;
      try {
      for (size_t part = 0; part < n_parts; ++part) {
        size_t qb = qa;
        if (part + 1 == n_parts) {
          qb = B;
        } else {  // whole runs up to this part's share of the results
          const uint64_t goal = (uint64_t)found * (part + 1) / n_parts;
          while (qb < B && (cmp_off[qb + 1] <= goal || qb == qa)) ++qb;
          qb = std::min(qb, B - (n_parts - 1 - part));
          qb = std::max(qb, qa + 1);
        }
        gs.run0 = (uint32_t)qa;
        gs.n_runs = (uint32_t)(qb - qa);
        gs.first = (uint32_t)cmp_off[qa];
        gs.n = (uint32_t)(cmp_off[qb] - cmp_off[qa]);
        const size_t o0 = offsets[qa], o1 = offsets[qb];
        if (gs.n) {
          size_t tb = m.d_sort_tmp.cap;
          PS_HIP(sort_runs_global(gs, m.d_sort_tmp.p, tb, st));
          PS_HIP(pack_sorted(gs, m.d_pack_off.p, o0, o1 - o0, m.d_keys, m.d_pack.p, st));
        }
        if (n_parts > 1 && o1 > o0) {
          PS_HIP(hipEventRecord(m.part_done[part], st));
          PS_HIP(hipStreamWaitEvent(m.copy_stream, m.part_done[part], 0));
          PS_HIP(hipMemcpyAsync(out.data() + o0, m.d_pack.p + o0, (o1 - o0) * sizeof(ps_result), hipMemcpyDeviceToHost, m.copy_stream));
        }
        qa = qb;
      }
      } catch (...) {  // nothing may still be writing into the caller's block when it goes back to the pool
        if (m.copy_stream) (void)hipStreamSynchronize(m.copy_stream);
        (void)hipStreamSynchronize(st);
        throw;
      }
      if (n_parts > 1) {
        downloaded = true;
        if (ftrace) { sync_stream(st); t_packed = now_ms(); }
        PS_HIP(hipStreamSynchronize(m.copy_stream));
        sync_stream(st);
      }
    } else {
      hipLaunchKernelGGL(k_pack_results, dim3(64, (uint32_t)B), dim3(256), 0, st, m.d_full_doc.p, m.d_full_score.p,
                         m.d_full_off.p, m.d_pack_off.p, m.d_keys, m.d_pack.p);
      PS_HIP(hipGetLastError());
    }
    if (downloaded) {
    } else if (pinned) {
      if (ftrace) { sync_stream(st); t_packed = now_ms(); }
      PS_HIP(hipMemcpyAsync(out.data(), m.d_pack.p, bytes, hipMemcpyDeviceToHost, st));
      sync_stream(st);
    } else {
      sync_stream(st);  // h_po (pinned) is reused as the download target
      if (ftrace) t_packed = now_ms();
      PS_HIP(hipMemcpyAsync(m.result.p, m.d_pack.p, bytes, hipMemcpyDeviceToHost, st));
      sync_stream(st);
      memcpy(out.data(), m.result.p, bytes);
    }
    if (ftrace)
      fprintf(stderr, "[full] to scored+counts %.2f ms, sort+pack %.2f, d2h%s %.2f (%zu results, %s block)\n",
              t_scored - t0, t_packed - t_scored, pinned ? "" : " + host copy", now_ms() - t_packed, total,
              pinned ? "pinned pool" : "malloc'd");
  }
  m.wc_results += total;
  fill_stats(m, stats, s, plan, total);
  stats.total_ms = now_ms() - t0;
}

}  // namespace ps
