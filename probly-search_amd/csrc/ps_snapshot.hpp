// ps_snapshot.hpp — immutable flattened view of an Index: CSR posting planes + frozen trie +
// host query planner.  Host-only (no HIP here); ps_engine.hip owns the device side.
//
// What is flattened (reference structures: src/index.rs:342-396):
//   * documents get dense ids in ASCENDING KEY order, so "doc id asc" == the canonical tie-break
//     "key asc" of test_util::test_score (src/lib.rs:54-58);
//   * every term's linked posting list becomes one (or, if a key was re-added without removal,
//     several "version layers" of) doc-id-sorted run(s) in three kinds of u32 planes:
//       doc[P], tf[F][P] (DocumentPointer::term_frequency), fl[F][P] (DocumentDetails::field_length,
//       denormalised per posting) — 4 + 8F bytes per posting, the north-star layout;
//   * removed documents are dropped; the per-occurrence pointer count survives as df_raw
//     (== Index::count_documents, src/index.rs:282-297) because BM25's idf needs it;
//   * per list, a table of offsets at LDS-tile granularity (tile t -> slot t >> shift), so a
//     wavefront finds its doc-range slice of a list with two scalar loads and no search;
//   * the trie is renumbered in DFS pre-order (children newest-first, src/query.rs:130-147), so
//     `expand_term(prefix)` is a contiguous range of term ordinals.
#pragma once
#include <cstdint>
#include <string>
#include <string_view>
#include <memory>
#include <utility>
#include <vector>

#include "ps_index.hpp"

namespace ps {

struct TermInfo {
  uint64_t df_raw;       // live DocumentPointer count of the term
  uint32_t byte_len;     // str::len() of the term (bm25.rs:48-53, zero_to_one.rs:57-58)
  uint32_t first_layer;  // index into Snapshot::layers
  uint32_t n_layers;     // 0 if no live posting in the base lists
  uint32_t fnode;        // frozen node id (unique per term; zero_to_one's index_node_id)
  uint32_t delta_head;   // first delta layer of the term (postings of documents added after the flatten), or NO_LAYER
  uint32_t _pad;
};
constexpr uint32_t NO_LAYER = 0xFFFFFFFFu;

struct LayerInfo {
  uint64_t post_off;  // multiple of 4 (16-byte aligned planes)
  uint32_t len;
  uint32_t tbl_off;
  uint32_t shift;
  uint32_t bm_off;    // first word of the list's membership bitmap in Snapshot::bits, or NO_BITMAP
  uint32_t next;      // delta layers: next (older) delta layer of the same term, or NO_LAYER
  uint32_t _pad;
};
constexpr uint32_t NO_BITMAP = 0xFFFFFFFFu;

struct FrozenNode {
  uint32_t child_begin, child_count;  // into fchar/fchild, sorted by char
  uint32_t term_begin, term_end;      // pre-order term ordinals of the subtree
};

struct Plan {
  std::vector<ps_plan_entry> entries;
  std::vector<uint32_t> qbeg;          // per query: first entry; size B+1
  std::vector<uint32_t> qterms_len;    // per query: TermData::query_terms_len
  std::vector<uint32_t> n_nodes;       // per query: distinct nodes (zero_to_one)
  uint64_t postings = 0;               // sum of entry lens
  uint32_t max_entries = 0, max_qterms = 0, max_nodes = 0;
  bool multi_expansion = false;        // some query term owns >1 entry -> visited tags needed
};

// Plane storage: resize() leaves new elements uninitialised, so the flattener's worker threads
// take the first-touch page faults in parallel instead of one serial zero-fill of ~1 GB.
template <typename T>
struct DefaultInitAllocator : std::allocator<T> {
  template <typename U> struct rebind { using other = DefaultInitAllocator<U>; };
  DefaultInitAllocator() = default;
  template <typename U> DefaultInitAllocator(const DefaultInitAllocator<U>&) {}
  template <typename U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
  template <typename U, typename... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
using PlaneVec = std::vector<uint32_t, DefaultInitAllocator<uint32_t>>;

// Read-only view of one snapshot array: the storage is either the flattener's vectors or a
// read-only mmap of a snapshot file (the two constructors of Snapshot).
template <typename T>
struct View {
  const T* p = nullptr;
  size_t n = 0;
  const T* data() const { return p; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  const T& operator[](size_t i) const { return p[i]; }
  const T* begin() const { return p; }
  const T* end() const { return p + n; }
  const T& back() const { return p[n - 1]; }
};

struct SnapshotStorage;  // the flattener's vectors (ps_snapshot.cpp)

// What Snapshot::apply_delta changed, for the device side (Engine::apply_delta uploads exactly this).
struct DeltaRanges {
  uint64_t plane_begin = 0, plane_end = 0;  // postings appended to the planes
  uint64_t table_begin = 0, table_end = 0;  // tile-offset table entries appended
  uint64_t key_begin = 0, key_end = 0;      // doc ids appended
  std::vector<uint32_t> alive_words;        // words of the alive bitmap that changed
  uint64_t docs_added = 0, docs_removed = 0;
  bool trie_refrozen = false;
};

class Snapshot {
 public:
  // headroom_pct > 0 reserves room (documents, postings, table entries) so that apply_delta can
  // append documents in place; 0 = exact fit (removals can still be applied as a delta).
  Snapshot(const Index& idx, uint32_t tile_docs, uint32_t headroom_pct = 0);
  // Incremental re-flatten: brings the snapshot from the index epoch it was built at
  // to the index's current one by (a) clearing the alive bit of removed documents and (b) appending
  // the postings of added documents as delta layers of their terms, re-freezing the trie (cheap, no
  // posting work) when they introduced new terms.  O(changes), not O(postings).  Returns false -
  // nothing modified - when the change set cannot be expressed (vacuum, re-added or out-of-order
  // keys, headroom exhausted, snapshot loaded from a file): the caller re-flattens.
  bool apply_delta(const Index& idx, DeltaRanges& out);
  // On-disk form of the flattened snapshot (the reference has no persistence at
  // all): a versioned little-endian file whose page-aligned sections ARE the arrays below, so
  // loading is one read-only mmap (no parse, no copy; the planes go from the page cache straight
  // to the device) plus a validation pass over every offset the planner or the kernels follow.
  explicit Snapshot(const std::string& path);
  ~Snapshot();
  Snapshot(const Snapshot&) = delete;
  Snapshot& operator=(const Snapshot&) = delete;
  void save(const std::string& path) const;
  bool mapped() const { return map_base_ != nullptr; }

  // host planner: tokenise -> expand_term -> before_each (src/query.rs:29-60, bm25.rs:35-58)
  void plan_query(const ps_scorer_desc& sc, std::string_view q, ps_tokenizer_fn tok, void* user, Plan& plan) const;

  uint32_t F, T, n_tiles;
  uint32_t tiles_cap = 0;   // tiles every table / bitmap / row is laid out for (>= n_tiles; headroom)
  uint64_t n_docs;  // docs.len(): LIVE documents (BM25's N)
  uint64_t n_ids = 0;       // doc id space = keys.size(): live + removed-by-delta documents
  uint64_t P_used = 0, table_used = 0;  // filled part of the planes / the table (the rest is headroom)
  uint32_t n_delta_layers = 0;
  uint64_t n_delta_postings = 0;
  // one bit per doc id: cleared by a delta removal (the kernels drop such documents when they emit;
  // null view = every document alive)
  View<uint32_t> alive;
  bool any_dead = false;
  // df_raw of the documents removed by deltas, per layer, is subtracted from TermInfo::df_raw by the
  // device side (Engine::apply_delta counts it with one pass over the planes and calls this)
  void set_removed_df(const std::vector<uint64_t>& removed_per_layer);
  std::vector<uint64_t> count_removed_df_host() const;
  uint32_t chain_length(uint32_t head) const;
  std::vector<uint64_t> removed_df;  // per layer
  View<uint64_t> keys;
  View<double> avg;
  View<TermInfo> terms;
  View<LayerInfo> layers;
  View<FrozenNode> fnodes;  // [0] = root
  View<uint32_t> fchar, fchild;
  // CSR planes (host copy)
  uint64_t P = 0;  // padded plane length
  View<uint32_t> doc, tf, fl, table;
  // Membership bitmaps of the denser lists (len >= n_docs / 128): per 32 documents one cell
  // {bits, postings of the list before these documents}.  K1d resolves "is document d in this list,
  // and where" with ONE 8-byte load instead of a table lookup + binary search.
  View<uint32_t> bits;
  uint64_t n_postings = 0, n_pointers = 0, n_live_terms = 0;
  uint32_t max_layers = 1;
  uint64_t src_epoch = 0;
  uint64_t src_uid = 0;  // Index::uid() of the source index (0: loaded from a file)
  // geometry of the BM25 saturated-tf LUT the engine builds per batch (rows of 16 doubles):
  // field x owns rows [lut_base[x], lut_base[x] + lut_cap[x]), one per field length < lut_cap[x]
  View<uint32_t> max_fl, lut_cap, lut_base;
  uint32_t lut_rows = 0;

 private:
  int64_t find_fnode(std::string_view term) const;
  void bind(const SnapshotStorage& st);
  void validate() const;  // throws std::invalid_argument on any out-of-range offset
  std::unique_ptr<SnapshotStorage> own_;  // set when flattened from an Index
  void* map_base_ = nullptr;              // set when mapped from a file
  size_t map_bytes_ = 0;
};

}  // namespace ps
