// ps_index.hpp — host-side mutable index (the build side of probly-search's Index<T>).
//
// Observable semantics follow src/index.rs of the reference (quantleaf/probly-search 2.0.1):
// field sum/avg update rules (:112-114, :176-186), newest-first child order of the trie
// (:409-419), per-occurrence document frequency (:119-157, :282-297), lazy removal + vacuum
// (:161-241).  The data structure is NOT the reference's: postings are stored as one compact
// record per (add_document call, term) carrying the per-field term frequencies — the reference's
// c identical per-occurrence DocumentPointers are one record of multiplicity c = sum(tf) — in
// flat per-term arrays that the flattener (ps_snapshot.cpp) turns into CSR planes for the GPU.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <string_view>
#include <unordered_map>
#include <unordered_set>
#include <mutex>
#include <vector>

#include "../../include/probly_search_amd.h"

namespace ps {

constexpr int32_t NIL = -1;

struct TrieNode {
  uint32_t ch;          // Unicode scalar value (Rust `char`)
  int32_t next;         // next sibling
  int32_t first_child;  // newest child first
  int32_t list;         // index into Index::lists, NIL if this node never held a posting
  int32_t parent;
};

// All postings of one term, in add order (the reference's list order is the reverse).
struct PostingList {
  std::vector<uint64_t> keys;  // one per record
  std::vector<uint32_t> tf;    // F per record
};

struct DocDetails {
  std::vector<uint32_t> field_length;
};

struct FieldDetails {
  uint64_t sum = 0;
  double avg = 0.0;
};

std::vector<std::string_view> tokenize(std::string_view s, ps_tokenizer_fn fn, void* user,
                                       std::vector<const char*>& scratch_p, std::vector<size_t>& scratch_l);
// Decodes one UTF-8 scalar starting at s[i]; advances i.
uint32_t next_char(std::string_view s, size_t& i);
void append_utf8(std::string& s, uint32_t cp);

// term bytes -> trie node, open addressing over a byte arena: one hash of the token's bytes and
// (almost always) one probe per indexed token, no std::string temporary.
class TermCache {
 public:
  int32_t find(std::string_view t) const {
    if (slots_.empty()) return -1;
    const uint64_t h = hash(t);
    for (size_t i = h & mask_;; i = (i + 1) & mask_) {
      const Slot& s = slots_[i];
      if (s.node < 0) return -1;
      if (s.hash == h && s.len == t.size() && memcmp(arena_.data() + s.off, t.data(), t.size()) == 0) return s.node;
    }
  }
  void insert(std::string_view t, int32_t node) {
    if ((used_ + 1) * 2 > slots_.size()) grow();
    Slot s{hash(t), arena_.size(), (uint32_t)t.size(), node};
    arena_.append(t.data(), t.size());
    place(s);
    ++used_;
  }
  void clear() {
    slots_.clear();
    arena_.clear();
    used_ = 0;
    mask_ = 0;
  }

 private:
  struct Slot { uint64_t hash; size_t off; uint32_t len; int32_t node; };
  static uint64_t hash(std::string_view t) {  // FNV-1a with a final mix
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : t) { h ^= c; h *= 1099511628211ull; }
    h ^= h >> 32;
    return h * 0x9E3779B97F4A7C15ull;
  }
  void place(const Slot& s) {
    for (size_t i = s.hash & mask_;; i = (i + 1) & mask_)
      if (slots_[i].node < 0) { slots_[i] = s; return; }
  }
  void grow() {
    std::vector<Slot> old;
    old.swap(slots_);
    slots_.assign(old.empty() ? 1024 : old.size() * 2, Slot{0, 0, 0, -1});
    mask_ = slots_.size() - 1;
    for (const Slot& s : old)
      if (s.node >= 0) place(s);
  }
  std::vector<Slot> slots_;
  std::string arena_;
  size_t used_ = 0, mask_ = 0;
};

// One logged mutation (the delta-snapshot path): enough to update a flattened snapshot
// without walking the posting lists again.
struct IndexChange {
  enum Kind : uint8_t { ADD = 0, REMOVE = 1, OTHER = 2 } kind = OTHER;  // OTHER (vacuum) forces a full re-flatten
  uint64_t key = 0;
  bool was_present = false;   // ADD: the key was already a live document (re-add without removal); REMOVE: it existed
  bool was_removed = false;   // ADD: the key sits in the lazily-removed set (it stays invisible until vacuum)
  std::vector<int32_t> nodes;         // ADD: trie nodes of the document's distinct terms
  std::vector<uint32_t> tf;           // ADD: F term frequencies per node
  std::vector<uint32_t> field_length; // ADD: DocumentDetails::field_length
};

class Index {
 public:
  explicit Index(size_t fields_num, size_t expected_index_size = 1000, size_t expected_documents_count = 10000);

  void add_document(uint64_t key, const ps_str* values, const size_t* n_values, ps_tokenizer_fn tok, void* user);
  void remove_document(uint64_t key);
  void vacuum();

  // read side
  size_t fields_len() const { return fields_.size(); }
  size_t docs_len() const { return docs_.size(); }
  const FieldDetails& field(size_t i) const { return fields_[i]; }
  const DocDetails* doc(uint64_t key) const;
  size_t count_nodes() const;
  size_t live_pointers() const;
  int32_t find_node(std::string_view term) const;                // find_inverted_index_node
  long count_documents(int32_t node) const;                      // Index::count_documents
  std::vector<std::string> expand_term(std::string_view term) const;
  std::vector<uint32_t> children(int32_t node) const;

  // Bulk form of n add_document calls on an EMPTY index (single-valued fields, whitespace tokenizer,
  // distinct keys) from a corpus the GPU already tokenised and grouped (ps_build.hip): the trie is
  // built by interning the distinct terms in first-occurrence order (same nodes, same newest-first
  // child lists as the incremental build), the posting lists are filled in document order.
  // The resulting index is indistinguishable from the incrementally built one.
  void bulk_load(const struct GroupedCorpus& g, size_t n_docs, const uint64_t* keys, const char* text);
  bool pristine() const { return docs_.empty() && nodes_.size() == 1 && !has_removed_ && lists_.empty(); }

  // Index::query for a caller-supplied ScoreCalculator (PS_SCORER_HOST_CALLBACKS): the reference's
  // driver loop (src/query.rs:29-105) over this index's posting lists, calling the three
  // callbacks in the reference's order.  Results in canonical order (score desc, key asc).
  // `handle` is what the before_each callback receives as `idx`.
  void query_callbacks(const ps_score_callbacks& cb, std::string_view query, ps_tokenizer_fn tok, void* tok_user,
                       const double* fields_boost, size_t n_boost, const ps_index* handle,
                       std::vector<ps_result>& out) const;

  // flattener access
  const std::vector<TrieNode>& nodes() const { return nodes_; }
  const std::vector<PostingList>& lists() const { return lists_; }
  const std::unordered_map<uint64_t, DocDetails>& docs() const { return docs_; }
  bool is_removed(uint64_t key) const { return has_removed_ && removed_.count(key) != 0; }
  bool any_removed() const { return has_removed_ && !removed_.empty(); }
  int32_t root() const { return 0; }
  uint64_t epoch() const { return epoch_; }  // bumped by every mutation
  // Starts the change log at the current epoch (called by the flattener: a snapshot exists that can replay it).
  // (const + mutable: snapshots take a const Index&, and two threads may take snapshots of one index at once - the
  // header only asks for external exclusion around MUTATION - so the switch is guarded)
  void enable_change_log() const {
    std::lock_guard<std::mutex> lock(log_mu_);
    if (log_enabled_) return;
    log_enabled_ = true;
    log_.clear();
    log_base_ = epoch_;
    log_postings_ = 0;
  }
  uint64_t uid() const { return uid_; }      // identity of this index object (a snapshot only takes deltas from its own source)
  // The mutations after `epoch`, oldest first, or nullptr if the log no longer reaches back that far
  // (it is bounded; a snapshot that old re-flattens).  *count = number of entries.
  const IndexChange* changes_since(uint64_t epoch, size_t* count) const;
  uint32_t node_char(int32_t node) const { return nodes_[(size_t)node].ch; }

 private:
  int32_t new_node(uint32_t ch, int32_t parent);
  int32_t find_child(int32_t node, uint32_t ch) const;
  int32_t find_or_create(std::string_view term);
  size_t vacuum_node(int32_t node);
  void expand_from(int32_t node, std::string& term, std::vector<std::string>& out) const;

  std::vector<TrieNode> nodes_;
  std::vector<int32_t> free_nodes_;
  std::vector<PostingList> lists_;
  std::vector<int32_t> free_lists_;
  std::unordered_map<uint64_t, DocDetails> docs_;
  std::vector<FieldDetails> fields_;
  bool has_removed_ = false;
  std::unordered_set<uint64_t> removed_;
  // term -> node cache so bulk indexing does one hash probe per token instead of a trie walk
  // over linked sibling lists; dropped whenever vacuum prunes nodes.
  TermCache term_cache_;
  uint64_t epoch_ = 0;
  uint64_t uid_ = 0;
  // change log: entry i is the mutation that moved the epoch from log_base_ + i to log_base_ + i + 1
  // The log is kept only once somebody can replay it: the first snapshot taken of this index switches
  // it on (enable_change_log); until then a mutation costs no copy of the document's postings.
  mutable std::vector<IndexChange> log_;
  mutable uint64_t log_base_ = 0;
  mutable size_t log_postings_ = 0;
  mutable bool log_enabled_ = false;
  mutable std::mutex log_mu_;
  void log_push(IndexChange&& c);
  // add_document scratch
  std::vector<const char*> sp_;
  std::vector<size_t> sl_;
  std::vector<int32_t> doc_nodes_;
  std::vector<uint32_t> doc_tf_;
};

}  // namespace ps
