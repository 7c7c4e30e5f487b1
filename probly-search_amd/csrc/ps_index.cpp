// ps_index.cpp — host-side mutable index.  See ps_index.hpp for the design notes and the
// reference citations (src/index.rs of quantleaf/probly-search 2.0.1).
#include "ps_index.hpp"
#include "ps_build.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace ps {

std::vector<std::string_view> tokenize(std::string_view s, ps_tokenizer_fn fn, void* user,
                                       std::vector<const char*>& sp, std::vector<size_t>& sl) {
  std::vector<std::string_view> out;
  if (fn == nullptr) {
    // `s.split(' ')` (src/lib.rs:42-44): k separators -> k+1 tokens, empty ones included.
    size_t start = 0;
    for (size_t i = 0; i <= s.size(); ++i) {
      if (i == s.size() || s[i] == ' ') {
        out.emplace_back(s.data() + start, i - start);
        start = i + 1;
      }
    }
    return out;
  }
  size_t cap = s.size() + 2;
  if (sp.size() < cap) { sp.resize(cap); sl.resize(cap); }
  size_t n = fn(s.data(), s.size(), sp.data(), sl.data(), sp.size(), user);
  if (n > sp.size()) {  // tokenizer produced more tokens than bytes+2: retry with the size it asked for
    sp.resize(n); sl.resize(n);
    n = fn(s.data(), s.size(), sp.data(), sl.data(), sp.size(), user);
    if (n > sp.size()) n = sp.size();
  }
  out.reserve(n);
  for (size_t i = 0; i < n; ++i) out.emplace_back(sp[i], sl[i]);
  return out;
}

uint32_t next_char(std::string_view s, size_t& i) {
  unsigned char c = (unsigned char)s[i++];
  if (c < 0x80) return c;
  int extra = (c >> 5) == 0x6 ? 1 : (c >> 4) == 0xE ? 2 : 3;
  uint32_t cp = extra == 1 ? (c & 0x1Fu) : extra == 2 ? (c & 0x0Fu) : (c & 0x07u);
  for (int k = 0; k < extra && i < s.size(); ++k) cp = (cp << 6) | ((unsigned char)s[i++] & 0x3Fu);
  return cp;
}

void append_utf8(std::string& s, uint32_t cp) {
  if (cp < 0x80) {
    s.push_back((char)cp);
  } else if (cp < 0x800) {
    s.push_back((char)(0xC0 | (cp >> 6)));
    s.push_back((char)(0x80 | (cp & 0x3F)));
  } else if (cp < 0x10000) {
    s.push_back((char)(0xE0 | (cp >> 12)));
    s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
    s.push_back((char)(0x80 | (cp & 0x3F)));
  } else {
    s.push_back((char)(0xF0 | (cp >> 18)));
    s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
    s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
    s.push_back((char)(0x80 | (cp & 0x3F)));
  }
}

namespace {
std::atomic<uint64_t> g_next_uid{1};
}

Index::Index(size_t fields_num, size_t expected_index_size, size_t expected_documents_count) {
  uid_ = g_next_uid.fetch_add(1);
  fields_.assign(fields_num, FieldDetails{});
  nodes_.reserve(std::max<size_t>(expected_index_size, 16));
  docs_.reserve(expected_documents_count);
  new_node(0, NIL);  // root, char 0 (src/index.rs:54)
}

int32_t Index::new_node(uint32_t ch, int32_t parent) {
  TrieNode n{ch, NIL, NIL, NIL, parent};
  if (!free_nodes_.empty()) {
    int32_t i = free_nodes_.back();
    free_nodes_.pop_back();
    nodes_[(size_t)i] = n;
    return i;
  }
  nodes_.push_back(n);
  return (int32_t)nodes_.size() - 1;
}

int32_t Index::find_child(int32_t node, uint32_t ch) const {
  for (int32_t it = nodes_[(size_t)node].first_child; it != NIL; it = nodes_[(size_t)it].next)
    if (nodes_[(size_t)it].ch == ch) return it;
  return NIL;
}

// Walk / extend the trie for one term.  New children are prepended to their parent's child
// list (src/index.rs:409-419, 437-452), which fixes the expansion order seen by queries.
int32_t Index::find_or_create(std::string_view term) {
  const int32_t hit = term_cache_.find(term);
  if (hit >= 0) return hit;
  int32_t node = 0;
  size_t i = 0;
  while (i < term.size()) {
    size_t j = i;
    uint32_t ch = next_char(term, j);
    int32_t c = nodes_[(size_t)node].first_child == NIL ? NIL : find_child(node, ch);
    if (c == NIL) {
      while (i < term.size()) {
        uint32_t cc = next_char(term, i);
        int32_t nn = new_node(cc, node);
        nodes_[(size_t)nn].next = nodes_[(size_t)node].first_child;
        nodes_[(size_t)node].first_child = nn;
        node = nn;
      }
      break;
    }
    node = c;
    i = j;
  }
  term_cache_.insert(term, node);
  return node;
}

void Index::add_document(uint64_t key, const ps_str* values, const size_t* n_values, ps_tokenizer_fn tok,
                         void* user) {
  const size_t F = fields_.size();
  std::vector<uint32_t> field_length(F, 0);
  doc_nodes_.clear();
  doc_tf_.clear();
  std::unordered_map<int32_t, uint32_t> big;  // only used once a document has many distinct terms
  size_t vi = 0;
  for (size_t i = 0; i < F; ++i) {
    for (size_t j = 0; j < n_values[i]; ++j, ++vi) {
      std::string_view value(values[vi].ptr, values[vi].len);
      uint32_t count = 0;
      for (std::string_view term : tokenize(value, tok, user, sp_, sl_)) {
        if (term.empty()) continue;  // src/index.rs:101
        ++count;
        int32_t node = find_or_create(term);
        size_t slot = doc_nodes_.size();
        if (doc_nodes_.size() <= 48) {
          for (size_t k = 0; k < doc_nodes_.size(); ++k)
            if (doc_nodes_[k] == node) { slot = k; break; }
        } else {
          if (big.empty())
            for (size_t k = 0; k < doc_nodes_.size(); ++k) big.emplace(doc_nodes_[k], (uint32_t)k);
          auto it = big.find(node);
          if (it != big.end()) slot = it->second;
        }
        if (slot == doc_nodes_.size()) {
          doc_nodes_.push_back(node);
          doc_tf_.insert(doc_tf_.end(), F, 0u);
          if (!big.empty() || doc_nodes_.size() > 49) big.emplace(node, (uint32_t)slot);
        }
        doc_tf_[slot * F + i] += 1;
      }
      fields_[i].sum += count;
      fields_[i].avg = (double)fields_[i].sum / ((double)docs_.size() + 1.0);  // len BEFORE insert (:113)
      field_length[i] = count;                                                // assignment (:114)
    }
  }
  if (log_enabled_) {
    IndexChange c;
    c.kind = IndexChange::ADD;
    c.key = key;
    c.was_present = docs_.count(key) != 0;
    c.was_removed = has_removed_ && removed_.count(key) != 0;
    c.nodes = doc_nodes_;
    c.tf = doc_tf_;
    c.field_length = field_length;
    log_push(std::move(c));
  } else {
    log_base_ = epoch_ + 1;  // nothing recorded: no snapshot exists that could replay it
  }
  docs_[key] = DocDetails{std::move(field_length)};
  for (size_t k = 0; k < doc_nodes_.size(); ++k) {
    TrieNode& n = nodes_[(size_t)doc_nodes_[k]];
    if (n.list == NIL) {
      if (!free_lists_.empty()) { n.list = free_lists_.back(); free_lists_.pop_back(); }
      else { lists_.emplace_back(); n.list = (int32_t)lists_.size() - 1; }
    }
    PostingList& pl = lists_[(size_t)n.list];
    pl.keys.push_back(key);
    pl.tf.insert(pl.tf.end(), doc_tf_.begin() + (long)(k * F), doc_tf_.begin() + (long)((k + 1) * F));
  }
  ++epoch_;
}

void Index::bulk_load(const GroupedCorpus& g, size_t n_docs, const uint64_t* keys, const char* text) {
  if (!pristine()) throw std::invalid_argument("bulk_load needs an empty index");
  {
    // duplicate keys are add_document's job (a re-add keeps both versions' pointers, index.rs:119-157):
    // found BEFORE anything is touched, and reported as "not expressible here" (PS_EUNSUPPORTED), so that
    // ps_index_add_documents_flat_gpu falls back to the incremental host path on a still-empty index
    std::vector<uint64_t> sorted(keys, keys + n_docs);
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
      throw std::length_error("bulk_load: duplicate keys (re-adding a key is add_document's job)");
  }
  const size_t F = fields_.size();
  const size_t n_terms = g.term_pos.size();
  // documents + field statistics: sum accumulates every field length; avg = sum / (docs.len() + 1) as
  // of the LAST add_document (index.rs:112-114), i.e. sum / n for distinct keys
  docs_.reserve(n_docs * 2);
  for (size_t d = 0; d < n_docs; ++d) {
    DocDetails dd;
    dd.field_length.assign(g.field_length.begin() + (long)(d * F), g.field_length.begin() + (long)((d + 1) * F));
    for (size_t x = 0; x < F; ++x) fields_[x].sum += dd.field_length[x];
    docs_.emplace(keys[d], std::move(dd));
  }
  if (n_docs)
    for (size_t x = 0; x < F; ++x) fields_[x].avg = (double)fields_[x].sum / (double)n_docs;
  // terms in first-occurrence order: the incremental build creates a term's missing trie nodes when it
  // first meets the term, and prepends each new node to its parent's child list (index.rs:409-419)
  std::vector<uint32_t> order(n_terms);
  for (size_t t = 0; t < n_terms; ++t) order[t] = (uint32_t)t;
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return g.term_first_token[a] < g.term_first_token[b]; });
  lists_.resize(n_terms);
  for (size_t k = 0; k < n_terms; ++k) {
    const uint32_t t = order[k];
    const int32_t node = find_or_create(std::string_view(text + g.term_pos[t], g.term_len[t]));
    nodes_[(size_t)node].list = (int32_t)k;
    PostingList& pl = lists_[k];
    const uint32_t b = g.term_post_begin[t], e = g.term_post_begin[t + 1];
    pl.keys.resize(e - b);
    for (uint32_t i = b; i < e; ++i) pl.keys[i - b] = keys[g.post_doc[i]];
    pl.tf.assign(g.post_tf.begin() + (long)((size_t)b * F), g.post_tf.begin() + (long)((size_t)e * F));
  }
  log_push(IndexChange{});  // OTHER: not replayable as a delta
  ++epoch_;
}

void Index::log_push(IndexChange&& c) {
  if (!log_enabled_) {  // the epoch is bumped right after this call: keep log_base_ + log_.size() == epoch_
    log_base_ = epoch_ + 1;
    return;
  }
  // bounded: a burst larger than this is cheaper to re-flatten than to replay
  constexpr size_t MAX_ENTRIES = 1u << 20, MAX_POSTINGS = 16u << 20;
  log_postings_ += c.nodes.size();
  log_.push_back(std::move(c));
  if (log_.size() > MAX_ENTRIES || log_postings_ > MAX_POSTINGS) {
    log_base_ += log_.size();
    log_.clear();
    log_.shrink_to_fit();
    log_postings_ = 0;
  }
}

const IndexChange* Index::changes_since(uint64_t epoch, size_t* count) const {
  *count = 0;
  if (epoch < log_base_ || epoch > epoch_) return nullptr;
  const size_t first = (size_t)(epoch - log_base_);
  if (first > log_.size() || log_base_ + log_.size() != epoch_) return nullptr;
  *count = log_.size() - first;
  return log_.data() + first;
}

void Index::remove_document(uint64_t key) {
  has_removed_ = true;
  {
    IndexChange c;
    c.kind = IndexChange::REMOVE;
    c.key = key;
    c.was_present = docs_.count(key) != 0;
    log_push(std::move(c));
  }
  auto it = docs_.find(key);
  if (it != docs_.end()) {
    removed_.insert(key);
    double new_len = (double)(docs_.size() - 1);
    for (size_t i = 0; i < fields_.size(); ++i) {
      uint32_t fl = it->second.field_length[i];
      if (fl > 0) {
        fields_[i].sum -= fl;
        fields_[i].avg = (double)fields_[i].sum / new_len;  // 0/0 -> NaN when the last doc goes (:643)
      }
    }
    docs_.erase(it);
  }
  ++epoch_;
}

// Returns 1 if the subtree still holds a posting (src/index.rs:203-241).
size_t Index::vacuum_node(int32_t node) {
  const size_t F = fields_.size();
  int32_t li = nodes_[(size_t)node].list;
  size_t ret = 0;
  if (li != NIL) {
    PostingList& pl = lists_[(size_t)li];
    size_t w = 0;
    for (size_t r = 0; r < pl.keys.size(); ++r) {
      if (removed_.count(pl.keys[r])) continue;
      if (w != r) {
        pl.keys[w] = pl.keys[r];
        std::copy(pl.tf.begin() + (long)(r * F), pl.tf.begin() + (long)((r + 1) * F), pl.tf.begin() + (long)(w * F));
      }
      ++w;
    }
    pl.keys.resize(w);
    pl.tf.resize(w * F);
    if (w == 0) {
      PostingList().keys.swap(pl.keys);
      std::vector<uint32_t>().swap(pl.tf);
      free_lists_.push_back(li);
      nodes_[(size_t)node].list = NIL;
    } else {
      ret = 1;
    }
  }
  int32_t prev = NIL;
  int32_t child = nodes_[(size_t)node].first_child;
  while (child != NIL) {
    size_t r = vacuum_node(child);
    ret |= r;
    int32_t nx = nodes_[(size_t)child].next;
    if (r == 0) {
      if (prev != NIL) nodes_[(size_t)prev].next = nx;
      else nodes_[(size_t)node].first_child = nx;
      free_nodes_.push_back(child);
    } else {
      prev = child;
    }
    child = nx;
  }
  return ret;
}

void Index::vacuum() {
  log_push(IndexChange{});  // OTHER: lists are compacted and nodes recycled - snapshots re-flatten
  vacuum_node(0);
  removed_.clear();
  has_removed_ = false;
  term_cache_.clear();
  ++epoch_;
}

const DocDetails* Index::doc(uint64_t key) const {
  auto it = docs_.find(key);
  return it == docs_.end() ? nullptr : &it->second;
}

size_t Index::count_nodes() const {
  size_t c = 0;
  std::vector<int32_t> st{0};
  while (!st.empty()) {
    int32_t n = st.back();
    st.pop_back();
    ++c;
    for (int32_t it = nodes_[(size_t)n].first_child; it != NIL; it = nodes_[(size_t)it].next) st.push_back(it);
  }
  return c;
}

size_t Index::live_pointers() const {
  size_t c = 0;
  for (const PostingList& pl : lists_)
    for (uint32_t t : pl.tf) c += t;
  return c;
}

int32_t Index::find_node(std::string_view term) const {
  int32_t node = 0;
  size_t i = 0;
  while (i < term.size() && node != NIL) node = find_child(node, next_char(term, i));
  return node;
}

long Index::count_documents(int32_t node) const {
  int32_t li = nodes_[(size_t)node].list;
  if (li == NIL) return 0;
  const PostingList& pl = lists_[(size_t)li];
  const size_t F = fields_.size();
  long df = 0;
  for (size_t r = 0; r < pl.keys.size(); ++r) {
    if (is_removed(pl.keys[r])) continue;
    for (size_t x = 0; x < F; ++x) df += pl.tf[r * F + x];  // one pointer per occurrence (:119-157)
  }
  return df;
}

void Index::expand_from(int32_t node, std::string& term, std::vector<std::string>& out) const {
  int32_t li = nodes_[(size_t)node].list;
  if (li != NIL && !lists_[(size_t)li].keys.empty()) out.push_back(term);  // first_doc.is_some()
  for (int32_t it = nodes_[(size_t)node].first_child; it != NIL; it = nodes_[(size_t)it].next) {
    size_t len = term.size();
    append_utf8(term, nodes_[(size_t)it].ch);
    expand_from(it, term, out);
    term.resize(len);
  }
}

std::vector<std::string> Index::expand_term(std::string_view term) const {
  std::vector<std::string> out;
  int32_t node = find_node(term);
  if (node != NIL) {
    std::string t(term);
    expand_from(node, t, out);
  }
  return out;
}

// The reference's Index::query driver (src/query.rs:29-105) for a custom ScoreCalculator handed
// over as C callbacks.  One `score` call per DocumentPointer: a (document, term) record stands
// for sum(tf) identical pointers, newest document first (src/index.rs:119-157, 422-433).
void Index::query_callbacks(const ps_score_callbacks& cb, std::string_view query, ps_tokenizer_fn tok, void* tok_user,
                            const double* fields_boost, size_t n_boost, const ps_index* handle,
                            std::vector<ps_result>& out) const {
  if (!cb.score) throw std::invalid_argument("ps_score_callbacks.score is required");
  const size_t F = fields_.size();
  std::vector<const char*> sp;
  std::vector<size_t> sl;
  const std::vector<std::string_view> terms = tokenize(query, tok, tok_user, sp, sl);
  std::vector<ps_field_details> fd(F);
  for (size_t x = 0; x < F; ++x) fd[x] = ps_field_details{fields_[x].sum, fields_[x].avg};
  const ps_field_data field_data{fields_boost, n_boost, fd.data(), F};
  std::unordered_map<uint64_t, double> scores;  // query.rs:31
  for (size_t qi = 0; qi < terms.size(); ++qi) {
    const std::string_view qt = terms[qi];
    if (qt.empty()) continue;  // query.rs:35
    std::unordered_set<uint64_t> visited;  // query.rs:37
    for (const std::string& expanded : expand_term(qt)) {
      const int32_t node = find_node(expanded);  // query.rs:39-43
      if (node == NIL) continue;
      const long df = count_documents(node);  // query.rs:45
      const int32_t li = nodes_[(size_t)node].list;
      if (li == NIL || lists_[(size_t)li].keys.empty() || df <= 0) continue;  // query.rs:47-48
      const ps_term_data td{qi, ps_str{qt.data(), qt.size()}, ps_str{expanded.data(), expanded.size()}, terms.size()};
      void* memory = nullptr;
      const bool some = cb.before_each && cb.before_each(cb.user, &td, (size_t)df, docs_.size(), handle, &memory) != 0;
      const PostingList& pl = lists_[(size_t)li];
      for (size_t r = pl.keys.size(); r-- > 0;) {  // newest first
        const uint64_t key = pl.keys[r];
        uint64_t copies = 0;
        for (size_t x = 0; x < F; ++x) copies += pl.tf[r * F + x];
        const bool live = !is_removed(key);
        const DocDetails* dd = live ? doc(key) : nullptr;
        for (uint64_t c = 0; c < copies; ++c) {
          if (dd) {  // query.rs:65-66
            const ps_document_pointer dp{key, pl.tf.data() + r * F};
            const ps_document_details det{key, dd->field_length.data()};
            double s = 0.0;
            if (cb.score(cb.user, some ? memory : nullptr, &dp, &det, (uint64_t)node, &field_data, &td, &s) != 0) {
              auto it = scores.find(key);  // max_score_merger, query.rs:150-164
              if (it == scores.end()) scores.emplace(key, s);
              else it->second = visited.count(key) ? std::fmax(it->second, s) : it->second + s;  // f64::max (query.rs:158): a NaN operand loses
            }
          }
          visited.insert(key);  // query.rs:87
        }
      }
      if (some && cb.drop_memory) cb.drop_memory(cb.user, memory);
    }
  }
  out.clear();
  out.reserve(scores.size());
  for (const auto& kv : scores) out.push_back(ps_result{kv.first, kv.second});
  std::sort(out.begin(), out.end(), [](const ps_result& a, const ps_result& b) { return a.key < b.key; });
  if (cb.finalize) {
    const size_t n = cb.finalize(cb.user, out.data(), out.size());
    if (n < out.size()) out.resize(n);
  }
  for (const ps_result& r : out)
    if (r.score != r.score) throw std::invalid_argument("a score is NaN (the reference panics: partial_cmp().unwrap(), query.rs:103)");
  // query.rs:103 is a stable sort by score desc; over key-ascending input that is the canonical order
  std::stable_sort(out.begin(), out.end(), [](const ps_result& a, const ps_result& b) { return a.score > b.score; });
}

std::vector<uint32_t> Index::children(int32_t node) const {
  std::vector<uint32_t> out;
  for (int32_t it = nodes_[(size_t)node].first_child; it != NIL; it = nodes_[(size_t)it].next)
    out.push_back(nodes_[(size_t)it].ch);
  return out;
}

}  // namespace ps
