// ============================================================================
// probly_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A literal, single-threaded C++ restatement of the probly-search 2.0.1 CPU
// query path (reference mounted at /root/reference; citations are relative to
// it).  It exists so that tests/, __graft_entry__.smoke() and bench.py's
// `cpu_baseline` leg have something to check / time the HIP path against.
// NOTHING in the product (probly-search_amd/) may include, link or call this.
//
// Parity pin: the Rust reference cannot be compiled in this image (no
// cargo/rustc, crates not vendored) so there is no oracle/_ref build.  The
// restatement is pinned instead against every known-answer test the reference
// itself holds for this path (tests/golden/reference_kats.json, rows R1–R24:
// src/score/default/bm25.rs:105-136, src/query.rs:182-387,
// src/score/default/zero_to_one.rs:139-404, tests/integrations_tests.rs:28-149,
// tests/document_frequency.rs:5-32, src/index.rs:497-658) — see
// tests/test_oracle_golden.py.
//
// Faithfulness rules (deliberately NOT optimised — this is also the timed
// "reference-faithful C++ restatement" CPU baseline):
//   * trie nodes and postings live in two slot arenas and are chained through
//     singly-linked lists; new children / postings are PREPENDED
//     (src/index.rs:409-433);
//   * one DocumentPointer per term OCCURRENCE, each owning its own heap
//     term-frequency vector (src/index.rs:119-157);
//   * query = count_documents pre-pass + posting walk with five hash
//     operations per pointer (src/query.rs:45,61-89);
//   * full result materialisation + stable sort (src/query.rs:97-105);
//   * arithmetic in the exact association of the Rust source, compiled with
//     -ffp-contract=off; ln == glibc log (Rust f64::ln lowers to libm log).
// ============================================================================
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <emmintrin.h>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

constexpr int64_t NONE = -1;  // Option<ArenaIndex<..>>::None

// src/index.rs:342-349
struct DocumentDetails {
  uint64_t key;
  std::vector<size_t> field_length;
};

// src/index.rs:354-361
struct DocumentPointer {
  int64_t next;
  uint64_t details_key;
  std::vector<size_t> term_frequency;
  bool live;  // arena slot occupied
};

// src/index.rs:364-373
struct InvertedIndexNode {
  uint32_t ch;  // Rust `char` (Unicode scalar value)
  int64_t next;
  int64_t first_child;
  int64_t first_doc;
  bool live;
};

// src/index.rs:391-396
struct FieldDetails {
  size_t sum;
  double avg;
};

// src/query.rs:10-15
struct QueryResult {
  uint64_t key;
  double score;
};

// src/score/calculator.rs:9-19
struct TermData {
  size_t query_term_index;
  const std::string* query_term;
  const std::string* query_term_expanded;
  size_t query_terms_len;
};

// Tokenizer (src/lib.rs:14).  Default == test_util::tokenizer, `s.split(' ')`
// (src/lib.rs:42-44): n separators always yield n+1 tokens, empties included.
typedef size_t (*orc_tokenizer_fn)(const char* s, size_t len, const char** tok_ptr,
                                   size_t* tok_len, size_t cap, void* user);

std::vector<std::string> tokenize(const char* s, size_t len, orc_tokenizer_fn fn, void* user) {
  std::vector<std::string> out;
  if (fn == nullptr) {
    size_t start = 0;
    for (size_t i = 0; i <= len; ++i) {
      if (i == len || s[i] == ' ') {
        out.emplace_back(s + start, i - start);
        start = i + 1;
      }
    }
    return out;
  }
  size_t cap = len + 2;
  std::vector<const char*> p(cap);
  std::vector<size_t> l(cap);
  size_t n = fn(s, len, p.data(), l.data(), cap, user);
  for (size_t i = 0; i < n && i < cap; ++i) out.emplace_back(p[i], l[i]);
  return out;
}

// `str::chars()`: decode (valid) UTF-8 into scalar values.
std::vector<uint32_t> chars_of(const std::string& s) {
  std::vector<uint32_t> out;
  size_t i = 0, n = s.size();
  while (i < n) {
    unsigned char c = (unsigned char)s[i];
    uint32_t cp;
    size_t extra;
    if (c < 0x80) { cp = c; extra = 0; }
    else if ((c >> 5) == 0x6) { cp = c & 0x1F; extra = 1; }
    else if ((c >> 4) == 0xE) { cp = c & 0x0F; extra = 2; }
    else { cp = c & 0x07; extra = 3; }
    ++i;
    for (size_t k = 0; k < extra && i < n; ++k, ++i) cp = (cp << 6) | ((unsigned char)s[i] & 0x3F);
    out.push_back(cp);
  }
  return out;
}

void push_utf8(std::string& s, uint32_t cp) {
  if (cp < 0x80) s.push_back((char)cp);
  else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
  else if (cp < 0x10000) {
    s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
    s.push_back((char)(0x80 | (cp & 0x3F)));
  } else {
    s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
    s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
  }
}

// ---------------------------------------------------------------------------
// Index<T> with T = u64 (src/index.rs:19-33)
// ---------------------------------------------------------------------------
struct Index {
  std::unordered_map<uint64_t, DocumentDetails> docs;
  int64_t root;
  std::vector<FieldDetails> fields;
  std::vector<InvertedIndexNode> arena_index;
  std::vector<int64_t> arena_index_free;
  std::vector<DocumentPointer> arena_doc;
  std::vector<int64_t> arena_doc_free;
  bool has_removed = false;  // Option<HashSet<T>>
  std::unordered_set<uint64_t> removed;

  // src/index.rs:37-60
  explicit Index(size_t fields_num) {
    fields.assign(fields_num, FieldDetails{0, 0.0});
    root = insert_node(0);
  }

  int64_t insert_node(uint32_t ch) {  // create_inverted_index_node, src/index.rs:399-406
    InvertedIndexNode n{ch, NONE, NONE, NONE, true};
    if (!arena_index_free.empty()) {
      int64_t i = arena_index_free.back();
      arena_index_free.pop_back();
      arena_index[(size_t)i] = n;
      return i;
    }
    arena_index.push_back(n);
    return (int64_t)arena_index.size() - 1;
  }
  int64_t insert_doc(DocumentPointer&& d) {
    if (!arena_doc_free.empty()) {
      int64_t i = arena_doc_free.back();
      arena_doc_free.pop_back();
      arena_doc[(size_t)i] = std::move(d);
      return i;
    }
    arena_doc.push_back(std::move(d));
    return (int64_t)arena_doc.size() - 1;
  }

  // src/index.rs:409-419 — PREPEND the child
  void add_inverted_index_child_node(int64_t parent, int64_t child) {
    int64_t first = arena_index[(size_t)parent].first_child;
    if (first != NONE) arena_index[(size_t)child].next = first;
    arena_index[(size_t)parent].first_child = child;
  }

  // src/index.rs:437-452
  int64_t create_inverted_index_nodes(int64_t parent, const std::vector<uint32_t>& term, size_t start) {
    for (size_t i = start; i < term.size(); ++i) {
      int64_t nn = insert_node(term[i]);
      add_inverted_index_child_node(parent, nn);
      parent = arena_index[(size_t)parent].first_child;
    }
    return parent;
  }

  // src/index.rs:321-337
  int64_t find_child_by_char(int64_t from, uint32_t ch) const {
    int64_t it = arena_index[(size_t)from].first_child;
    while (it != NONE) {
      const InvertedIndexNode& n = arena_index[(size_t)it];
      if (n.ch == ch) return it;
      it = n.next;
    }
    return NONE;
  }

  // src/index.rs:300-318
  int64_t find_inverted_index_node(int64_t node, const std::string& term) const {
    int64_t it = node;
    for (uint32_t ch : chars_of(term)) {
      if (it != NONE) it = find_child_by_char(it, ch);
      else break;
    }
    return it;
  }

  // src/index.rs:77-158.  `values[i]` = what field accessor i returned.
  void add_document(uint64_t key, const std::vector<std::vector<std::string>>& values,
                    orc_tokenizer_fn tok, void* user) {
    std::vector<size_t> field_length(fields.size(), 0);
    std::unordered_map<std::string, std::vector<size_t>> term_counts;
    std::vector<std::string> all_terms;
    for (size_t i = 0; i < fields.size(); ++i) {
      size_t fields_len = fields.size();
      for (const std::string& field_value : values[i]) {
        std::vector<std::string> terms = tokenize(field_value.data(), field_value.size(), tok, user);
        size_t filtered_terms_count = 0;
        for (std::string& term : terms) {
          if (!term.empty()) {
            filtered_terms_count += 1;
            all_terms.push_back(term);
            auto it = term_counts.find(term);
            if (it == term_counts.end()) it = term_counts.emplace(term, std::vector<size_t>(fields_len, 0)).first;
            it->second[i] += 1;
          }
        }
        fields[i].sum += filtered_terms_count;
        fields[i].avg = (double)fields[i].sum / ((double)docs.size() + 1.0);  // :113, len BEFORE insert
        field_length[i] = filtered_terms_count;                                // :114, assignment
      }
    }
    docs[key] = DocumentDetails{key, field_length};  // HashMap::insert overwrites
    for (const std::string& term : all_terms) {      // one iteration per OCCURRENCE
      int64_t node_index = root;
      std::vector<uint32_t> cs = chars_of(term);
      for (size_t i = 0; i < cs.size(); ++i) {
        if (arena_index[(size_t)node_index].first_child == NONE) {
          node_index = create_inverted_index_nodes(node_index, cs, i);
          break;
        }
        int64_t next_node = find_child_by_char(node_index, cs[i]);
        if (next_node == NONE) {
          node_index = create_inverted_index_nodes(node_index, cs, i);
          break;
        }
        node_index = next_node;
      }
      // add_inverted_index_doc, src/index.rs:422-433 — PREPEND
      DocumentPointer dp{NONE, key, term_counts[term], true};
      int64_t first = arena_index[(size_t)node_index].first_doc;
      if (first != NONE) dp.next = first;
      int64_t di = insert_doc(std::move(dp));
      arena_index[(size_t)node_index].first_doc = di;
    }
  }

  // src/index.rs:161-191
  void remove_document(uint64_t key) {
    has_removed = true;
    auto it = docs.find(key);
    if (it != docs.end()) {
      removed.insert(key);
      double new_len = (double)(docs.size() - 1);
      for (size_t i = 0; i < fields.size(); ++i) {
        size_t fl = it->second.field_length[i];
        if (fl > 0) {
          fields[i].sum -= fl;
          fields[i].avg = (double)fields[i].sum / new_len;
        }
      }
      docs.erase(it);
    }
  }

  // src/index.rs:245-279
  size_t disconnect_and_count_documents(int64_t node_index, const std::unordered_set<uint64_t>* rem) {
    int64_t prev = NONE;
    int64_t p = arena_index[(size_t)node_index].first_doc;
    size_t df = 0;
    while (p != NONE) {
      bool is_removed = rem != nullptr && rem->count(arena_doc[(size_t)p].details_key) != 0;
      int64_t nx = arena_doc[(size_t)p].next;
      if (is_removed) {
        if (prev == NONE) arena_index[(size_t)node_index].first_doc = nx;
        else arena_doc[(size_t)prev].next = nx;
      } else {
        df += 1;
        prev = p;
      }
      if (is_removed) {
        arena_doc[(size_t)p].live = false;
        arena_doc[(size_t)p].term_frequency = std::vector<size_t>();
        arena_doc_free.push_back(p);
      }
      p = nx;
    }
    return df;
  }

  // src/index.rs:203-241
  size_t vacuum_node(int64_t node_index, const std::unordered_set<uint64_t>& rem) {
    disconnect_and_count_documents(node_index, &rem);
    int64_t prev_child = NONE;
    size_t ret = 0;
    if (arena_index[(size_t)node_index].first_doc != NONE) ret = 1;
    int64_t child = arena_index[(size_t)node_index].first_child;
    while (child != NONE) {
      size_t r = vacuum_node(child, rem);
      ret |= r;
      int64_t child_next = arena_index[(size_t)child].next;
      if (r == 0) {
        if (prev_child != NONE) arena_index[(size_t)prev_child].next = child_next;
        else arena_index[(size_t)node_index].first_child = child_next;
      } else {
        prev_child = child;
      }
      if (r == 0) {
        arena_index[(size_t)child].live = false;
        arena_index_free.push_back(child);
      }
      child = child_next;
    }
    return ret;
  }

  // src/index.rs:194-199
  void vacuum() {
    std::unordered_set<uint64_t> rem;
    rem.swap(removed);
    has_removed = false;
    vacuum_node(root, rem);
  }

  // src/index.rs:282-297
  size_t count_documents(int64_t node_index) const {
    int64_t p = arena_index[(size_t)node_index].first_doc;
    size_t df = 0;
    while (p != NONE) {
      bool is_removed = has_removed && removed.count(arena_doc[(size_t)p].details_key) != 0;
      if (!is_removed) df += 1;
      p = arena_doc[(size_t)p].next;
    }
    return df;
  }

  // src/query.rs:130-147
  void expand_term_from_node(int64_t node, std::vector<std::string>& results, const std::string& term) const {
    if (arena_index[(size_t)node].first_doc != NONE) results.push_back(term);
    int64_t child = arena_index[(size_t)node].first_child;
    while (child != NONE) {
      std::string inter = term;
      push_utf8(inter, arena_index[(size_t)child].ch);
      expand_term_from_node(child, results, inter);
      child = arena_index[(size_t)child].next;
    }
  }

  // src/query.rs:109-126
  std::vector<std::string> expand_term(const std::string& term) const {
    std::vector<std::string> results;
    int64_t node = find_inverted_index_node(root, term);
    if (node != NONE) expand_term_from_node(node, results, term);
    return results;
  }
};

// ---------------------------------------------------------------------------
// ScoreCalculator<T, M> (src/score/calculator.rs:33-70), dynamic dispatch here
// ---------------------------------------------------------------------------
struct PreCalc {  // the `M` of both shipped scorers, as a tagged bag
  bool some = false;
  double idf = 0.0;
  double expansion_boost = 0.0;
};

struct ScoreCalculator {
  virtual ~ScoreCalculator() {}
  virtual PreCalc before_each(const TermData&, size_t /*document_frequency*/,
                              const std::unordered_map<uint64_t, DocumentDetails>&) {
    return PreCalc{};  // trait default: None (calculator.rs:43-50)
  }
  // returns true + *out for Some(score)
  virtual bool score(const PreCalc* before_output, const DocumentPointer& dp, const DocumentDetails& dd,
                     int64_t index_node, const double* fields_boost, const std::vector<FieldDetails>& fields,
                     const TermData& td, double* out) = 0;
  virtual void finalize(std::vector<QueryResult>&) {}  // trait default: no-op (calculator.rs:69)
};

// src/score/default/bm25.rs:14-93
struct BM25 : ScoreCalculator {
  double bm25k1 = 1.2, bm25b = 0.75;
  PreCalc before_each(const TermData& te, size_t document_frequency,
                      const std::unordered_map<uint64_t, DocumentDetails>& documents) override {
    size_t frequency = std::min(documents.size(), document_frequency);
    size_t diff = documents.size() - frequency;
    PreCalc m;
    m.some = true;
    if (*te.query_term_expanded == *te.query_term) {
      m.expansion_boost = 1.0;
    } else {
      m.expansion_boost =
          std::log(1.0 + (1.0 / (1.0 + (double)te.query_term_expanded->size() - (double)te.query_term->size())));
    }
    m.idf = std::log(1.0 + ((double)diff + 0.5) / ((double)frequency + 0.5));
    return m;
  }
  bool score(const PreCalc* pre, const DocumentPointer& dp, const DocumentDetails& dd, int64_t,
             const double* fields_boost, const std::vector<FieldDetails>& fields, const TermData&,
             double* out) override {
    double score = 0.0;
    for (size_t x = 0; x < dd.field_length.size(); ++x) {
      double tf = (double)dp.term_frequency[x];
      if (tf > 0.0) {
        double avg_field_length = fields[x].avg;
        tf = ((bm25k1 + 1.0) * tf) /
             (bm25k1 * ((1.0 - bm25b) + bm25b * ((double)dd.field_length[x] / avg_field_length)) + tf);
        score += tf * pre->idf * fields_boost[x] * pre->expansion_boost;
      }
    }
    if (score > 0.0) { *out = score; return true; }
    return false;
  }
};

// src/score/default/zero_to_one.rs:24-126
struct ScoreByTerm {
  size_t query_term_index, all_query_terms_len, field_length, index_node_id, term_frequency;
  double score;
};
struct ZeroToOne : ScoreCalculator {
  std::unordered_map<uint64_t, std::vector<std::vector<ScoreByTerm>>> score_by_document_and_field;
  bool nan_seen = false;
  bool score(const PreCalc*, const DocumentPointer& dp, const DocumentDetails& dd, int64_t index_node,
             const double*, const std::vector<FieldDetails>&, const TermData& td, double* out) override {
    uint64_t key = dd.key;
    for (size_t x = 0; x < dd.field_length.size(); ++x) {
      size_t tf = dp.term_frequency[x];
      if (tf > 0) {
        double term_exp_len = (double)td.query_term_expanded->size();
        double term_len = (double)td.query_term->size();
        size_t field_length = dd.field_length[x];
        if (score_by_document_and_field.find(key) == score_by_document_and_field.end()) {
          score_by_document_and_field.emplace(key, std::vector<std::vector<ScoreByTerm>>(dd.field_length.size()));
        }
        score_by_document_and_field[key][x].push_back(ScoreByTerm{
            td.query_term_index, td.query_terms_len, field_length, (size_t)index_node, tf,
            1.0 - std::fabs(term_exp_len - term_len) / term_exp_len});
      }
    }
    *out = 0.0;  // dummy (zero_to_one.rs:81)
    return true;
  }
  void finalize(std::vector<QueryResult>& results) override {
    for (QueryResult& result : results) {
      for (std::vector<ScoreByTerm>& field_scores : score_by_document_and_field.at(result.key)) {
        std::unordered_map<size_t, size_t> df_pool_by_id;
        std::unordered_set<size_t> consumed_index;
        std::stable_sort(field_scores.begin(), field_scores.end(),
                         [](const ScoreByTerm& a, const ScoreByTerm& b) { return b.score < a.score; });
        double score_by_pool = 0.0;
        for (const ScoreByTerm& s : field_scores) {
          if (consumed_index.count(s.query_term_index)) continue;
          auto it = df_pool_by_id.find(s.index_node_id);
          if (it != df_pool_by_id.end()) {
            if (it->second <= 0) continue;
            it->second -= 1;
          } else {
            df_pool_by_id.emplace(s.index_node_id, s.term_frequency - 1);
          }
          consumed_index.insert(s.query_term_index);
          double df = (double)s.term_frequency;
          score_by_pool += std::fmin(s.score / df, 1.0) * (double)s.term_frequency /
                           (double)std::max(s.field_length, s.all_query_terms_len);
        }
        result.score = std::fmax(score_by_pool, result.score);
      }
    }
    score_by_document_and_field.clear();
  }
};

// src/query.rs:150-164
inline double max_score_merger(double score, const double* previous_score, bool document_visited_for_term) {
  if (previous_score != nullptr) {
    if (document_visited_for_term) return std::fmax(*previous_score, score);
    return *previous_score + score;
  }
  return score;
}

// src/query.rs:21-106.  `canonical` additionally applies test_util::test_score's
// re-sort (score desc, then key asc; src/lib.rs:54-58) because hashbrown's
// iteration order (and hence the reference's tie order) is unspecified.
// Returns false if the reference would have panicked in partial_cmp().unwrap().
bool query(const Index& idx, const char* q, size_t qlen, ScoreCalculator& sc, orc_tokenizer_fn tok, void* user,
           const double* fields_boost, bool canonical, std::vector<QueryResult>& result) {
  std::vector<std::string> query_terms = tokenize(q, qlen, tok, user);
  std::unordered_map<uint64_t, double> scores;
  size_t query_terms_len = query_terms.size();
  for (size_t query_term_index = 0; query_term_index < query_terms.size(); ++query_term_index) {
    const std::string& query_term = query_terms[query_term_index];
    if (query_term.empty()) continue;
    std::vector<std::string> expanded_terms = idx.expand_term(query_term);
    std::unordered_set<uint64_t> visited_documents_for_term;
    for (const std::string& query_term_expanded : expanded_terms) {
      int64_t term_node_index = idx.find_inverted_index_node(idx.root, query_term_expanded);
      if (term_node_index == NONE) continue;
      size_t document_frequency = idx.count_documents(term_node_index);
      int64_t first_doc = idx.arena_index[(size_t)term_node_index].first_doc;
      if (first_doc == NONE || document_frequency == 0) continue;
      TermData td{query_term_index, &query_term, &query_term_expanded, query_terms_len};
      PreCalc pre = sc.before_each(td, document_frequency, idx.docs);
      int64_t pointer = first_doc;
      while (pointer != NONE) {
        const DocumentPointer& pb = idx.arena_doc[(size_t)pointer];
        uint64_t key = pb.details_key;
        if (!idx.has_removed || idx.removed.count(key) == 0) {
          double s;
          bool some = sc.score(pre.some ? &pre : nullptr, pb, idx.docs.at(key), term_node_index, fields_boost,
                               idx.fields, td, &s);
          if (some) {
            auto it = scores.find(key);
            double new_score = max_score_merger(s, it == scores.end() ? nullptr : &it->second,
                                                visited_documents_for_term.count(key) != 0);
            scores[key] = new_score;
          }
        }
        visited_documents_for_term.insert(key);
        pointer = pb.next;
      }
    }
  }
  result.clear();
  result.reserve(scores.size());
  for (const auto& kv : scores) result.push_back(QueryResult{kv.first, kv.second});
  sc.finalize(result);
  for (const QueryResult& r : result)
    if (std::isnan(r.score)) return false;  // partial_cmp(..).unwrap() would panic (query.rs:103)
  std::stable_sort(result.begin(), result.end(),
                   [](const QueryResult& a, const QueryResult& b) { return b.score < a.score; });
  if (canonical) {
    std::sort(result.begin(), result.end(), [](const QueryResult& a, const QueryResult& b) {
      if (a.score != b.score) return a.score > b.score;
      return a.key < b.key;
    });
  }
  return true;
}

// A TEST PLUGIN, not a reference scorer: a ScoreCalculator whose `score` returns NaN for some (document, expansion)
// pairs, to pin what max_score_merger does with NaN operands (f64::max returns the non-NaN one, query.rs:158; `+` poisons,
// and a NaN that survives to the sort panics, query.rs:103).  tests/test_host_callbacks.py runs the same function as a
// product plugin.  score(key, expanded) = NaN if (key + len(expanded)) % 3 == 0 else 0.25 * (key + 1) + len(expanded).
struct NanProbe : ScoreCalculator {
  bool score(const PreCalc*, const DocumentPointer&, const DocumentDetails& dd, int64_t, const double*,
             const std::vector<FieldDetails>&, const TermData& td, double* out) override {
    const size_t le = td.query_term_expanded->size();
    *out = ((dd.key + le) % 3 == 0) ? std::nan("") : 0.25 * (double)(dd.key + 1) + (double)le;
    return true;
  }
};

// ---------------------------------------------------------------------------
// SECOND CPU-BASELINE LEG ("flat"): the same walk, the same five hash operations per pointer in the same order
// (src/query.rs:63-88), but on hashbrown-class containers - the reference uses hashbrown 0.14 (src/query.rs:1,31,37,
// src/index.rs:8), i.e. SwissTable: open addressing, 7-bit control bytes probed 16 at a time with SSE2, load factor 7/8,
// triangular probing - and a per-thread bump arena behind the per-query tables, so that growing a table costs the
// rehash and not malloc / mmap / page faults.  `std::unordered_*` (node-based, one allocation per insertion) is what the
// literal leg above uses; it says so and is a pessimistic stand-in.  This leg is the stronger baseline.  Its results are
// bit-identical to query() (tests/test_oracle_golden.py::test_flat_leg_matches_literal).
// ---------------------------------------------------------------------------
struct Arena {  // per thread; reset() keeps the memory
  std::vector<std::pair<char*, size_t>> chunks;
  size_t cur = 0, off = 0;
  ~Arena() { for (auto& c : chunks) free(c.first); }
  void* alloc(size_t bytes) {
    bytes = (bytes + 63) & ~(size_t)63;
    while (cur < chunks.size() && off + bytes > chunks[cur].second) { ++cur; off = 0; }
    if (cur == chunks.size()) {
      const size_t sz = std::max<size_t>(bytes, (size_t)32 << 20);
      chunks.emplace_back((char*)aligned_alloc(64, sz), sz);
      off = 0;
    }
    void* r = chunks[cur].first + off;
    off += bytes;
    return r;
  }
  void reset() { cur = 0; off = 0; }
};

inline uint64_t swiss_hash(uint64_t k) {  // ahash's fallback: one folded 64 x 64 -> 128 multiply
  const unsigned __int128 m = (unsigned __int128)(k ^ 0x243F6A8885A308D3ull) * 0x5851F42D4C957F2Dull;
  return (uint64_t)m ^ (uint64_t)(m >> 64);
}

template <class V>
struct Swiss {  // uint64_t -> V; insert and lookup only (the path never erases)
  struct Slot { uint64_t key; V val; };
  static constexpr uint8_t EMPTY = 0xFF;
  uint8_t* ctrl = nullptr;
  Slot* slots = nullptr;
  size_t mask = 0, items = 0, growth_left = 0;
  Arena* arena = nullptr;  // nullptr: malloc, released by the destructor
  Swiss() {}
  explicit Swiss(Arena* a) : arena(a) {}
  Swiss(const Swiss&) = delete;
  Swiss& operator=(const Swiss&) = delete;
  ~Swiss() {
    if (arena || !ctrl) return;
    for (size_t i = 0; i <= mask; ++i)
      if (ctrl[i] != EMPTY) slots[i].~Slot();
    free(ctrl);
    free(slots);
  }
  void* raw(size_t bytes) { return arena ? arena->alloc(bytes) : aligned_alloc(64, (bytes + 63) & ~(size_t)63); }
  void set_ctrl(size_t i, uint8_t c) {
    ctrl[i] = c;
    if (i < 16) ctrl[mask + 1 + i] = c;  // the first group is mirrored behind the last one
  }
  const Slot* find(uint64_t k) const {
    if (!ctrl) return nullptr;
    const uint64_t h = swiss_hash(k);
    const __m128i tag = _mm_set1_epi8((char)(h >> 57)), empty = _mm_set1_epi8((char)EMPTY);
    size_t pos = h & mask, stride = 0;
    for (;;) {
      const __m128i g = _mm_loadu_si128((const __m128i*)(ctrl + pos));
      unsigned bits = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(g, tag));
      while (bits) {
        const size_t i = (pos + (size_t)__builtin_ctz(bits)) & mask;
        if (slots[i].key == k) return &slots[i];
        bits &= bits - 1;
      }
      if (_mm_movemask_epi8(_mm_cmpeq_epi8(g, empty))) return nullptr;
      stride += 16;
      pos = (pos + stride) & mask;
    }
  }
  Slot* place(uint64_t k) {  // a free slot for a key known to be absent
    const uint64_t h = swiss_hash(k);
    const __m128i empty = _mm_set1_epi8((char)EMPTY);
    size_t pos = h & mask, stride = 0;
    for (;;) {
      const unsigned bits = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(ctrl + pos)), empty));
      if (bits) {
        const size_t i = (pos + (size_t)__builtin_ctz(bits)) & mask;
        if (ctrl[i] == EMPTY) {  // (a bit of the mirrored tail can alias a full slot of the first group)
          set_ctrl(i, (uint8_t)(h >> 57));
          return &slots[i];
        }
      }
      stride += 16;
      pos = (pos + stride) & mask;
    }
  }
  void grow() {
    const size_t old_n = ctrl ? mask + 1 : 0, n = old_n ? old_n * 2 : 16;
    uint8_t* oc = ctrl;
    Slot* os = slots;
    ctrl = (uint8_t*)raw(n + 16);
    memset(ctrl, EMPTY, n + 16);
    slots = (Slot*)raw(n * sizeof(Slot));
    mask = n - 1;
    growth_left = n / 8 * 7 - items;
    for (size_t i = 0; i < old_n; ++i)
      if (oc[i] != EMPTY) {
        Slot* d = place(os[i].key);
        new (d) Slot(std::move(os[i]));
        os[i].~Slot();
      }
    if (!arena) { free(oc); free(os); }
  }
  // HashMap::insert / HashSet::insert: overwrite or add
  void insert(uint64_t k, V v) {
    if (Slot* s = const_cast<Slot*>(find(k))) { s->val = std::move(v); return; }
    if (growth_left == 0) grow();
    new (place(k)) Slot{k, std::move(v)};
    ++items;
    --growth_left;
  }
};

struct FlatView {  // built once per index state by orc_index_build_flat; any mutation drops it
  Swiss<DocumentDetails> docs;  // bucket = (key, DocumentDetails) inline, its Vec on the heap: hashbrown's layout
  Swiss<char> removed;
};

// Index::count_documents (src/index.rs:282-297) on the flat `removed` set
size_t count_documents_flat(const Index& idx, const FlatView& fv, int64_t node_index) {
  int64_t p = idx.arena_index[(size_t)node_index].first_doc;
  size_t df = 0;
  while (p != NONE) {
    const DocumentPointer& dp = idx.arena_doc[(size_t)p];
    if (!idx.has_removed || fv.removed.find(dp.details_key) == nullptr) df += 1;
    p = dp.next;
  }
  return df;
}

// query() above, line by line, on the flat containers
bool query_flat(const Index& idx, const FlatView& fv, Arena& arena, const char* q, size_t qlen, ScoreCalculator& sc,
                const double* fields_boost, bool canonical, std::vector<QueryResult>& result) {
  arena.reset();
  std::vector<std::string> query_terms = tokenize(q, qlen, nullptr, nullptr);
  Swiss<double> scores(&arena);
  size_t query_terms_len = query_terms.size();
  for (size_t query_term_index = 0; query_term_index < query_terms.size(); ++query_term_index) {
    const std::string& query_term = query_terms[query_term_index];
    if (query_term.empty()) continue;
    std::vector<std::string> expanded_terms = idx.expand_term(query_term);
    Swiss<char> visited_documents_for_term(&arena);
    for (const std::string& query_term_expanded : expanded_terms) {
      int64_t term_node_index = idx.find_inverted_index_node(idx.root, query_term_expanded);
      if (term_node_index == NONE) continue;
      size_t document_frequency = count_documents_flat(idx, fv, term_node_index);
      int64_t first_doc = idx.arena_index[(size_t)term_node_index].first_doc;
      if (first_doc == NONE || document_frequency == 0) continue;
      TermData td{query_term_index, &query_term, &query_term_expanded, query_terms_len};
      PreCalc pre = sc.before_each(td, document_frequency, idx.docs);
      int64_t pointer = first_doc;
      while (pointer != NONE) {
        const DocumentPointer& pb = idx.arena_doc[(size_t)pointer];
        uint64_t key = pb.details_key;
        if (!idx.has_removed || fv.removed.find(key) == nullptr) {
          double s;
          bool some = sc.score(pre.some ? &pre : nullptr, pb, fv.docs.find(key)->val, term_node_index, fields_boost,
                               idx.fields, td, &s);  // docs.get(key).unwrap()
          if (some) {
            const Swiss<double>::Slot* prev = scores.find(key);                          // scores.get(key)
            const bool visited = visited_documents_for_term.find(key) != nullptr;          // visited.contains(key)
            scores.insert(key, max_score_merger(s, prev ? &prev->val : nullptr, visited));  // scores.insert(..)
          }
        }
        visited_documents_for_term.insert(key, 0);  // visited.insert(key)
        pointer = pb.next;
      }
    }
  }
  result.clear();
  result.reserve(scores.items);
  if (scores.ctrl)
    for (size_t i = 0; i <= scores.mask; ++i)
      if (scores.ctrl[i] != Swiss<double>::EMPTY) result.push_back(QueryResult{scores.slots[i].key, scores.slots[i].val});
  sc.finalize(result);
  for (const QueryResult& r : result)
    if (std::isnan(r.score)) return false;
  std::stable_sort(result.begin(), result.end(),
                   [](const QueryResult& a, const QueryResult& b) { return b.score < a.score; });
  if (canonical) {
    std::sort(result.begin(), result.end(), [](const QueryResult& a, const QueryResult& b) {
      if (a.score != b.score) return a.score > b.score;
      return a.key < b.key;
    });
  }
  return true;
}

void build_flat(const Index& idx, FlatView& fv) {
  for (const auto& kv : idx.docs) fv.docs.insert(kv.first, kv.second);
  for (const uint64_t k : idx.removed) fv.removed.insert(k, 0);
}

ScoreCalculator* make_scorer(int kind, double k1, double b) {
  if (kind == 1) { BM25* s = new BM25(); s->bm25k1 = k1; s->bm25b = b; return s; }
  if (kind == 2) return new ZeroToOne();
  if (kind == 3) return new NanProbe();  // test plugin
  return nullptr;
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI for the ctypes test/bench drivers
// ---------------------------------------------------------------------------
extern "C" {

struct orc_str { const char* ptr; size_t len; };
struct orc_result { uint64_t key; double score; };

void* orc_index_new(size_t fields_num) { return new Index(fields_num); }
void orc_index_free(void* h) { delete (Index*)h; }

// values: concatenation over fields of each field's value list; n_values[i] = #values of field i.
int orc_index_add_document(void* h, uint64_t key, const orc_str* values, const size_t* n_values,
                           orc_tokenizer_fn tok, void* user) {
  Index* idx = (Index*)h;
  std::vector<std::vector<std::string>> v(idx->fields.size());
  size_t k = 0;
  for (size_t i = 0; i < idx->fields.size(); ++i)
    for (size_t j = 0; j < n_values[i]; ++j, ++k) v[i].emplace_back(values[k].ptr, values[k].len);
  idx->add_document(key, v, tok, user);
  return 0;
}

// Bulk form: n_docs documents, each with exactly one value per field; value (d, f) is
// text[offsets[d*F+f] .. offsets[d*F+f+1]).  Default tokenizer.
int orc_index_add_documents_flat(void* h, size_t n_docs, const uint64_t* keys, const char* text,
                                 const uint64_t* offsets) {
  Index* idx = (Index*)h;
  size_t F = idx->fields.size();
  std::vector<std::vector<std::string>> v(F);
  for (size_t d = 0; d < n_docs; ++d) {
    for (size_t f = 0; f < F; ++f) {
      v[f].clear();
      v[f].emplace_back(text + offsets[d * F + f], (size_t)(offsets[d * F + f + 1] - offsets[d * F + f]));
    }
    idx->add_document(keys[d], v, nullptr, nullptr);
  }
  return 0;
}

void orc_index_remove_document(void* h, uint64_t key) { ((Index*)h)->remove_document(key); }
void orc_index_vacuum(void* h) { ((Index*)h)->vacuum(); }
size_t orc_index_docs_len(void* h) { return ((Index*)h)->docs.size(); }
size_t orc_index_fields_len(void* h) { return ((Index*)h)->fields.size(); }
void orc_index_field_details(void* h, size_t i, uint64_t* sum, double* avg) {
  Index* idx = (Index*)h;
  *sum = idx->fields[i].sum;
  *avg = idx->fields[i].avg;
}
// field_length of a document; returns 0 if the key is absent, 1 otherwise.
int orc_index_doc_field_length(void* h, uint64_t key, uint64_t* out) {
  Index* idx = (Index*)h;
  auto it = idx->docs.find(key);
  if (it == idx->docs.end()) return 0;
  for (size_t i = 0; i < it->second.field_length.size(); ++i) out[i] = it->second.field_length[i];
  return 1;
}
// count_nodes of src/index.rs:464-481 (live nodes reachable from the root, root included)
static size_t count_rec(const Index* idx, int64_t n) {
  size_t c = 1;
  const InvertedIndexNode& node = idx->arena_index[(size_t)n];
  if (node.first_child != NONE) c += count_rec(idx, node.first_child);
  if (node.next != NONE) c += count_rec(idx, node.next);
  return c;
}
size_t orc_index_count_nodes(void* h) { return count_rec((Index*)h, ((Index*)h)->root); }
size_t orc_index_arena_doc_live(void* h) {
  Index* idx = (Index*)h;
  return idx->arena_doc.size() - idx->arena_doc_free.size();
}
// Children (chars, in list order) of the node reached by `term` ("" = root).  Returns the count
// or -1 if the path does not exist.
long orc_index_children(void* h, const char* term, size_t len, uint32_t* out, size_t cap) {
  Index* idx = (Index*)h;
  int64_t n = idx->find_inverted_index_node(idx->root, std::string(term, len));
  if (n == NONE) return -1;
  long c = 0;
  for (int64_t it = idx->arena_index[(size_t)n].first_child; it != NONE; it = idx->arena_index[(size_t)it].next) {
    if ((size_t)c < cap) out[c] = idx->arena_index[(size_t)it].ch;
    ++c;
  }
  return c;
}
// Posting list of `term` in list order: keys[i], tf[i*F..]; returns pointer count (removed ones
// included, as stored) or -1 if no such node.
long orc_index_postings(void* h, const char* term, size_t len, uint64_t* keys, uint64_t* tf, size_t cap) {
  Index* idx = (Index*)h;
  int64_t n = idx->find_inverted_index_node(idx->root, std::string(term, len));
  if (n == NONE) return -1;
  size_t F = idx->fields.size();
  long c = 0;
  for (int64_t p = idx->arena_index[(size_t)n].first_doc; p != NONE; p = idx->arena_doc[(size_t)p].next) {
    if ((size_t)c < cap) {
      keys[c] = idx->arena_doc[(size_t)p].details_key;
      for (size_t f = 0; f < F; ++f) tf[(size_t)c * F + f] = idx->arena_doc[(size_t)p].term_frequency[f];
    }
    ++c;
  }
  return c;
}
long orc_index_count_documents(void* h, const char* term, size_t len) {
  Index* idx = (Index*)h;
  int64_t n = idx->find_inverted_index_node(idx->root, std::string(term, len));
  if (n == NONE) return -1;
  return (long)idx->count_documents(n);
}

// expand_term: results are written NUL-separated into buf; returns number of terms, *need = bytes needed.
size_t orc_index_expand_term(void* h, const char* term, size_t len, char* buf, size_t cap, size_t* need) {
  std::vector<std::string> r = ((Index*)h)->expand_term(std::string(term, len));
  size_t off = 0;
  for (const std::string& s : r) {
    if (off + s.size() + 1 <= cap) { memcpy(buf + off, s.data(), s.size()); buf[off + s.size()] = 0; }
    off += s.size() + 1;
  }
  *need = off;
  return r.size();
}

// Index::query.  scorer_kind 1 = bm25 {k1,b}, 2 = zero_to_one.  Output is malloc'd; free with
// orc_results_free.  Returns 0 ok, 2 = reference would have panicked (NaN in sort), 1 = bad args.
int orc_index_query(void* h, int scorer_kind, double k1, double b, const char* q, size_t qlen,
                    const double* fields_boost, size_t n_boost, orc_tokenizer_fn tok, void* user, int canonical,
                    orc_result** out, size_t* out_len) {
  Index* idx = (Index*)h;
  if (n_boost < idx->fields.size()) return 1;  // reference: index-out-of-bounds panic (bm25.rs:85)
  ScoreCalculator* sc = make_scorer(scorer_kind, k1, b);
  if (!sc) return 1;
  std::vector<QueryResult> res;
  bool ok = query(*idx, q, qlen, *sc, tok, user, fields_boost, canonical != 0, res);
  delete sc;
  if (!ok) return 2;
  *out_len = res.size();
  *out = (orc_result*)malloc(sizeof(orc_result) * (res.size() ? res.size() : 1));
  for (size_t i = 0; i < res.size(); ++i) { (*out)[i].key = res[i].key; (*out)[i].score = res[i].score; }
  return 0;
}
// The same query through the flat leg (default tokenizer); builds the flat view for this call.
int orc_index_query_flat(void* h, int scorer_kind, double k1, double b, const char* q, size_t qlen,
                         const double* fields_boost, size_t n_boost, int canonical, orc_result** out, size_t* out_len) {
  Index* idx = (Index*)h;
  if (n_boost < idx->fields.size()) return 1;
  ScoreCalculator* sc = make_scorer(scorer_kind, k1, b);
  if (!sc) return 1;
  FlatView fv;
  build_flat(*idx, fv);
  Arena arena;
  std::vector<QueryResult> res;
  bool ok = query_flat(*idx, fv, arena, q, qlen, *sc, fields_boost, canonical != 0, res);
  delete sc;
  if (!ok) return 2;
  *out_len = res.size();
  *out = (orc_result*)malloc(sizeof(orc_result) * (res.size() ? res.size() : 1));
  for (size_t i = 0; i < res.size(); ++i) { (*out)[i].key = res[i].key; (*out)[i].score = res[i].score; }
  return 0;
}
void orc_results_free(orc_result* p) { free(p); }

// CPU-baseline timing leg: runs `n` queries (default tokenizer) with `threads` worker threads, one
// query per thread at a time over the shared read-only index (legal: query(&self), src/query.rs:21-27;
// threads == 1 is the reference's own execution model).  seconds[i] = wall time of query i;
// n_results[i] = result count; returns total wall seconds.  If top_k > 0 the first top_k canonical
// results per query are written to out_topk[i*top_k ..] (key=~0 padding) for cross-checks.
// flavor 0 = the literal leg (std::unordered_*), 1 = the flat leg (SwissTable-class containers + a per-thread arena; its
// flat view of `docs` / `removed` is built before the clock starts, *flat_build_s says how long that took).
double orc_bench_queries(void* h, int scorer_kind, double k1, double b, const orc_str* queries, size_t n,
                         const double* fields_boost, unsigned threads, double* seconds, uint64_t* n_results,
                         size_t top_k, orc_result* out_topk, int flavor, double* flat_build_s) {
  Index* idx = (Index*)h;
  if (threads == 0) threads = 1;
  std::unique_ptr<FlatView> fv;
  if (flavor == 1) {
    auto b0 = std::chrono::steady_clock::now();
    fv.reset(new FlatView());
    build_flat(*idx, *fv);
    if (flat_build_s) *flat_build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - b0).count();
  }
  std::atomic<size_t> next{0};
  auto t0 = std::chrono::steady_clock::now();
  auto worker = [&](unsigned tid) {
    Arena arena;
    (void)tid;
    for (size_t i; (i = next.fetch_add(1, std::memory_order_relaxed)) < n;) {  // a shared queue: a thread that drew a cheap query takes another
      ScoreCalculator* sc = make_scorer(scorer_kind, k1, b);
      std::vector<QueryResult> res;
      auto a = std::chrono::steady_clock::now();
      if (flavor == 1) query_flat(*idx, *fv, arena, queries[i].ptr, queries[i].len, *sc, fields_boost, true, res);
      else query(*idx, queries[i].ptr, queries[i].len, *sc, nullptr, nullptr, fields_boost, true, res);
      auto z = std::chrono::steady_clock::now();
      seconds[i] = std::chrono::duration<double>(z - a).count();
      n_results[i] = res.size();
      if (top_k && out_topk) {
        for (size_t k = 0; k < top_k; ++k) {
          if (k < res.size()) { out_topk[i * top_k + k].key = res[k].key; out_topk[i * top_k + k].score = res[k].score; }
          else { out_topk[i * top_k + k].key = ~0ull; out_topk[i * top_k + k].score = 0.0; }
        }
      }
      delete sc;
    }
  };
  if (threads == 1) worker(0);
  else {
    std::vector<std::thread> ts;
    for (unsigned t = 0; t < threads; ++t) ts.emplace_back(worker, t);
    for (auto& t : ts) t.join();
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
