// ps_snapshot.cpp — flattener (Index -> CSR planes + tile tables + frozen trie) and host query
// planner.  See ps_snapshot.hpp for the layout; reference citations inline.
#include "ps_snapshot.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace ps {

namespace {
struct Frame {
  int32_t node;
  int32_t child;  // next child to visit
  uint32_t fid;
};
inline uint32_t utf8_len(uint32_t cp) { return cp < 0x80 ? 1 : cp < 0x800 ? 2 : cp < 0x10000 ? 3 : 4; }
}  // namespace

namespace {
struct PhaseTimer {  // PS_TRACE=1: where the flattener's time goes
  bool on = getenv("PS_TRACE") && *getenv("PS_TRACE") == '1';
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void mark(const char* what) {
    if (!on) return;
    auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[ps] flatten %-10s %.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};
}  // namespace

Snapshot::Snapshot(const Index& idx, uint32_t tile_docs) {
  PhaseTimer pt;
  F = (uint32_t)idx.fields_len();
  T = tile_docs ? tile_docs : 1024;
  if (T < 256 || T > 4096 || (T & (T - 1))) throw std::invalid_argument("tile_docs must be a power of two in [256, 4096]");
  src_epoch = idx.epoch();
  n_docs = idx.docs_len();
  if (n_docs >= 0xFFFFFFF0ull) throw std::length_error("more than 2^32-16 documents");
  n_tiles = (uint32_t)((n_docs + T - 1) / T);
  if (n_tiles == 0) n_tiles = 1;
  avg.resize(F);
  for (uint32_t x = 0; x < F; ++x) avg[x] = idx.field(x).avg;

  unsigned n_thr = std::thread::hardware_concurrency();
  if (const char* e = getenv("PS_FLATTEN_THREADS")) n_thr = (unsigned)strtoul(e, nullptr, 10);
  n_thr = std::max(1u, std::min(n_thr, 32u));
  // body(i) for i in [0, n), in blocks of 16 handed out to a few threads
  auto for_range = [&](size_t n, bool serial, const std::function<void(size_t)>& body) {
    std::atomic<size_t> next{0};
    std::exception_ptr err;
    std::mutex err_mu;
    auto worker = [&]() {
      try {
        for (;;) {
          const size_t b0 = next.fetch_add(16);
          if (b0 >= n) break;
          for (size_t o = b0; o < std::min(n, b0 + 16); ++o) body(o);
        }
      } catch (...) {
        std::lock_guard<std::mutex> l(err_mu);
        if (!err) err = std::current_exception();
      }
    };
    std::vector<std::thread> th;
    for (unsigned i = 1; i < (serial ? 1u : n_thr); ++i) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    if (err) std::rethrow_exception(err);
  };

  // ---- dense doc ids in ascending key order ------------------------------------------------
  keys.reserve(n_docs);
  for (const auto& kv : idx.docs()) keys.push_back(kv.first);
  std::sort(keys.begin(), keys.end());
  std::vector<uint32_t> fl_by_doc((size_t)n_docs * F);
  const uint64_t max_key = keys.empty() ? 0 : keys.back();
  const bool direct = max_key < 4 * n_docs + 1024;
  std::vector<uint32_t> direct_id;
  std::unordered_map<uint64_t, uint32_t> hashed_id;
  if (direct) direct_id.assign((size_t)max_key + 1, 0xFFFFFFFFu);
  else hashed_id.reserve((size_t)n_docs * 2);
  if (!direct)
    for (size_t i = 0; i < keys.size(); ++i) hashed_id.emplace(keys[i], (uint32_t)i);
  for_range(keys.size(), keys.size() < 4096, [&](size_t i) {
    if (direct) direct_id[(size_t)keys[i]] = (uint32_t)i;
    const DocDetails* d = idx.doc(keys[i]);
    for (uint32_t x = 0; x < F; ++x) fl_by_doc[i * F + x] = d->field_length[x];
  });
  pt.mark("doc ids");
  // LUT geometry: cover field lengths 0..max_fl[x] where the row budget (64 rows = 8 KiB of LDS)
  // allows; longer documents take the inline arithmetic.
  max_fl.assign(F, 0);
  for (size_t i = 0; i < (size_t)n_docs; ++i)
    for (uint32_t x = 0; x < F; ++x) max_fl[x] = std::max(max_fl[x], fl_by_doc[i * F + x]);
  lut_cap.assign(F, 0);
  lut_base.assign(F, 0);
  {
    uint32_t budget = 64;
    for (uint32_t x = 0; x < F; ++x) {
      uint32_t fair = budget / (F - x);
      uint32_t want = (uint32_t)std::min<uint64_t>((uint64_t)max_fl[x] + 1, fair);
      // fields are visited in order; what a short field leaves unused goes to the later ones
      lut_base[x] = lut_rows;
      lut_cap[x] = n_docs ? want : 0;
      lut_rows += lut_cap[x];
      budget -= lut_cap[x];
    }
  }
  auto id_of = [&](uint64_t key) -> uint32_t {
    if (direct) return key <= max_key ? direct_id[(size_t)key] : 0xFFFFFFFFu;
    auto it = hashed_id.find(key);
    return it == hashed_id.end() ? 0xFFFFFFFFu : it->second;
  };

  // ---- DFS pre-order over the trie: frozen node ids + term ordinals ------------------------
  const std::vector<TrieNode>& nodes = idx.nodes();
  const std::vector<PostingList>& lists = idx.lists();
  std::vector<int32_t> term_node;  // term ordinal -> source trie node
  std::vector<uint32_t> node_bytes;  // per frozen node: byte length of its path
  {
    std::vector<Frame> st;
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> kids;  // per frozen node: (char, fid)
    auto enter = [&](int32_t node, uint32_t bytes) -> uint32_t {
      uint32_t fid = (uint32_t)fnodes.size();
      fnodes.push_back(FrozenNode{0, 0, (uint32_t)terms.size(), 0});
      node_bytes.push_back(bytes);
      kids.emplace_back();
      int32_t li = nodes[(size_t)node].list;
      if (li != NIL && !lists[(size_t)li].keys.empty()) {  // first_doc.is_some()  (query.rs:136-138)
        terms.push_back(TermInfo{0, bytes, 0, 0, fid});
        term_node.push_back(node);
      }
      st.push_back(Frame{node, nodes[(size_t)node].first_child, fid});
      return fid;
    };
    enter(idx.root(), 0);
    while (!st.empty()) {
      Frame& f = st.back();
      if (f.child == NIL) {
        fnodes[f.fid].term_end = (uint32_t)terms.size();
        st.pop_back();
        continue;
      }
      int32_t c = f.child;
      uint32_t parent_fid = f.fid;
      f.child = nodes[(size_t)c].next;
      uint32_t bytes = node_bytes[parent_fid] + utf8_len(nodes[(size_t)c].ch);
      uint32_t cf = enter(c, bytes);  // invalidates f
      kids[parent_fid].emplace_back(nodes[(size_t)c].ch, cf);
    }
    for (size_t n = 0; n < fnodes.size(); ++n) {
      std::sort(kids[n].begin(), kids[n].end());
      fnodes[n].child_begin = (uint32_t)fchar.size();
      fnodes[n].child_count = (uint32_t)kids[n].size();
      for (auto& kc : kids[n]) { fchar.push_back(kc.first); fchild.push_back(kc.second); }
    }
  }

  pt.mark("trie");
  // ---- postings: per term, newest-first walk -> doc-sorted layer(s) -------------------------
  // Two passes over the terms, both spread over a few threads (terms are independent):
  //   1. count the live records of every term and notice whether its walk is already doc-sorted;
  //      the general case (keys added out of order / re-added) builds its version layers here
  //   2. after one serial prefix sum has fixed every layer's place, write the postings and the
  //      tile-offset tables straight into the final planes (no staging copy)
  const bool any_removed = idx.any_removed();
  struct TermFlat {
    uint32_t live = 0;
    bool sorted_desc = true;
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> lay;  // general case: (doc id, record) per layer
  };
  std::vector<TermFlat> flat(terms.size());
  auto for_terms = [&](const std::function<void(size_t)>& body) { for_range(terms.size(), terms.size() < 256, body); };

  for_terms([&](size_t o) {
    const PostingList& pl = lists[(size_t)nodes[(size_t)term_node[o]].list];
    TermInfo& ti = terms[o];
    TermFlat& tfl = flat[o];
    const size_t nrec = pl.keys.size();
    uint32_t prev = 0xFFFFFFFFu;
    for (size_t r = nrec; r-- > 0;) {  // newest -> oldest
      const uint64_t key = pl.keys[r];
      if (any_removed && idx.is_removed(key)) continue;  // query.rs:65, index.rs:287-293
      const uint32_t id = id_of(key);
      if (id == 0xFFFFFFFFu) continue;
      for (uint32_t x = 0; x < F; ++x) ti.df_raw += pl.tf[r * F + x];
      if (tfl.live && id >= prev) tfl.sorted_desc = false;  // doc ids must strictly descend
      prev = id;
      ++tfl.live;
    }
    if (tfl.live == 0 || tfl.sorted_desc) return;
    // General case (keys added out of order, or a key re-added without removal).  Per doc the
    // distinct adjacent tf versions, newest first, become layer 0, 1, ...; the planner emits
    // the layers as consecutive entries of the same query term, which reproduces the
    // reference's walk: first version adds (or assigns), later ones take max (query.rs:150-164).
    std::vector<std::pair<uint32_t, uint32_t>> tmp;
    tmp.reserve(tfl.live);
    for (size_t r = nrec; r-- > 0;) {
      const uint64_t key = pl.keys[r];
      if (any_removed && idx.is_removed(key)) continue;
      const uint32_t id = id_of(key);
      if (id != 0xFFFFFFFFu) tmp.emplace_back(id, (uint32_t)r);
    }
    std::sort(tmp.begin(), tmp.end(), [](const auto& a, const auto& b) {
      return a.first != b.first ? a.first < b.first : a.second > b.second;
    });
    for (size_t i = 0; i < tmp.size();) {
      size_t j = i;
      uint32_t v = 0;
      while (j < tmp.size() && tmp[j].first == tmp[i].first) {
        const bool same = j > i && std::equal(pl.tf.begin() + (long)((size_t)tmp[j].second * F),
                                              pl.tf.begin() + (long)((size_t)(tmp[j].second + 1) * F),
                                              pl.tf.begin() + (long)((size_t)tmp[j - 1].second * F));
        if (!same) {
          if (tfl.lay.size() <= v) tfl.lay.emplace_back();
          tfl.lay[v].push_back(tmp[j]);
          ++v;
        }
        ++j;
      }
      i = j;
    }
  });
  pt.mark("count");

  // serial: place every layer (4-aligned starts, so 16-byte vector loads never straddle lists)
  // and its tile-offset table
  uint64_t cursor = 0, tcursor = 0;
  for (size_t o = 0; o < terms.size(); ++o) {
    TermInfo& ti = terms[o];
    TermFlat& tfl = flat[o];
    if (tfl.live == 0) continue;
    ++n_live_terms;
    n_pointers += ti.df_raw;
    ti.first_layer = (uint32_t)layers.size();
    ti.n_layers = tfl.sorted_desc ? 1u : (uint32_t)tfl.lay.size();
    max_layers = std::max(max_layers, ti.n_layers);
    for (uint32_t l = 0; l < ti.n_layers; ++l) {
      LayerInfo L{0, 0, 0, 0};
      L.post_off = cursor;
      L.len = tfl.sorted_desc ? tfl.live : (uint32_t)tfl.lay[l].size();
      cursor = (cursor + L.len + 3) & ~(uint64_t)3;
      n_postings += L.len;
      // smallest shift with slots <= max(1, len/2): <= 2 table bytes per posting overall
      uint32_t shift = 0;
      const uint64_t want = std::max<uint64_t>(1, L.len / 2);
      while ((((uint64_t)n_tiles - 1) >> shift) + 1 > want) ++shift;
      const uint32_t slots = ((n_tiles - 1) >> shift) + 1;
      L.shift = shift;
      if (tcursor + slots + 1 >= 0xFFFFFFFFull) throw std::length_error("tile-offset table exceeds 2^32 entries");
      L.tbl_off = (uint32_t)tcursor;
      tcursor += slots + 1;
      layers.push_back(L);
    }
  }
  P = std::max<uint64_t>(cursor, 4);  // keep planes non-empty so device pointers are always valid
  doc.resize(P);
  tf.resize((size_t)P * F);
  fl.resize((size_t)P * F);
  table.resize(std::max<uint64_t>(tcursor, 1));
  if (tcursor == 0) table[0] = 0;
  if (cursor == 0) {
    std::fill(doc.begin(), doc.end(), 0xFFFFFFFFu);
    std::fill(tf.begin(), tf.end(), 0u);
    std::fill(fl.begin(), fl.end(), 0u);
  }
  pt.mark("place");

  for_terms([&](size_t o) {
    const TermInfo& ti = terms[o];
    const TermFlat& tfl = flat[o];
    if (tfl.live == 0) return;
    const PostingList& pl = lists[(size_t)nodes[(size_t)term_node[o]].list];
    for (uint32_t l = 0; l < ti.n_layers; ++l) {
      const LayerInfo& L = layers[ti.first_layer + l];
      uint32_t* d = doc.data() + L.post_off;
      auto put = [&](uint64_t at, uint32_t id, uint32_t r) {
        d[at] = id;
        for (uint32_t x = 0; x < F; ++x) {
          tf[(size_t)x * P + L.post_off + at] = pl.tf[(size_t)r * F + x];
          fl[(size_t)x * P + L.post_off + at] = fl_by_doc[(size_t)id * F + x];
        }
      };
      if (tfl.sorted_desc) {
        // the newest -> oldest walk descends in doc id: fill from the back
        uint64_t at = L.len;
        for (size_t r = pl.keys.size(); r-- > 0;) {
          const uint64_t key = pl.keys[r];
          if (any_removed && idx.is_removed(key)) continue;
          const uint32_t id = id_of(key);
          if (id == 0xFFFFFFFFu) continue;
          put(--at, id, (uint32_t)r);
        }
      } else {
        for (size_t i = 0; i < tfl.lay[l].size(); ++i) put(i, tfl.lay[l][i].first, tfl.lay[l][i].second);
      }
      for (uint64_t at = L.len; at < ((L.len + 3) & ~(uint64_t)3); ++at) {  // pad postings match nothing
        d[at] = 0xFFFFFFFFu;
        for (uint32_t x = 0; x < F; ++x) { tf[(size_t)x * P + L.post_off + at] = 0; fl[(size_t)x * P + L.post_off + at] = 0; }
      }
      // tile-offset table: first posting of every (group of) tile(s), + the end
      const uint32_t slots = ((n_tiles - 1) >> L.shift) + 1;
      uint32_t pos = 0;
      for (uint32_t sl = 0; sl < slots; ++sl) {
        const uint64_t first_doc = ((uint64_t)sl << L.shift) * T;
        while (pos < L.len && d[pos] < first_doc) ++pos;
        table[L.tbl_off + sl] = pos;
      }
      table[L.tbl_off + slots] = L.len;
    }
  });
  { std::vector<TermFlat>().swap(flat); }
  pt.mark("planes");
}

// ---- on-disk snapshot ---------------------------------------------------------------------------
namespace {
constexpr char MAGIC[8] = {'P', 'S', 'N', 'A', 'P', '0', '0', '1'};

struct File {
  FILE* f;
  File(const std::string& path, const char* mode) : f(fopen(path.c_str(), mode)) {
    if (!f) throw std::invalid_argument("cannot open snapshot file: " + path);
  }
  ~File() { if (f) fclose(f); }
  void put(const void* p, size_t n) { if (n && fwrite(p, 1, n, f) != n) throw std::runtime_error("snapshot write failed"); }
  void get(void* p, size_t n) { if (n && fread(p, 1, n, f) != n) throw std::invalid_argument("snapshot file truncated"); }
  template <typename Vec> void put_vec(const Vec& v) {
    uint64_t n = v.size();
    put(&n, 8);
    put(v.data(), n * sizeof(typename Vec::value_type));
  }
  template <typename Vec> void get_vec(Vec& v) {
    uint64_t n = 0;
    get(&n, 8);
    if (n > (1ull << 40) / sizeof(typename Vec::value_type)) throw std::invalid_argument("snapshot file corrupt");
    v.resize((size_t)n);
    get(v.data(), n * sizeof(typename Vec::value_type));
  }
};
}  // namespace

void Snapshot::save(const std::string& path) const {
  File f(path, "wb");
  f.put(MAGIC, 8);
  const uint64_t hdr[10] = {F, T, n_tiles, n_docs, P, n_postings, n_pointers, n_live_terms, max_layers, lut_rows};
  f.put(hdr, sizeof(hdr));
  f.put_vec(keys); f.put_vec(avg); f.put_vec(terms); f.put_vec(layers); f.put_vec(fnodes);
  f.put_vec(fchar); f.put_vec(fchild); f.put_vec(doc); f.put_vec(tf); f.put_vec(fl); f.put_vec(table);
  f.put_vec(max_fl); f.put_vec(lut_cap); f.put_vec(lut_base);
}

Snapshot::Snapshot(const std::string& path) {
  File f(path, "rb");
  char magic[8];
  f.get(magic, 8);
  if (memcmp(magic, MAGIC, 8) != 0) throw std::invalid_argument("not a probly-search_amd snapshot (bad magic/version)");
  uint64_t hdr[10];
  f.get(hdr, sizeof(hdr));
  F = (uint32_t)hdr[0]; T = (uint32_t)hdr[1]; n_tiles = (uint32_t)hdr[2]; n_docs = hdr[3]; P = hdr[4];
  n_postings = hdr[5]; n_pointers = hdr[6]; n_live_terms = hdr[7]; max_layers = (uint32_t)hdr[8];
  lut_rows = (uint32_t)hdr[9];
  f.get_vec(keys); f.get_vec(avg); f.get_vec(terms); f.get_vec(layers); f.get_vec(fnodes);
  f.get_vec(fchar); f.get_vec(fchild); f.get_vec(doc); f.get_vec(tf); f.get_vec(fl); f.get_vec(table);
  f.get_vec(max_fl); f.get_vec(lut_cap); f.get_vec(lut_base);
  if (keys.size() != n_docs || avg.size() != F || doc.size() != P || tf.size() != (size_t)P * F ||
      fl.size() != (size_t)P * F || fnodes.empty() || T < 256 || (T & (T - 1)))
    throw std::invalid_argument("snapshot file inconsistent");
  src_epoch = ~0ull;
}

int64_t Snapshot::find_fnode(std::string_view term) const {
  uint32_t n = 0;
  size_t i = 0;
  while (i < term.size()) {
    uint32_t ch = next_char(term, i);
    const FrozenNode& fn = fnodes[n];
    const uint32_t* b = fchar.data() + fn.child_begin;
    const uint32_t* e = b + fn.child_count;
    const uint32_t* it = std::lower_bound(b, e, ch);
    if (it == e || *it != ch) return -1;
    n = fchild[fn.child_begin + (uint32_t)(it - b)];
  }
  return (int64_t)n;
}

void Snapshot::plan_query(const ps_scorer_desc& sc, std::string_view q, ps_tokenizer_fn tok, void* user,
                          Plan& plan) const {
  thread_local std::vector<const char*> sp;
  thread_local std::vector<size_t> sl;
  std::vector<std::string_view> tokens = tokenize(q, tok, user, sp, sl);
  if (plan.qbeg.empty()) plan.qbeg.push_back(0);
  const size_t e_begin = plan.entries.size();
  uint32_t qord = 0;
  std::vector<uint32_t> seen_nodes;  // zero_to_one: distinct trie nodes hit by this query
  for (size_t qi = 0; qi < tokens.size(); ++qi) {
    std::string_view qt = tokens[qi];
    if (qt.empty()) continue;  // query.rs:35 (still counted in query_terms_len, query.rs:32)
    int64_t fn = find_fnode(qt);
    size_t before = plan.entries.size();
    if (fn >= 0) {
      const FrozenNode& node = fnodes[(size_t)fn];
      for (uint32_t o = node.term_begin; o < node.term_end; ++o) {  // == expand_term order
        const TermInfo& t = terms[o];
        if (t.df_raw == 0 || t.n_layers == 0) continue;  // query.rs:47-48
        ps_plan_entry e;
        memset(&e, 0, sizeof(e));
        e.qterm = qord;
        e.qterm_index = (uint32_t)qi;
        if (sc.kind == PS_SCORER_BM25) {
          // BM25::before_each, src/score/default/bm25.rs:35-58
          uint64_t frequency = std::min<uint64_t>(n_docs, t.df_raw);
          uint64_t diff = n_docs - frequency;
          e.boost = (t.fnode == (uint32_t)fn)
                        ? 1.0
                        : std::log(1.0 + (1.0 / (1.0 + (double)t.byte_len - (double)qt.size())));
          e.idf = std::log(1.0 + ((double)diff + 0.5) / ((double)frequency + 0.5));
        } else {
          // ScoreByTerm::score, src/score/default/zero_to_one.rs:57-73
          double term_exp_len = (double)t.byte_len, term_len = (double)qt.size();
          e.boost = 1.0 - std::fabs(term_exp_len - term_len) / term_exp_len;
          size_t k = 0;
          while (k < seen_nodes.size() && seen_nodes[k] != t.fnode) ++k;
          if (k == seen_nodes.size()) seen_nodes.push_back(t.fnode);
          e.node = (uint32_t)k;
        }
        for (uint32_t l = 0; l < t.n_layers; ++l) {
          const LayerInfo& L = layers[t.first_layer + l];
          e.post_off = L.post_off;
          e.len = L.len;
          e.tbl_off = L.tbl_off;
          e.shift = L.shift | (l << 8);  // bits 8.. = version layer (0 = newest)
          plan.entries.push_back(e);
          plan.postings += L.len;
        }
      }
    }
    if (plan.entries.size() - before > 1) plan.multi_expansion = true;
    ++qord;
  }
  if (sc.kind == PS_SCORER_ZERO_TO_ONE && plan.entries.size() - e_begin <= 64) {
    // zero_to_one's per-node pool (zero_to_one.rs:104-113) is tracked on the GPU as "how many
    // already-consumed entries of this query share my trie node": ship that relation as a bitmask
    // over the query's entries in the otherwise unused idf slot.
    for (size_t i = e_begin; i < plan.entries.size(); ++i) {
      uint64_t mask = 0;
      for (size_t j = e_begin; j < plan.entries.size(); ++j)
        if (plan.entries[j].node == plan.entries[i].node) mask |= 1ull << (j - e_begin);
      memcpy(&plan.entries[i].idf, &mask, 8);
    }
  }
  plan.qbeg.push_back((uint32_t)plan.entries.size());
  plan.qterms_len.push_back((uint32_t)tokens.size());
  plan.n_nodes.push_back((uint32_t)seen_nodes.size());
  plan.max_entries = std::max<uint32_t>(plan.max_entries, (uint32_t)(plan.entries.size() - e_begin));
  plan.max_qterms = std::max(plan.max_qterms, qord);
  plan.max_nodes = std::max<uint32_t>(plan.max_nodes, (uint32_t)seen_nodes.size());
}

}  // namespace ps
