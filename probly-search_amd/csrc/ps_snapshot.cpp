// ps_snapshot.cpp — flattener (Index -> CSR planes + tile tables + frozen trie) and host query
// planner.  See ps_snapshot.hpp for the layout; reference citations inline.
#include "ps_snapshot.hpp"

#include <algorithm>
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

Snapshot::Snapshot(const Index& idx, uint32_t tile_docs) {
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
  for (size_t i = 0; i < keys.size(); ++i) {
    if (direct) direct_id[(size_t)keys[i]] = (uint32_t)i;
    else hashed_id.emplace(keys[i], (uint32_t)i);
    const DocDetails* d = idx.doc(keys[i]);
    for (uint32_t x = 0; x < F; ++x) fl_by_doc[i * F + x] = d->field_length[x];
  }
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

  // ---- postings: per term, newest-first walk -> doc-sorted layer(s) -------------------------
  std::vector<std::vector<uint32_t>> tfv(F), flv(F);
  std::vector<std::pair<uint32_t, uint32_t>> tmp;  // (doc id, record)
  const bool any_removed = idx.any_removed();
  auto pad4 = [&]() {
    while (doc.size() & 3) {
      doc.push_back(0xFFFFFFFFu);
      for (uint32_t x = 0; x < F; ++x) { tfv[x].push_back(0); flv[x].push_back(0); }
    }
  };
  auto emit_table = [&](LayerInfo& L) {
    // smallest shift with slots <= max(1, len/2): <= 2 table bytes per posting overall
    uint32_t shift = 0;
    uint64_t want = std::max<uint64_t>(1, L.len / 2);
    while ((((uint64_t)n_tiles - 1) >> shift) + 1 > want) ++shift;
    uint32_t slots = ((n_tiles - 1) >> shift) + 1;
    L.shift = shift;
    if (table.size() + slots + 1 >= 0xFFFFFFFFull) throw std::length_error("tile-offset table exceeds 2^32 entries");
    L.tbl_off = (uint32_t)table.size();
    const uint32_t* d = doc.data() + L.post_off;
    uint32_t pos = 0;
    for (uint32_t s = 0; s < slots; ++s) {
      uint64_t first_doc = ((uint64_t)s << shift) * T;
      while (pos < L.len && d[pos] < first_doc) ++pos;
      table.push_back(pos);
    }
    table.push_back(L.len);
  };

  for (size_t o = 0; o < terms.size(); ++o) {
    const PostingList& pl = lists[(size_t)nodes[(size_t)term_node[o]].list];
    TermInfo& ti = terms[o];
    const size_t nrec = pl.keys.size();
    tmp.clear();
    bool sorted_desc = true;  // doc ids strictly descending while walking newest -> oldest
    uint32_t prev = 0xFFFFFFFFu;
    for (size_t r = nrec; r-- > 0;) {
      uint64_t key = pl.keys[r];
      if (any_removed && idx.is_removed(key)) continue;  // query.rs:65, index.rs:287-293
      uint32_t id = id_of(key);
      if (id == 0xFFFFFFFFu) continue;
      for (uint32_t x = 0; x < F; ++x) ti.df_raw += pl.tf[r * F + x];
      if (!tmp.empty() && id >= prev) sorted_desc = false;
      prev = id;
      tmp.emplace_back(id, (uint32_t)r);
    }
    if (tmp.empty()) continue;
    ++n_live_terms;
    n_pointers += ti.df_raw;
    ti.first_layer = (uint32_t)layers.size();
    auto push = [&](uint32_t id, uint32_t r) {
      doc.push_back(id);
      for (uint32_t x = 0; x < F; ++x) {
        tfv[x].push_back(pl.tf[(size_t)r * F + x]);
        flv[x].push_back(fl_by_doc[(size_t)id * F + x]);
      }
    };
    if (sorted_desc) {
      pad4();
      LayerInfo L{doc.size(), (uint32_t)tmp.size(), 0, 0};
      for (size_t i = tmp.size(); i-- > 0;) push(tmp[i].first, tmp[i].second);
      layers.push_back(L);
      ti.n_layers = 1;
    } else {
      // General case (keys added out of order, or a key re-added without removal).  Per doc the
      // distinct adjacent tf versions, newest first, become layer 0, 1, ...; the planner emits
      // the layers as consecutive entries of the same query term, which reproduces the
      // reference's walk: first version adds (or assigns), later ones take max (query.rs:150-164).
      std::sort(tmp.begin(), tmp.end(), [](const auto& a, const auto& b) {
        return a.first != b.first ? a.first < b.first : a.second > b.second;
      });
      std::vector<std::vector<std::pair<uint32_t, uint32_t>>> lay;
      for (size_t i = 0; i < tmp.size();) {
        size_t j = i;
        uint32_t v = 0;
        while (j < tmp.size() && tmp[j].first == tmp[i].first) {
          bool same = j > i && std::equal(pl.tf.begin() + (long)((size_t)tmp[j].second * F),
                                          pl.tf.begin() + (long)((size_t)(tmp[j].second + 1) * F),
                                          pl.tf.begin() + (long)((size_t)tmp[j - 1].second * F));
          if (!same) {
            if (lay.size() <= v) lay.emplace_back();
            lay[v].push_back(tmp[j]);
            ++v;
          }
          ++j;
        }
        i = j;
      }
      for (auto& lv : lay) {
        pad4();
        LayerInfo L{doc.size(), (uint32_t)lv.size(), 0, 0};
        for (auto& pr : lv) push(pr.first, pr.second);
        layers.push_back(L);
      }
      ti.n_layers = (uint32_t)lay.size();
      max_layers = std::max(max_layers, ti.n_layers);
    }
    for (uint32_t l = 0; l < ti.n_layers; ++l) n_postings += layers[ti.first_layer + l].len;
  }
  pad4();
  P = doc.size();
  if (P == 0) {  // keep planes non-empty so device pointers are always valid
    P = 4;
    doc.assign(4, 0xFFFFFFFFu);
    for (uint32_t x = 0; x < F; ++x) { tfv[x].assign(4, 0); flv[x].assign(4, 0); }
  }
  for (LayerInfo& L : layers) emit_table(L);
  if (table.empty()) table.push_back(0);
  tf.resize((size_t)P * F);
  fl.resize((size_t)P * F);
  for (uint32_t x = 0; x < F; ++x) {
    memcpy(tf.data() + (size_t)x * P, tfv[x].data(), (size_t)P * 4);
    memcpy(fl.data() + (size_t)x * P, flv[x].data(), (size_t)P * 4);
    std::vector<uint32_t>().swap(tfv[x]);
    std::vector<uint32_t>().swap(flv[x]);
  }
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
  template <typename V> void put_vec(const std::vector<V>& v) {
    uint64_t n = v.size();
    put(&n, 8);
    put(v.data(), n * sizeof(V));
  }
  template <typename V> void get_vec(std::vector<V>& v) {
    uint64_t n = 0;
    get(&n, 8);
    if (n > (1ull << 40) / sizeof(V)) throw std::invalid_argument("snapshot file corrupt");
    v.resize((size_t)n);
    get(v.data(), n * sizeof(V));
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
