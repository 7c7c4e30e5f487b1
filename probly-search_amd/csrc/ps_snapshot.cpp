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
#include <type_traits>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

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

struct SnapshotStorage {
  std::vector<uint64_t> keys;
  std::vector<double> avg;
  std::vector<TermInfo> terms;
  std::vector<LayerInfo> layers;
  std::vector<FrozenNode> fnodes;
  std::vector<uint32_t> fchar, fchild;
  PlaneVec doc, tf, fl, table;
  std::vector<uint32_t> bits;
  std::vector<uint32_t> max_fl, lut_cap, lut_base;
  std::vector<uint32_t> alive;
  // delta bookkeeping (never saved): source trie node of every term ordinal, df_raw before the
  // removed documents are subtracted
  std::vector<int32_t> term_node;
  std::vector<uint64_t> df_total;
  std::unordered_map<int32_t, uint32_t> ord_of;  // source trie node -> term ordinal: built by the first delta with additions, redone when the trie is re-frozen
  // doc ids for which the flattener folded a duplicate record (a key re-added without removal whose
  // term frequencies did not change): df_raw counts that record (index.rs:282-297 counts every
  // pointer), but no posting stands for it, so the per-posting re-count of a delta removal would
  // leave df too high.  Removing such a document is "not expressible": apply_delta re-flattens.
  std::vector<uint32_t> folded_ids;  // sorted, unique
};

namespace {
// DFS pre-order over the index's trie, children newest-first (query.rs:130-147): frozen node ids,
// term ordinals (a term = a node whose list holds a posting, query.rs:136-138), children sorted by
// char for binary search.  No posting work: cheap enough to redo on every delta.
struct FrozenTrie {
  std::vector<FrozenNode> fnodes;
  std::vector<uint32_t> fchar, fchild;
  std::vector<int32_t> term_node;    // term ordinal -> source trie node
  std::vector<uint32_t> term_bytes;  // byte length of the term
  std::vector<uint32_t> term_fnode;
};

void freeze_trie(const Index& idx, FrozenTrie& ft) {
  const std::vector<TrieNode>& nodes = idx.nodes();
  const std::vector<PostingList>& lists = idx.lists();
  std::vector<uint32_t> node_bytes;  // per frozen node: byte length of its path
  std::vector<Frame> st;
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> kids;  // per frozen node: (char, fid)
  auto enter = [&](int32_t node, uint32_t bytes) -> uint32_t {
    uint32_t fid = (uint32_t)ft.fnodes.size();
    ft.fnodes.push_back(FrozenNode{0, 0, (uint32_t)ft.term_node.size(), 0});
    node_bytes.push_back(bytes);
    kids.emplace_back();
    int32_t li = nodes[(size_t)node].list;
    if (li != NIL && !lists[(size_t)li].keys.empty()) {  // first_doc.is_some()  (query.rs:136-138)
      ft.term_node.push_back(node);
      ft.term_bytes.push_back(bytes);
      ft.term_fnode.push_back(fid);
    }
    st.push_back(Frame{node, nodes[(size_t)node].first_child, fid});
    return fid;
  };
  enter(idx.root(), 0);
  while (!st.empty()) {
    Frame& f = st.back();
    if (f.child == NIL) {
      ft.fnodes[f.fid].term_end = (uint32_t)ft.term_node.size();
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
  for (size_t n = 0; n < ft.fnodes.size(); ++n) {
    std::sort(kids[n].begin(), kids[n].end());
    ft.fnodes[n].child_begin = (uint32_t)ft.fchar.size();
    ft.fnodes[n].child_count = (uint32_t)kids[n].size();
    for (auto& kc : kids[n]) { ft.fchar.push_back(kc.first); ft.fchild.push_back(kc.second); }
  }
}
}  // namespace

void Snapshot::bind(const SnapshotStorage& st) {
  auto v = [](const auto& vec) { return View<typename std::decay_t<decltype(vec)>::value_type>{vec.data(), vec.size()}; };
  keys = v(st.keys); avg = v(st.avg); terms = v(st.terms); layers = v(st.layers); fnodes = v(st.fnodes);
  fchar = v(st.fchar); fchild = v(st.fchild); doc = v(st.doc); tf = v(st.tf); fl = v(st.fl); table = v(st.table); bits = v(st.bits);
  max_fl = v(st.max_fl); lut_cap = v(st.lut_cap); lut_base = v(st.lut_base); alive = v(st.alive);
}

Snapshot::Snapshot(const Index& idx, uint32_t tile_docs, uint32_t headroom_pct) {
  PhaseTimer pt;
  idx.enable_change_log();  // from here on this snapshot can be brought up to date by replaying the index's log
  own_.reset(new SnapshotStorage());
  // the flattener fills the owned vectors (these references shadow the read-only views, which are
  // bound to the finished vectors at the end)
  auto& keys = own_->keys; auto& avg = own_->avg; auto& terms = own_->terms; auto& layers = own_->layers;
  auto& fnodes = own_->fnodes; auto& fchar = own_->fchar; auto& fchild = own_->fchild;
  auto& doc = own_->doc; auto& tf = own_->tf; auto& fl = own_->fl; auto& table = own_->table; auto& bits = own_->bits;
  auto& max_fl = own_->max_fl; auto& lut_cap = own_->lut_cap; auto& lut_base = own_->lut_base;
  F = (uint32_t)idx.fields_len();
  T = tile_docs ? tile_docs : 1024;
  if (T < 256 || T > 4096 || (T & (T - 1))) throw std::invalid_argument("tile_docs must be a power of two in [256, 4096]");
  src_epoch = idx.epoch();
  src_uid = idx.uid();
  n_docs = idx.docs_len();
  if (n_docs >= 0xFFFFFFF0ull) throw std::length_error("more than 2^32-16 documents");
  n_ids = n_docs;
  n_tiles = (uint32_t)((n_docs + T - 1) / T);
  if (n_tiles == 0) n_tiles = 1;
  // headroom: tables, bitmaps and rows are laid out for tiles_cap tiles, the planes for extra postings
  headroom_pct = std::min(headroom_pct, 400u);
  tiles_cap = n_tiles;
  if (headroom_pct) tiles_cap = (uint32_t)std::min<uint64_t>(0xFFFFFFF0ull / T, n_tiles + std::max<uint64_t>(1, (uint64_t)n_tiles * headroom_pct / 100));
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
  std::vector<int32_t>& term_node = own_->term_node;  // term ordinal -> source trie node
  {
    FrozenTrie ft;
    freeze_trie(idx, ft);
    fnodes = std::move(ft.fnodes);
    fchar = std::move(ft.fchar);
    fchild = std::move(ft.fchild);
    term_node = std::move(ft.term_node);
    terms.resize(term_node.size());
    for (size_t o = 0; o < terms.size(); ++o) terms[o] = TermInfo{0, ft.term_bytes[o], 0, 0, ft.term_fnode[o], NO_LAYER, 0};
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
    std::vector<uint32_t> folded;  // doc ids with a duplicate record that became no posting
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
        } else {
          tfl.folded.push_back(tmp[j].first);
        }
        ++j;
      }
      i = j;
    }
  });
  for (const TermFlat& tf_ : flat) own_->folded_ids.insert(own_->folded_ids.end(), tf_.folded.begin(), tf_.folded.end());
  std::sort(own_->folded_ids.begin(), own_->folded_ids.end());
  own_->folded_ids.erase(std::unique(own_->folded_ids.begin(), own_->folded_ids.end()), own_->folded_ids.end());
  pt.mark("count");

  // serial: place every layer (4-aligned starts, so 16-byte vector loads never straddle lists)
  // and its tile-offset table
  uint64_t cursor = 0, tcursor = 0, bcursor = 0;
  const uint64_t bm_words = 2 * (((uint64_t)tiles_cap * T + 31) / 32);  // {bits, postings before} per 32 documents
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
      LayerInfo L{0, 0, 0, 0, NO_BITMAP, NO_LAYER, 0};
      L.post_off = cursor;
      L.len = tfl.sorted_desc ? tfl.live : (uint32_t)tfl.lay[l].size();
      cursor = (cursor + L.len + 3) & ~(uint64_t)3;
      n_postings += L.len;
      // smallest shift with slots <= max(1, len/2): <= 2 table bytes per posting overall
      uint32_t shift = 0;
      const uint64_t want = std::max<uint64_t>(1, L.len / 2);
      while ((((uint64_t)tiles_cap - 1) >> shift) + 1 > want) ++shift;
      const uint32_t slots = ((tiles_cap - 1) >> shift) + 1;
      L.shift = shift;
      if (tcursor + slots + 1 >= 0xFFFFFFFFull) throw std::length_error("tile-offset table exceeds 2^32 entries");
      L.tbl_off = (uint32_t)tcursor;
      tcursor += slots + 1;
      if ((uint64_t)L.len * 128 >= n_docs && L.len >= 64 && bcursor + bm_words < 0xFFFFFFF0ull) {
        L.bm_off = (uint32_t)bcursor;
        bcursor += bm_words;
      }
      layers.push_back(L);
    }
  }
  P_used = std::max<uint64_t>(cursor, 4);  // keep planes non-empty so device pointers are always valid
  P = P_used;
  if (headroom_pct) P = (P_used + std::max<uint64_t>(4096, P_used * headroom_pct / 100) + 3) & ~(uint64_t)3;
  table_used = std::max<uint64_t>(tcursor, 1);
  doc.resize(P);
  tf.resize((size_t)P * F);
  fl.resize((size_t)P * F);
  table.resize(headroom_pct ? table_used + std::max<uint64_t>(65536, table_used * headroom_pct / 100) : table_used);
  if (tcursor == 0) table[0] = 0;
  bits.assign(std::max<uint64_t>(bcursor, 2), 0u);
  if (P > P_used) {  // the headroom is part of the saved file: keep it deterministic
    std::fill(doc.begin() + (long)P_used, doc.end(), 0xFFFFFFFFu);
    for (uint32_t x = 0; x < F; ++x) {
      std::fill(tf.begin() + (long)((size_t)x * P + P_used), tf.begin() + (long)((size_t)(x + 1) * P), 0u);
      std::fill(fl.begin() + (long)((size_t)x * P + P_used), fl.begin() + (long)((size_t)(x + 1) * P), 0u);
    }
  }
  if (table.size() > table_used) std::fill(table.begin() + (long)table_used, table.end(), 0u);
  keys.reserve((size_t)tiles_cap * T);
  own_->alive.assign(((size_t)tiles_cap * T + 31) / 32, 0u);
  for (uint64_t i = 0; i < n_ids; ++i) own_->alive[i >> 5] |= 1u << (i & 31u);
  own_->df_total.assign(terms.size(), 0);
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
      const uint32_t slots = ((tiles_cap - 1) >> L.shift) + 1;
      uint32_t pos = 0;
      for (uint32_t sl = 0; sl < slots; ++sl) {
        const uint64_t first_doc = ((uint64_t)sl << L.shift) * T;
        while (pos < L.len && d[pos] < first_doc) ++pos;
        table[L.tbl_off + sl] = pos;
      }
      table[L.tbl_off + slots] = L.len;
      if (L.bm_off != NO_BITMAP) {
        uint32_t* bm = bits.data() + L.bm_off;
        for (uint32_t i = 0; i < L.len; ++i) bm[2 * (size_t)(d[i] >> 5)] |= 1u << (d[i] & 31u);
        uint32_t before = 0;
        for (uint64_t w = 0; w < bm_words / 2; ++w) { bm[2 * w + 1] = before; before += (uint32_t)__builtin_popcount(bm[2 * w]); }
      }
    }
  });
  for (size_t o = 0; o < terms.size(); ++o) own_->df_total[o] = terms[o].df_raw;
  removed_df.assign(layers.size(), 0);
  { std::vector<TermFlat>().swap(flat); }
  pt.mark("planes");
  bind(*own_);
}

// ---- on-disk snapshot ---------------------------------------------------------------------------
// File = one 4096-byte header page + 16 sections, each starting on a 4096-byte boundary and holding
// one array exactly as it lives in memory (little-endian, natural alignment):
//   header: magic "PSNAP004" | u64 file_bytes | u64 scalars[16] | 16 x {u64 offset, u64 bytes, u64 checksum}
// Loading maps the file read-only and points the views at the sections; nothing is parsed or copied.
namespace {
constexpr char MAGIC[8] = {'P', 'S', 'N', 'A', 'P', '0', '0', '4'};
constexpr size_t PAGE = 4096;
constexpr int N_SECTIONS = 16;
struct SectionRef { uint64_t offset, bytes, checksum; };
struct FileHeader {
  char magic[8];
  uint64_t file_bytes;
  uint64_t scalars[16];  // F, T, n_tiles, n_docs, P, n_postings, n_pointers, n_live_terms, max_layers, lut_rows,
                         // n_ids, tiles_cap, P_used, table_used, any_dead, (reserved)
  SectionRef sec[N_SECTIONS];
};
static_assert(sizeof(FileHeader) <= PAGE, "header fits one page");

// word-wise multiply-xor checksum (not cryptographic: catches truncation, bit rot, spliced files)
uint64_t checksum(const void* p, size_t bytes) {
  const unsigned char* b = static_cast<const unsigned char*>(p);
  uint64_t h = 0x9E3779B97F4A7C15ull ^ bytes;
  size_t i = 0;
  for (; i + 8 <= bytes; i += 8) {
    uint64_t w;
    memcpy(&w, b + i, 8);
    h = (h ^ w) * 0xFF51AFD7ED558CCDull;
    h ^= h >> 29;
  }
  for (; i < bytes; ++i) h = (h ^ b[i]) * 0x100000001B3ull;
  return h;
}
}  // namespace

void Snapshot::save(const std::string& path) const {
  struct Sec { const void* p; size_t bytes; };
  const Sec secs[N_SECTIONS] = {
      {keys.data(), keys.size() * 8}, {avg.data(), avg.size() * 8}, {terms.data(), terms.size() * sizeof(TermInfo)},
      {layers.data(), layers.size() * sizeof(LayerInfo)}, {fnodes.data(), fnodes.size() * sizeof(FrozenNode)},
      {fchar.data(), fchar.size() * 4}, {fchild.data(), fchild.size() * 4}, {doc.data(), doc.size() * 4},
      {tf.data(), tf.size() * 4}, {fl.data(), fl.size() * 4}, {table.data(), table.size() * 4},
      {max_fl.data(), max_fl.size() * 4}, {lut_cap.data(), lut_cap.size() * 4}, {lut_base.data(), lut_base.size() * 4}, {bits.data(), bits.size() * 4}, {alive.data(), alive.size() * 4}};
  FileHeader h;
  memset(&h, 0, sizeof(h));
  memcpy(h.magic, MAGIC, 8);
  const uint64_t sc[16] = {F, T, n_tiles, n_docs, P, n_postings, n_pointers, n_live_terms, max_layers, lut_rows,
                           n_ids, tiles_cap, P_used, table_used, any_dead ? 1u : 0u, 0};
  memcpy(h.scalars, sc, sizeof(sc));
  uint64_t off = PAGE;
  for (int i = 0; i < N_SECTIONS; ++i) {
    h.sec[i] = SectionRef{off, secs[i].bytes, checksum(secs[i].p, secs[i].bytes)};
    off = (off + secs[i].bytes + PAGE - 1) / PAGE * PAGE;
  }
  h.file_bytes = off;
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) throw std::invalid_argument("cannot open snapshot file: " + path);
  bool ok = true;
  std::vector<unsigned char> page(PAGE, 0);
  memcpy(page.data(), &h, sizeof(h));
  ok = ok && fwrite(page.data(), 1, PAGE, f) == PAGE;
  std::fill(page.begin(), page.end(), 0);
  for (int i = 0; i < N_SECTIONS && ok; ++i) {
    if (secs[i].bytes) ok = fwrite(secs[i].p, 1, secs[i].bytes, f) == secs[i].bytes;
    const size_t pad = (PAGE - secs[i].bytes % PAGE) % PAGE;
    if (ok && pad) ok = fwrite(page.data(), 1, pad, f) == pad;
  }
  ok = (fclose(f) == 0) && ok;
  if (!ok) throw std::runtime_error("snapshot write failed: " + path);
}

Snapshot::~Snapshot() {
  if (map_base_) munmap(map_base_, map_bytes_);
}

Snapshot::Snapshot(const std::string& path) {
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) throw std::invalid_argument("cannot open snapshot file: " + path);
  struct stat st;
  if (fstat(fd, &st) != 0 || (size_t)st.st_size < PAGE) {
    close(fd);
    throw std::invalid_argument("snapshot file truncated");
  }
  map_bytes_ = (size_t)st.st_size;
  void* m = mmap(nullptr, map_bytes_, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) throw std::invalid_argument("cannot map snapshot file: " + path);
  // (a constructor that throws runs no destructor: the mapping is released here until the file is accepted)
  struct MapGuard {
    void* p; size_t n;
    ~MapGuard() { if (p) munmap(p, n); }
  } guard{m, map_bytes_};
  const unsigned char* base = static_cast<const unsigned char*>(m);
  FileHeader h;
  memcpy(&h, base, sizeof(h));
  if (memcmp(h.magic, MAGIC, 8) != 0) throw std::invalid_argument("not a probly-search_amd snapshot (bad magic/version)");
  if (h.file_bytes != map_bytes_) throw std::invalid_argument("snapshot file truncated or padded");
  F = (uint32_t)h.scalars[0]; T = (uint32_t)h.scalars[1]; n_tiles = (uint32_t)h.scalars[2]; n_docs = h.scalars[3];
  P = h.scalars[4]; n_postings = h.scalars[5]; n_pointers = h.scalars[6]; n_live_terms = h.scalars[7];
  max_layers = (uint32_t)h.scalars[8]; lut_rows = (uint32_t)h.scalars[9];
  n_ids = h.scalars[10]; tiles_cap = (uint32_t)h.scalars[11]; P_used = h.scalars[12]; table_used = h.scalars[13];
  any_dead = h.scalars[14] != 0;
  const size_t elem[N_SECTIONS] = {8, 8, sizeof(TermInfo), sizeof(LayerInfo), sizeof(FrozenNode), 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};
  for (int i = 0; i < N_SECTIONS; ++i) {
    const SectionRef& s = h.sec[i];
    if (s.offset % PAGE || s.offset < PAGE || s.offset > map_bytes_ || s.bytes > map_bytes_ - s.offset || s.bytes % elem[i])
      throw std::invalid_argument("snapshot file corrupt: section out of range");
    if (checksum(base + s.offset, s.bytes) != s.checksum) throw std::invalid_argument("snapshot file corrupt: checksum mismatch");
  }
  auto view = [&](int i, auto* tag) {
    using E = std::remove_pointer_t<decltype(tag)>;
    return View<E>{reinterpret_cast<const E*>(base + h.sec[i].offset), (size_t)(h.sec[i].bytes / sizeof(E))};
  };
  keys = view(0, (uint64_t*)nullptr); avg = view(1, (double*)nullptr); terms = view(2, (TermInfo*)nullptr);
  layers = view(3, (LayerInfo*)nullptr); fnodes = view(4, (FrozenNode*)nullptr); fchar = view(5, (uint32_t*)nullptr);
  fchild = view(6, (uint32_t*)nullptr); doc = view(7, (uint32_t*)nullptr); tf = view(8, (uint32_t*)nullptr);
  fl = view(9, (uint32_t*)nullptr); table = view(10, (uint32_t*)nullptr); max_fl = view(11, (uint32_t*)nullptr);
  lut_cap = view(12, (uint32_t*)nullptr); lut_base = view(13, (uint32_t*)nullptr); bits = view(14, (uint32_t*)nullptr); alive = view(15, (uint32_t*)nullptr);
  validate();
  removed_df.assign(layers.size(), 0);
  src_epoch = ~0ull;
  map_base_ = m;
  guard.p = nullptr;
}

// Every index the planner or the kernels follow is checked against the array it points into, so a
// damaged file is refused here instead of turning into out-of-bounds reads on the host or the GPU.
void Snapshot::validate() const {
  auto bad = [](const char* what) { throw std::invalid_argument(std::string("snapshot file inconsistent: ") + what); };
  if (F > 8) bad("more than 8 fields");
  if (T < 256 || T > 4096 || (T & (T - 1))) bad("tile_docs");
  if (n_ids >= 0xFFFFFFF0ull || n_docs > n_ids) bad("n_docs / n_ids");
  const uint64_t want_tiles = std::max<uint64_t>(1, (n_ids + T - 1) / T);
  if (n_tiles != want_tiles || tiles_cap < n_tiles || (uint64_t)tiles_cap * T >= 0xFFFFFFF0ull + T) bad("n_tiles / tiles_cap");
  if (keys.size() != n_ids || avg.size() != F) bad("keys / avg size");
  if (P < 4 || P % 4 || doc.size() != P || tf.size() != (size_t)P * F || fl.size() != (size_t)P * F) bad("plane sizes");
  if (P_used > P || P_used % 4 || table_used > table.size()) bad("used sizes");
  if (table.empty() || bits.empty()) bad("empty table");
  if (alive.size() != ((size_t)tiles_cap * T + 31) / 32) bad("alive bitmap size");
  {
    uint64_t live = 0;
    for (size_t w = 0; w < alive.size(); ++w) live += (uint64_t)__builtin_popcount(alive[w]);
    if (live > n_ids || (!any_dead && live != n_ids)) bad("alive bitmap population");
    for (uint64_t i = n_ids; i < (uint64_t)alive.size() * 32; ++i)
      if ((alive[i >> 5] >> (i & 31u)) & 1u) bad("alive bit beyond the id space");
  }
  if (max_fl.size() != F || lut_cap.size() != F || lut_base.size() != F) bad("LUT vectors");
  uint64_t rows = 0;
  for (uint32_t x = 0; x < F; ++x) {
    if (lut_base[x] != rows) bad("lut_base");
    rows += lut_cap[x];
  }
  if (rows != lut_rows || lut_rows > 64) bad("lut_rows");
  for (size_t i = 1; i < keys.size(); ++i)
    if (keys[i - 1] >= keys[i]) bad("keys not strictly ascending");
  if (fnodes.empty() || fchar.size() != fchild.size()) bad("frozen trie");
  for (size_t n = 0; n < fnodes.size(); ++n) {
    const FrozenNode& fn = fnodes[n];
    if ((uint64_t)fn.child_begin + fn.child_count > fchar.size()) bad("fnode child range");
    if (fn.term_begin > fn.term_end || fn.term_end > terms.size()) bad("fnode term range");
    for (uint32_t c = 0; c < fn.child_count; ++c) {
      if (fchild[fn.child_begin + c] >= fnodes.size() || fchild[fn.child_begin + c] <= n) bad("fchild target");
      if (c && fchar[fn.child_begin + c - 1] >= fchar[fn.child_begin + c]) bad("fchar order");
    }
  }
  for (size_t o = 0; o < terms.size(); ++o) {
    const TermInfo& t = terms[o];
    if (t.fnode >= fnodes.size()) bad("term fnode");
    if (t.n_layers && ((uint64_t)t.first_layer + t.n_layers > layers.size())) bad("term layer range");
    uint32_t hops = 0;
    for (uint32_t l = t.delta_head; l != NO_LAYER; l = layers[l].next)
      if (l >= layers.size() || ++hops > layers.size()) bad("term delta chain");
  }
  for (size_t l = 0; l < layers.size(); ++l) {
    const LayerInfo& L = layers[l];
    if (L.post_off % 4 || L.post_off > P_used || L.len > P_used - L.post_off) bad("layer posting range");
    if (L.shift > 31) bad("layer shift");
    if (L.next != NO_LAYER && L.next >= layers.size()) bad("layer chain");
    const uint64_t slots = (((uint64_t)tiles_cap - 1) >> L.shift) + 1;
    if ((uint64_t)L.tbl_off + slots + 1 > table_used) bad("layer table range");
    uint32_t prev = 0;
    for (uint64_t sl = 0; sl <= slots; ++sl) {
      const uint32_t v = table[L.tbl_off + sl];
      if (v < prev || v > L.len) bad("table offsets");
      prev = v;
    }
    if (table[L.tbl_off + slots] != L.len) bad("table end");
    const uint64_t bm_words = 2 * (((uint64_t)tiles_cap * T + 31) / 32);
    if (L.bm_off != NO_BITMAP && ((L.bm_off & 1u) || (uint64_t)L.bm_off + bm_words > bits.size())) bad("layer bitmap range");
    uint64_t set = 0;
    for (uint32_t i = 0; i < L.len; ++i) {
      const uint32_t d = doc[L.post_off + i];
      if (d >= n_ids || (i && doc[L.post_off + i - 1] >= d)) bad("posting doc ids");
      if (L.bm_off != NO_BITMAP) {
        const uint32_t w = bits[(size_t)L.bm_off + 2 * (size_t)(d >> 5)], before = bits[(size_t)L.bm_off + 2 * (size_t)(d >> 5) + 1];
        if (!((w >> (d & 31u)) & 1u) || before + (uint32_t)__builtin_popcount(w & ((1u << (d & 31u)) - 1u)) != i)
          bad("bitmap does not lead to the posting");
      }
    }
    if (L.bm_off != NO_BITMAP) {
      for (uint64_t w = 0; w < bm_words / 2; ++w) set += (uint64_t)__builtin_popcount(bits[(size_t)L.bm_off + 2 * w]);
      if (set != L.len) bad("bitmap holds documents the list does not");
    }
  }
}

uint32_t Snapshot::chain_length(uint32_t head) const {
  uint32_t n = 0;
  for (uint32_t l = head; l != NO_LAYER; l = layers[l].next) ++n;
  return n;
}

// ---- delta snapshot --------------------------------------------------------------
bool Snapshot::apply_delta(const Index& idx, DeltaRanges& out) {
  out = DeltaRanges{};
  if (!own_ || src_epoch == ~0ull || src_uid != idx.uid()) return false;  // mapped from a file / another index: no source bookkeeping
  if (idx.epoch() == src_epoch) return true;
  if (idx.fields_len() != F) return false;
  size_t n_changes = 0;
  const IndexChange* log = idx.changes_since(src_epoch, &n_changes);
  if (!log) return false;
  SnapshotStorage& st = *own_;
  // ---- is the change set expressible? ----
  std::vector<const IndexChange*> adds;
  std::vector<uint32_t> removes;  // doc ids
  bool have_key = !st.keys.empty();
  uint64_t last_key = have_key ? st.keys.back() : 0;
  size_t new_postings = 0;
  for (size_t i = 0; i < n_changes; ++i) {
    const IndexChange& c = log[i];
    if (c.kind == IndexChange::OTHER) return false;
    if (c.kind == IndexChange::ADD) {
      // appended documents only: a key above every key the snapshot knows (doc ids stay key-ordered),
      // not a re-add, not a lazily removed key (that one stays invisible until vacuum)
      if (c.was_present || c.was_removed || (have_key && c.key <= last_key)) return false;
      adds.push_back(&c);
      last_key = c.key;
      have_key = true;
      new_postings += c.nodes.size();
    } else if (c.was_present) {
      auto it = std::lower_bound(st.keys.begin(), st.keys.end(), c.key);
      if (it == st.keys.end() || *it != c.key) return false;  // a document added earlier in this very delta
      const uint32_t rid = (uint32_t)(it - st.keys.begin());
      // (a duplicate record of this document was folded at flatten time: its df share has no posting)
      if (std::binary_search(st.folded_ids.begin(), st.folded_ids.end(), rid)) return false;
      removes.push_back(rid);
    }
  }
  if (n_ids + adds.size() > (uint64_t)tiles_cap * T) return false;
  // room in the planes (4-aligned starts) and the table (<= len/2 + 2 slots per new layer, bounded below)
  if (new_postings && P_used + 4 * new_postings > P) return false;  // (every touched term starts a 4-aligned run)
  // ---- removals: clear the alive bit; df is re-counted by the caller (set_removed_df) ----
  for (uint32_t id : removes) {
    uint32_t& w = st.alive[id >> 5];
    if (w & (1u << (id & 31u))) {
      w &= ~(1u << (id & 31u));
      out.alive_words.push_back(id >> 5);
      ++out.docs_removed;
      any_dead = true;
    }
  }
  // ---- additions ----
  out.key_begin = st.keys.size();
  out.plane_begin = P_used;
  out.table_begin = table_used;
  if (!adds.empty()) {
    // group the new postings by source trie node, in doc id order
    struct NewPosting { uint32_t id; const IndexChange* c; uint32_t slot; };
    std::unordered_map<int32_t, std::vector<NewPosting>> by_node;
    for (const IndexChange* c : adds) {
      const uint32_t id = (uint32_t)st.keys.size();
      st.keys.push_back(c->key);
      st.alive[id >> 5] |= 1u << (id & 31u);
      out.alive_words.push_back(id >> 5);
      for (uint32_t x = 0; x < F; ++x) st.max_fl[x] = std::max(st.max_fl[x], c->field_length[x]);
      for (size_t k = 0; k < c->nodes.size(); ++k) by_node[c->nodes[k]].push_back(NewPosting{id, c, (uint32_t)k});
    }
    // table room: one slot table per new layer
    uint64_t need_tbl = 0;
    for (auto& kv : by_node) {
      uint32_t shift = 0;
      const uint64_t want = std::max<uint64_t>(1, kv.second.size() / 2);
      while ((((uint64_t)tiles_cap - 1) >> shift) + 1 > want) ++shift;
      need_tbl += (((uint64_t)tiles_cap - 1) >> shift) + 2;
    }
    if (table_used + need_tbl > st.table.size() || st.layers.size() + by_node.size() >= 0xFFFFFFF0ull) {
      // undo the additions made so far (nothing else was touched)
      for (size_t i = 0; i < adds.size(); ++i) {
        const uint32_t id = (uint32_t)st.keys.size() - 1;
        st.alive[id >> 5] &= ~(1u << (id & 31u));
        st.keys.pop_back();
      }
      // (removals already applied stay: the caller re-flattens anyway)
      bind(st);
      return false;
    }
    // terms: does every touched node already have a term ordinal?
    // (100 k insertions: 3-5 ms if redone for every delta - it only changes when the trie is re-frozen)
    std::unordered_map<int32_t, uint32_t>& ord_of = st.ord_of;
    if (ord_of.size() != st.term_node.size()) {
      ord_of.clear();
      ord_of.reserve(st.term_node.size() * 2);
      for (size_t o = 0; o < st.term_node.size(); ++o) ord_of.emplace(st.term_node[o], (uint32_t)o);
    }
    bool new_terms = false;
    for (auto& kv : by_node)
      if (!ord_of.count(kv.first)) { new_terms = true; break; }
    if (new_terms) {
      // re-freeze the trie (no posting work) and carry every term's lists over by source node
      FrozenTrie ft;
      freeze_trie(idx, ft);
      std::vector<TermInfo> nt(ft.term_node.size());
      std::vector<uint64_t> ndf(ft.term_node.size(), 0);
      for (size_t o = 0; o < nt.size(); ++o) {
        auto it = ord_of.find(ft.term_node[o]);
        if (it != ord_of.end()) {
          nt[o] = st.terms[it->second];
          ndf[o] = st.df_total[it->second];
        } else {
          nt[o] = TermInfo{0, ft.term_bytes[o], 0, 0, 0, NO_LAYER, 0};
        }
        nt[o].byte_len = ft.term_bytes[o];
        nt[o].fnode = ft.term_fnode[o];
      }
      st.terms = std::move(nt);
      st.df_total = std::move(ndf);
      st.fnodes = std::move(ft.fnodes);
      st.fchar = std::move(ft.fchar);
      st.fchild = std::move(ft.fchild);
      st.term_node = std::move(ft.term_node);
      ord_of.clear();
      for (size_t o = 0; o < st.term_node.size(); ++o) ord_of.emplace(st.term_node[o], (uint32_t)o);
      out.trie_refrozen = true;
    }
    // one delta layer per touched term, appended to the planes / the table
    std::vector<int32_t> touched;
    touched.reserve(by_node.size());
    for (auto& kv : by_node) touched.push_back(kv.first);
    std::sort(touched.begin(), touched.end());  // deterministic layout
    for (int32_t node : touched) {
      const std::vector<NewPosting>& np = by_node[node];
      auto it = ord_of.find(node);
      if (it == ord_of.end()) throw std::logic_error("delta: a touched trie node has no term (index / log out of step)");
      TermInfo& ti = st.terms[it->second];
      LayerInfo L{P_used, (uint32_t)np.size(), (uint32_t)table_used, 0, NO_BITMAP, ti.delta_head, 0};
      const uint64_t want = std::max<uint64_t>(1, L.len / 2);
      while ((((uint64_t)tiles_cap - 1) >> L.shift) + 1 > want) ++L.shift;
      const uint32_t slots = ((tiles_cap - 1) >> L.shift) + 1;
      uint64_t added_df = 0;
      for (size_t i = 0; i < np.size(); ++i) {
        const uint64_t at = P_used + i;
        st.doc[at] = np[i].id;
        for (uint32_t x = 0; x < F; ++x) {
          const uint32_t t = np[i].c->tf[(size_t)np[i].slot * F + x];
          st.tf[(size_t)x * P + at] = t;
          st.fl[(size_t)x * P + at] = np[i].c->field_length[x];
          added_df += t;
        }
      }
      const uint64_t padded = (L.len + 3) & ~(uint64_t)3;
      for (uint64_t at = P_used + L.len; at < P_used + padded; ++at) {
        st.doc[at] = 0xFFFFFFFFu;
        for (uint32_t x = 0; x < F; ++x) { st.tf[(size_t)x * P + at] = 0; st.fl[(size_t)x * P + at] = 0; }
      }
      uint32_t pos = 0;
      for (uint32_t sl = 0; sl < slots; ++sl) {
        const uint64_t first_doc = ((uint64_t)sl << L.shift) * T;
        while (pos < L.len && st.doc[P_used + pos] < first_doc) ++pos;
        st.table[table_used + sl] = pos;
      }
      st.table[table_used + slots] = L.len;
      P_used += padded;
      table_used += slots + 1;
      ti.delta_head = (uint32_t)st.layers.size();
      st.layers.push_back(L);
      st.df_total[it->second] += added_df;
      ti.df_raw += added_df;
      n_postings += L.len;
      n_pointers += added_df;
      n_delta_postings += L.len;
      ++n_delta_layers;
    }
    removed_df.resize(st.layers.size(), 0);
    n_ids = st.keys.size();
    n_tiles = (uint32_t)std::max<uint64_t>(1, (n_ids + T - 1) / T);
    out.docs_added = adds.size();
  }
  out.key_end = st.keys.size();
  out.plane_end = P_used;
  out.table_end = table_used;
  std::sort(out.alive_words.begin(), out.alive_words.end());
  out.alive_words.erase(std::unique(out.alive_words.begin(), out.alive_words.end()), out.alive_words.end());
  // scalars the scorers read (index.rs:112-114, 176-186; bm25.rs:41): current as of this epoch
  n_docs = idx.docs_len();
  for (uint32_t x = 0; x < F; ++x) st.avg[x] = idx.field(x).avg;
  n_live_terms = 0;
  for (const TermInfo& t : st.terms)
    if (t.n_layers || t.delta_head != NO_LAYER) ++n_live_terms;
  src_epoch = idx.epoch();
  bind(st);
  return true;
}

// Per layer, the pointer count (sum of tf) of the postings whose document was removed by a delta:
// Index::count_documents skips removed documents (index.rs:287-293), so df_raw shrinks by it.
void Snapshot::set_removed_df(const std::vector<uint64_t>& removed_per_layer) {
  if (!own_) return;
  SnapshotStorage& st = *own_;
  removed_df = removed_per_layer;
  removed_df.resize(st.layers.size(), 0);
  for (size_t o = 0; o < st.terms.size(); ++o) {
    TermInfo& t = st.terms[o];
    uint64_t gone = 0;
    for (uint32_t l = 0; l < t.n_layers; ++l) gone += removed_df[t.first_layer + l];
    for (uint32_t l = t.delta_head; l != NO_LAYER; l = st.layers[l].next) gone += removed_df[l];
    t.df_raw = st.df_total[o] > gone ? st.df_total[o] - gone : 0;
  }
  bind(st);
}

// Host fallback of the same count (host-only snapshots; the engine does it on the device).
std::vector<uint64_t> Snapshot::count_removed_df_host() const {
  std::vector<uint64_t> r(layers.size(), 0);
  if (!any_dead) return r;
  for (size_t l = 0; l < layers.size(); ++l) {
    const LayerInfo& L = layers[l];
    for (uint32_t i = 0; i < L.len; ++i) {
      const uint32_t d = doc[L.post_off + i];
      if ((alive[d >> 5] >> (d & 31u)) & 1u) continue;
      for (uint32_t x = 0; x < F; ++x) r[l] += tf[(size_t)x * P + L.post_off + i];
    }
  }
  return r;
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
        if (t.df_raw == 0 || (t.n_layers == 0 && t.delta_head == NO_LAYER)) continue;  // query.rs:47-48
        ps_plan_entry e;
        memset(&e, 0, sizeof(e));
        e.qterm = qord;
        e.qterm_index = (uint32_t)qi;
        e.bm_off = NO_BITMAP;
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
        // base layers (newest version first), then the delta layers (documents added after the flatten:
        // disjoint from the base documents, so "another entry of the same query term" is exact)
        const uint32_t n_chain = t.n_layers + chain_length(t.delta_head);
        for (uint32_t l = 0, dl = t.delta_head; l < n_chain; ++l) {
          const uint32_t li = l < t.n_layers ? t.first_layer + l : dl;
          const LayerInfo& L = layers[li];
          if (l >= t.n_layers) dl = L.next;
          e.post_off = L.post_off;
          e.len = L.len;
          e.tbl_off = L.tbl_off;
          e.shift = L.shift | (l << 8);  // bits 8.. = version layer (0 = newest)
          if (sc.kind == PS_SCORER_BM25) e.node = li;  // ordinal of the list (the engine's per-list bounds)
          e.layer = li;
          e.bm_off = L.bm_off;           // the list's membership bitmap (K1d lookups)
          plan.entries.push_back(e);
          plan.postings += L.len;
        }
      }
    }
    if (plan.entries.size() - before > 1) plan.multi_expansion = true;
    ++qord;
  }
  plan.qbeg.push_back((uint32_t)plan.entries.size());
  plan.qterms_len.push_back((uint32_t)tokens.size());
  plan.n_nodes.push_back((uint32_t)seen_nodes.size());
  plan.max_entries = std::max<uint32_t>(plan.max_entries, (uint32_t)(plan.entries.size() - e_begin));
  plan.max_qterms = std::max(plan.max_qterms, qord);
  plan.max_nodes = std::max<uint32_t>(plan.max_nodes, (uint32_t)seen_nodes.size());
}

}  // namespace ps
