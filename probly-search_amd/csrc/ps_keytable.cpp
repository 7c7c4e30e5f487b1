// ps_keytable.cpp — the side table a binding needs for document keys that are not u64.
//
// The reference's index is generic over its key, `Index<T: Eq + Hash + Copy + Debug>` (src/index.rs:19-33;
// `QueryResult<T>`, src/query.rs:10-17); the ABI carries `uint64_t` keys.  A binding for another `T` (a uuid, a
// (shard, row) pair, a short string) hands the key's bytes to this table and gets a dense id back - ids count up from
// 0 in first-seen order and are never reused, so a removed and re-added key meets the index under the id it had
// (the reference's `removed_documents` test is by key, src/index.rs:161-191) - and turns result ids back into key
// bytes after a query.  Host code only; nothing here touches the device.
//
// Layout: key bytes back to back in one arena + an offsets column (id -> [begin, end)), and an open-addressing table
// of {hash tag, id + 1} cells probed linearly; the table doubles at 5/8 load.  Readers (find / key / resolve) need no
// lock while nobody interns, the same rule the index itself has (`&self` / `&mut self`).
#include <cstdint>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/probly_search_amd.h"
#include "ps_capi_internal.hpp"

namespace {
template <typename Fn>
ps_status guarded(Fn&& fn) {
  try {
    return fn();
  } catch (const std::bad_alloc&) {
    return ps::set_error(PS_ENOMEM, "out of memory");
  } catch (const std::length_error& e) {
    return ps::set_error(PS_EUNSUPPORTED, e.what());
  } catch (const std::exception& e) {
    return ps::set_error(PS_EINVAL, e.what());
  }
}
}  // namespace

struct ps_keytable {
  std::vector<char> arena;
  std::vector<uint64_t> off{0};  // n + 1 entries
  struct Cell {
    uint32_t tag;  // high hash bits; compared before the bytes
    uint32_t id1;  // id + 1, 0 = empty
  };
  std::vector<Cell> cells;  // power-of-two size
  size_t n = 0;

  static uint64_t hash(const unsigned char* p, size_t len) {
    // 8 bytes at a time, multiply-xorshift mix per word (keys are short: the point is few instructions per key)
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t(len) * 0xD6E8FEB86659FD93ull);
    while (len >= 8) {
      uint64_t w;
      std::memcpy(&w, p, 8);
      h = (h ^ w) * 0xD6E8FEB86659FD93ull;
      h ^= h >> 32;
      p += 8;
      len -= 8;
    }
    if (len) {
      uint64_t w = 0;
      std::memcpy(&w, p, len);
      h = (h ^ w) * 0xD6E8FEB86659FD93ull;
      h ^= h >> 32;
    }
    h *= 0xFF51AFD7ED558CCDull;
    return h ^ (h >> 29);
  }

  bool same(uint32_t id, const unsigned char* p, size_t len) const {
    const uint64_t b = off[id], e = off[id + 1];
    return e - b == len && (len == 0 || std::memcmp(arena.data() + b, p, len) == 0);
  }

  void grow() {
    const size_t cap = cells.empty() ? 1024 : cells.size() * 2;
    std::vector<Cell> nc(cap, Cell{0, 0});
    for (size_t id = 0; id < n; ++id) {
      const uint64_t h = hash(reinterpret_cast<const unsigned char*>(arena.data()) + off[id], off[id + 1] - off[id]);
      size_t s = size_t(h) & (cap - 1);
      while (nc[s].id1) s = (s + 1) & (cap - 1);
      nc[s] = Cell{uint32_t(h >> 32), uint32_t(id + 1)};
    }
    cells.swap(nc);
  }

  // returns id, or UINT64_MAX when absent
  uint64_t find(const unsigned char* p, size_t len) const {
    if (cells.empty()) return UINT64_MAX;
    const uint64_t h = hash(p, len);
    const uint32_t tag = uint32_t(h >> 32);
    const size_t mask = cells.size() - 1;
    for (size_t s = size_t(h) & mask;; s = (s + 1) & mask) {
      const Cell c = cells[s];
      if (!c.id1) return UINT64_MAX;
      if (c.tag == tag && same(c.id1 - 1, p, len)) return c.id1 - 1;
    }
  }

  uint64_t intern(const unsigned char* p, size_t len, bool* inserted) {
    if ((n + 1) * 8 > cells.size() * 5) grow();
    const uint64_t h = hash(p, len);
    const uint32_t tag = uint32_t(h >> 32);
    const size_t mask = cells.size() - 1;
    size_t s = size_t(h) & mask;
    for (;; s = (s + 1) & mask) {
      const Cell c = cells[s];
      if (!c.id1) break;
      if (c.tag == tag && same(c.id1 - 1, p, len)) {
        if (inserted) *inserted = false;
        return c.id1 - 1;
      }
    }
    if (n >= 0xFFFFFFF0u) throw std::length_error("key table holds at most 2^32-16 keys (the engine's document limit)");
    // Room for the new end offset first: if that allocation throws, the arena has not grown (a stray tail would be
    // glued to the next key).  A key that points into the arena itself (a ps_str handed out by ps_keytable_key /
    // ps_keytable_resolve) is copied before the arena may reallocate under it.
    off.reserve(off.size() + 1);
    const char* src = reinterpret_cast<const char*>(p);
    std::string own;
    if (!arena.empty() && src >= arena.data() && src < arena.data() + arena.size()) {
      own.assign(src, len);
      src = own.data();
    }
    arena.insert(arena.end(), src, src + len);
    off.push_back(arena.size());
    cells[s] = Cell{tag, uint32_t(n + 1)};
    if (inserted) *inserted = true;
    return n++;
  }
};

extern "C" {

ps_status ps_keytable_new(ps_keytable** out) {
  return guarded([&]() -> ps_status {
    if (!out) return ps::set_error(PS_EINVAL, "null out pointer");
    *out = new ps_keytable();
    return PS_OK;
  });
}

void ps_keytable_free(ps_keytable* kt) { delete kt; }

size_t ps_keytable_len(const ps_keytable* kt) { return kt ? kt->n : 0; }

ps_status ps_keytable_intern(ps_keytable* kt, const void* key, size_t len, uint64_t* id, int* inserted) {
  return guarded([&]() -> ps_status {
    if (!kt || !id || (len && !key)) return ps::set_error(PS_EINVAL, "null key table, key or id pointer");
    bool ins = false;
    *id = kt->intern(static_cast<const unsigned char*>(key), len, &ins);
    if (inserted) *inserted = ins ? 1 : 0;
    return PS_OK;
  });
}

ps_status ps_keytable_intern_flat(ps_keytable* kt, size_t n_keys, const void* bytes, const uint64_t* offsets,
                                  uint64_t* ids) {
  return guarded([&]() -> ps_status {
    if (!kt || (n_keys && (!offsets || !ids))) return ps::set_error(PS_EINVAL, "null key table, offsets or ids pointer");
    for (size_t i = 0; i < n_keys; ++i)
      if (offsets[i + 1] < offsets[i]) return ps::set_error(PS_EINVAL, "key offsets must be non-decreasing");
    if (n_keys && offsets[n_keys] > offsets[0] && !bytes) return ps::set_error(PS_EINVAL, "null key bytes");
    const unsigned char* b = static_cast<const unsigned char*>(bytes);
    for (size_t i = 0; i < n_keys; ++i) ids[i] = kt->intern(b + offsets[i], size_t(offsets[i + 1] - offsets[i]), nullptr);
    return PS_OK;
  });
}

int ps_keytable_find(const ps_keytable* kt, const void* key, size_t len, uint64_t* id) {
  if (!kt || (len && !key)) return 0;
  const uint64_t r = kt->find(static_cast<const unsigned char*>(key), len);
  if (r == UINT64_MAX) return 0;
  if (id) *id = r;
  return 1;
}

ps_status ps_keytable_key(const ps_keytable* kt, uint64_t id, ps_str* out) {
  if (!kt || !out) return ps::set_error(PS_EINVAL, "null key table or out pointer");
  if (id >= kt->n) return ps::set_error(PS_EINVAL, "id was not handed out by this key table");
  out->ptr = kt->arena.data() + kt->off[id];
  out->len = size_t(kt->off[id + 1] - kt->off[id]);
  return PS_OK;
}

ps_status ps_keytable_resolve(const ps_keytable* kt, const ps_result* results, size_t n, ps_str* keys) {
  if (!kt || (n && (!results || !keys))) return ps::set_error(PS_EINVAL, "null key table, results or keys pointer");
  for (size_t i = 0; i < n; ++i) {
    const uint64_t id = results[i].key;
    if (id >= kt->n) return ps::set_error(PS_EINVAL, "a result key was not handed out by this key table");
    keys[i].ptr = kt->arena.data() + kt->off[id];
    keys[i].len = size_t(kt->off[id + 1] - kt->off[id]);
  }
  return PS_OK;
}

}  // extern "C"
