"""Document keys that are not u64: `Index<T>` of the reference (src/index.rs:19-33) over the u64 ABI.

`KeyTable` wraps `ps_keytable_*` (csrc/ps_keytable.cpp): key bytes -> dense id, id -> key bytes.  `KeyedIndex` is the
binding a maintainer would write for a `T` that is not an integer: same methods as `Index`, keys are `bytes` / `str`
(or anything, through `encode` / `decode`), every call goes through the u64 entry points underneath and results come
back as `QueryResult(key: T, score)` (src/query.rs:10-17).
"""
import ctypes as C

from . import _lib
from .index import Index, QueryResult


class _RawStr(C.Structure):  # ps_str with the pointer kept as an address: key bytes may hold NULs
    _fields_ = [("ptr", C.c_void_p), ("len", C.c_size_t)]


class KeyTable:
    def __init__(self):
        self._L = _lib.load()
        h = C.c_void_p()
        _lib.check(self._L.ps_keytable_new(C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if self._h:
                self._L.ps_keytable_free(self._h)
                self._h = None
        except Exception:
            pass

    def __len__(self):
        return self._L.ps_keytable_len(self._h)

    def intern(self, key):
        """-> (id, inserted)"""
        out, ins = C.c_uint64(), C.c_int()
        _lib.check(self._L.ps_keytable_intern(self._h, key, len(key), C.byref(out), C.byref(ins)))
        return out.value, bool(ins.value)

    def intern_flat(self, data, offsets):
        """keys as one uint8 buffer + u64 offsets[n + 1] (numpy) -> ids u64[n]"""
        import numpy as np
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        data = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else data)
        ids = np.empty(len(offsets) - 1, dtype=np.uint64)
        _lib.check(self._L.ps_keytable_intern_flat(self._h, len(ids), data.ctypes.data, offsets.ctypes.data,
                                                   ids.ctypes.data))
        return ids

    def find(self, key):
        out = C.c_uint64()
        return out.value if self._L.ps_keytable_find(self._h, key, len(key), C.byref(out)) else None

    def key(self, id_):
        s = _RawStr()
        _lib.check(self._L.ps_keytable_key(self._h, id_, C.byref(s)))
        return C.string_at(s.ptr, s.len) if s.len else b""

    def resolve(self, results):
        """ctypes array of ps_result (or a list of QueryResult with id keys) -> list of key bytes"""
        if isinstance(results, list):
            arr = (_lib.Result * max(1, len(results)))()
            for i, r in enumerate(results):
                arr[i].key, arr[i].score = r.key, r.score
            n = len(results)
        else:
            arr, n = results, len(results)
        keys = (_RawStr * max(1, n))()
        _lib.check(self._L.ps_keytable_resolve(self._h, arr, n, keys))
        return [C.string_at(keys[i].ptr, keys[i].len) if keys[i].len else b"" for i in range(n)]


def _enc(k):
    return k.encode("utf-8") if isinstance(k, str) else bytes(k)


class KeyedIndex:
    """Index<T> for a T that is not an integer.  `encode(key) -> bytes` must agree with T's equality; `decode` turns
    the bytes back (default: str keys as UTF-8 in, `bytes` out unless decode is given)."""

    def __init__(self, fields_num, encode=_enc, decode=None, **kw):
        self.index = Index(fields_num, **kw)
        self.keys = KeyTable()
        self._enc, self._dec = encode, decode or (lambda b: b)

    def add_document(self, field_accessors, tokenizer, key, doc):
        id_, _ = self.keys.intern(self._enc(key))
        self.index.add_document(field_accessors, tokenizer, id_, doc)

    def add_field_values(self, key, values, tokenizer=None):
        id_, _ = self.keys.intern(self._enc(key))
        self.index.add_field_values(id_, values, tokenizer)

    def remove_document(self, key):
        id_ = self.keys.find(self._enc(key))
        if id_ is not None:  # a key never added: no-op in the reference too (src/index.rs:161-164)
            self.index.remove_document(id_)

    def vacuum(self):
        self.index.vacuum()

    def _back(self, res):
        ks = self.keys.resolve(res)
        return [QueryResult(self._dec(k), r.score) for k, r in zip(ks, res)]

    def query(self, query, score_calculator, tokenizer, fields_boost, top_k=0):
        return self._back(self.index.query(query, score_calculator, tokenizer, fields_boost, top_k))

    def snapshot(self, **kw):
        return KeyedSnapshot(self.index.snapshot(**kw), self)


class KeyedSnapshot:
    def __init__(self, snap, owner):
        self.snapshot, self._o = snap, owner

    def query(self, query, score_calculator, tokenizer, fields_boost, top_k=0):
        return self._o._back(self.snapshot.query(query, score_calculator, tokenizer, fields_boost, top_k))

    def query_batch(self, queries, score_calculator, tokenizer, fields_boost, top_k=0):
        return [self._o._back(r) for r in
                self.snapshot.query_batch(queries, score_calculator, tokenizer, fields_boost, top_k)]
