"""ctypes driver for the CPU oracle (oracle/probly_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg as the *checker*.  The product package
(probly-search_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libprobly_oracle.so")

BM25 = 1
ZERO_TO_ONE = 2
NAN_PROBE = 3  # a test plugin (probly_oracle.cpp, NanProbe), not a reference scorer


class _Str(C.Structure):
    _fields_ = [("ptr", C.c_char_p), ("len", C.c_size_t)]


class _Res(C.Structure):
    _fields_ = [("key", C.c_uint64), ("score", C.c_double)]


TOKENIZER_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p),
                           C.POINTER(C.c_size_t), C.c_size_t, C.c_void_p)


def build(force=False):
    """Compile the oracle with g++ (building the checker is not using it)."""
    src = os.path.join(_HERE, "probly_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_index_new.restype = C.c_void_p
        L.orc_index_new.argtypes = [C.c_size_t]
        L.orc_index_free.argtypes = [C.c_void_p]
        L.orc_index_add_document.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(_Str), C.POINTER(C.c_size_t),
                                             C.c_void_p, C.c_void_p]
        L.orc_index_add_documents_flat.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_index_remove_document.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_index_vacuum.argtypes = [C.c_void_p]
        L.orc_index_docs_len.restype = C.c_size_t
        L.orc_index_docs_len.argtypes = [C.c_void_p]
        L.orc_index_fields_len.restype = C.c_size_t
        L.orc_index_fields_len.argtypes = [C.c_void_p]
        L.orc_index_field_details.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.orc_index_doc_field_length.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.orc_index_count_nodes.restype = C.c_size_t
        L.orc_index_count_nodes.argtypes = [C.c_void_p]
        L.orc_index_arena_doc_live.restype = C.c_size_t
        L.orc_index_arena_doc_live.argtypes = [C.c_void_p]
        L.orc_index_children.restype = C.c_long
        L.orc_index_children.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t]
        L.orc_index_postings.restype = C.c_long
        L.orc_index_postings.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64), C.c_size_t]
        L.orc_index_count_documents.restype = C.c_long
        L.orc_index_count_documents.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.orc_index_expand_term.restype = C.c_size_t
        L.orc_index_expand_term.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                            C.POINTER(C.c_size_t)]
        L.orc_index_query.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_char_p, C.c_size_t,
                                      C.POINTER(C.c_double), C.c_size_t, C.c_void_p, C.c_void_p, C.c_int,
                                      C.POINTER(C.POINTER(_Res)), C.POINTER(C.c_size_t)]
        L.orc_results_free.argtypes = [C.POINTER(_Res)]
        L.orc_index_query_flat.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_char_p, C.c_size_t,
                                           C.POINTER(C.c_double), C.c_size_t, C.c_int,
                                           C.POINTER(C.POINTER(_Res)), C.POINTER(C.c_size_t)]
        L.orc_bench_queries.restype = C.c_double
        L.orc_bench_queries.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.POINTER(_Str), C.c_size_t,
                                        C.POINTER(C.c_double), C.c_uint, C.POINTER(C.c_double),
                                        C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(_Res), C.c_int,
                                        C.POINTER(C.c_double)]
        _lib = L
    return _lib


class Scorer:
    def __init__(self, kind, k1=1.2, b=0.75):
        self.kind, self.k1, self.b = kind, k1, b


def bm25(k1=1.2, b=0.75):
    """score::bm25::new() (src/score/default/bm25.rs:21-26) with public k1/b fields."""
    return Scorer(BM25, k1, b)


def zero_to_one():
    """score::zero_to_one::new() (src/score/default/zero_to_one.rs:35-39)."""
    return Scorer(ZERO_TO_ONE)


def nan_probe():
    """The oracle's test plugin: a ScoreCalculator returning NaN for some (document, expansion) pairs."""
    return Scorer(NAN_PROBE)


def _wrap_tokenizer(py_tok):
    """py_tok: str -> list[str].  Keeps the produced byte strings alive for the call."""
    if py_tok is None:
        return None, None
    keep = []

    def cb(ptr, n, out_ptr, out_len, cap, _user):
        s = C.string_at(ptr, n).decode("utf-8")
        toks = [t.encode("utf-8") for t in py_tok(s)]
        for i, t in enumerate(toks[:cap]):
            buf = C.create_string_buffer(t, len(t) + 1)
            keep.append(buf)
            out_ptr[i] = C.cast(buf, C.c_void_p).value
            out_len[i] = len(t)
        return len(toks)

    fn = TOKENIZER_FN(cb)
    return fn, keep


class Index:
    """Index<u64> of the reference (src/index.rs:19-33) — oracle edition."""

    def __init__(self, fields_num):
        self._L = lib()
        self._h = self._L.orc_index_new(fields_num)
        self.fields_num = fields_num

    def __del__(self):
        try:
            if self._h:
                self._L.orc_index_free(self._h)
                self._h = None
        except Exception:
            pass

    def add_document(self, key, fields, tokenizer=None):
        """fields[i] is a str (single-valued accessor) or a list[str] (multi-valued accessor)."""
        vals, counts = [], []
        for f in fields:
            vs = [f] if isinstance(f, str) else list(f)
            counts.append(len(vs))
            vals.extend(v.encode("utf-8") for v in vs)
        arr = (_Str * max(1, len(vals)))()
        for i, v in enumerate(vals):
            arr[i].ptr, arr[i].len = v, len(v)
        cnt = (C.c_size_t * len(counts))(*counts)
        fn, keep = _wrap_tokenizer(tokenizer)
        self._L.orc_index_add_document(self._h, key, arr, cnt, C.cast(fn, C.c_void_p) if fn else None, None)

    def add_documents_flat(self, keys, text, offsets):
        """numpy bulk add: keys u64[n], text bytes/uint8 array, offsets u64[n*F+1]."""
        import numpy as np
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        text = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else text)
        self._L.orc_index_add_documents_flat(self._h, len(keys), keys.ctypes.data, text.ctypes.data,
                                             offsets.ctypes.data)

    def remove_document(self, key):
        self._L.orc_index_remove_document(self._h, key)

    def vacuum(self):
        self._L.orc_index_vacuum(self._h)

    def docs_len(self):
        return self._L.orc_index_docs_len(self._h)

    def field_details(self, i):
        s, a = C.c_uint64(), C.c_double()
        self._L.orc_index_field_details(self._h, i, C.byref(s), C.byref(a))
        return s.value, a.value

    def doc_field_length(self, key):
        out = (C.c_uint64 * self.fields_num)()
        if not self._L.orc_index_doc_field_length(self._h, key, out):
            return None
        return list(out)

    def count_nodes(self):
        return self._L.orc_index_count_nodes(self._h)

    def arena_doc_live(self):
        return self._L.orc_index_arena_doc_live(self._h)

    def children(self, term=""):
        t = term.encode("utf-8")
        buf = (C.c_uint32 * 4096)()
        n = self._L.orc_index_children(self._h, t, len(t), buf, 4096)
        return None if n < 0 else [chr(buf[i]) for i in range(n)]

    def postings(self, term):
        t = term.encode("utf-8")
        n = self._L.orc_index_postings(self._h, t, len(t), None, None, 0)
        if n < 0:
            return None
        keys = (C.c_uint64 * max(1, n))()
        tf = (C.c_uint64 * max(1, n * self.fields_num))()
        self._L.orc_index_postings(self._h, t, len(t), keys, tf, n)
        F = self.fields_num
        return [(keys[i], [tf[i * F + f] for f in range(F)]) for i in range(n)]

    def count_documents(self, term):
        t = term.encode("utf-8")
        return self._L.orc_index_count_documents(self._h, t, len(t))

    def expand_term(self, term):
        t = term.encode("utf-8")
        need = C.c_size_t()
        n = self._L.orc_index_expand_term(self._h, t, len(t), None, 0, C.byref(need))
        buf = C.create_string_buffer(max(1, need.value))
        self._L.orc_index_expand_term(self._h, t, len(t), buf, need.value, C.byref(need))
        return [s.decode("utf-8") for s in buf.raw[:need.value].split(b"\0")[:n]]

    def query(self, q, scorer, fields_boost, tokenizer=None, canonical=True):
        """Index::query (src/query.rs:21-106) -> list[(key, score)]."""
        qb = q.encode("utf-8")
        boosts = (C.c_double * len(fields_boost))(*fields_boost)
        out, n = C.POINTER(_Res)(), C.c_size_t()
        fn, keep = _wrap_tokenizer(tokenizer)
        rc = self._L.orc_index_query(self._h, scorer.kind, scorer.k1, scorer.b, qb, len(qb), boosts,
                                     len(fields_boost), C.cast(fn, C.c_void_p) if fn else None, None,
                                     1 if canonical else 0, C.byref(out), C.byref(n))
        if rc == 1:
            raise IndexError("fields_boost shorter than fields_num (reference: index out of bounds panic)")
        if rc == 2:
            raise ValueError("NaN score (reference: partial_cmp().unwrap() panic)")
        res = [(out[i].key, out[i].score) for i in range(n.value)]
        self._L.orc_results_free(out)
        return res

    def query_flat(self, q, scorer, fields_boost, canonical=True):
        """The same query through the second CPU-baseline leg (SwissTable-class containers): must equal query()."""
        qb = q.encode("utf-8")
        boosts = (C.c_double * len(fields_boost))(*fields_boost)
        out, n = C.POINTER(_Res)(), C.c_size_t()
        rc = self._L.orc_index_query_flat(self._h, scorer.kind, scorer.k1, scorer.b, qb, len(qb), boosts,
                                          len(fields_boost), 1 if canonical else 0, C.byref(out), C.byref(n))
        if rc == 1:
            raise IndexError("fields_boost shorter than fields_num (reference: index out of bounds panic)")
        if rc == 2:
            raise ValueError("NaN score (reference: partial_cmp().unwrap() panic)")
        res = [(out[i].key, out[i].score) for i in range(n.value)]
        self._L.orc_results_free(out)
        return res

    def bench_queries(self, queries, scorer, fields_boost, threads=1, top_k=0, flat=False):
        """Time queries inside C++ (cpu_baseline leg).  Returns (wall_s, per_query_s, n_results, topk); flat=True runs the
        SwissTable-class leg (self.flat_build_s = seconds spent building its flat view, outside the clock)."""
        import numpy as np
        qb = [q.encode("utf-8") for q in queries]
        arr = (_Str * max(1, len(qb)))()
        for i, v in enumerate(qb):
            arr[i].ptr, arr[i].len = v, len(v)
        boosts = (C.c_double * len(fields_boost))(*fields_boost)
        secs = np.zeros(len(qb), dtype=np.float64)
        nres = np.zeros(len(qb), dtype=np.uint64)
        topk = (_Res * max(1, len(qb) * top_k))() if top_k else None
        fb = C.c_double(0.0)
        wall = self._L.orc_bench_queries(self._h, scorer.kind, scorer.k1, scorer.b, arr, len(qb), boosts, threads,
                                         secs.ctypes.data_as(C.POINTER(C.c_double)),
                                         nres.ctypes.data_as(C.POINTER(C.c_uint64)), top_k, topk, 1 if flat else 0,
                                         C.byref(fb))
        self.flat_build_s = fb.value
        tk = None
        if top_k:
            tk = [[(topk[i * top_k + k].key, topk[i * top_k + k].score) for k in range(top_k)
                   if topk[i * top_k + k].key != 2 ** 64 - 1] for i in range(len(qb))]
        return wall, secs, nres, tk
