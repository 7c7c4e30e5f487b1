"""Host-side mirror of `probly_search::Index<T>` (src/index.rs, src/query.rs of probly-search
2.0.1) over the C ABI.  Same method names, argument order and meaning as the reference:

    index = Index(2)                                                      # Index::<usize>::new(2)
    index.add_document([title_extract, description_extract], tokenizer, doc.id, doc)
    result = index.query("abc", bm25.new(), tokenizer, [1., 1.])          # -> [QueryResult]
    index.remove_document(doc.id); index.vacuum()

Keys are u64 at the ABI (the reference's tests use usize).  Where the reference panics
(fields_boost shorter than fields_num, src/score/default/bm25.rs:85) an IndexError is raised.
"""
import ctypes as C
import os

from . import _lib
from ._lib import PsError


def whitespace_tokenizer(s):
    """test_util::tokenizer — `s.split(' ')` (src/lib.rs:42-44).  Passing this function (or
    None) selects the identical built-in C++ tokenizer instead of a Python callback."""
    return s.split(" ")


class QueryResult:
    """QueryResult<T> {key, score} (src/query.rs:10-15); == is exact f64 equality like the
    derived PartialEq the reference's tests rely on."""
    __slots__ = ("key", "score")

    def __init__(self, key, score):
        self.key = key
        self.score = score

    def __eq__(self, other):
        return isinstance(other, QueryResult) and self.key == other.key and self.score == other.score

    def __iter__(self):
        return iter((self.key, self.score))

    def __repr__(self):
        return "QueryResult { key: %r, score: %r }" % (self.key, self.score)


class FieldDetails:
    """FieldDetails {sum, avg} (src/index.rs:391-396)."""
    __slots__ = ("sum", "avg")

    def __init__(self, sum, avg):
        self.sum = sum
        self.avg = avg

    def __eq__(self, other):
        return isinstance(other, FieldDetails) and self.sum == other.sum and self.avg == other.avg

    def __repr__(self):
        return "FieldDetails { sum: %r, avg: %r }" % (self.sum, self.avg)


def _scorer_desc(score_calculator):
    kind = getattr(score_calculator, "kind", None)
    if kind not in (1, 2):
        raise TypeError("score_calculator must be score.bm25.new() or score.zero_to_one.new() here; a custom "
                        "ScoreCalculator runs through Index.query (host callbacks), not on a snapshot")
    return _lib.ScorerDesc(kind, 0, float(score_calculator.bm25k1), float(score_calculator.bm25b), None)


class _Documents:
    """What before_each sees as `documents`: len() == documents.len()."""

    def __init__(self, n):
        self._n = n

    def __len__(self):
        return self._n


def _callbacks_desc(calc, fields_num):
    """ps_scorer_desc {PS_SCORER_HOST_CALLBACKS} around a Python ScoreCalculator.  Returns the
    descriptor plus the objects that must stay alive during the call; exceptions raised inside a
    callback are re-raised by the caller after the C call returns."""
    from . import score as sc
    memories, errors = {}, []

    def term_data(td):
        t = td.contents
        return sc.TermData(t.query_term_index, C.string_at(t.query_term.ptr, t.query_term.len).decode("utf-8"),
                           C.string_at(t.query_term_expanded.ptr, t.query_term_expanded.len).decode("utf-8"),
                           t.query_terms_len)

    def before_each(_user, td, df, n_docs, _idx, mem_out):
        try:
            m = calc.before_each(term_data(td), df, _Documents(n_docs))
        except Exception as e:  # noqa: BLE001 - must not unwind through C
            errors.append(e)
            return 0
        if m is None:
            return 0
        tok = len(memories) + 1
        memories[tok] = m
        mem_out[0] = tok
        return 1

    def score(_user, mem, dp, dd, node, fd, td, out):
        try:
            f = fd.contents
            field_data = sc.FieldData([f.fields_boost[i] for i in range(f.n_boost)],
                                      [FieldDetails(f.fields[i].sum, f.fields[i].avg) for i in range(f.n_fields)])
            p, d = dp.contents, dd.contents
            s = calc.score(memories.get(mem) if mem else None,
                           sc.DocumentPointer(p.details_key, [p.term_frequency[i] for i in range(fields_num)]),
                           sc.DocumentDetails(d.key, [d.field_length[i] for i in range(fields_num)]),
                           node, field_data, term_data(td))
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            return 0
        if s is None:
            return 0
        out[0] = float(s)
        return 1

    def finalize(_user, res, n):
        try:
            lst = [QueryResult(res[i].key, res[i].score) for i in range(n)]
            calc.finalize(lst)
            for i, r in enumerate(lst[:n]):
                res[i].key, res[i].score = r.key, r.score
            return min(len(lst), n)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            return n

    def drop(_user, mem):
        memories.pop(mem, None)

    cbs = _lib.ScoreCallbacks(_lib.BEFORE_EACH_FN(before_each), _lib.SCORE_FN(score), _lib.FINALIZE_FN(finalize),
                              _lib.DROP_FN(drop), None)
    desc = _lib.ScorerDesc(3, 0, 0.0, 0.0, C.pointer(cbs))
    return desc, (cbs, memories, errors)


class _Tok:
    """Wraps a Python tokenizer (str -> list[str]) as a ps_tokenizer_fn; None / the whitespace
    tokenizer map to NULL (built-in split(' '))."""

    def __init__(self, tokenizer):
        self.fn = None
        self._keep = []
        if tokenizer is None or tokenizer is whitespace_tokenizer:
            return
        keep = self._keep

        def cb(ptr, n, out_ptr, out_len, cap, _user):
            del keep[:]
            toks = [t.encode("utf-8") for t in tokenizer(C.string_at(ptr, n).decode("utf-8"))]
            for i, t in enumerate(toks[:cap]):
                buf = C.create_string_buffer(t, len(t) + 1)
                keep.append(buf)
                out_ptr[i] = C.cast(buf, C.c_void_p).value
                out_len[i] = len(t)
            return len(toks)

        self.fn = _lib.TOKENIZER_FN(cb)

    @property
    def ptr(self):
        return C.cast(self.fn, C.c_void_p) if self.fn is not None else None


def _boosts(fields_boost):
    arr = (C.c_double * max(1, len(fields_boost)))(*[float(b) for b in fields_boost])
    return arr, len(fields_boost)


def _raise(e):
    if e.status == _lib.PS_EINVAL and "fields_boost" in str(e):
        raise IndexError(str(e))
    raise e


def _take_results(L, out, n):
    res = [QueryResult(out[i].key, out[i].score) for i in range(n)]
    L.ps_free(out)
    return res


class Snapshot:
    """Immutable flattened CSR view of an Index, resident in HBM (ps_snapshot)."""

    def __init__(self, handle, owner):
        self._L = _lib.load()
        self._h = handle
        self._owner = owner  # keep the Index alive

    def __del__(self):
        try:
            if self._h:
                self._L.ps_snapshot_free(self._h)
                self._h = None
        except Exception:
            pass

    def update(self):
        """ps_snapshot_update: bring the snapshot up to its Index's current state - a delta (alive bits,
        appended postings; O(changes)) when expressible, a full re-flatten otherwise.  -> stats dict."""
        st = _lib.UpdateStats()
        _lib.check(self._L.ps_snapshot_update(self._h, self._owner._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def save(self, path):
        """Write the flattened snapshot to disk (versioned binary dump)."""
        _lib.check(self._L.ps_snapshot_save(self._h, os.fsencode(path)))

    @classmethod
    def load(cls, path, device=0):
        """Load a snapshot written by save(); no Index needed (device=-1: host-only)."""
        L = _lib.load()
        h = C.c_void_p()
        _lib.check(L.ps_snapshot_load(os.fsencode(path), device, C.byref(h)))
        return cls(h, None)

    def info(self):
        i = _lib.SnapshotInfo()
        _lib.check(self._L.ps_snapshot_get_info(self._h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in i._fields_}

    def query(self, query, score_calculator, tokenizer, fields_boost, top_k=0):
        qb = query.encode("utf-8")
        desc = _scorer_desc(score_calculator)
        b, nb = _boosts(fields_boost)
        tok = _Tok(tokenizer)
        out, n = C.POINTER(_lib.Result)(), C.c_size_t()
        try:
            _lib.check(self._L.ps_snapshot_query(self._h, C.byref(desc), qb, len(qb), b, nb, tok.ptr, None, top_k,
                                                 C.byref(out), C.byref(n)))
        except PsError as e:
            _raise(e)
        return _take_results(self._L, out, n.value)

    @staticmethod
    def _pack_queries(queries):
        qb = [q.encode("utf-8") if isinstance(q, str) else bytes(q) for q in queries]
        arr = (_lib.Str * max(1, len(qb)))()
        for i, v in enumerate(qb):
            arr[i].ptr, arr[i].len = v, len(v)
        return qb, arr

    def query_batch(self, queries, score_calculator, tokenizer, fields_boost, top_k=0):
        """B independent Index::query calls in one kernel pass -> list[list[QueryResult]]."""
        qb, arr = self._pack_queries(queries)
        desc = _scorer_desc(score_calculator)
        b, nb = _boosts(fields_boost)
        tok = _Tok(tokenizer)
        out, offs = C.POINTER(_lib.Result)(), C.POINTER(C.c_size_t)()
        try:
            _lib.check(self._L.ps_snapshot_query_batch(self._h, C.byref(desc), arr, len(qb), b, nb, tok.ptr, None,
                                                       top_k, C.byref(out), C.byref(offs)))
        except PsError as e:
            _raise(e)
        res = [[QueryResult(out[j].key, out[j].score) for j in range(offs[i], offs[i + 1])] for i in range(len(qb))]
        self._L.ps_free(out)
        self._L.ps_free(offs)
        return res

    def query_batch_arrays(self, queries, score_calculator, tokenizer, fields_boost, top_k):
        """Like query_batch but returns numpy arrays (keys u64[n], scores f64[n], offsets[B+1])."""
        import numpy as np
        qb, arr = self._pack_queries(queries)
        desc = _scorer_desc(score_calculator)
        b, nb = _boosts(fields_boost)
        tok = _Tok(tokenizer)
        out, offs = C.POINTER(_lib.Result)(), C.POINTER(C.c_size_t)()
        try:
            _lib.check(self._L.ps_snapshot_query_batch(self._h, C.byref(desc), arr, len(qb), b, nb, tok.ptr, None,
                                                       top_k, C.byref(out), C.byref(offs)))
        except PsError as e:
            _raise(e)
        o = np.ctypeslib.as_array(offs, shape=(len(qb) + 1,)).copy()
        n = int(o[-1])
        keys, scores = np.empty(n, np.uint64), np.empty(n, np.float64)
        self._L.ps_results_split(out, n, keys.ctypes.data, scores.ctypes.data)
        self._L.ps_free(out)
        self._L.ps_free(offs)
        return keys, scores, o

    def query_batch_device(self, queries, score_calculator, tokenizer, fields_boost, top_k, d_keys, d_scores,
                           d_counts, stream=None):
        """Device-resident top-k: d_* are device pointers (ints) on this snapshot's device, e.g.
        torch tensors' data_ptr(); `stream` a hipStream_t handle (torch.cuda.Stream.cuda_stream)
        or None for the engine's own stream (synchronous)."""
        qb, arr = self._pack_queries(queries)
        desc = _scorer_desc(score_calculator)
        b, nb = _boosts(fields_boost)
        tok = _Tok(tokenizer)
        try:
            _lib.check(self._L.ps_snapshot_query_batch_device(self._h, C.byref(desc), arr, len(qb), b, nb, tok.ptr,
                                                              None, top_k, d_keys, d_scores, d_counts,
                                                              stream if stream else None))
        except PsError as e:
            _raise(e)

    def query_batch_device_flat(self, text, offsets, score_calculator, fields_boost, top_k, d_keys, d_scores,
                                d_counts, stream=None):
        """query_batch_device with the batch as one contiguous uint8 buffer + u64 offsets[n+1]
        (numpy arrays): no per-query marshalling on the Python side."""
        desc = _scorer_desc(score_calculator)
        b, nb = _boosts(fields_boost)
        try:
            _lib.check(self._L.ps_snapshot_query_batch_device_flat(
                self._h, C.byref(desc), text.ctypes.data, offsets.ctypes.data, len(offsets) - 1, b, nb, None, None,
                top_k, d_keys, d_scores, d_counts, stream if stream else None))
        except PsError as e:
            _raise(e)

    def _cached_args(self, score_calculator, fields_boost):
        """ctypes descriptor + boosts array for a serving loop that repeats the same scorer / boosts."""
        key = (score_calculator.kind, float(score_calculator.bm25k1), float(score_calculator.bm25b), tuple(fields_boost))
        hit = getattr(self, "_arg_cache", None)
        if hit is None or hit[0] != key:
            b, nb = _boosts(fields_boost)
            hit = (key, _scorer_desc(score_calculator), b, nb)
            self._arg_cache = hit
        return hit[1], hit[2], hit[3]

    def query_batch_allgather_flat(self, comm, text, offsets, score_calculator, fields_boost, top_k, d_local_block,
                                   d_all_blocks, stream=None):
        """ps_snapshot_query_batch_allgather_flat: score this rank's shard into d_local_block and
        all-gather every rank's block into d_all_blocks (ncclAllGather inside the library, ordered on
        `stream`).  comm: dist.Comm or None (one rank: no collective)."""
        desc, b, nb = self._cached_args(score_calculator, fields_boost)
        try:
            _lib.check(self._L.ps_snapshot_query_batch_allgather_flat(
                self._h, comm._h if comm is not None else None, C.byref(desc), text.ctypes.data, offsets.ctypes.data,
                len(offsets) - 1, b, nb, None, None, top_k, d_local_block, d_all_blocks, stream if stream else None))
        except PsError as e:
            _raise(e)

    def plan_ahead_flat(self, text, offsets, score_calculator, fields_boost=None):
        """ps_snapshot_plan_ahead_flat: announce the next flat batch (its planner count pass starts now); the flat query
        call that follows with the same (text, offsets) finds the totals ready.  -> accepted (bool)."""
        desc = self._cached_args(score_calculator, fields_boost)[0] if fields_boost is not None else _scorer_desc(score_calculator)
        ok = C.c_int(0)
        _lib.check(self._L.ps_snapshot_plan_ahead_flat(self._h, C.byref(desc), text.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                                                       C.byref(ok)))
        return bool(ok.value)

    def query_batch_device_planned_flat(self, text, offsets, score_calculator, fields_boost, top_k, d_keys, d_scores,
                                        d_counts, stream=None):
        """ps_snapshot_query_batch_device_planned_flat: like query_batch_device_flat, but the query planner
        (tokenise, trie lookup, expansion, before_each) runs on the device as well (BM25)."""
        desc = _scorer_desc(score_calculator)
        b, nb = _boosts(fields_boost)
        try:
            _lib.check(self._L.ps_snapshot_query_batch_device_planned_flat(
                self._h, C.byref(desc), text.ctypes.data, offsets.ctypes.data, len(offsets) - 1, b, nb, top_k, d_keys,
                d_scores, d_counts, stream if stream else None))
        except PsError as e:
            _raise(e)

    def plan_device(self, queries, score_calculator):
        """The plans the DEVICE planner builds for `queries` -> list of (entries, query_terms_len), the
        same shape Snapshot.plan returns per query."""
        from . import synth
        text, offsets = synth.pack_queries(list(queries))
        desc = _scorer_desc(score_calculator)
        ent, n = C.POINTER(_lib.PlanEntry)(), C.c_size_t()
        qb, ql = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
        _lib.check(self._L.ps_snapshot_plan_device(self._h, C.byref(desc), text.ctypes.data, offsets.ctypes.data,
                                                   len(queries), C.byref(ent), C.byref(n), C.byref(qb), C.byref(ql)))
        out = []
        for q in range(len(queries)):
            es = [{k: getattr(ent[i], k) for k, _ in _lib.PlanEntry._fields_} for i in range(qb[q], qb[q + 1])]
            out.append((es, ql[q]))
        for p in (ent, qb, ql):
            self._L.ps_free(p)
        return out

    def last_stats(self):
        s = _lib.BatchStats()
        _lib.check(self._L.ps_snapshot_last_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def kernel_times(self, reset=False):
        """(total_ms, launches) of the posting-accumulate kernel since the last reset (HIP events)."""
        t, n = C.c_double(), C.c_uint64()
        _lib.check(self._L.ps_snapshot_kernel_times(self._h, C.byref(t), C.byref(n), 1 if reset else 0))
        return t.value, n.value

    def kernel_breakdown(self, reset=False):
        """dict(score_ms, rows_ms, launches, score_kernel, score_busy_ms): the scoring kernel alone, K0/K0b in front
        of it, the scoring kernel's demangled symbol, and the wall-clock during which at least one scoring launch was
        executing (== score_ms unless consecutive batches' kernels overlap; ps_snapshot_kernel_breakdown)."""
        kt = _lib.KernelTimes()
        _lib.check(self._L.ps_snapshot_kernel_breakdown(self._h, C.byref(kt), 1 if reset else 0))
        return {"score_ms": kt.score_ms, "rows_ms": kt.rows_ms, "launches": kt.launches,
                "score_kernel": kt.score_kernel.decode("utf-8", "replace"), "score_busy_ms": kt.score_busy_ms}

    def work_counters(self, reset=False):
        """What the scoring kernels counted themselves since the last reset (ps_snapshot_work_counters):
        postings scanned, lookups by kind, offers, bytes touched.  Waits for outstanding work."""
        w = _lib.WorkCounters()
        _lib.check(self._L.ps_snapshot_work_counters(self._h, C.byref(w), 1 if reset else 0))
        return {k: getattr(w, k) for k, _ in w._fields_}

    def plan(self, query, score_calculator, tokenizer=None):
        """Host query plan (tokenise -> expand_term -> before_each): (entries, query_terms_len)."""
        qb = query.encode("utf-8")
        desc = _scorer_desc(score_calculator)
        tok = _Tok(tokenizer)
        out, n, qtl = C.POINTER(_lib.PlanEntry)(), C.c_size_t(), C.c_size_t()
        _lib.check(self._L.ps_snapshot_plan(self._h, C.byref(desc), qb, len(qb), tok.ptr, None, C.byref(out),
                                            C.byref(n), C.byref(qtl)))
        ents = [{k: getattr(out[i], k) for k, _ in _lib.PlanEntry._fields_} for i in range(n.value)]
        self._L.ps_free(out)
        return ents, qtl.value

    def host_csr(self):
        """numpy views of the host copy of the CSR planes (valid while the snapshot lives)."""
        import numpy as np
        c = _lib.HostCsr()
        _lib.check(self._L.ps_snapshot_host_csr(self._h, C.byref(c)))
        inf = self.info()
        P, F = int(c.plane_stride), inf["fields_num"]
        arr = np.ctypeslib.as_array
        return {"doc": arr(c.doc, shape=(P,)), "tf": arr(c.tf, shape=(max(F, 1), P))[:F],
                "fl": arr(c.fl, shape=(max(F, 1), P))[:F],
                "table": arr(c.table, shape=(max(1, inf["n_table_entries"]),)),
                "keys": arr(c.keys, shape=(max(1, inf["n_ids"]),))[:inf["n_ids"]],
                "alive": arr(c.alive, shape=(max(1, (inf["tiles_cap"] * inf["tile_docs"] + 31) // 32),)),
                "avg": arr(c.avg, shape=(max(F, 1),))[:F].copy(), "tile_docs": inf["tile_docs"]}


class Index:
    """Index<u64> (src/index.rs:19-33)."""

    def __init__(self, fields_num, expected_index_size=None, expected_documents_count=None):
        self._L = _lib.load()
        h = C.c_void_p()
        if expected_index_size is None:
            _lib.check(self._L.ps_index_new(fields_num, C.byref(h)))            # Index::new
        else:
            _lib.check(self._L.ps_index_new_with_capacity(fields_num, expected_index_size,
                                                          expected_documents_count or 10000, C.byref(h)))
        self._h = h
        self.fields_num = fields_num

    @classmethod
    def new(cls, fields_num):
        return cls(fields_num)

    @classmethod
    def new_with_capacity(cls, fields_num, expected_index_size, expected_documents_count):
        return cls(fields_num, expected_index_size, expected_documents_count)

    def __del__(self):
        try:
            if self._h:
                self._L.ps_index_free(self._h)
                self._h = None
        except Exception:
            pass

    # ---- build side ------------------------------------------------------------------------
    def add_document(self, field_accessors, tokenizer, key, doc):
        """Index::add_document(&[FieldAccessor<D>], Tokenizer, key, &doc) (src/index.rs:77-83).
        A field accessor is `fn(&D) -> Vec<&str>`: a callable returning a list of strings."""
        values = [list(acc(doc)) for acc in field_accessors]
        self.add_field_values(key, values, tokenizer)

    def add_field_values(self, key, values, tokenizer=None):
        """add_document with the accessors already applied: values[i] = str or list[str]."""
        vals, counts = [], []
        for f in values:
            vs = [f] if isinstance(f, str) else list(f)
            counts.append(len(vs))
            vals.extend(v.encode("utf-8") for v in vs)
        if len(counts) < self.fields_num:
            raise IndexError("fewer field accessors than fields_num (reference: index out of bounds, src/index.rs:91)")
        arr = (_lib.Str * max(1, len(vals)))()
        for i, v in enumerate(vals):
            arr[i].ptr, arr[i].len = v, len(v)
        cnt = (C.c_size_t * len(counts))(*counts)
        tok = _Tok(tokenizer)
        _lib.check(self._L.ps_index_add_document(self._h, key, arr, cnt, tok.ptr, None))

    def add_documents_flat(self, keys, text, offsets):
        """Bulk add (single-valued fields, whitespace tokenizer) from numpy arrays: keys u64[n],
        text uint8[...], offsets u64[n*F+1]."""
        import numpy as np
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        text = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray))
                                    else text)
        _lib.check(self._L.ps_index_add_documents_flat(self._h, len(keys), keys.ctypes.data, text.ctypes.data,
                                                       offsets.ctypes.data))

    def add_documents_flat_gpu(self, keys, text, offsets, device=0):
        """GPU bulk indexing of an EMPTY index (ps_index_add_documents_flat_gpu): same result as
        add_documents_flat, the per-token work runs on the device.  -> True if the GPU path ran."""
        import numpy as np
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        text = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray))
                                    else text)
        used = C.c_int(0)
        _lib.check(self._L.ps_index_add_documents_flat_gpu(self._h, len(keys), keys.ctypes.data, text.ctypes.data,
                                                           offsets.ctypes.data, device, C.byref(used)))
        return bool(used.value)

    def remove_document(self, key):
        _lib.check(self._L.ps_index_remove_document(self._h, key))

    def vacuum(self):
        _lib.check(self._L.ps_index_vacuum(self._h))

    # ---- query ------------------------------------------------------------------------------
    def query(self, query, score_calculator, tokenizer, fields_boost, top_k=0):
        """Index::query(query, &mut score_calculator, tokenizer, fields_boost) (src/query.rs:21-27):
        every matching document, score desc (ties: key asc).  Runs on the GPU; the flattened
        snapshot is rebuilt lazily after mutations."""
        qb = query.encode("utf-8")
        keep = None
        if getattr(score_calculator, "kind", None) == 3:  # a custom ScoreCalculator: host callbacks
            desc, keep = _callbacks_desc(score_calculator, self.fields_num)
        else:
            desc = _scorer_desc(score_calculator)
        b, nb = _boosts(fields_boost)
        tok = _Tok(tokenizer)
        out, n = C.POINTER(_lib.Result)(), C.c_size_t()
        try:
            _lib.check(self._L.ps_index_query(self._h, C.byref(desc), qb, len(qb), b, nb, tok.ptr, None, top_k,
                                              C.byref(out), C.byref(n)))
        except PsError as e:
            if keep and keep[2]:
                raise keep[2][0]
            _raise(e)
        if keep and keep[2]:
            self._L.ps_free(out)
            raise keep[2][0]
        return _take_results(self._L, out, n.value)

    def snapshot(self, device=0, tile_docs=0, headroom_pct=0):
        """Flatten to CSR planes and upload to `device` (-1: host-only, for inspection).
        headroom_pct > 0 reserves room so that Snapshot.update can append documents in place."""
        h = C.c_void_p()
        _lib.check(self._L.ps_index_snapshot_ex(self._h, device, tile_docs, headroom_pct, C.byref(h)))
        return Snapshot(h, self)

    # ---- read-side state the reference's unit tests look at -----------------------------------
    @property
    def fields(self):
        out = []
        for i in range(self.fields_num):
            s, a = C.c_uint64(), C.c_double()
            _lib.check(self._L.ps_index_field_details(self._h, i, C.byref(s), C.byref(a)))
            out.append(FieldDetails(s.value, a.value))
        return out

    def docs_len(self):
        return self._L.ps_index_docs_len(self._h)

    def doc_field_length(self, key):
        out = (C.c_uint64 * max(1, self.fields_num))()
        if not self._L.ps_index_doc_field_length(self._h, key, out):
            return None
        return list(out)[:self.fields_num]

    def count_nodes(self):
        return self._L.ps_index_count_nodes(self._h)

    def live_pointers(self):
        return self._L.ps_index_live_pointers(self._h)

    def children(self, term=""):
        t = term.encode("utf-8")
        n = self._L.ps_index_children(self._h, t, len(t), None, 0)
        if n < 0:
            return None
        buf = (C.c_uint32 * max(1, n))()
        self._L.ps_index_children(self._h, t, len(t), buf, n)
        return [chr(buf[i]) for i in range(n)]

    def count_documents(self, term):
        t = term.encode("utf-8")
        return self._L.ps_index_count_documents(self._h, t, len(t))

    def expand_term(self, term):
        """Index::expand_term (src/query.rs:109-126)."""
        t = term.encode("utf-8")
        need = C.c_size_t()
        n = self._L.ps_index_expand_term(self._h, t, len(t), None, 0, C.byref(need))
        buf = C.create_string_buffer(max(1, need.value))
        self._L.ps_index_expand_term(self._h, t, len(t), buf, need.value, C.byref(need))
        return [s.decode("utf-8") for s in buf.raw[:need.value].split(b"\0")[:n]]
