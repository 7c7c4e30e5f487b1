"""Key table (`ps_keytable_*`, include/probly_search_amd.h): the side table a binding for `Index<T>` with a non-integer
`T` (src/index.rs:19-33, src/query.rs:10-17) keeps.  Host logic against a Python dict as the model; on the GPU, an
index keyed by strings against the oracle keyed by the same strings."""
import ctypes as C
import random

import numpy as np
import pytest

import probly_search_amd as psa
from probly_search_amd import _lib
from probly_search_amd.keys import KeyTable, KeyedIndex


def test_intern_find_key_roundtrip_against_a_dict():
    rng = random.Random(7)
    kt, model, back = KeyTable(), {}, []
    keys = [b"", b"\0", b"\0\0", b"a", b"a\0", b"abcdefgh", b"abcdefgh\0", b"abcdefghi"]
    keys += [bytes(rng.randrange(256) for _ in range(rng.randrange(0, 40))) for _ in range(20000)]
    keys += [b"%d" % i for i in range(20000)] + keys[:500]  # repeats
    for k in keys:
        id_, ins = kt.intern(k)
        if k in model:
            assert (id_, ins) == (model[k], False)
        else:
            assert (id_, ins) == (len(model), True)  # dense, first-seen order
            model[k] = id_
            back.append(k)
    assert len(kt) == len(model)
    for k in rng.sample(keys, 3000):
        assert kt.find(k) == model[k]
    assert kt.find(b"never seen") is None and kt.find(b"abcdefg") is None
    for id_ in rng.sample(range(len(back)), 3000):
        assert kt.key(id_) == back[id_]
    res = [psa.QueryResult(i, 0.5) for i in (0, 1, 2, len(back) - 1, 7)]
    assert kt.resolve(res) == [back[r.key] for r in res]


def test_flat_interning_is_n_single_calls_in_order():
    words = [b"doc-%d" % (i % 700) for i in range(2000)] + [b""]
    data = np.frombuffer(b"".join(words), dtype=np.uint8)
    offs = np.zeros(len(words) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(w) for w in words])
    a, b = KeyTable(), KeyTable()
    ids = a.intern_flat(data, offs)
    assert ids.tolist() == [b.intern(w)[0] for w in words]
    assert len(a) == len(b) == 701
    ids2 = a.intern_flat(data, offs)  # again: nothing new
    assert ids2.tolist() == ids.tolist() and len(a) == 701


def test_errors_are_statuses_not_crashes():
    L = _lib.load()
    kt = KeyTable()
    kt.intern(b"x")
    with pytest.raises(psa.PsError) as e:
        kt.key(1)
    assert e.value.status == _lib.PS_EINVAL
    with pytest.raises(psa.PsError):
        kt.resolve([psa.QueryResult(5, 1.0)])
    bad = np.array([0, 3, 2], dtype=np.uint64)  # decreasing offsets
    with pytest.raises(psa.PsError):
        kt.intern_flat(np.zeros(4, dtype=np.uint8), bad)
    out = C.c_uint64()
    assert L.ps_keytable_intern(None, b"x", 1, C.byref(out), None) == _lib.PS_EINVAL
    assert L.ps_keytable_find(None, b"x", 1, None) == 0 and L.ps_keytable_len(None) == 0
    L.ps_keytable_free(None)


def test_keyed_index_host_side_state():
    """add / re-add / remove by string key reach the index under one id per key (src/index.rs:77-83,161-191)."""
    idx = KeyedIndex(1)
    idx.add_field_values("uuid-b", ["a b"])
    idx.add_field_values("uuid-a", ["a c"])
    assert idx.index.docs_len() == 2 and len(idx.keys) == 2
    idx.remove_document("uuid-zzz")  # never added: no-op
    idx.remove_document("uuid-b")
    assert idx.index.docs_len() == 1
    idx.vacuum()
    idx.add_field_values("uuid-b", ["a d"])  # re-added after removal: same id
    assert len(idx.keys) == 2 and idx.keys.find(b"uuid-b") == 0 and idx.index.docs_len() == 2


@pytest.mark.gpu
def test_string_keyed_index_matches_the_oracle_keyed_by_the_same_strings():
    from oracle import oracle as orc
    rng = random.Random(3)
    vocab = ["w%d" % i for i in range(60)]
    idx = KeyedIndex(2, decode=lambda b: b.decode("utf-8"))
    o, names = orc.Index(2), []
    for d in range(400):
        name = "doc/%04x/%s" % (rng.randrange(1 << 16), "x" * (d % 5))
        while name in names:
            name += "'"
        names.append(name)
        t = " ".join(rng.choice(vocab) for _ in range(rng.randrange(1, 9)))
        b = " ".join(rng.choice(vocab) for _ in range(rng.randrange(1, 30)))
        idx.add_field_values(name, [t, b])
        o.add_document(d, [[t], [b]])
    for d in (3, 77, 399):
        idx.remove_document(names[d])
        o.remove_document(d)
    snap = idx.snapshot()
    for calc, ocalc in ((psa.bm25.new(), orc.bm25()), (psa.zero_to_one.new(), orc.zero_to_one())):
        for q in ("w1 w2", "w5", "w3 w3 w40 nope", "w"):
            exp = [(names[k], s) for k, s in o.query(q, ocalc, [1.0, 2.0])]
            for got in (idx.query(q, calc, None, [1.0, 2.0]), snap.query(q, calc, None, [1.0, 2.0])):
                assert [(r.key, r.score) for r in got] == exp
    got = snap.query_batch(["w1", "w2 w7"], psa.bm25.new(), None, [1.0, 1.0], top_k=5)
    for q, g in zip(["w1", "w2 w7"], got):
        assert [(r.key, r.score) for r in g] == [(names[k], s) for k, s in o.query(q, orc.bm25(), [1.0, 1.0])][:5]
