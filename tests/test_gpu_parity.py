"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle.
Bit-exact f64 scores and identical canonical ordering (BASELINE.md §5 asks for 1e-9; the design
goal — same operation order, no FMA contraction, ln on the host — is bit identity, and that is
what is asserted here)."""
import math

import pytest

import probly_search_amd as psa
from adapters import ProductIndex, oracle_scorer, product_scorer, replay
from corpus_util import build_script, random_queries
from emu import bits
from kat_runner import load_cases, run_case
from oracle import oracle as orc
from probly_search_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["daat", "k_score"])
def scoring_kernel(request):
    """Every test of this module runs twice: BM25 top-k batches on K1d k_daat (the default) and on
    K1 k_score (PS_DAAT=0; still the kernel of full-result mode, zero_to_one, small batches and
    non-positive boosts).  The knobs go through ps_set_option: engines re-read them at the next batch,
    so module-scoped snapshots (the full-size fixtures) really run under both settings."""
    L = psa.load()
    L.ps_set_option(b"PS_DAAT", 1 if request.param == "daat" else 0)
    L.ps_set_option(b"PS_DAAT_MULTI", 1)
    yield request.param
    L.ps_set_option(b"PS_DAAT", 1)


def row_stats(snap):
    """(rows read, rows scored) by the most recent batch: chosen on the host for K1 (batch stats), on the
    device for K1d (work counters; call snap.work_counters(reset=True) before the batch)."""
    st = snap.last_stats()
    if st["dense_rows"]:
        return st["dense_rows"], st["dense_rows_built"]
    wc = snap.work_counters(reset=True)
    return wc["rows_used"], wc["rows_built"]


def assert_same(got, exp, ctx):
    assert [k for k, _ in got] == [k for k, _ in exp], (ctx, got[:5], exp[:5], len(got), len(exp))
    for (k, a), (_, b) in zip(got, exp):
        assert bits(a) == bits(b), (ctx, k, a.hex(), b.hex())


@pytest.mark.parametrize("case", load_cases("reference_kats.json"), ids=lambda c: c["id"])
def test_reference_kats_on_gpu(case):
    """Every known-answer test of the reference, through Index.query on the GPU, bit-exact."""
    run_case(ProductIndex, product_scorer, case, force_exact=True)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("tile", [256, 2048])
def test_random_corpora_full_and_topk(seed, tile):
    F, steps, vocab = build_script(200 + seed, n_docs=300, fields=1 + seed % 3, vocab_size=40,
                                   shuffle_keys=seed % 2 == 1, multi_valued=seed % 3 == 2)
    o, p = orc.Index(F), ProductIndex(F)
    replay(steps, F, o, p)
    snap = p.idx.snapshot(device=0, tile_docs=tile)
    boosts = [1.0] * F if seed % 2 else [2.0, 0.5, 1.5][:F]
    queries = random_queries(seed, vocab, n=30) + ["", " ", "zzzz", "a"]
    scorers = [("bm25", {}), ("zero_to_one", {}), ("bm25", {"k1": 2.0, "b": 0.1})]
    for name, kw in scorers:
        sc = product_scorer(name, **kw)
        full = snap.query_batch(queries, sc, None, boosts, top_k=0)
        top3 = snap.query_batch(queries, sc, None, boosts, top_k=3)
        top64 = snap.query_batch(queries, sc, None, boosts, top_k=64)
        top100 = snap.query_batch(queries, sc, None, boosts, top_k=100)  # > 64: full mode + host cut
        for q, f, t3, t64, t100 in zip(queries, full, top3, top64, top100):
            exp = o.query(q, oracle_scorer(name, **kw), boosts)
            assert_same([tuple(r) for r in f], exp, (seed, name, q))
            assert_same([tuple(r) for r in t3], exp[:3], (seed, name, q, "top3"))
            assert_same([tuple(r) for r in t64], exp[:64], (seed, name, q, "top64"))
            assert_same([tuple(r) for r in t100], exp[:100], (seed, name, q, "top100"))
            single = snap.query(q, sc, None, boosts)
            assert_same([tuple(r) for r in single], exp, (seed, name, q, "single"))


def test_readd_layers_and_query_term_order_on_gpu():
    o, p = orc.Index(1), ProductIndex(1)
    for ix in (o, p):
        ix.add_document(1, ["a a b"])
        ix.add_document(2, ["a c"])
        ix.add_document(1, ["a c c"])
        ix.add_document(3, ["ab a"])
    # merge order matters ("x a" vs "a x")
    o2, p2 = orc.Index(1), ProductIndex(1)
    for ix in (o2, p2):
        for k, t in enumerate(["x ac ab", "x ab", "x ab", "x ab", "x ab", "x ab"]):
            ix.add_document(k, [t])
    for oi, pi, qs in ((o, p, ["a", "a c", "c a", "b", "a a"]), (o2, p2, ["x a", "a x"])):
        for q in qs:
            for name in ("bm25", "zero_to_one"):
                exp = oi.query(q, oracle_scorer(name), [1.0])
                got = pi.query(q, product_scorer(name), [1.0])
                assert_same(got, exp, (q, name))
    assert p2.query("x a", product_scorer("bm25"), [1.0])[0] != p2.query("a x", product_scorer("bm25"), [1.0])[0]


def test_edge_cases():
    p = ProductIndex(2)
    assert p.query("anything", product_scorer("bm25"), [1.0, 1.0]) == []  # empty index
    p.add_document(5, ["a b", "c"])
    assert p.query("", product_scorer("bm25"), [1.0, 1.0]) == []
    assert p.query("zzz", product_scorer("zero_to_one"), [1.0, 1.0]) == []
    with pytest.raises(IndexError):  # reference: fields_boost[x] out of bounds panic (bm25.rs:85)
        p.query("a", product_scorer("bm25"), [1.0])
    # zero / negative boosts: score() returns None (bm25.rs:89-92) -> no result
    assert p.query("a", product_scorer("bm25"), [0.0, 0.0]) == []
    p.remove_document(5)  # last doc removed: avg becomes NaN (index.rs:643), N = 0
    assert p.query("a", product_scorer("bm25"), [1.0, 1.0]) == []
    p.vacuum()
    assert p.query("a", product_scorer("bm25"), [1.0, 1.0]) == []
    o = orc.Index(2)
    o.add_document(5, ["a b", "c"])
    o.remove_document(5)
    assert o.query("a", orc.bm25(), [1.0, 1.0]) == []


def test_custom_tokenizer_roundtrip():
    p, o = psa.Index(1), orc.Index(1)
    tok = lambda s: [t for t in s.lower().replace(",", " ").split(" ")]
    for k, t in enumerate(["Hello, World", "hello again", "WORLD wide web"]):
        p.add_field_values(k, [t], tok)
        o.add_document(k, [t], tokenizer=tok)
    for q in ["HELLO", "wor", "web hello"]:
        exp = o.query(q, orc.bm25(), [1.0], tokenizer=tok)
        got = [tuple(r) for r in p.query(q, psa.bm25.new(), tok, [1.0])]
        assert_same(got, exp, q)


@pytest.mark.parametrize("cfg_name,n_docs,vocab", [("C2", 60_000, 5_000), ("C5", 40_000, 3_000), ("C1", 50_000, 20_000)])
def test_synthetic_configs_vs_oracle(cfg_name, n_docs, vocab):
    """BASELINE configs at sizes the oracle finishes in seconds: full lists on a subsample,
    top-10 on the whole batch, both scorers."""
    cfg = dict(synth.CONFIGS[cfg_name], n_docs=n_docs, vocab=vocab)
    corpus = synth.Corpus(**cfg)
    F = cfg["fields"]
    p, o = synth.fill(psa.Index(F), corpus), synth.fill(orc.Index(F), corpus)
    snap = p.snapshot(device=0)
    queries = corpus.queries(48, cfg["q_terms"])
    boosts = [1.0] * F
    for name in ("bm25", "zero_to_one"):
        sc = product_scorer(name)
        top = snap.query_batch(queries, sc, None, boosts, top_k=10)
        full = snap.query_batch(queries[:6], sc, None, boosts, top_k=0)
        _, _, _, otop = o.bench_queries(queries, oracle_scorer(name), boosts, threads=4, top_k=10)
        for q, t, e in zip(queries, top, otop):
            assert_same([tuple(r) for r in t], e, (cfg_name, name, q))
        for q, f in zip(queries[:6], full):
            assert_same([tuple(r) for r in f], o.query(q, oracle_scorer(name), boosts), (cfg_name, name, q, "full"))
    st = snap.last_stats()
    assert st["postings_visited"] > 0 and st["score_kernel_ms"] > 0


def test_deterministic_and_device_topk_buffers():
    import torch
    cfg = dict(synth.CONFIGS["C2"], n_docs=30_000, vocab=2_000)
    corpus = synth.Corpus(**cfg)
    p = synth.fill(psa.Index(2), corpus)
    snap = p.snapshot(device=0)
    queries = corpus.queries(64, 3)
    a = snap.query_batch(queries, psa.bm25.new(), None, [1.0, 1.0], top_k=10)
    b = snap.query_batch(queries, psa.bm25.new(), None, [1.0, 1.0], top_k=10)
    assert a == b
    K = 10
    dk = torch.zeros(64 * K, dtype=torch.int64, device="cuda:0")
    ds = torch.zeros(64 * K, dtype=torch.float64, device="cuda:0")
    dc = torch.zeros(64, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()  # (the zero fills run on torch's stream; the library's own streams are non-blocking: no implicit order)
    st = torch.cuda.current_stream()
    snap.query_batch_device(queries, psa.bm25.new(), None, [1.0, 1.0], K, dk.data_ptr(), ds.data_ptr(),
                            dc.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
    hk, hs, hc = dk.cpu().view(64, K), ds.cpu().view(64, K), dc.cpu()
    for i, res in enumerate(a):
        assert int(hc[i]) == len(res)
        for k, r in enumerate(res):
            assert int(hk[i, k]) == r.key and bits(float(hs[i, k])) == bits(r.score)


def test_batches_on_different_streams_are_ordered():
    """Batches share the engine's per-batch device buffers; enqueued back to back on different
    caller streams (and mixed with synchronous host calls) they must still come out right."""
    import torch
    cfg = dict(synth.CONFIGS["C2"], n_docs=120_000, vocab=3_000)
    corpus = synth.Corpus(**cfg)
    snap = synth.fill(psa.Index(2), corpus).snapshot(device=0)
    K, B, rounds = 10, 256, 6
    batches = [corpus.queries(B, 3, salt=r) for r in range(rounds)]
    expect = [snap.query_batch(b, psa.bm25.new(), None, [1.0, 1.0], top_k=K) for b in batches]
    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(3)]
    outs = []
    for r, b in enumerate(batches):
        dk = torch.zeros(B * K, dtype=torch.int64, device="cuda:0")
        ds = torch.zeros(B * K, dtype=torch.float64, device="cuda:0")
        dc = torch.zeros(B, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        outs.append((dk, ds, dc))
    for r, b in enumerate(batches):  # no host synchronisation between the enqueues
        dk, ds, dc = outs[r]
        snap.query_batch_device(b, psa.bm25.new(), None, [1.0, 1.0], K, dk.data_ptr(), ds.data_ptr(), dc.data_ptr(),
                                stream=streams[r % 3].cuda_stream)
        if r == 3:  # a synchronous call on the engine's own stream in the middle
            mid = snap.query_batch(batches[0], psa.bm25.new(), None, [1.0, 1.0], top_k=K)
            assert mid == expect[0]
    torch.cuda.synchronize()
    for r in range(rounds):
        dk, ds, dc = outs[r]
        hk, hs, hc = dk.cpu().view(B, K), ds.cpu().view(B, K), dc.cpu()
        for i, res in enumerate(expect[r]):
            assert int(hc[i]) == len(res), (r, i)
            for k, x in enumerate(res):
                assert int(hk[i, k]) == x.key and bits(float(hs[i, k])) == bits(x.score), (r, i, k)


def test_snapshot_queries_from_several_threads():
    """A ps_snapshot is immutable and its query entry points are thread-safe (`query(&self)`,
    src/query.rs:21-27): four threads hammering one snapshot with single queries, small batches
    and full-result calls get exactly the answers of a sequential run."""
    import threading
    cfg = dict(synth.CONFIGS["C2"], n_docs=40_000, vocab=2_000)
    corpus = synth.Corpus(**cfg)
    snap = synth.fill(psa.Index(2), corpus).snapshot(device=0)
    queries = corpus.queries(96, 3)
    b1 = [1.0, 1.0]

    def work(tid):
        out = []
        for i in range(tid, len(queries), 4):
            sc = psa.bm25.new() if (i // 4) % 2 == 0 else psa.zero_to_one.new()
            out.append((i, "one", snap.query(queries[i], sc, None, b1, top_k=10)))
            if i % 3 == 0:
                out.append((i, "batch", snap.query_batch(queries[i:i + 5], sc, None, b1, top_k=5)))
            if i % 8 == 0:
                out.append((i, "full", snap.query(queries[i], sc, None, b1)))
        return out

    expect = {tid: work(tid) for tid in range(4)}
    got = {}
    threads = [threading.Thread(target=lambda t=t: got.__setitem__(t, work(t))) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert got == expect


def test_full_size_properties_c2_slice():
    """Size-independent properties on a larger index (no oracle): top-k is a prefix of the full
    list, full list is sorted canonically, scores of a 1-term query are invariant to batching."""
    cfg = dict(synth.CONFIGS["C2"], n_docs=200_000, vocab=20_000)
    corpus = synth.Corpus(**cfg)
    p = synth.fill(psa.Index(2), corpus)
    snap = p.snapshot(device=0)
    queries = corpus.queries(32, 3)
    full = snap.query_batch(queries, psa.bm25.new(), None, [1.0, 1.0], top_k=0)
    top = snap.query_batch(queries, psa.bm25.new(), None, [1.0, 1.0], top_k=10)
    for f, t in zip(full, top):
        assert t == f[:10]
        keys = [(-r.score, r.key) for r in f]
        assert keys == sorted(keys) and len({r.key for r in f}) == len(f)
    # additivity: for single-expansion terms the 2-term score is the f64 sum of the 1-term scores
    q = queries[0].split(" ")
    s1 = {r.key: r.score for r in snap.query(q[0], psa.bm25.new(), None, [1.0, 1.0])}
    s2 = {r.key: r.score for r in snap.query(q[1], psa.bm25.new(), None, [1.0, 1.0])}
    both = snap.query(q[0] + " " + q[1], psa.bm25.new(), None, [1.0, 1.0])
    for r in both:
        exp = (s1[r.key] + s2[r.key]) if (r.key in s1 and r.key in s2) else s1.get(r.key, s2.get(r.key))
        assert bits(r.score) == bits(exp)


@pytest.fixture(scope="module")
def c2_full():
    """BASELINE config 2/3 at its full size: 1 M documents, 2 fields, ~33 M postings."""
    cfg = dict(synth.CONFIGS["C2"])
    corpus = synth.Corpus(**cfg)
    snap = synth.fill(psa.Index(2), corpus).snapshot(device=0)
    return corpus, snap


def test_full_size_c2_properties(c2_full, scoring_kernel):
    """Full BASELINE size, no oracle (it would take minutes): properties that hold at any size.
    (a) the batched top-10 (dense rows, fused / written row uses, LPT order) is the prefix of the
    full sorted list of the same query asked alone; (b) the 3-term score is ((s1 + s2) + s3) of the
    1-term scores, bit for bit, for every matching document; (c) splitting the batch (different
    hot-list selection) changes nothing; (d) doubling the boosts doubles every score exactly."""
    corpus, snap = c2_full
    sc = psa.bm25.new()
    b1 = [1.0, 1.0]
    queries = corpus.queries(1024, 3)
    snap.work_counters(reset=True)
    top = snap.query_batch(queries, sc, None, b1, top_k=10)
    assert snap.kernel_breakdown(reset=True)["score_kernel"].startswith("ps::k_daat" if scoring_kernel == "daat" else "ps::k_score")
    # dense score rows were in play: chosen on the host for K1 (batch stats), on the device for K1d (work counters)
    assert row_stats(snap)[0] > 0
    for qi in (0, 17, 333, 1023):
        full = snap.query(queries[qi], sc, None, b1)
        assert top[qi] == full[:10], qi
        keys = [(-r.score, r.key) for r in full]
        assert keys == sorted(keys) and len({r.key for r in full}) == len(full)
        terms = queries[qi].split(" ")
        parts = [{r.key: r.score for r in snap.query(t, sc, None, b1)} for t in terms]
        assert len(full) == len(set().union(*[set(p_) for p_ in parts]))
        for r in full:
            acc = None
            for p_ in parts:  # plan order == query term order; absent terms contribute nothing
                if r.key in p_:
                    acc = p_[r.key] if acc is None else acc + p_[r.key]
            assert bits(r.score) == bits(acc), (qi, r.key)
    halves = snap.query_batch(queries[:300], sc, None, b1, top_k=10) + snap.query_batch(queries[300:], sc, None, b1, top_k=10)
    assert halves == top
    twice = snap.query_batch(queries[:256], sc, None, [2.0, 2.0], top_k=10)
    for a, b in zip(top[:256], twice):
        assert [r.key for r in a] == [r.key for r in b]
        assert all(bits(2.0 * x.score) == bits(y.score) for x, y in zip(a, b))


def test_full_size_c3_properties(c2_full):
    """zero_to_one at the full size: batched top-10 == prefix of the query asked alone (full list),
    scores in (0, 1], batch split invariance."""
    corpus, snap = c2_full
    sc = psa.zero_to_one.new()
    b1 = [1.0, 1.0]
    queries = corpus.queries(512, 3, salt=5)
    top = snap.query_batch(queries, sc, None, b1, top_k=10)
    for qi in (1, 100, 511):
        full = snap.query(queries[qi], sc, None, b1)
        assert top[qi] == full[:10], qi
        assert all(0.0 < r.score <= 1.0 for r in full)
        keys = [(-r.score, r.key) for r in full]
        assert keys == sorted(keys)
    assert snap.query_batch(queries[:200], sc, None, b1, top_k=10) + snap.query_batch(queries[200:], sc, None, b1, top_k=10) == top


def test_sharded_entry_point_single_rank():
    """probly_search_amd.dist.query_batch_sharded at world_size 1 (no collective) == query_batch."""
    from probly_search_amd import dist as psd
    cfg = dict(synth.CONFIGS["C2"], n_docs=20_000, vocab=1_500)
    corpus = synth.Corpus(**cfg)
    p = synth.fill(psa.Index(2), corpus)
    snap = p.snapshot(device=0)
    queries = corpus.queries(33, 3)
    a = snap.query_batch(queries, psa.bm25.new(), None, [1.0, 1.0], top_k=7)
    b = psd.query_batch_sharded(snap, queries, psa.bm25.new(), [1.0, 1.0], 7)
    assert [[tuple(r) for r in x] for x in a] == b


@pytest.mark.parametrize("seed", range(4))
def test_dense_rows_path_forced(seed, monkeypatch):
    """K0b/K1 dense-row path (hot lists scored once per batch) forced on for every list: results
    must stay bit-identical, single- and multi-expansion (visited tags) queries alike."""
    monkeypatch.setenv("PS_DENSE_MIN_USES", "1")
    monkeypatch.setenv("PS_DENSE_MIN_DENSITY_PCT", "0")
    monkeypatch.setenv("PS_DAAT_DENSE_MIN_DENSITY_PCT", "0")
    # (the zero_to_one batch whole on the streaming kernels - the rows are theirs; split, its complex queries alone take k_z21)
    psa.load().ps_set_option(b"PS_DAAT_Z_SPLIT", 0)
    F, steps, vocab = build_script(300 + seed, n_docs=400, fields=1 + seed % 2, vocab_size=25, shuffle_keys=seed % 2 == 1)
    o, p = orc.Index(F), ProductIndex(F)
    replay(steps, F, o, p)
    snap = p.idx.snapshot(device=0, tile_docs=256)
    boosts = [1.0] * F if seed % 2 else [2.0, 0.5][:F]
    queries = random_queries(seed, vocab, n=40)
    for name, kw in (("bm25", {}), ("bm25", {"k1": 0.7, "b": 0.3}), ("zero_to_one", {})):
        sc = product_scorer(name, **kw)
        full = snap.query_batch(queries, sc, None, boosts, top_k=0)
        snap.work_counters(reset=True)
        top = snap.query_batch(queries, sc, None, boosts, top_k=5)
        assert row_stats(snap)[0] > 0
        for q, f, t in zip(queries, full, top):
            exp = o.query(q, oracle_scorer(name, **kw), boosts)
            assert_same([tuple(r) for r in f], exp, (seed, name, q, "dense-full"))
            assert_same([tuple(r) for r in t], exp[:5], (seed, name, q, "dense-top5"))


@pytest.mark.parametrize("zero_copy", ["1", "0"])
def test_latency_path_small_batches(zero_copy, monkeypatch):
    """The single-query / tiny-batch path: plan read in place from pinned memory, results written
    straight to pinned memory, no memset between batches (K3 leaves the control words clean),
    BM25 table reused until (k1, b) change.  Batches of 1..6 queries (the in-place threshold is 4),
    scorers and parameters alternating from call to call, top-k and full results interleaved."""
    monkeypatch.setenv("PS_ZERO_COPY", zero_copy)
    F, steps, vocab = build_script(911, n_docs=2500, fields=2, vocab_size=60)
    o, p = orc.Index(F), ProductIndex(F)
    replay(steps, F, o, p)
    snap = p.idx.snapshot(device=0, tile_docs=256)
    boosts = [1.5, 0.75]
    queries = random_queries(5, vocab, n=36) + ["", "zzzz"]
    variants = [("bm25", {}), ("zero_to_one", {}), ("bm25", {"k1": 2.0, "b": 0.1}), ("bm25", {}),
                ("bm25", {"k1": 0.5, "b": 1.0})]
    call = 0
    for nb in (1, 2, 3, 4, 5, 6, 1, 1):
        for start in range(0, len(queries) - nb, 7):
            name, kw = variants[call % len(variants)]
            call += 1
            qs = queries[start:start + nb]
            sc = product_scorer(name, **kw)
            k = (3, 10, 0)[call % 3]
            got = snap.query_batch(qs, sc, None, boosts, top_k=k) if nb > 1 else [snap.query(qs[0], sc, None, boosts, top_k=k)]
            for q, g in zip(qs, got):
                exp = o.query(q, oracle_scorer(name, **kw), boosts)
                assert_same([tuple(r) for r in g], exp[:k] if k else exp, (zero_copy, nb, name, kw, q, k))


@pytest.mark.parametrize("fields", [2, 1])
@pytest.mark.parametrize("fuse", ["3", "2", "1", "0"])
def test_dense_rows_first_written_last_fused(fuse, fields, monkeypatch):
    """BM25, one list per query term, every list dense: the query's last entry is added while the
    tile is harvested and a dense entry in position 0 / 1 is written first (swapped in front of a
    sparse e0) - the f64 sum must keep the reference's bits for 1..5 terms per query, repeated
    terms included, non-unit boosts, tiles of 256 and 1024 documents."""
    monkeypatch.setenv("PS_DENSE_MIN_USES", "1")
    monkeypatch.setenv("PS_DENSE_MIN_DENSITY_PCT", "30")  # only the head lists become rows: mixed plans
    monkeypatch.setenv("PS_DAAT_DENSE_MIN_DENSITY_PCT", "30")
    monkeypatch.setenv("PS_DENSE_FUSE", fuse)
    psa.load().ps_set_option(b"PS_DAAT_Z", 0)  # (the zero_to_one half is here for the streaming kernels' rows: K1dz - which takes queries of up to 8 lists - reads none; reset by conftest)
    cfg = dict(synth.CONFIGS["C2" if fields == 2 else "C1"], n_docs=6_000, vocab=300)
    corpus = synth.Corpus(**cfg)
    p, o = synth.fill(psa.Index(fields), corpus), synth.fill(orc.Index(fields), corpus)
    boosts = [1.25, 0.5][:fields]
    queries = []
    for nt in (1, 2, 3, 4, 5):
        queries += corpus.queries(24, nt, salt=nt)
    head = corpus.queries(1, 1)[0]
    queries += [head + " " + head, head + " zzzz " + head, "zzzz " + head]
    for tile in (256, 1024):
        snap = p.snapshot(device=0, tile_docs=tile)
        for name in ("bm25", "zero_to_one"):  # zero_to_one: the same tricks on its per-field planes (sorted record order)
            sc = product_scorer(name)
            snap.work_counters(reset=True)
            top = snap.query_batch(queries, sc, None, boosts, top_k=10)
            assert row_stats(snap)[0] > 0
            full = snap.query_batch(queries[::9], sc, None, boosts, top_k=0)
            for q, t in zip(queries, top):
                assert_same([tuple(r) for r in t], o.query(q, oracle_scorer(name), boosts)[:10], (fuse, tile, name, q))
            for q, f in zip(queries[::9], full):
                assert_same([tuple(r) for r in f], o.query(q, oracle_scorer(name), boosts), (fuse, tile, name, q, "full"))


def test_resident_rows_reuse_eviction_and_invalidation(monkeypatch):
    """Dense rows stay resident in the snapshot's row slab across batches (LRU).  A slab of a few
    slots and every list forced dense: rows are reused, evicted and rebuilt over a sequence of
    batches; changing k1 / b / the boosts / the scorer drops them.  Every answer bit-exact."""
    monkeypatch.setenv("PS_DENSE_MIN_USES", "1")
    monkeypatch.setenv("PS_DENSE_MIN_DENSITY_PCT", "0")
    monkeypatch.setenv("PS_DAAT_DENSE_MIN_DENSITY_PCT", "0")
    monkeypatch.setenv("PS_DENSE_MAX_ROWS", "6")
    monkeypatch.setenv("PS_ROW_CACHE_MB", "1")
    psa.load().ps_set_option(b"PS_DAAT_Z", 0)  # (the zero_to_one steps are here for the row slab: K1dz reads no rows; reset by conftest)
    cfg = dict(synth.CONFIGS["C2"], n_docs=30_000, vocab=400)   # 240 KB per row: 6 slots (the per-batch minimum)
    corpus = synth.Corpus(**cfg)
    p, o = synth.fill(psa.Index(2), corpus), synth.fill(orc.Index(2), corpus)
    snap = p.snapshot(device=0, tile_docs=512)
    seq = [("bm25", {}, [1.0, 1.0]), ("bm25", {}, [1.0, 1.0]), ("bm25", {}, [1.0, 1.0]), ("bm25", {"k1": 2.0, "b": 0.5}, [1.0, 1.0]),
           ("bm25", {"k1": 2.0, "b": 0.5}, [1.0, 3.0]), ("zero_to_one", {}, [1.0, 1.0]), ("zero_to_one", {}, [1.0, 1.0]),
           ("bm25", {}, [1.0, 1.0])]
    built = []
    for step, (name, kw, boosts) in enumerate(seq):
        queries = corpus.queries(12, 2, salt=step % 3)  # salts repeat: the same hot lists come back
        snap.work_counters(reset=True)
        top = snap.query_batch(queries, product_scorer(name, **kw), None, boosts, top_k=10)
        built.append(row_stats(snap))
        for q, t in zip(queries, top):
            assert_same([tuple(r) for r in t], o.query(q, oracle_scorer(name, **kw), boosts)[:10], (step, name, kw, boosts, q))
    assert all(u > 0 for u, _ in built)
    assert built[0][1] == built[0][0]      # cold: every row scored
    assert built[3][1] == built[3][0]      # new k1 / b: nothing reusable
    assert built[4][1] == built[4][0]      # new boosts
    assert any(b < u for u, b in built), built   # and somewhere a resident row was reused


def test_wide_prefix_expansion_many_entries():
    """A 1-2 character prefix expanding to hundreds of indexed terms: plans far larger than the
    register-resident group size, table slices disabled (too many entries for LDS), visited-tag
    merge across every expansion of the term — BM25 and the general zero_to_one kernel limits."""
    import random
    rng = random.Random(5)
    vocab = sorted({"ab" + "".join(rng.choice("abcdefgh") for _ in range(rng.randint(0, 4))) for _ in range(400)})
    o, p = orc.Index(1), ProductIndex(1)
    for k in range(600):
        text = " ".join(rng.choice(vocab) for _ in range(rng.randint(2, 9)))
        o.add_document(k, [text])
        p.add_document(k, [text])
    snap = p.idx.snapshot(device=0, tile_docs=256)
    for q in ["a", "ab", "abc x", "abd abe", "ab ab"]:
        exp = o.query(q, orc.bm25(), [1.0])
        got = [tuple(r) for r in snap.query(q, psa.bm25.new(), None, [1.0])]
        assert_same(got, exp, ("wide", q))
        top = [tuple(r) for r in snap.query(q, psa.bm25.new(), None, [1.0], top_k=10)]
        assert_same(top, exp[:10], ("wide-top", q))
    ents, _ = snap.plan("a", psa.bm25.new())
    assert len(ents) > 200
    # zero_to_one: hundreds of expansions of ONE query term go through the consumed-term-mask path
    for q in ("a", "ab", "abc x"):
        exp = o.query(q, orc.zero_to_one(), [1.0])
        got = [tuple(r) for r in snap.query(q, psa.zero_to_one.new(), None, [1.0])]
        assert_same(got, exp, ("wide-z21", q))
    # ... the same nodes under two query terms need the general kernel (per-node pools): any number of
    # expanded lists per query (round 1 refused more than 64 with PS_EUNSUPPORTED)
    for q in ("ab ab", "a ab abc", "abc ab a ab"):
        exp = o.query(q, orc.zero_to_one(), [1.0])
        got = [tuple(r) for r in snap.query(q, psa.zero_to_one.new(), None, [1.0])]
        assert_same(got, exp, ("wide-z21-general", q))
        top = [tuple(r) for r in snap.query(q, psa.zero_to_one.new(), None, [1.0], top_k=10)]
        assert_same(top, exp[:10], ("wide-z21-general-top", q))
    ents, _ = snap.plan("ab ab", psa.zero_to_one.new())
    assert len(ents) > 64


def test_zero_to_one_many_query_terms_mostly_out_of_vocabulary():
    """consumed_index is tracked per query term WITH entries: a query of 80 tokens, most of them
    matching nothing, whose matching terms sit at token ordinals >= 64 (the advisor's aliasing case:
    `1ull << qterm` wrapped), combined with prefix expansion and repeated terms (general kernel)."""
    o, p = orc.Index(2), ProductIndex(2)
    import random
    rng = random.Random(11)
    vocab = ["abc", "abcd", "abce", "abd", "xyz", "xyzz", "q", "qq"]
    for k in range(300):
        f0 = " ".join(rng.choice(vocab) for _ in range(rng.randint(1, 4)))
        f1 = " ".join(rng.choice(vocab) for _ in range(rng.randint(2, 7)))
        o.add_document(k, [f0, f1])
        p.add_document(k, [f0, f1])
    snap = p.idx.snapshot(device=0, tile_docs=256)
    oov = ["zz%d" % i for i in range(70)]
    queries = [" ".join(oov + ["ab", "abc", "ab"]), " ".join(oov[:66] + ["abc"] + oov[66:] + ["abc", "xy", "q"]),
               " ".join(["ab"] + oov + ["ab", "xyz"])]
    for q in queries:
        for top_k in (0, 5):
            exp = o.query(q, orc.zero_to_one(), [1.0, 1.0])
            got = [tuple(r) for r in snap.query(q, psa.zero_to_one.new(), None, [1.0, 1.0], top_k=top_k)]
            assert_same(got, exp if top_k == 0 else exp[:top_k], ("many-terms", q[:20], top_k))
    got = snap.query_batch(queries, psa.zero_to_one.new(), None, [1.0, 1.0], top_k=0)
    for q, g in zip(queries, got):
        assert_same([tuple(r) for r in g], o.query(q, orc.zero_to_one(), [1.0, 1.0]), ("many-terms batch", q[:20]))


def test_snapshot_loaded_from_disk_scores_identically(tmp_path):
    cfg = dict(synth.CONFIGS["C2"], n_docs=20_000, vocab=1_500)
    corpus = synth.Corpus(**cfg)
    p = synth.fill(psa.Index(2), corpus)
    snap = p.snapshot(device=0)
    path = str(tmp_path / "c2.snap")
    snap.save(path)
    back = psa.Snapshot.load(path, device=0)
    queries = corpus.queries(40, 3)
    for sc in (psa.bm25.new(), psa.zero_to_one.new()):
        assert snap.query_batch(queries, sc, None, [1.0, 1.0], top_k=10) == back.query_batch(queries, sc, None, [1.0, 1.0], top_k=10)
        assert snap.query(queries[0], sc, None, [1.0, 1.0]) == back.query(queries[0], sc, None, [1.0, 1.0])


def test_zero_to_one_large_term_frequencies():
    """k_score's zero_to_one arm divides once per (posting, field) while every term frequency of the trip
    is <= the entry's exact-numerator limit (48 for an exact-match expansion) and evaluates the full
    zero_to_one.rs:117-120 expression otherwise: documents on both sides of the limit, exact terms and
    prefix expansions, top-k and full results."""
    F = 2
    o, p = orc.Index(F), ProductIndex(F)
    tfs = [1, 2, 3, 7, 47, 48, 49, 50, 63, 64, 65, 98, 120, 255, 256, 300]
    for k, tf in enumerate(tfs * 3):
        a = " ".join(["alpha"] * tf + ["beta"] * (k % 5) + ["alphabet"] * (k % 3))
        b = " ".join(["beta"] * (tf // 2 + 1) + ["alpha"] * (k % 4) + ["gamma"])
        for ix in (o, p):
            ix.add_document(1000 + k, [a, b])
    snap = p.idx.snapshot(device=0, tile_docs=256)
    queries = ["alpha", "alpha beta", "alph", "beta gamma alpha", "alphabet alpha", "a b g"] * 3
    # (bm25 too: term frequencies >= 255 do not fit the packed {tf, field length} posting words the
    # kernels stream and are fetched from the exact planes)
    for name in ("zero_to_one", "bm25"):
        sc = product_scorer(name)
        for boosts in ([1.0, 1.0], [2.0, 0.5]):
            full = snap.query_batch(queries, sc, None, boosts, top_k=0)
            top5 = snap.query_batch(queries, sc, None, boosts, top_k=5)
            for q, f, t5 in zip(queries, full, top5):
                exp = o.query(q, oracle_scorer(name), boosts)
                assert_same([tuple(r) for r in f], exp, (name, q, boosts))
                assert_same([tuple(r) for r in t5], exp[:5], (name, q, boosts, "top5"))


@pytest.fixture(scope="module")
def huge_field_pair():
    """(oracle index, snapshot) of four documents, one with a field of 2^24 + 50 tokens (tokenising 32 MB of text twice takes
    ~15 s: built once for both kernel settings of this module)."""
    F = 2
    n = (1 << 24) + 50
    o, p = orc.Index(F), ProductIndex(F)
    big = "x " * (n - 3) + "rare y x"
    docs = [(1, [big, "x y"]), (2, ["x y rare", "y"]), (3, ["y y", "rare x x"]), (4, ["z", big[: 2 * 300]])]
    for key, fields in docs:
        for ix in (o, p):
            ix.add_document(key, fields)
    return o, p.idx.snapshot(device=0, tile_docs=256)


def test_field_longer_than_the_packed_posting_word_holds(huge_field_pair):
    """A field of 2^24 + 50 tokens: its length (and the term frequency) saturate the packed posting word
    (8-bit tf, 24-bit field length) and every kernel has to fetch the exact values from the planes."""
    o, snap = huge_field_pair
    queries = ["x", "rare", "x rare y", "y z", "ra"] * 2  # (>= PS_DAAT_MIN_BATCH queries: the top-k batch takes the pruning kernels)
    for name in ("bm25", "zero_to_one"):
        sc = product_scorer(name)
        full = snap.query_batch(queries, sc, None, [1.0, 1.5], top_k=0)
        top2 = snap.query_batch(queries, sc, None, [1.0, 1.5], top_k=2)
        want = {}  # (the oracle walks one pointer per occurrence - 16 M per query that holds "x": once per distinct query)
        for q, f, t2 in zip(queries, full, top2):
            if q not in want:
                want[q] = o.query(q, oracle_scorer(name), [1.0, 1.5])
            exp = want[q]
            assert_same([tuple(r) for r in f], exp, (name, q))
            assert_same([tuple(r) for r in t2], exp[:2], (name, q, "top2"))


def test_full_result_blocks_from_the_pinned_pool():
    """Large result blocks are pinned pool blocks the device writes into (ps_engine.hpp: ResultBuf): forced for
    every size here.  Full lists of 1 / 5 / 12 queries (one set of device-wide sorts over all runs) and the
    truncated full mode (top_k = 100) against the oracle; a block handed back with ps_free is the next call's block."""
    import ctypes as C
    from probly_search_amd import _lib
    L = psa.load()
    cfg = dict(synth.CONFIGS["C2"], n_docs=60_000, vocab=5_000)
    corpus = synth.Corpus(**cfg)
    p, o = synth.fill(psa.Index(2), corpus), synth.fill(orc.Index(2), corpus)
    snap = p.snapshot(device=0)
    queries = corpus.queries(12, 3) + ["", "zzzzzz"]
    boosts = [1.0, 1.5]
    L.ps_set_option(b"PS_RESULT_PINNED_MIN_KB", 0)
    L.ps_set_option(b"PS_FULL_PARTS_MIN_KB", 0)  # ... and in up to 4 parts (downloads beside the next part's sorts)
    try:
        for name in ("bm25", "zero_to_one"):
            sc = product_scorer(name)
            exp = [o.query(q, oracle_scorer(name), boosts) for q in queries]
            assert sum(len(e) for e in exp[:12]) > 12 * 32768  # the runs are of the size that takes the device-wide sorts
            for sel in (slice(0, 1), slice(3, 8), slice(0, 14), slice(11, 14), slice(12, 14)):  # (the last two: parts that are empty runs)
                for k in (0, 100):
                    got = snap.query_batch(queries[sel], sc, None, boosts, top_k=k)
                    for q, g, e in zip(queries[sel], got, exp[sel]):
                        assert_same([tuple(r) for r in g], e[:k] if k else e, (name, q, k))
        # the block of one call, freed, is the block of the next call of the same size
        qb, arr = snap._pack_queries(queries[:4])
        desc = psa.index._scorer_desc(product_scorer("bm25"))
        b, nb = psa.index._boosts(boosts)
        seen = []
        for _ in range(3):
            out, offs = C.POINTER(_lib.Result)(), C.POINTER(C.c_size_t)()
            _lib.check(L.ps_snapshot_query_batch(snap._h, C.byref(desc), arr, len(qb), b, nb, None, None, 0,
                                                 C.byref(out), C.byref(offs)))
            seen.append(C.cast(out, C.c_void_p).value)
            assert offs[4] > 4 * 1000
            L.ps_free(out)
            L.ps_free(offs)
        assert seen[0] == seen[1] == seen[2], seen
    finally:
        L.ps_set_option(b"PS_RESULT_PINNED_MIN_KB", 4096)
        L.ps_set_option(b"PS_FULL_PARTS_MIN_KB", 32768)
