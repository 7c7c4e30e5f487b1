"""SURVEY 8f N2 - device-side term lookup / prefix expansion: the planner kernel (k_plan over the
frozen trie in HBM) must emit exactly the plan the host planner emits - same expansion order
(newest child first, DFS: query.rs:130-147, the reference's test query.rs:344-364), same idf /
expansion_boost bits, same token accounting - and a batch scored from a device-built plan must get
the oracle's answers."""
import ctypes as C

import numpy as np
import pytest

import probly_search_amd as psa
from adapters import ProductIndex, replay, run_device_planned
from corpus_util import build_script, random_queries
from emu import bits
from oracle import oracle as orc
from probly_search_amd import dist as psd, synth

pytestmark = pytest.mark.gpu


def _same_plans(snap, queries):
    dev = snap.plan_device(queries, psa.bm25.new())
    for q, (ents, qtl) in zip(queries, dev):
        h_ents, h_qtl = snap.plan(q, psa.bm25.new())
        assert qtl == h_qtl, q
        assert len(ents) == len(h_ents), (q, len(ents), len(h_ents))
        for a, b in zip(ents, h_ents):
            for k in a:
                if k in ("idf", "boost"):
                    assert bits(a[k]) == bits(b[k]), (q, k, a[k], b[k])
                else:
                    assert a[k] == b[k], (q, k, a[k], b[k])


def test_reference_expand_term_order_on_device():
    """query.rs:344-364: expand_term("a") == ["adef", "abc"] (newest child first)."""
    p = psa.Index(2)
    p.add_field_values(1, ["abc", "hello world"])
    p.add_field_values(2, ["adef", "lorem ipsum"])
    snap = p.snapshot(device=0, tile_docs=256)
    (ents, qtl), (none, _), (h, _) = snap.plan_device(["a", "x", "h  w"], psa.bm25.new())
    host = snap.plan("a", psa.bm25.new())[0]
    assert [e["post_off"] for e in ents] == [e["post_off"] for e in host] and len(ents) == 2 and qtl == 1
    assert none == []
    assert [e["qterm_index"] for e in h] == [0, 2] and [e["qterm"] for e in h] == [0, 1]
    _same_plans(snap, ["a", "x", "h  w", "", " ", "abc adef a", "hello wor lorem"])


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_device_plan_equals_host_plan_random_corpora(seed):
    F, steps, vocab = build_script(300 + seed, n_docs=200, fields=2, vocab_size=60, shuffle_keys=seed % 2 == 1)
    p = ProductIndex(F)
    replay(steps, F, p)
    snap = p.idx.snapshot(device=0, tile_docs=256)
    _same_plans(snap, random_queries(seed, vocab, n=60) + ["", "  ", "é", "日", "日本 é a ab abc"])


def _run_planned(snap, queries, boosts, top_k):
    return run_device_planned(snap, queries, boosts, top_k)


@pytest.mark.parametrize("kernel", ["daat", "k_score"])
@pytest.mark.parametrize("cfg_name,n_docs", [("C2", 40_000), ("C5", 20_000)])
def test_batches_scored_from_a_device_built_plan(cfg_name, n_docs, kernel):
    """Device-built plan -> device-built K1d descriptors -> k_daat (default), or -> k_score (PS_DAAT=0):
    the oracle's answers either way, also after a delta (new terms, removals) and with other boosts."""
    psa.load().ps_set_option(b"PS_DAAT", 1 if kernel == "daat" else 0)
    try:
        _device_planned_batches(cfg_name, n_docs, kernel)
    finally:
        psa.load().ps_set_option(b"PS_DAAT", 1)


def _device_planned_batches(cfg_name, n_docs, kernel):
    cfg = dict(synth.CONFIGS[cfg_name], n_docs=n_docs, vocab=2_000)
    corpus = synth.Corpus(**cfg)
    p, o = synth.fill(psa.Index(2), corpus), synth.fill(orc.Index(2), corpus)
    snap = p.snapshot(device=0, headroom_pct=20)
    queries = corpus.queries(64, cfg["q_terms"]) + ["", "zzzzzz"]
    _same_plans(snap, queries[:20])
    got = _run_planned(snap, queries, [1.0, 1.0], 10)
    for q, g in zip(queries, got):
        exp = o.query(q, orc.bm25(), [1.0, 1.0])[:10]
        assert [(k, bits(s)) for k, s in g] == [(k, bits(s)) for k, s in exp], q
    assert snap.last_stats()["device_planned"] == 1
    assert snap.kernel_breakdown()["score_kernel"].startswith("ps::k_daat" if kernel == "daat" else "ps::k_score")
    # the device trie follows a delta (new terms re-freeze the trie on the host; the device copy is refreshed)
    for i in range(50):
        f = ["fresh term%d" % (i % 3), queries[0]]
        o.add_document(n_docs + i, [[f[0]], [f[1]]]); p.add_field_values(n_docs + i, f)
    o.remove_document(3); p.remove_document(3)
    assert snap.update()["mode"] == 1
    qs = queries[:16] + ["fresh", "term", "fre ter"]
    _same_plans(snap, qs)
    got = _run_planned(snap, qs, [2.0, 0.5], 7)
    for q, g in zip(qs, got):
        exp = o.query(q, orc.bm25(), [2.0, 0.5])[:7]
        assert [(k, bits(s)) for k, s in g] == [(k, bits(s)) for k, s in exp], q


def test_batches_in_flight_keep_their_results_apart():
    """The library overlaps consecutive K1d batches of one snapshot (three batch contexts, its own streams): nine
    different batches submitted back to back on ONE caller stream into nine output blocks, then one synchronisation -
    every block must hold its own batch's answer (the synchronous call's), device- and host-planned alike."""
    hip = psd._DeviceBuffer.hip()
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    cfg = dict(synth.CONFIGS["C2"], n_docs=60_000, vocab=3_000)
    corpus = synth.Corpus(**cfg)
    snap = synth.fill(psa.Index(2), corpus).snapshot(device=0)
    sc, K, B = psa.bm25.new(), 10, 96
    batches = [corpus.queries(B, 3, salt=s) for s in range(9)]
    want = [[[(r.key, bits(r.score)) for r in rs] for rs in snap.query_batch(b, sc, None, [1.0, 1.0], top_k=K)] for b in batches]
    for planner in (1, 0):
        psa.load().ps_set_option(b"PS_DEVICE_PLAN", planner)
        try:
            st = C.c_void_p()
            assert hip.hipStreamCreate(C.byref(st)) == 0
            bufs = [psd._DeviceBuffer(psd.block_bytes(B, K)) for _ in batches]
            packed = [synth.pack_queries(b) for b in batches]
            for (text, offsets), buf in zip(packed, bufs):
                snap.query_batch_allgather_flat(None, text, offsets, sc, [1.0, 1.0], K, buf.ptr.value, buf.ptr.value, stream=st.value)
                assert snap.last_stats()["device_planned"] == planner
            assert hip.hipStreamSynchronize(st) == 0
            for i, buf in enumerate(bufs):
                got = psd.unpack_blocks(buf.to_host(), 1, B, K, [B])
                assert [[(k, bits(s)) for k, s in rs] for rs in got] == want[i], (planner, i)
            hip.hipStreamDestroy(st)
        finally:
            psa.load().ps_set_option(b"PS_DEVICE_PLAN", 1)


def test_batches_in_flight_with_a_different_boost_vector_each():
    """fields_boost is per call (src/query.rs:26).  The per-boost score planes are recycled (LRU); a batch still in its
    scoring kernel must not see its plane rewritten for a later batch's boosts: nine batches, five distinct boost vectors
    (and a changed k1 in between), back to back on one caller stream - every block equals the synchronous answer."""
    hip = psd._DeviceBuffer.hip()
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    cfg = dict(synth.CONFIGS["C2"], n_docs=60_000, vocab=3_000)
    corpus = synth.Corpus(**cfg)
    snap = synth.fill(psa.Index(2), corpus).snapshot(device=0)
    K, B = 10, 96
    boosts = [[1.0, 1.0], [2.0, 0.5], [0.25, 3.0], [5.0, 5.0], [1.5, 0.75]]
    scorers = [psa.bm25.new(), psa.bm25.BM25(0.9, 0.4)]
    batches = [(corpus.queries(B, 3, salt=40 + s), boosts[s % 5], scorers[(s // 4) % 2]) for s in range(9)]
    want = [[[(r.key, bits(r.score)) for r in rs] for rs in snap.query_batch(b, sc, None, bo, top_k=K)] for b, bo, sc in batches]
    st = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0
    bufs = [psd._DeviceBuffer(psd.block_bytes(B, K)) for _ in batches]
    for (b, bo, sc), buf in zip(batches, bufs):
        text, offsets = synth.pack_queries(b)
        snap.query_batch_allgather_flat(None, text, offsets, sc, bo, K, buf.ptr.value, buf.ptr.value, stream=st.value)
    assert hip.hipStreamSynchronize(st) == 0
    for i, buf in enumerate(bufs):
        got = psd.unpack_blocks(buf.to_host(), 1, B, K, [B])
        assert [[(k, bits(s)) for k, s in rs] for rs in got] == want[i], i
    hip.hipStreamDestroy(st)


def test_batches_announced_ahead():
    """ps_snapshot_plan_ahead_flat: the planner's count pass of the NEXT batch starts while the current ones are scored.
    Announced and asked for, announced and NOT asked for (another batch, another scorer, a host-planned batch, nothing
    at all), announced twice - every query call returns its own batch's answer, back to back on one caller stream."""
    hip = psd._DeviceBuffer.hip()
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    cfg = dict(synth.CONFIGS["C2"], n_docs=60_000, vocab=3_000)
    corpus = synth.Corpus(**cfg)
    snap = synth.fill(psa.Index(2), corpus).snapshot(device=0)
    sc, z, K, B = psa.bm25.new(), psa.zero_to_one.new(), 10, 96
    batches = [corpus.queries(B, 3, salt=70 + s) for s in range(8)]
    packed = [synth.pack_queries(b) for b in batches]
    want = [[[(r.key, bits(r.score)) for r in rs] for rs in snap.query_batch(b, sc, None, [1.0, 1.0], top_k=K)] for b in batches]
    want_z = [[(r.key, bits(r.score)) for r in rs] for rs in snap.query_batch(batches[0], z, None, [1.0, 1.0], top_k=K)]
    st = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0
    bufs = [psd._DeviceBuffer(psd.block_bytes(B, K)) for _ in range(12)]
    got_order = []

    def run(i, buf, scorer=sc):
        text, offsets = packed[i]
        snap.query_batch_allgather_flat(None, text, offsets, scorer, [1.0, 1.0], K, buf.ptr.value, buf.ptr.value, stream=st.value)
        got_order.append((i, scorer is z))

    assert snap.plan_ahead_flat(*packed[0], sc) is True
    run(0, bufs[0])                                   # announced and asked for
    assert snap.last_stats()["device_planned"] == 1 and snap.last_stats()["plan_ms"] < 200.0  # (other test processes may share the GPU)
    for i in range(1, 4):                             # the serving loop: announce s + 1, then ask for it
        snap.plan_ahead_flat(*packed[i], sc)
        run(i, bufs[i])
    snap.plan_ahead_flat(*packed[4], sc)
    run(5, bufs[4])                                   # announced 4, asked for 5
    run(4, bufs[5])                                   # ... then 4 after all (planned afresh)
    snap.plan_ahead_flat(*packed[6], sc)
    snap.plan_ahead_flat(*packed[7], sc)              # two announced, the SECOND asked for first: out of order - both are dropped
    run(7, bufs[6])
    snap.plan_ahead_flat(*packed[6], sc)
    run(0, bufs[7], scorer=z)                         # another scorer and another batch in between
    run(6, bufs[8])
    assert snap.plan_ahead_flat(*packed[0], z) is True    # zero_to_one batches of simple queries are planned on the device too
    run(0, bufs[9], scorer=z)
    assert snap.last_stats()["device_planned"] == 1
    snap.plan_ahead_flat(*packed[0], sc)              # announced for one scorer, asked for with the other: the count pass is the same
    run(0, bufs[10], scorer=z)
    assert snap.last_stats()["device_planned"] == 1
    snap.plan_ahead_flat(*packed[1], sc)              # announced and never asked for before the snapshot is queried synchronously
    sync = [[(r.key, bits(r.score)) for r in rs] for rs in snap.query_batch(batches[2], sc, None, [1.0, 1.0], top_k=K)]
    assert sync == want[2]
    assert hip.hipStreamSynchronize(st) == 0
    for (i, is_z), buf in zip(got_order, bufs):
        got = psd.unpack_blocks(buf.to_host(), 1, B, K, [B])
        exp = want_z if is_z else want[i]
        assert [[(k, bits(s)) for k, s in rs] for rs in got] == exp, (i, is_z)
    hip.hipStreamDestroy(st)


def test_several_batches_announced_in_a_row_and_scoring_stream_variants():
    """The announcement queue (PS_PLAN_AHEAD_DEPTH, 3): three batches announced, then asked for in order - each finds its count
    pass done; a fourth announcement is refused (accepted = 0) without dropping anything.  And every way of placing consecutive
    batches on the scoring streams (PS_SCORE_ALT 0 / 1 / 2 / 3 / 4) returns the same bits."""
    hip = psd._DeviceBuffer.hip()
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    cfg = dict(synth.CONFIGS["C2"], n_docs=60_000, vocab=3_000)
    corpus = synth.Corpus(**cfg)
    snap = synth.fill(psa.Index(2), corpus).snapshot(device=0)
    sc, K, B = psa.bm25.new(), 10, 96
    batches = [corpus.queries(B, 3, salt=170 + s) for s in range(6)]
    packed = [synth.pack_queries(b) for b in batches]
    want = [[[(r.key, bits(r.score)) for r in rs] for rs in snap.query_batch(b, sc, None, [1.0, 1.0], top_k=K)] for b in batches]
    st = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0
    L = psa.load()
    try:
        for alt in (1, 0, 2, 3, 4):
            L.ps_set_option(b"PS_SCORE_ALT", alt)
            bufs = [psd._DeviceBuffer(psd.block_bytes(B, K)) for _ in batches]
            snap.query_batch(batches[0][:8], sc, None, [1.0, 1.0], top_k=K)  # (the knob is read at the next batch; drops stale announcements)
            assert [snap.plan_ahead_flat(*packed[i], sc) for i in range(4)] == [True, True, True, False]
            for i in range(6):
                text, offsets = packed[i]
                snap.query_batch_allgather_flat(None, text, offsets, sc, [1.0, 1.0], K, bufs[i].ptr.value, bufs[i].ptr.value, stream=st.value)
                assert snap.last_stats()["device_planned"] == 1
                if i + 3 < 6:
                    assert snap.plan_ahead_flat(*packed[i + 3], sc) is True
            assert hip.hipStreamSynchronize(st) == 0
            for i, buf in enumerate(bufs):
                got = psd.unpack_blocks(buf.to_host(), 1, B, K, [B])
                assert [[(k, bits(s)) for k, s in rs] for rs in got] == want[i], (alt, i)
    finally:
        L.ps_set_option(b"PS_SCORE_ALT", 1)
        hip.hipStreamDestroy(st)


def test_work_counters_of_a_batch():
    """ps_snapshot_work_counters: what the kernels counted is consistent with the plan (K1d scans a part of the
    postings the reference walks; k_score streams all of them), bytes follow the documented formula, reset works."""
    cfg = dict(synth.CONFIGS["C2"], n_docs=80_000, vocab=3_000)
    corpus = synth.Corpus(**cfg)
    snap = synth.fill(psa.Index(2), corpus).snapshot(device=0)
    queries = corpus.queries(128, 3)
    sc = psa.bm25.new()
    F = 2
    snap.work_counters(reset=True)
    snap.query_batch(queries, sc, None, [1.0, 1.0], top_k=10)
    walked = snap.last_stats()["postings_visited"]
    w = snap.work_counters(reset=True)
    assert snap.kernel_breakdown()["score_kernel"].startswith("ps::k_daat")
    assert w["launches"] == 1 and 0 < w["items_run"] <= w["items"]
    assert 0 < w["postings_scanned"] < walked and w["postings_reached_lookups"] <= w["postings_scanned"]
    assert w["results"] == 128 * 10 and w["k1_postings"] == 0
    want = (w["postings_scanned"] * (4 + 8 * F) + (w["lookups_row"] + w["lookups_cell"]) * 8 + w["lookups_probe"] * 4 +
            w["lookup_hits"] * 8 * F + w["items_run"] * 10 * 12 + w["results"] * 16)
    assert w["bytes_touched"] == want
    psa.load().ps_set_option(b"PS_DAAT", 0)
    try:
        snap.query_batch(queries, sc, None, [1.0, 1.0], top_k=10)
        w = snap.work_counters(reset=True)
        assert w["postings_scanned"] == 0 and w["k1_items"] > 0
        # every posting of every (query, list) is streamed, unless its list was read as a dense row
        assert 0 < w["k1_postings"] <= walked and (w["k1_postings"] == walked or w["k1_row_slices"] > 0)
    finally:
        psa.load().ps_set_option(b"PS_DAAT", 1)
    assert snap.work_counters()["launches"] == 0
