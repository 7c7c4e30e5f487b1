"""K1dz (k_daat_z, ps_z21_daat.hpp): exact top-K with dynamic pruning for zero_to_one
(src/score/default/zero_to_one.rs:44-126) - bit-identical to the oracle and to the streaming kernel
(k_score<MODE_Z21S>, PS_DAAT_Z=0), ties at the threshold included: the scorer's scores are small rationals,
thousands of documents tie, and the order among them is key ascending (src/lib.rs:54-58)."""
import random

import pytest

import probly_search_amd as psa
from emu import bits
from oracle import oracle as orc
from probly_search_amd import synth

pytestmark = pytest.mark.gpu


def _opt(name, v):
    psa.load().ps_set_option(name, v)


@pytest.fixture(autouse=True)
def _restore():
    yield
    _opt(b"PS_DAAT_Z", 1)
    _opt(b"PS_DAAT_CHUNK", 4096)


def _topk(snap, queries, K, boosts):
    return [[(r.key, bits(r.score)) for r in rs] for rs in snap.query_batch(queries, psa.zero_to_one.new(), None, boosts, top_k=K)]


def _check(snap, o, queries, K, F, expect_kernel="ps::k_daat_z"):
    boosts = [1.0] * F
    _opt(b"PS_DAAT_Z", 1)
    got = _topk(snap, queries, K, boosts)
    name = snap.kernel_breakdown()["score_kernel"]
    assert name.startswith(expect_kernel), name
    for _ in range(3):  # (thresholds are published by racing waves: the answer must not depend on who wins)
        assert _topk(snap, queries, K, boosts) == got
    _opt(b"PS_DAAT_Z", 0)
    ref = _topk(snap, queries, K, boosts)
    assert snap.kernel_breakdown()["score_kernel"].startswith(("ps::k_score", "ps::k_z21"))
    _opt(b"PS_DAAT_Z", 1)
    for q, g, r in zip(queries, got, ref):
        assert g == r, (q, K, g[:4], r[:4])
    for q, g in list(zip(queries, got))[:24]:
        exp = [(k, bits(s)) for k, s in o.query(q, orc.zero_to_one(), boosts)[:K]]
        assert g == exp, (q, K, g[:4], exp[:4])
    return got


def _tie_corpus(n_docs, fields, seed, vocab=40, short=(1, 4), long=(6, 14)):
    """Few distinct terms, short fields: nearly every score is one of a handful of rationals."""
    rng = random.Random(seed)
    words = ["w%02d" % i for i in range(vocab)] + ["wa", "wab", "wabc"]
    docs = []
    for _ in range(n_docs):
        vals = []
        for f in range(fields):
            lo, hi = short if f == 0 else long
            vals.append(" ".join(rng.choice(words[: 6 + 3 * f] if rng.random() < 0.7 else words) for _ in range(rng.randint(lo, hi))))
        docs.append(vals)
    return words, docs


def _build(docs, fields):
    p, o = psa.Index(fields), orc.Index(fields)
    for k, vals in enumerate(docs):
        p.add_field_values(k * 3 + 1, vals)
        o.add_document(k * 3 + 1, vals)
    return p, o


@pytest.mark.parametrize("fields", [1, 2, 3])
@pytest.mark.parametrize("K", [1, 10, 64])
def test_ties_everywhere(fields, K):
    words, docs = _tie_corpus(30_000, fields, seed=fields * 10 + K)
    p, o = _build(docs, fields)
    snap = p.snapshot(device=0, tile_docs=256)
    rng = random.Random(K)
    queries = [" ".join(rng.choice(words[:8]) for _ in range(rng.randint(1, 4))) for _ in range(40)]
    queries += ["w00 w00", "w01 w01 w01", "w00 w01 w00", "w02  w03", "zzz w00", "w00"] + ["w0%d w0%d w0%d" % (a, b, c) for a, b, c in [(0, 1, 2), (1, 2, 3), (3, 4, 5)]]
    _opt(b"PS_DAAT_CHUNK", 256)  # several chunks per list, on both sides of D0
    _check(snap, o, queries, K, fields)


@pytest.mark.parametrize("fields,kernel", [(4, "ps::k_daat_z"), (5, ("ps::k_score", "ps::k_z21"))])
def test_field_count_gate_edge(fields, kernel):
    """K1dz keeps its per-field state in registers for F <= 4 (ps_engine.hip: the `s.F <= 4` gates): F = 4 is taken, F = 5 goes to
    the streaming kernels - both against the oracle, ties included."""
    words, docs = _tie_corpus(12_000, fields, seed=40 + fields)
    p, o = _build(docs, fields)
    snap = p.snapshot(device=0, tile_docs=256)
    rng = random.Random(fields)
    queries = [" ".join(rng.choice(words[:8]) for _ in range(rng.randint(1, 4))) for _ in range(32)] + ["w00 w01 w00", "zzz w00", "w00"]
    _opt(b"PS_DAAT_CHUNK", 256)
    boosts = [1.0] * fields
    got = _topk(snap, queries, 10, boosts)
    assert snap.kernel_breakdown()["score_kernel"].startswith(kernel), snap.kernel_breakdown()["score_kernel"]
    for q, g in zip(queries, got):
        exp = [(k, bits(s)) for k, s in o.query(q, orc.zero_to_one(), boosts)[:10]]
        assert g == exp, (fields, q, g[:4], exp[:4])
    if fields == 4:
        _opt(b"PS_DAAT_Z", 0)
        assert _topk(snap, queries, 10, boosts) == got


@pytest.mark.parametrize("fields", [1, 2])
@pytest.mark.parametrize("planner", [1, 0])
def test_five_to_eight_lists_take_the_wide_instantiation(fields, planner):
    """zero_to_one.rs:84-126 takes any number of records.  Queries of five to eight lists run on k_daat_z<F, WC, 8> (the same
    kernel with eight lists of per-document state); a batch that mixes them with shorter queries takes that instantiation
    whole.  Against the streaming kernels and the oracle, ties included; device- and host-planned."""
    words, docs = _tie_corpus(30_000, fields, seed=70 + fields)
    p, o = _build(docs, fields)
    snap = p.snapshot(device=0, tile_docs=256)
    rng = random.Random(fields)
    queries = [" ".join(rng.sample(words[:12], rng.randint(5, 8))) for _ in range(24)]
    queries += [" ".join(rng.choice(words[:8]) for _ in range(rng.randint(1, 4))) for _ in range(12)]
    queries += ["w00 w01 w02 w03 w00 w01", "wa w00 w01 w02", "w00 w01 w02 w03 w04 w05 w06 w07", "zzz w00 w01 w02 w03 w04", ""]
    plans = [snap.plan(q, psa.zero_to_one.new())[0] for q in queries]
    assert 4 < max(len(e) for e in plans) <= 8
    _opt(b"PS_DAAT_CHUNK", 256)
    _opt(b"PS_DEVICE_PLAN", planner)
    boosts = [1.0] * fields
    try:
        for K in (1, 10, 64):
            got = _planned(snap, queries, K, boosts)
            assert snap.last_stats()["device_planned"] == planner
            name = snap.kernel_breakdown()["score_kernel"]
            assert name.startswith("ps::k_daat_z<") and name.endswith(", 8>"), name
            for _ in range(2):
                assert _planned(snap, queries, K, boosts) == got
            _opt(b"PS_DAAT_Z", 0)
            ref = _planned(snap, queries, K, boosts)
            assert snap.kernel_breakdown()["score_kernel"].startswith(("ps::k_score", "ps::k_z21"))
            _opt(b"PS_DAAT_Z", 1)
            for q, g, r in zip(queries, got, ref):
                assert g == r, (q, K, g[:4], r[:4])
            for q, g in zip(queries, got):
                exp = [(k, bits(s)) for k, s in o.query(q, orc.zero_to_one(), boosts)[:K]]
                assert g == exp, (q, K, g[:4], exp[:4])
    finally:
        _opt(b"PS_DEVICE_PLAN", 1)
        _opt(b"PS_DAAT_Z", 1)


def test_prefix_expansions_within_four_lists():
    """Several expansions of one query term (consumed_index, zero_to_one.rs:101-103) and the same node under
    two query terms (the pool rule, :104-113) - as long as a query has at most 4 lists it stays on K1dz."""
    words, docs = _tie_corpus(20_000, 2, seed=5)
    p, o = _build(docs, 2)
    snap = p.snapshot(device=0, tile_docs=256)
    queries = ["wa", "wab w00", "wa w01", "wabc wabc", "wabc w00 wabc", "w00 wab", "wa w00"] * 2
    plans = [snap.plan(q, psa.zero_to_one.new())[0] for q in queries]
    assert max(len(e) for e in plans) <= 4 and any(len(e) > len(q.split()) for e, q in zip(plans, queries))
    _opt(b"PS_DAAT_CHUNK", 256)
    for K in (1, 10, 64):
        _check(snap, o, queries, K, 2)
    # several expansions of a term AND a node under two records ("wab wab": wab, wabc twice): the general kernel's case
    got = _topk(snap, ["wab wab"] * 8, 10, [1.0, 1.0])
    assert snap.kernel_breakdown()["score_kernel"].startswith(("ps::k_score", "ps::k_z21"))
    assert got[0] == [(k, bits(s)) for k, s in o.query("wab wab", orc.zero_to_one(), [1.0, 1.0])[:10]]


def test_batches_that_do_not_qualify_keep_the_streaming_kernels():
    words, docs = _tie_corpus(5_000, 2, seed=6)
    p, o = _build(docs, 2)
    snap = p.snapshot(device=0, tile_docs=256)
    queries = ["w"] * 4 + ["w00 w01"] * 8  # "w" expands to every word: far more than 4 lists
    got = _topk(snap, queries, 10, [1.0, 1.0])
    assert snap.kernel_breakdown()["score_kernel"].startswith(("ps::k_score", "ps::k_z21"))
    for q, g in zip(queries[:5], got):
        assert g == [(k, bits(s)) for k, s in o.query(q, orc.zero_to_one(), [1.0, 1.0])[:10]]


def test_c3_shape_at_oracle_size_repeated_runs_are_bit_identical():
    cfg = dict(synth.CONFIGS["C3"], n_docs=120_000, vocab=8_000)
    corpus = synth.Corpus(**cfg)
    p, o = synth.fill(psa.Index(2), corpus), synth.fill(orc.Index(2), corpus)
    snap = p.snapshot(device=0)
    queries = corpus.queries(256, 3)
    first = _check(snap, o, queries, 10, 2)
    for _ in range(3):
        assert _topk(snap, queries, 10, [1.0, 1.0]) == first
    w = snap.work_counters(reset=True)
    assert w["z_postings_scanned"] > 0 and w["z_postings_scanned"] == w["postings_scanned"]


def test_removed_documents_through_a_delta():
    words, docs = _tie_corpus(20_000, 2, seed=9)
    p, o = _build(docs, 2)
    snap = p.snapshot(device=0, tile_docs=256, headroom_pct=10)
    rng = random.Random(3)
    for k in rng.sample(range(20_000), 700):
        p.remove_document(k * 3 + 1)
        o.remove_document(k * 3 + 1)
    snap.update()
    queries = [" ".join(rng.choice(words[:8]) for _ in range(3)) for _ in range(32)]
    _opt(b"PS_DAAT_CHUNK", 256)
    _check(snap, o, queries, 10, 2)


# ---- the same batches planned on the device (k_plan zmode -> k_zplan_arrange -> K1dz; SURVEY 8f N2 for zero_to_one) ----

def _planned(snap, queries, K, boosts):
    from adapters import run_device_planned
    got = run_device_planned(snap, queries, boosts, K, scorer=psa.zero_to_one.new())
    return [[(k, bits(s)) for k, s in rs] for rs in got]


def _check_planned(snap, o, queries, K, F, device_planned=1):
    """Device-planned == host-planned (both through K1dz when the batch qualifies) == oracle."""
    boosts = [1.0] * F
    got = _planned(snap, queries, K, boosts)
    st = snap.last_stats()
    assert st["device_planned"] == device_planned, st
    name = snap.kernel_breakdown()["score_kernel"]
    assert name.startswith("ps::k_daat_z" if device_planned else ("ps::k_score", "ps::k_z21", "ps::k_daat_z")), name
    for _ in range(2):
        assert _planned(snap, queries, K, boosts) == got
    _opt(b"PS_DEVICE_PLAN", 0)
    try:
        ref = _planned(snap, queries, K, boosts)
        assert snap.last_stats()["device_planned"] == 0
    finally:
        _opt(b"PS_DEVICE_PLAN", 1)
    for q, g, r in zip(queries, got, ref):
        assert g == r, (q, K, g[:4], r[:4])
    for q, g in list(zip(queries, got))[:24]:
        exp = [(k, bits(s)) for k, s in o.query(q, orc.zero_to_one(), boosts)[:K]]
        assert g == exp, (q, K, g[:4], exp[:4])


@pytest.mark.parametrize("fields", [1, 2, 3])
@pytest.mark.parametrize("K", [1, 10, 64])
def test_device_planned_ties_everywhere(fields, K):
    words, docs = _tie_corpus(30_000, fields, seed=fields * 10 + K)
    p, o = _build(docs, fields)
    snap = p.snapshot(device=0, tile_docs=256)
    rng = random.Random(K)
    queries = [" ".join(rng.choice(words[:8]) for _ in range(rng.randint(1, 4))) for _ in range(40)]
    queries += ["w00 w00", "w01 w01 w01", "w00 w01 w00", "w02  w03", "zzz w00", "w00", "", " ", "w00 w06 w06 w00"]
    _opt(b"PS_DAAT_CHUNK", 256)
    _check_planned(snap, o, queries, K, fields)


def test_device_planned_prefix_expansions_and_hand_back():
    """Expansions within 4 lists stay on the device planner; a query that is not simple ("wab wab": two expansions of a
    term AND a term under two records), or one with more than 4 lists, sends the batch back to the host planner - same
    answers either way."""
    words, docs = _tie_corpus(20_000, 2, seed=5)
    p, o = _build(docs, 2)
    snap = p.snapshot(device=0, tile_docs=256)
    queries = ["wa", "wab w00", "wa w01", "wabc wabc", "wabc w00 wabc", "w00 wab", "wa w00"] * 2
    _opt(b"PS_DAAT_CHUNK", 256)
    for K in (1, 10):
        _check_planned(snap, o, queries, K, 2)
    _check_planned(snap, o, ["wab wab"] * 8 + ["w00"], 10, 2, device_planned=0)
    _check_planned(snap, o, ["w"] * 4 + ["w00 w01"] * 8, 10, 2, device_planned=0)
    _check_planned(snap, o, ["w00 w01"] * 4, 10, 2, device_planned=0)  # (fewer than PS_DAAT_MIN_BATCH queries)
    # and the device planner is taken again afterwards (the abandoned contexts went back into the rotation)
    _check_planned(snap, o, queries, 10, 2)


def test_device_planned_after_a_delta():
    """New documents, new terms and removals through ps_snapshot_update: a term with a delta layer has two lists - the batch
    goes to the host planner; terms without one keep the device planner."""
    words, docs = _tie_corpus(20_000, 2, seed=9)
    p, o = _build(docs, 2)
    snap = p.snapshot(device=0, tile_docs=256, headroom_pct=10)
    rng = random.Random(3)
    for k in rng.sample(range(20_000), 300):
        p.remove_document(k * 3 + 1)
        o.remove_document(k * 3 + 1)
    for i in range(40):
        vals = ["fresh%d w39" % (i % 3), "w38 fresh%d" % (i % 2)]
        p.add_field_values(100_000 + i, vals)
        o.add_document(100_000 + i, vals)
    snap.update()
    _opt(b"PS_DAAT_CHUNK", 256)
    untouched = [" ".join(rng.choice(words[:8]) for _ in range(3)) for _ in range(32)]
    _check_planned(snap, o, untouched, 10, 2)
    _check_planned(snap, o, untouched[:12] + ["w39 w00", "fresh1", "w38"], 10, 2, device_planned=0)


def test_device_planned_c3_shape_and_batches_in_flight():
    """C3's shape at a size the oracle finishes; then six different batches back to back on one caller stream, announced
    ahead (ps_snapshot_plan_ahead_flat), every block compared with the synchronous answer."""
    import ctypes as C
    from probly_search_amd import dist as psd
    cfg = dict(synth.CONFIGS["C3"], n_docs=120_000, vocab=8_000)
    corpus = synth.Corpus(**cfg)
    p, o = synth.fill(psa.Index(2), corpus), synth.fill(orc.Index(2), corpus)
    snap = p.snapshot(device=0)
    _check_planned(snap, o, corpus.queries(256, 3), 10, 2)
    sc, K, B = psa.zero_to_one.new(), 10, 128
    batches = [corpus.queries(B, 3, salt=s) for s in range(6)]
    want = [_topk(snap, b, K, [1.0, 1.0]) for b in batches]
    hip = psd._DeviceBuffer.hip()
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    st = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0
    bufs = [psd._DeviceBuffer(psd.block_bytes(B, K)) for _ in batches]
    packed = [synth.pack_queries(b) for b in batches]
    for i, ((text, offsets), buf) in enumerate(zip(packed, bufs)):
        snap.query_batch_allgather_flat(None, text, offsets, sc, [1.0, 1.0], K, buf.ptr.value, buf.ptr.value, stream=st.value)
        assert snap.last_stats()["device_planned"] == 1
        if i + 1 < len(packed):
            assert snap.plan_ahead_flat(packed[i + 1][0], packed[i + 1][1], sc, [1.0, 1.0])
    assert hip.hipStreamSynchronize(st) == 0
    for i, buf in enumerate(bufs):
        got = psd.unpack_blocks(buf.to_host(), 1, B, K, [B])
        assert [[(k, bits(s)) for k, s in rs] for rs in got] == want[i], i
    hip.hipStreamDestroy(st)


# ---- mixed batches: the queries K1dz takes go there, the others through the streaming kernels (KParams::out_row) ----

@pytest.mark.parametrize("planner", [1, 0])
def test_mixed_batches_are_split(planner):
    """A zero_to_one batch in which some queries are not for K1dz (not simple: "wab wab"; more than 4 lists: "w", five
    terms) is split - every query's row must hold its own answer, whichever kernel wrote it; device planner (hands the
    batch back) and host planner alike; with PS_DAAT_Z_SPLIT=0 the whole batch takes the streaming kernels: same answers."""
    words, docs = _tie_corpus(20_000, 2, seed=11)
    p, o = _build(docs, 2)
    snap = p.snapshot(device=0, tile_docs=256)
    rng = random.Random(7)
    simple = [" ".join(rng.choice(words[:8]) for _ in range(rng.randint(1, 4))) for _ in range(40)]
    other = ["wab wab", "w", "w00 w01 w02 w03 w04", "wa wab", "w0 w1", "wab wab w00"]
    queries = []
    for i, q in enumerate(simple):
        queries.append(q)
        if i % 5 == 2:
            queries.append(other[(i // 5) % len(other)])
    queries += ["", "w00"]
    _opt(b"PS_DAAT_CHUNK", 256)
    _opt(b"PS_DEVICE_PLAN", planner)
    try:
        for K in (1, 10, 64):
            boosts = [1.0, 1.0]
            snap.work_counters(reset=True)
            got = _planned(snap, queries, K, boosts)
            assert snap.last_stats()["device_planned"] == 0  # (handed back to the host planner, which splits it)
            assert snap.work_counters(reset=True)["z_postings_scanned"] > 0  # (K1dz took its part)
            for _ in range(2):
                assert _planned(snap, queries, K, boosts) == got
            _opt(b"PS_DAAT_Z_SPLIT", 0)
            snap.work_counters(reset=True)
            whole = _planned(snap, queries, K, boosts)
            assert snap.kernel_breakdown()["score_kernel"].startswith(("ps::k_score", "ps::k_z21"))
            assert snap.work_counters(reset=True)["z_postings_scanned"] == 0
            _opt(b"PS_DAAT_Z_SPLIT", 1)
            for q, g, w in zip(queries, got, whole):
                assert g == w, (q, K, g[:3], w[:3])
            for q, g in zip(queries, got):
                exp = [(k, bits(s)) for k, s in o.query(q, orc.zero_to_one(), boosts)[:K]]
                assert g == exp, (q, K, g[:3], exp[:3])
        # the host-result entry point (ps_snapshot_query_batch) takes the same path
        assert _topk(snap, queries, 10, [1.0, 1.0]) == _planned(snap, queries, 10, [1.0, 1.0])
        # fewer than PS_DAAT_MIN_BATCH queries for K1dz: not split
        few = ["wab wab"] * 6 + ["w00 w01"] * 3
        assert _planned(snap, few, 10, [1.0, 1.0]) == [[(k, bits(s)) for k, s in o.query(q, orc.zero_to_one(), [1.0, 1.0])[:10]] for q in few]
    finally:
        _opt(b"PS_DEVICE_PLAN", 1)
        _opt(b"PS_DAAT_Z_SPLIT", 1)


def test_mixed_batches_in_flight():
    """Split batches back to back on one caller stream, pure batches between them."""
    import ctypes as C
    from probly_search_amd import dist as psd
    words, docs = _tie_corpus(30_000, 2, seed=12)
    p, o = _build(docs, 2)
    snap = p.snapshot(device=0, tile_docs=256)
    sc, K, B = psa.zero_to_one.new(), 10, 64
    rng = random.Random(5)
    batches = []
    for s in range(6):
        qs = [" ".join(rng.choice(words[:10]) for _ in range(rng.randint(1, 4))) for _ in range(B)]
        if s % 2 == 0:
            for i in range(3, B, 7):
                qs[i] = ["wab wab", "w", "w00 w01 w02 w03 w04"][i % 3]
        batches.append(qs)
    _opt(b"PS_DAAT_Z_SPLIT", 0)
    want = [_topk(snap, b, K, [1.0, 1.0]) for b in batches]
    _opt(b"PS_DAAT_Z_SPLIT", 1)
    hip = psd._DeviceBuffer.hip()
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    st = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0
    bufs = [psd._DeviceBuffer(psd.block_bytes(B, K)) for _ in batches]
    for b, buf in zip(batches, bufs):
        text, offsets = synth.pack_queries(b)
        snap.query_batch_allgather_flat(None, text, offsets, sc, [1.0, 1.0], K, buf.ptr.value, buf.ptr.value, stream=st.value)
    assert hip.hipStreamSynchronize(st) == 0
    for i, buf in enumerate(bufs):
        got = psd.unpack_blocks(buf.to_host(), 1, B, K, [B])
        assert [[(k, bits(s)) for k, s in rs] for rs in got] == want[i], i
    hip.hipStreamDestroy(st)
