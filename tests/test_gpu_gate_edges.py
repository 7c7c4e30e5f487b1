"""Edges of the gates in front of the pruning kernels (ps_engine.hip: daat_eligible / bm25_params_sane): K1d takes BM25
top-k batches with k1 >= 0, 0 <= b <= 1, positive finite boosts; everything else must fall through to K1 k_score and
still equal the oracle bit for bit (src/score/default/bm25.rs:60-93, src/query.rs:150-164).  The admitted corners
are the hard ones for exact pruning: k1 = 0 makes every saturated term frequency exactly 1, so whole lists tie and
the result order is decided by the key-ascending tie-break alone (src/lib.rs:54-58)."""
import math
import random

import pytest

import probly_search_amd as psa
from adapters import oracle_scorer, product_scorer
from emu import bits
from oracle import oracle as orc
from probly_search_amd import synth

pytestmark = pytest.mark.gpu

N_DOCS = 50_000


@pytest.fixture(scope="module")
def corpus_pair():
    cfg = dict(synth.CONFIGS["C2"], n_docs=N_DOCS, vocab=2_000)
    corpus = synth.Corpus(**cfg)
    p, o = synth.fill(psa.Index(2), corpus), synth.fill(orc.Index(2), corpus)
    snap = p.snapshot(device=0, headroom_pct=10)
    return corpus, p, o, snap


def _run(snap, o, queries, kw, boosts, K, want_kernel):
    sc = product_scorer("bm25", **kw)
    got = snap.query_batch(queries, sc, None, boosts, top_k=K)
    name = snap.kernel_breakdown()["score_kernel"]
    assert name.startswith(want_kernel), (name, kw, boosts)
    osc = oracle_scorer("bm25", **kw)
    for q, g in list(zip(queries, got))[:10]:
        exp = o.query(q, osc, boosts)[:K]
        assert [(r.key, bits(r.score)) for r in g] == [(k, bits(s)) for k, s in exp], (q, kw, boosts, K, [tuple(r) for r in g][:3], exp[:3])
    return got


ADMITTED = [
    ({"k1": 0.0}, [1.0, 1.0]),            # every tfn == 1: all documents of a list tie
    ({"k1": 0.0, "b": 0.0}, [1.0, 1.0]),
    ({"b": 0.0}, [1.0, 1.0]),             # no length normalisation
    ({"b": 1.0}, [1.0, 1.0]),
    ({"k1": 1e-300}, [1.0, 1.0]),         # k1 * (...) underflows towards 0 without being 0
    ({"k1": 1e300}, [1.0, 1.0]),
    ({}, [5e-324, 1.0]),                  # subnormal boost: products underflow to 0 -> score() returns None for that field
    ({}, [1e300, 1e-300]),
    ({}, [1e300, 1e300]),
    ({}, [3e-322, 2e-322]),               # both subnormal: every product is a subnormal or 0, a rounding is an absolute 2^-1074
    ({}, [2.5e-308, 1e-310]),             # at the edge of the normal range
]
REJECTED = [
    ({"k1": -0.5}, [1.0, 1.0]),           # negative k1: tfn no longer monotone
    ({"b": 1.5}, [1.0, 1.0]),
    ({"b": -0.25}, [1.0, 1.0]),
    ({}, [math.inf, 1.0]),                # +inf boost: inf scores, inf - inf free but no finite bound
    ({}, [0.0, 1.0]),
    ({}, [-1.0, 1.0]),
    ({"k1": math.inf}, [1.0, 1.0]),
]


@pytest.mark.parametrize("K", [1, 10, 64])
@pytest.mark.parametrize("case", range(len(ADMITTED)))
def test_admitted_corners_run_on_k1d_and_match_the_oracle(corpus_pair, case, K):
    corpus, p, o, snap = corpus_pair
    kw, boosts = ADMITTED[case]
    queries = corpus.queries(16, 3, salt=case) + ["", "zzzz"]
    _run(snap, o, queries, kw, boosts, K, "ps::k_daat")


@pytest.mark.parametrize("case", range(len(REJECTED)))
def test_rejected_parameters_fall_through_to_k_score(corpus_pair, case):
    corpus, p, o, snap = corpus_pair
    kw, boosts = REJECTED[case]
    queries = corpus.queries(16, 3, salt=100 + case)
    _run(snap, o, queries, kw, boosts, 10, "ps::k_score")


def test_all_ties_whole_batch_equals_the_streaming_kernel(corpus_pair):
    """k1 = 0: the K-th best score is shared by thousands of documents; the pruned kernel and the streaming kernel must
    return the same keys in the same order for every query of a larger batch."""
    corpus, p, o, snap = corpus_pair
    queries = corpus.queries(256, 3, salt=7)
    sc = product_scorer("bm25", k1=0.0)
    L = psa.load()
    for K in (1, 10, 64):
        a = snap.query_batch(queries, sc, None, [1.0, 1.0], top_k=K)
        assert snap.kernel_breakdown()["score_kernel"].startswith("ps::k_daat")
        L.ps_set_option(b"PS_DAAT", 0)
        try:
            b = snap.query_batch(queries, sc, None, [1.0, 1.0], top_k=K)
            assert snap.kernel_breakdown()["score_kernel"].startswith("ps::k_score")
        finally:
            L.ps_set_option(b"PS_DAAT", 1)
        assert [[(r.key, bits(r.score)) for r in rs] for rs in a] == [[(r.key, bits(r.score)) for r in rs] for rs in b], K


def test_k_daat_small_is_instantiated_for_the_batchs_longest_plan(corpus_pair):
    """Batches whose queries have at most three lists run k_daat_small<F, WC, 3> (two other lists of per-query state), one four-term
    query moves the batch to <F, WC, 4>; PS_DAAT_SMALL_NL=0 always takes the four-list instantiation.  Same bits every way, and the
    oracle's."""
    corpus, p, o, snap = corpus_pair
    three = corpus.queries(48, 3, salt=41) + corpus.queries(8, 2, salt=42) + corpus.queries(8, 1, salt=43) + ["", "zzzz"]
    four = three[:40] + corpus.queries(4, 4, salt=44) + three[40:]
    sc, osc = product_scorer("bm25"), oracle_scorer("bm25")
    L = psa.load()

    def run(queries):
        return [[(r.key, bits(r.score)) for r in rs] for rs in snap.query_batch(queries, sc, None, [1.0, 1.0], top_k=10)]

    a = run(three)
    assert snap.kernel_breakdown()["score_kernel"].startswith("ps::k_daat_small<2, ") and snap.kernel_breakdown()["score_kernel"].endswith(", 3>")
    b = run(four)
    assert snap.kernel_breakdown()["score_kernel"].endswith(", 4>")
    L.ps_set_option(b"PS_DAAT_SMALL_NL", 0)
    try:
        assert run(three) == a
        assert snap.kernel_breakdown()["score_kernel"].endswith(", 4>")
    finally:
        L.ps_set_option(b"PS_DAAT_SMALL_NL", 1)
    for queries, got in ((three, a), (four, b)):
        for q, g in zip(queries, got):
            exp = [(k, bits(s_)) for k, s_ in o.query(q, osc, [1.0, 1.0])[:10]]
            assert g == exp, (q, g[:3], exp[:3])


BOOST_VECTORS = [[1.0, 1.0], [2.0, 0.5], [0.25, 3.0], [1.0, 1e-3], [1e-3, 1.0], [7.0, 7.0], [1.0, 0.0985], [3.0, 2.9], [1e6, 1.0], [0.3, 0.3000001],
                 [1.0, 1.0]]


def test_fresh_fields_boost_every_batch(corpus_pair):
    """fields_boost is a per-call argument (src/query.rs:26).  The score plane holds tfn * idf (boost-free) and the two-field
    joint bound comes from direction supports, so a boost vector never seen before costs no pass over the postings
    (`bounds_recomputed` stays 0 after the first batch) - and prunes exactly: every batch equals the streaming kernel, and
    the oracle on a sample; vectors on a stored direction ([1, 1]), between two, and at the ends of the cone."""
    corpus, p, o, snap = corpus_pair
    queries = corpus.queries(192, 3, salt=21) + ["", "zzzz"]
    sc, osc = product_scorer("bm25"), oracle_scorer("bm25")
    L = psa.load()
    snap.query_batch(queries, sc, None, [1.0, 1.0], top_k=10)  # (whatever pass the scorer parameters need happens here)
    for i, bs in enumerate(BOOST_VECTORS):
        a = snap.query_batch(queries, sc, None, bs, top_k=10)
        assert snap.kernel_breakdown()["score_kernel"].startswith("ps::k_daat"), bs
        assert snap.last_stats()["bounds_recomputed"] == 0, (bs, "a new boost vector must not trigger a pass over the postings")
        L.ps_set_option(b"PS_DAAT", 0)
        try:
            b = snap.query_batch(queries, sc, None, bs, top_k=10)
            assert snap.kernel_breakdown()["score_kernel"].startswith("ps::k_score")
        finally:
            L.ps_set_option(b"PS_DAAT", 1)
        assert [[(r.key, bits(r.score)) for r in rs] for rs in a] == [[(r.key, bits(r.score)) for r in rs] for rs in b], bs
        for q, g in list(zip(queries, a))[:6]:
            exp = o.query(q, osc, bs)[:10]
            assert [(r.key, bits(r.score)) for r in g] == [(k, bits(s)) for k, s in exp], (q, bs)


def test_fresh_fields_boost_three_fields():
    """Three fields: the joint bound of a new boost vector is one pass over the packed words (no plane rewrite, nobody
    drained); the three most recent vectors stay resident."""
    cfg = dict(synth.CONFIGS["C2"], n_docs=20_000, vocab=1_500, fields=3)
    corpus = synth.Corpus(**cfg)
    p, o = synth.fill(psa.Index(3), corpus), synth.fill(orc.Index(3), corpus)
    snap = p.snapshot(device=0)
    queries = corpus.queries(64, 3, salt=5)
    sc, osc = product_scorer("bm25"), oracle_scorer("bm25")
    seen = []
    for bs in ([1.0, 1.0, 1.0], [2.0, 0.5, 1.0], [0.1, 3.0, 9.0], [1.0, 1.0, 1.0], [5.0, 5.0, 0.01], [2.0, 0.5, 1.0]):
        a = snap.query_batch(queries, sc, None, bs, top_k=10)
        assert snap.kernel_breakdown()["score_kernel"].startswith("ps::k_daat"), bs
        assert snap.last_stats()["bounds_recomputed"] == (0 if bs in seen[-3:] else 1), (bs, seen)
        if bs in seen:
            seen.remove(bs)
        seen.append(bs)
        for q, g in list(zip(queries, a))[:12]:
            exp = o.query(q, osc, bs)[:10]
            assert [(r.key, bits(r.score)) for r in g] == [(k, bits(s)) for k, s in exp], (q, bs)


@pytest.mark.parametrize("device_plan", [False, True])
def test_mixed_batch_is_split_between_k_daat_small_and_k_daat(corpus_pair, device_plan):
    """src/query.rs:33-60 takes any number of terms and expansions.  k_daat_small takes queries of <= 4 lists, one per query term;
    a batch that also holds other queries - five terms, a prefix with several expansions, a term repeated - is scored by both
    kernels, each over its part of the item array (PS_DAAT_SPLIT).  Every query: == the unsplit batch (k_daat alone), == the
    streaming kernel, == the oracle."""
    from adapters import run_device_planned
    corpus, p, o, snap = corpus_pair
    base = corpus.queries(96, 3, salt=31)
    stems = [q.split(" ")[0] for q in base]
    odd = [" ".join(stems[i:i + 5]) for i in (0, 7)] + [stems[3][:3] + " " + stems[4], stems[5][:2], stems[6] + " " + stems[6], " ".join(stems[10:18])]
    # duplicate tokens and dead tokens (no such term: no entry) inside both kinds of queries - the "which kernel takes this query"
    # rule is applied by the planner's count pass, by k_plan and by k_prep_query, and the launch grids are sized from the first
    # (if they ever disagree the preparation now raises the engine's fault word and the next call fails loudly)
    odd += [" ".join(stems[20:24] + [stems[20]]), stems[25] + " qqqq " + stems[26], "qqqq " + " ".join(stems[27:32]),
            stems[33] + " " + stems[33] + " " + stems[34], " ".join(stems[35:39]) + " qqqq"]
    queries = base[:40] + odd[:3] + base[40:] + odd[3:] + ["", "zzzz"]
    sc, osc = product_scorer("bm25"), oracle_scorer("bm25")
    plans = [snap.plan(q, sc)[0] for q in queries]
    assert max(len(e) for e in plans) > 4 and any(len(e) > len(q.split()) for e, q in zip(plans, queries) if q)
    L = psa.load()

    def run():
        if device_plan:
            return [[(k, bits(s_)) for k, s_ in rs] for rs in run_device_planned(snap, queries, [1.0, 1.0], 10)]
        return [[(r.key, bits(r.score)) for r in rs] for rs in snap.query_batch(queries, sc, None, [1.0, 1.0], top_k=10)]

    L.ps_set_option(b"PS_DEVICE_PLAN", 1 if device_plan else 0)
    try:
        got = run()
        name = snap.kernel_breakdown()["score_kernel"]
        assert name.startswith("ps::k_daat_small") and " + ps::k_daat<" in name, name
        for _ in range(2):
            assert run() == got
        L.ps_set_option(b"PS_DAAT_SPLIT", 0)
        whole = run()
        name = snap.kernel_breakdown()["score_kernel"]
        assert name.startswith("ps::k_daat<"), name
        assert whole == got
        L.ps_set_option(b"PS_DAAT", 0)
        assert run() == got
    finally:
        L.ps_set_option(b"PS_DAAT", 1)
        L.ps_set_option(b"PS_DAAT_SPLIT", 1)
        L.ps_set_option(b"PS_DEVICE_PLAN", 1)
    for q, g in zip(queries, got):
        exp = [(k, bits(s_)) for k, s_ in o.query(q, osc, [1.0, 1.0])[:10]]
        assert g == exp, (q, g[:3], exp[:3])


@pytest.mark.parametrize("K", [1, 3, 10, 17, 64])
def test_threshold_priming_changes_work_not_results(corpus_pair, K):
    """PS_DAAT_PRIME: a query's threshold starts at the K-th best posting score of its best single list and lists that are
    non-essential under it get no work items (k_list_kth / k_prep_query).  Rank-safe: the batch with priming == without ==
    the streaming kernel == the oracle, for K at, between and at the end of the stored ranks, other boosts (the plane-sum
    direction), lists shorter than K, and one- to six-term queries; and priming must actually drop work items."""
    corpus, p, o, snap = corpus_pair
    base = corpus.queries(160, 3, salt=77)
    stems = [q.split(" ")[0] for q in base]
    queries = base + stems[:24] + [" ".join(stems[i:i + 2]) for i in range(24, 48, 2)] + [" ".join(stems[i:i + 6]) for i in (50, 60)] + [stems[70][:3]]
    sc, osc = product_scorer("bm25"), oracle_scorer("bm25")
    L = psa.load()

    def run(boosts):
        snap.work_counters(reset=True)
        got = [[(r.key, bits(r.score)) for r in rs] for rs in snap.query_batch(queries, sc, None, boosts, top_k=K)]
        return got, snap.work_counters(reset=True)

    try:
        for boosts in ([1.0, 1.0], [0.5, 3.0], [2.0, 0.25]):
            L.ps_set_option(b"PS_DAAT_PRIME", 1)
            primed, w1 = run(boosts)
            assert snap.kernel_breakdown()["score_kernel"].startswith("ps::k_daat")
            L.ps_set_option(b"PS_DAAT_PRIME", 0)
            plain, w0 = run(boosts)
            assert primed == plain, (K, boosts)
            assert w1["items_run"] <= w0["items_run"] and w1["postings_scanned"] <= w0["postings_scanned"], (K, w1, w0)
            if K <= 10 and boosts == [1.0, 1.0]:
                assert w1["postings_scanned"] < w0["postings_scanned"], (K, w1["postings_scanned"], w0["postings_scanned"])
            L.ps_set_option(b"PS_DAAT", 0)
            streamed, _ = run(boosts)
            L.ps_set_option(b"PS_DAAT", 1)
            assert streamed == primed, (K, boosts)
            for q, g in list(zip(queries, primed))[::9]:
                exp = [(k, bits(s_)) for k, s_ in o.query(q, osc, boosts)[:K]]
                assert g == exp, (K, boosts, q, g[:3], exp[:3])
    finally:
        L.ps_set_option(b"PS_DAAT", 1)
        L.ps_set_option(b"PS_DAAT_PRIME", 1)


def test_corners_under_a_delta_with_removals(corpus_pair):
    """(last: it mutates the module's index)  Removed documents are tombstones until the next flatten; the tie order of
    the survivors must not change."""
    corpus, p, o, snap = corpus_pair
    rng = random.Random(11)
    for k in rng.sample(range(N_DOCS), 1500):
        p.remove_document(k)
        o.remove_document(k)
    st = snap.update()
    assert st["mode"] in (1, 2), st
    queries = corpus.queries(24, 3, salt=9)
    for kw, boosts in [({"k1": 0.0}, [1.0, 1.0]), ({"b": 0.0}, [2.0, 0.5]), ({}, [1.0, 1.0])]:
        for K in (1, 10, 64):
            _run(snap, o, queries, kw, boosts, K, "ps::k_daat")
    _run(snap, o, queries, {"k1": -0.5}, [1.0, 1.0], 10, "ps::k_score")
    # threshold priming stays on under tombstones: the per-list tables skip removed documents (k_list_kth reads the alive bitmap) -
    # primed == unprimed == oracle for every query, and priming still drops work.  The best documents of a few lists are removed
    # on purpose: exactly the postings a stale table would still count on.
    sc, osc = product_scorer("bm25"), oracle_scorer("bm25")
    qs = corpus.queries(96, 3, salt=41)
    victims = set()
    for q in qs[:24]:
        for r in o.query(q, osc, [1.0, 1.0])[:10]:
            victims.add(r[0])
    for k in sorted(victims):
        p.remove_document(k)
        o.remove_document(k)
    st = snap.update()
    assert st["mode"] in (1, 2) and st["docs_removed"] == len(victims), st
    L = psa.load()
    try:
        out = {}
        for prime in (1, 0):
            L.ps_set_option(b"PS_DAAT_PRIME", prime)
            snap.work_counters(reset=True)
            out[prime] = ([[(r.key, bits(r.score)) for r in rs] for rs in snap.query_batch(qs, sc, None, [1.0, 1.0], top_k=10)],
                          snap.work_counters(reset=True))
            assert snap.kernel_breakdown()["score_kernel"].startswith("ps::k_daat")
        assert out[1][0] == out[0][0]
        assert out[1][1]["items_run"] < out[0][1]["items_run"], (out[1][1]["items_run"], out[0][1]["items_run"])
        for q, g in zip(qs, out[1][0]):
            exp = [(k, bits(s_)) for k, s_ in o.query(q, osc, [1.0, 1.0])[:10]]
            assert g == exp, (q, g[:3], exp[:3])
            assert not ({k for k, _ in g} & victims)
    finally:
        L.ps_set_option(b"PS_DAAT_PRIME", 1)


def test_doc_ordered_filters_with_clustered_lists():
    """The Bloom filters of sparse lists are laid out in document order (word = doc id >> shift, bits by hash): a list whose
    documents sit in one narrow id range loads a few filter words with all of its keys - more "maybe" answers there, never a
    wrong one.  Rare terms confined to 200-document windows, a common term everywhere, queries that pair them: the pruning
    kernels (lookups through filters -> table slots) == the streaming kernel == the oracle."""
    rng = random.Random(5)
    n = 40_000
    p, o = psa.Index(2), orc.Index(2)
    rare = ["rare%02d" % i for i in range(24)]
    window = {t: rng.randrange(0, n - 200) for t in rare}
    for k in range(n):
        f0 = ["common"] if k % 3 else ["common", "other"]
        f1 = ["filler%d" % (k % 7), "common" if k % 5 == 0 else "pad"]
        for t in rare:
            if window[t] <= k < window[t] + 200 and rng.random() < 0.8:
                (f0 if rng.random() < 0.5 else f1).append(t)
        if k % 97 == 0:
            f1.append(rng.choice(rare))  # a few stragglers outside the windows
        vals = [" ".join(f0), " ".join(f1)]
        p.add_field_values(k, vals)
        o.add_document(k, vals)
    snap = p.snapshot(device=0)
    queries = [a + " " + b for a in rare[:12] for b in rare[12:16]] + [t + " common" for t in rare] + [t + " other filler3" for t in rare[:8]]
    sc, osc = product_scorer("bm25"), oracle_scorer("bm25")
    L = psa.load()
    for K in (1, 10):
        a = [[(r.key, bits(r.score)) for r in rs] for rs in snap.query_batch(queries, sc, None, [1.0, 1.0], top_k=K)]
        assert snap.kernel_breakdown()["score_kernel"].startswith("ps::k_daat")
        L.ps_set_option(b"PS_DAAT", 0)
        try:
            b = [[(r.key, bits(r.score)) for r in rs] for rs in snap.query_batch(queries, sc, None, [1.0, 1.0], top_k=K)]
        finally:
            L.ps_set_option(b"PS_DAAT", 1)
        assert a == b, K
        for q, g in list(zip(queries, a))[::3]:
            exp = [(k, bits(s_)) for k, s_ in o.query(q, osc, [1.0, 1.0])[:K]]
            assert g == exp, (K, q, g[:3], exp[:3])
