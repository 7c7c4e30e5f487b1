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
                elif k != "_pad":
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
