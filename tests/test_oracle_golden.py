"""Pins the CPU oracle (oracle/) against every known-answer test the reference holds
for the Index::query / ScoreCalculator path (tests/golden/reference_kats.json)."""
import pytest

from kat_runner import load_cases, run_case
from oracle import oracle as orc


def scorer_factory(name, **kw):
    return orc.bm25(**kw) if name == "bm25" else orc.zero_to_one()


@pytest.mark.parametrize("case", load_cases("reference_kats.json"), ids=lambda c: c["id"])
def test_oracle_reproduces_reference_kat(case):
    # force_exact: the restatement must hit the printed constants bit-for-bit even where the
    # reference itself only asserts 8 decimals (R3-R6).
    run_case(orc.Index, scorer_factory, case, force_exact=True)


def test_oracle_df_exceeds_n_lists():
    # tests/document_frequency.rs:5-32: "the," occurs 3x -> 3 pointers (one per OCCURRENCE,
    # src/index.rs:119-157), "the" once; df("the,")=3 > N=1 is clamped by min(N, df) (bm25.rs:41).
    idx = orc.Index(1)
    idx.add_document(0, ["this is text with lots of the, the, the, the"])
    assert idx.expand_term("the") == ["the", "the,"]
    assert idx.count_documents("the,") == 3 and idx.count_documents("the") == 1
    assert [p[1] for p in idx.postings("the,")] == [[3]] * 3
    res = idx.query("What did the author do growing up?", orc.bm25(), [1.0])
    assert res == [(0, 0.28768207245178085)]  # SURVEY App. B derived vector D2


def test_oracle_custom_tokenizer_and_short_boosts():
    idx = orc.Index(1)
    idx.add_document(7, ["Hello,World"], tokenizer=lambda s: s.lower().split(","))
    assert idx.expand_term("hel") == ["hello"]
    res = idx.query("WORLD", orc.bm25(), [1.0], tokenizer=lambda s: [s.lower()])
    assert [k for k, _ in res] == [7]
    with pytest.raises(IndexError):
        idx.query("world", orc.bm25(), [])


def test_flat_leg_matches_literal():
    """The second CPU-baseline leg (SwissTable-class containers + arena, bench.py `cpu_baseline.flat`) must return the
    literal restatement's results bit for bit - removals, re-adds and prefix expansions included."""
    from corpus_util import build_script, random_queries
    from adapters import replay
    for seed in (1, 2, 3, 4):
        fields, steps, vocab = build_script(seed, n_docs=200, fields=2, vocab_size=40, mutate=True)
        o = orc.Index(fields)
        replay(steps, fields, o)
        for q in random_queries(seed, vocab, n=20):
            for sc in (orc.bm25(), orc.zero_to_one(), orc.bm25(k1=0.4, b=0.3)):
                assert o.query_flat(q, sc, [1.0, 0.5]) == o.query(q, sc, [1.0, 0.5]), (seed, q, sc.kind)
    # timed entry point: per-query times and top-k of both flavours on one index
    o = orc.Index(1)
    for k in range(300):
        o.add_document(k, ["t%d u%d v%d" % (k % 7, k % 11, k % 3)])
    qs = ["t1 u2", "v0", "t3 u3 v1", "zz"]
    lit = o.bench_queries(qs, orc.bm25(), [1.0], threads=2, top_k=5)
    flat = o.bench_queries(qs, orc.bm25(), [1.0], threads=2, top_k=5, flat=True)
    assert lit[3] == flat[3] and list(lit[2]) == list(flat[2])


def test_oracle_under_address_and_ub_sanitizers(tmp_path):
    """an -fsanitize=address,undefined build of the oracle (oracle/Makefile `asan`) replays every
    reference KAT, a mutation script and both bench legs in a child process; any report fails the test."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "asan"])
    so = os.path.join(root, "oracle", "_build", "libprobly_oracle_asan.so")
    asan_rt = subprocess.check_output(["g++", "-print-file-name=libasan.so"], text=True).strip()
    code = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from oracle import oracle as orc
orc._SO = %r
orc.build = lambda force=False: orc._SO
from kat_runner import load_cases, run_case
from corpus_util import build_script, random_queries
from adapters import replay
for case in load_cases("reference_kats.json"):
    run_case(orc.Index, lambda name, **kw: orc.bm25(**kw) if name == "bm25" else orc.zero_to_one(), case, force_exact=True)
fields, steps, vocab = build_script(5, n_docs=150, fields=2, vocab_size=30, mutate=True)
o = orc.Index(fields)
replay(steps, fields, o)
qs = random_queries(5, vocab, n=25)
for q in qs:
    for sc in (orc.bm25(), orc.zero_to_one()):
        assert o.query_flat(q, sc, [1.0, 2.0]) == o.query(q, sc, [1.0, 2.0])
for flat in (False, True):
    o.bench_queries(qs, orc.bm25(), [1.0, 1.0], threads=3, top_k=4, flat=flat)
del o
print("sanitized oracle ok")
''' % (root, os.path.join(root, "tests"), so)
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=67")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "sanitized oracle ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-4000:]
