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
