"""The reference's own integration tests (tests/integrations_tests.rs:28-149 and
tests/document_frequency.rs:5-32 of probly-search 2.0.1), rewritten line for line against the
host-side mirror: same Doc struct, field accessors, tokenizer, calls and exact expectations."""
import pytest

import probly_search_amd as psa
from probly_search_amd import Index, QueryResult
from probly_search_amd.score import bm25, zero_to_one

pytestmark = pytest.mark.gpu


class Doc:
    def __init__(self, id, title, description):
        self.id, self.title, self.description = id, title, description


def tokenizer(s):
    return s.split(" ")


def title_extract(d):
    return [d.title]


def description_extract(d):
    return [d.description]


def build():
    index = Index.new(2)
    doc_1 = Doc(0, "abc", "dfg")
    doc_2 = Doc(1, "dfgh", "abcd")
    index.add_document([title_extract, description_extract], tokenizer, doc_1.id, doc_1)
    index.add_document([title_extract, description_extract], tokenizer, doc_2.id, doc_2)
    return index, doc_1, doc_2


def test_add_query_delete_bm25():
    index, doc_1, _ = build()
    result = index.query("abc", bm25.new(), tokenizer, [1., 1.])
    assert len(result) == 2
    assert result[0] == QueryResult(0, 0.6931471805599453)
    assert result[1] == QueryResult(1, 0.28104699650060755)
    index.remove_document(doc_1.id)
    index.vacuum()
    result = index.query("abc", bm25.new(), tokenizer, [1., 1.])
    assert len(result) == 1
    assert result[0] == QueryResult(1, 0.1166450426074421)


def test_add_query_delete_zero_to_one():
    index, doc_1, _ = build()
    result = index.query("abc", zero_to_one.new(), tokenizer, [1., 1.])
    assert len(result) == 2
    assert result[0] == QueryResult(0, 1.)
    assert result[1] == QueryResult(1, 0.75)
    index.remove_document(doc_1.id)
    result = index.query("abc", zero_to_one.new(), tokenizer, [1., 1.])
    assert len(result) == 1
    assert result[0] == QueryResult(1, 0.75)


def test_should_not_panic_when_document_frequency_gt_documents_len():
    index = Index.new(1)
    doc = Doc(0, "this is text with lots of the, the, the, the", "")
    index.add_document([title_extract], tokenizer, doc.id, doc)
    index.query("What did the author do growing up?", bm25.new(), tokenizer, [1.])


def test_mutable_bm25_parameters_and_short_boosts_panic():
    index, _, _ = build()
    doc_3 = Doc(2, "abc abc xyz", "abcd q r s t")  # makes field lengths differ from the averages
    index.add_document([title_extract, description_extract], tokenizer, doc_3.id, doc_3)
    s = bm25.new()
    s.bm25k1, s.bm25b = 2.0, 0.5  # BM25's fields are public (bm25.rs:14-20)
    a = index.query("abc", s, tokenizer, [1., 1.])
    b = index.query("abc", bm25.new(), tokenizer, [1., 1.])
    assert sorted(r.key for r in a) == sorted(r.key for r in b) == [0, 1, 2]
    assert {r.key: r.score for r in a} != {r.key: r.score for r in b}
    with pytest.raises(IndexError):  # fields_boost[x] out of bounds (bm25.rs:85)
        index.query("abc", bm25.new(), tokenizer, [1.])
    with pytest.raises(TypeError):   # custom ScoreCalculator callbacks cannot cross to the device
        index.query("abc", object(), tokenizer, [1., 1.])
