#!/usr/bin/env python3
"""Writes tests/golden/reference_kats.json.

These are the known-answer tests the reference (quantleaf/probly-search 2.0.1)
holds for the Index::query / ScoreCalculator path, transcribed as DATA (inputs
and the expected outputs its own #[test]s assert); `source` cites the reference
file:line of each.  Nothing here executes or embeds reference code.

A case is a list of steps replayed against an Index<u64>:
  {"add": [key, [field0, field1, ...]]}   field = str, or [str, ...] for a multi-valued accessor
  {"remove": key} / {"vacuum": true}
  {"query": q, "scorer": "bm25"|"zero_to_one", "boosts": [...], "expected": [[key, score], ...],
   "exact": bool}    expected is in canonical order (score desc, key asc; src/lib.rs:54-58);
                     exact=true -> the reference asserts bit-exact f64 equality (assert_eq! on
                     QueryResult, src/lib.rs:62-65); exact=false -> it asserts to 8 decimals.
  {"query": ..., "no_panic": true}        only "must not panic" is asserted
  {"expand": term, "expected": [...]}     Index::expand_term order
  {"fields": [[sum, avg|"nan"], ...]}     FieldDetails state
  {"docs_len": n} / {"doc_field_length": [key, [..]]} / {"children": [term, [chars...]]}
  {"count_nodes": n} / {"arena_doc_live": n}
"""
import json
import os

TWO_DOCS = [{"add": [1, ["a b c", "hello world"]]}, {"add": [2, ["c d e", "lorem ipsum"]]}]
INTEG = [{"add": [0, ["abc", "dfg"]]}, {"add": [1, ["dfgh", "abcd"]]}]
FIVE = ["abcdef", "abc abcdef", "abcdef abcdef", "abcdef abcdefghi", "def abcdef"]


def one_field(titles):  # test_util::build_test_index, src/lib.rs:72-83
    return [{"add": [i, [t]]} for i, t in enumerate(titles)]


def q(query, scorer, boosts, expected, exact=True):
    return {"query": query, "scorer": scorer, "boosts": boosts, "expected": expected, "exact": exact}


CASES = [
    {"id": "R1", "source": "src/score/default/bm25.rs:105-116", "fields_num": 1,
     "steps": one_field(["a b c", "c d e"]) + [q("a", "bm25", [1.0], [[0, 0.6931471805599453]])]},
    {"id": "R2", "source": "src/score/default/bm25.rs:119-136", "fields_num": 1,
     "steps": one_field(["a b c", "c d e"]) +
     [q("c", "bm25", [1.0], [[0, 0.1823215567939546], [1, 0.1823215567939546]])]},
    {"id": "R3", "source": "src/query.rs:182-211", "fields_num": 2,
     "steps": TWO_DOCS + [q("a", "bm25", [1.0, 1.0], [[1, 0.6931471805599453]], exact=False)]},
    {"id": "R4", "source": "src/query.rs:214-258", "fields_num": 2,
     "steps": TWO_DOCS + [q("c", "bm25", [1.0, 1.0], [[1, 0.1823215567939546], [2, 0.1823215567939546]],
                            exact=False)]},
    {"id": "R5", "source": "src/query.rs:261-292", "fields_num": 2,
     "steps": TWO_DOCS + [q("h", "bm25", [1.0, 1.0], [[1, 0.12637567304702957]], exact=False)]},
    {"id": "R6", "source": "src/query.rs:295-338", "fields_num": 2,
     "steps": TWO_DOCS + [q("a d", "bm25", [1.0, 1.0], [[1, 0.6931471805599453], [2, 0.6931471805599453]],
                            exact=False)]},
    {"id": "R7_R8", "source": "tests/integrations_tests.rs:28-93 (= README.md:99-141)", "fields_num": 2,
     "steps": INTEG + [
         q("abc", "bm25", [1.0, 1.0], [[0, 0.6931471805599453], [1, 0.28104699650060755]]),
         {"remove": 0}, {"vacuum": True},
         q("abc", "bm25", [1.0, 1.0], [[1, 0.1166450426074421]])]},
    {"id": "R9_R10", "source": "tests/integrations_tests.rs:96-149", "fields_num": 2,
     "steps": INTEG + [
         q("abc", "zero_to_one", [1.0, 1.0], [[0, 1.0], [1, 0.75]]),
         {"remove": 0},  # NO vacuum: lazily-deleted postings must be filtered
         q("abc", "zero_to_one", [1.0, 1.0], [[1, 0.75]])]},
    {"id": "R11", "source": "src/score/default/zero_to_one.rs:139-157", "fields_num": 1,
     "steps": one_field(["abc", "abcefg", "abcefghij"]) +
     [q("abc", "zero_to_one", [1.0], [[0, 1.0], [1, 0.5], [2, 0.33333333333333337]])]},
    {"id": "R12", "source": "src/score/default/zero_to_one.rs:160-171", "fields_num": 1,
     "steps": one_field(["abcdef abcdefghi"]) + [q("abc abc", "zero_to_one", [1.0], [[0, 0.4166666666666667]])]},
    {"id": "R13", "source": "src/score/default/zero_to_one.rs:174-182", "fields_num": 1,
     "steps": one_field(["abc"]) + [q("abc abc", "zero_to_one", [1.0], [[0, 0.5]])]},
    {"id": "R14", "source": "src/score/default/zero_to_one.rs:185-192", "fields_num": 1,
     "steps": one_field(["abc abc"]) + [q("abc", "zero_to_one", [1.0], [[0, 0.5]])]},
    {"id": "R15", "source": "src/score/default/zero_to_one.rs:195-206", "fields_num": 1,
     "steps": one_field(["abc abc"]) + [q("abc ab", "zero_to_one", [1.0], [[0, 0.8333333333333334]])]},
    {"id": "R16", "source": "src/score/default/zero_to_one.rs:209-217", "fields_num": 1,
     "steps": one_field(["abc ab"]) + [q("abc abc", "zero_to_one", [1.0], [[0, 0.5]])]},
    {"id": "R17", "source": "src/score/default/zero_to_one.rs:220-231", "fields_num": 1,
     "steps": one_field(["oy oy oysters"]) + [q("oy oy oysters", "zero_to_one", [1.0], [[0, 1.0]])]},
    {"id": "R18", "source": "src/score/default/zero_to_one.rs:234-267", "fields_num": 1,
     "steps": one_field(FIVE) +
     [q("abc", "zero_to_one", [1.0], [[0, 0.5], [1, 0.5], [2, 0.25], [3, 0.25], [4, 0.25]])]},
    {"id": "R19", "source": "src/score/default/zero_to_one.rs:270-306", "fields_num": 1,
     "steps": one_field(FIVE) +
     [q("abc abc", "zero_to_one", [1.0],
        [[1, 0.75], [2, 0.5], [3, 0.4166666666666667], [0, 0.25], [4, 0.25]])]},
    {"id": "R20a", "source": "src/score/default/zero_to_one.rs:309-356", "fields_num": 2,
     "steps": [{"add": [i, [t, t]]} for i, t in enumerate(["abc", "abcefg", "abcefghij"])] +
     [q("abc", "zero_to_one", [1.0, 1.0], [[0, 1.0], [1, 0.5], [2, 0.33333333333333337]])]},
    {"id": "R20b", "source": "src/score/default/zero_to_one.rs:359-404", "fields_num": 2,
     "steps": [{"add": [i, [t, "a"]]} for i, t in enumerate(["abc", "abcefg", "abcefghij"])] +
     [q("abc", "zero_to_one", [1.0, 1.0], [[0, 1.0], [1, 0.5], [2, 0.33333333333333337]])]},
    {"id": "R21a", "source": "src/query.rs:344-364", "fields_num": 2,
     "steps": [{"add": [1, ["abc", "hello world"]]}, {"add": [2, ["adef", "lorem ipsum"]]},
               {"expand": "a", "expected": ["adef", "abc"]}]},
    {"id": "R21b", "source": "src/query.rs:367-387", "fields_num": 2,
     "steps": [{"add": [1, ["abc def", "hello world"]]}, {"add": [2, ["adef abc", "lorem ipsum"]]},
               {"expand": "x", "expected": []}]},
    {"id": "R22", "source": "tests/document_frequency.rs:5-32", "fields_num": 1,
     "steps": [{"add": [0, ["this is text with lots of the, the, the, the"]]},
               {"query": "What did the author do growing up?", "scorer": "bm25", "boosts": [1.0],
                "no_panic": True}]},
    {"id": "R23a", "source": "src/index.rs:497-545", "fields_num": 1,
     "steps": [{"add": [1, ["a b c"]]}, {"docs_len": 1}, {"doc_field_length": [1, [3]]},
               {"fields": [[3, 3.0]]}, {"children": ["", ["c", "b", "a"]]}, {"children": ["c", []]}]},
    {"id": "R23b", "source": "src/index.rs:548-604", "fields_num": 1,
     "steps": [{"add": [1, ["a b c"]]}, {"add": [2, ["b c d"]]}, {"docs_len": 2},
               {"doc_field_length": [1, [3]]}, {"doc_field_length": [2, [3]]},
               {"fields": [[6, 3.0]]}, {"children": ["", ["d", "c", "b", "a"]]}]},
    {"id": "R23c", "source": "src/index.rs:607-617", "fields_num": 1,
     "steps": [{"add": [1, ["a  b"]]}, {"docs_len": 1}, {"doc_field_length": [1, [2]]}]},
    {"id": "R24", "source": "src/index.rs:624-658", "fields_num": 1,
     "steps": [{"arena_doc_live": 0}, {"add": [1, ["a"]]}, {"remove": 1}, {"vacuum": True},
               {"docs_len": 0}, {"fields": [[0, "nan"]]}, {"children": ["", []]},
               {"arena_doc_live": 0}, {"count_nodes": 1}]},
    {"id": "R25a", "source": "src/index.rs:739-762", "fields_num": 1,
     "steps": [{"add": [1, ["abc"]]}, {"add": [1, ["abe"]]}, {"count_nodes": 5}]},
    {"id": "R25b", "source": "src/index.rs:765-782", "fields_num": 1,
     "steps": [{"add": [1, ["ab cd"]]}, {"add": [1, ["ab ef"]]}, {"count_nodes": 7}]},
    {"id": "R25c", "source": "src/index.rs:785-789", "fields_num": 1, "steps": [{"count_nodes": 1}]},
]

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
    with open(out, "w") as f:
        json.dump({"_about": "Known-answer tests held by the reference's own #[test]s for the "
                             "Index::query / ScoreCalculator path; see make_reference_kats.py",
                   "cases": CASES}, f, indent=1)
    print("wrote", out, len(CASES), "cases")
