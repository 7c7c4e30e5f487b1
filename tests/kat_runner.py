"""Replays a golden case (tests/golden/*.json step list) against an Index implementation.

Works for both the CPU oracle (oracle/oracle.py) and the product binding
(probly_search_amd.Index): they expose the same reference-shaped surface.
"""
import json
import math
import os
import struct

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_cases(name):
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return json.load(f)["cases"]


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def canonical(res):
    """test_util::test_score's ordering: score desc, then key asc (src/lib.rs:54-58)."""
    return sorted(res, key=lambda r: (-r[1], r[0]))


def run_case(index_cls, scorer_factory, case, force_exact=False):
    """scorer_factory(name, **params) -> scorer object accepted by index.query."""
    idx = index_cls(case["fields_num"])
    for step in case["steps"]:
        if "add" in step:
            key, fields = step["add"]
            idx.add_document(key, fields)
        elif "remove" in step:
            idx.remove_document(step["remove"])
        elif "vacuum" in step:
            idx.vacuum()
        elif "query" in step:
            scorer = scorer_factory(step["scorer"], **step.get("scorer_params", {}))
            res = idx.query(step["query"], scorer, step["boosts"])
            if step.get("no_panic"):
                continue
            res = canonical(res)
            exp = step["expected"]
            assert [r[0] for r in res] == [e[0] for e in exp], (case["id"], step["query"], res, exp)
            for (k, s), (ek, es) in zip(res, exp):
                es = float(es)
                if step.get("exact", True) or force_exact:
                    assert bits(s) == bits(es), (case["id"], step["query"], k, s.hex(), es.hex())
                else:  # the reference asserts 8 decimals for these rows
                    assert abs(s - es) < 1e-8, (case["id"], step["query"], k, s, es)
        elif "expand" in step:
            assert idx.expand_term(step["expand"]) == step["expected"], case["id"]
        elif "fields" in step:
            for i, (esum, eavg) in enumerate(step["fields"]):
                s, a = idx.field_details(i)
                assert s == esum, (case["id"], i, s, esum)
                if eavg == "nan":
                    assert math.isnan(a), (case["id"], a)
                else:
                    assert bits(a) == bits(float(eavg)), (case["id"], a, eavg)
        elif "docs_len" in step:
            assert idx.docs_len() == step["docs_len"], case["id"]
        elif "doc_field_length" in step:
            key, fl = step["doc_field_length"]
            assert idx.doc_field_length(key) == fl, case["id"]
        elif "children" in step:
            term, chars = step["children"]
            assert idx.children(term) == chars, (case["id"], term, idx.children(term))
        elif "count_nodes" in step:
            assert idx.count_nodes() == step["count_nodes"], (case["id"], idx.count_nodes())
        elif "arena_doc_live" in step:
            assert idx.arena_doc_live() == step["arena_doc_live"], case["id"]
        else:
            raise ValueError("unknown step %r" % (step,))
    return idx
