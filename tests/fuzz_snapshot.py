"""Corrupted-file fuzz of ps_snapshot_load.  Not a test module itself: called by
tests/test_sanitizers.py both against the normal library and - in a child process - against the ASan / UBSan build, where
any out-of-bounds read of the validator, the planner or the CSR accessors aborts the child.

A valid snapshot file is damaged `n` times: random bytes / extreme words inside a section WITH the section checksum
re-computed (so the damage reaches the structural validation, Snapshot::validate, and - if that accepts it - the planner),
header scalars, section descriptors, truncation and padding.  Every load must either fail with PS_EINVAL or succeed; a file
that loads must then survive planning queries (trie walk, layer tables) and a CSR dump."""
import random
import struct

import probly_search_amd as psa
from adapters import ProductIndex, replay
from corpus_util import build_script, random_queries

HDR_SECS = 8 + 8 + 128  # magic | file_bytes | scalars[16]
N_SECTIONS = 16
EXTREME = (0, 1, 2, 0x7F, 0xFF, 0x100, 0xFFFF, 0x10000, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFE, 0xFFFFFFFF)


def checksum(blob):
    """The file's section checksum (ps_snapshot.cpp), restated."""
    M = (1 << 64) - 1
    h = 0x9E3779B97F4A7C15 ^ len(blob)
    n8 = len(blob) // 8 * 8
    for (w,) in struct.iter_unpack("<Q", bytes(blob[:n8])):
        h = ((h ^ w) * 0xFF51AFD7ED558CCD) & M
        h ^= h >> 29
    for b in blob[n8:]:
        h = ((h ^ b) * 0x100000001B3) & M
    return h


def fuzz(tmp_dir, n=300, seed=1):
    rng = random.Random(seed)
    F, steps, vocab = build_script(seed, n_docs=80, fields=2, vocab_size=25)
    p = ProductIndex(F)
    replay(steps, F, p)
    good_path = "%s/good_%d.snap" % (tmp_dir, seed)
    p.idx.snapshot(device=-1, tile_docs=256).save(good_path)
    good = open(good_path, "rb").read()
    queries = random_queries(seed, vocab, n=6)
    sc = psa.bm25.new()
    secs = [struct.unpack_from("<QQQ", good, HDR_SECS + 24 * i) for i in range(N_SECTIONS)]
    nonempty = [i for i in range(N_SECTIONS) if secs[i][1] >= 4]
    stats = {"rejected": 0, "accepted": 0, "accepted_and_planned": 0, "kinds": {}}
    bad_path = "%s/bad_%d.snap" % (tmp_dir, seed)
    for it in range(n):
        blob = bytearray(good)
        kind = rng.choice(("word", "word", "word", "bytes", "swap", "scalar", "secref", "truncate", "pad"))
        if kind in ("word", "bytes", "swap"):
            i = rng.choice(nonempty)
            off, nbytes, _ = secs[i]
            if kind == "word":
                for _ in range(rng.choice((1, 1, 2, 4))):
                    at = off + 4 * rng.randrange(nbytes // 4)
                    struct.pack_into("<I", blob, at, rng.choice(EXTREME + (rng.getrandbits(32), struct.unpack_from("<I", blob, at)[0] + rng.choice((-1, 1)) & 0xFFFFFFFF)))
            elif kind == "bytes":
                for _ in range(rng.randint(1, 8)):
                    blob[off + rng.randrange(nbytes)] = rng.getrandbits(8)
            else:  # two words of the section exchanged (breaks monotone tables / sorted ids without extreme values)
                a, b = off + 4 * rng.randrange(nbytes // 4), off + 4 * rng.randrange(nbytes // 4)
                blob[a:a + 4], blob[b:b + 4] = blob[b:b + 4], blob[a:a + 4]
            struct.pack_into("<Q", blob, HDR_SECS + 24 * i + 16, checksum(blob[off:off + nbytes]))
        elif kind == "scalar":
            k = rng.randrange(15)
            old = struct.unpack_from("<Q", blob, 16 + 8 * k)[0]
            struct.pack_into("<Q", blob, 16 + 8 * k, rng.choice((0, 1, old + 1, max(0, old - 1), old * 2, 1 << 32, (1 << 64) - 1, rng.getrandbits(20))))
        elif kind == "secref":
            i = rng.randrange(N_SECTIONS)
            f = rng.randrange(2)
            old = struct.unpack_from("<Q", blob, HDR_SECS + 24 * i + 8 * f)[0]
            struct.pack_into("<Q", blob, HDR_SECS + 24 * i + 8 * f, rng.choice((0, 4096, old + 4096, max(0, old - 4096), old + 4, len(blob), (1 << 63), old // 2 // 4 * 4)))
        elif kind == "truncate":
            blob = blob[:rng.choice((0, 8, 4095, 4096, len(blob) - 4096, len(blob) - 1, rng.randrange(len(blob))))]
        else:
            blob += bytes(rng.choice((1, 4096)))
        open(bad_path, "wb").write(bytes(blob))
        stats["kinds"][kind] = stats["kinds"].get(kind, 0) + 1
        try:
            snap = psa.Snapshot.load(bad_path, device=-1)
        except psa.PsError as e:
            assert e.status == 1, (kind, it, e.status, str(e))  # PS_EINVAL: refused, nothing else
            stats["rejected"] += 1
            continue
        stats["accepted"] += 1  # the damage left a structurally valid snapshot (or changed only padding / unused words)
        for q in queries:
            snap.plan(q, sc)
        snap.host_csr()
        snap.info()
        stats["accepted_and_planned"] += 1
        del snap
    return stats
