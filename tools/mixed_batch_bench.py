#!/usr/bin/env python3
"""What a zero_to_one batch costs when a few of its queries are not for K1dz (five terms: more than 4 lists): split
between K1dz and the streaming kernels (PS_DAAT_Z_SPLIT=1, default) against the whole batch on the streaming kernels (0).
C3's corpus, 1024-query batches, `--other` of them five-term queries.  One JSON line per leg."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import probly_search_amd as psa
from probly_search_amd import dist as psd, synth

ap = argparse.ArgumentParser()
ap.add_argument("--other", type=int, default=32)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--n-docs", type=int, default=0)
args = ap.parse_args()
cfg = dict(synth.CONFIGS["C3"])
if args.n_docs:
    cfg["n_docs"] = args.n_docs
corpus = synth.Corpus(**cfg)
snap = synth.fill(psa.Index(2), corpus).snapshot(device=0)
sc, K, B = psa.zero_to_one.new(), 10, 1024
batches = []
for s in range(4):
    qs = corpus.queries(B, 3, salt=s)
    five = corpus.queries(args.other, 5, salt=100 + s)
    for i, q in enumerate(five):
        qs[(i * B) // max(1, args.other)] = q
    batches.append(synth.pack_queries(qs))
buf = psd._DeviceBuffer(psd.block_bytes(B, K))
L = psa.load()
for leg in (1, 0, 1, 0):
    L.ps_set_option(b"PS_DAAT_Z_SPLIT", leg)
    for w in range(3):
        snap.query_batch_allgather_flat(None, *batches[w % 4], sc, [1.0, 1.0], K, buf.ptr.value, buf.ptr.value, stream=None)
    t0 = time.perf_counter()
    for i in range(args.steps):
        snap.query_batch_allgather_flat(None, *batches[i % 4], sc, [1.0, 1.0], K, buf.ptr.value, buf.ptr.value, stream=None)
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    print(json.dumps({"PS_DAAT_Z_SPLIT": leg, "other_queries": args.other, "ms_per_1024_query_batch_synchronous": round(ms, 3),
                      "queries_per_s": round(B / ms * 1e3)}))
