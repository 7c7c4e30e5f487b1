#!/usr/bin/env python3
"""What a batch costs when some of its queries are wider than the narrow pruning kernels take (five terms: more than 4 lists).
C3's corpus, 1024-query batches, `--other` of them five-term queries (1024: every query).  zero_to_one: K1dz's wide instantiation
(k_daat_z<F, WC, 8>, PS_DAAT_Z=1) against the streaming kernels (PS_DAAT_Z=0); BM25 (`--scorer bm25`): the batch split between
k_daat_small and k_daat (PS_DAAT_SPLIT=1) against the whole batch on k_daat (0).  One JSON line per leg."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import probly_search_amd as psa
from probly_search_amd import dist as psd, synth

ap = argparse.ArgumentParser()
ap.add_argument("--other", type=int, default=32)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--n-docs", type=int, default=0)
ap.add_argument("--scorer", default="zero_to_one")
args = ap.parse_args()
cfg = dict(synth.CONFIGS["C3"])
if args.n_docs:
    cfg["n_docs"] = args.n_docs
corpus = synth.Corpus(**cfg)
snap = synth.fill(psa.Index(2), corpus).snapshot(device=0)
bm25 = args.scorer == "bm25"
sc, K, B = (psa.bm25.new() if bm25 else psa.zero_to_one.new()), 10, 1024
knob = b"PS_DAAT_SPLIT" if bm25 else b"PS_DAAT_Z"
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
    L.ps_set_option(knob, leg)
    for w in range(3):
        snap.query_batch_allgather_flat(None, *batches[w % 4], sc, [1.0, 1.0], K, buf.ptr.value, buf.ptr.value, stream=None)
    snap.kernel_breakdown(reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        snap.query_batch_allgather_flat(None, *batches[i % 4], sc, [1.0, 1.0], K, buf.ptr.value, buf.ptr.value, stream=None)
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    kt = snap.kernel_breakdown(reset=True)
    print(json.dumps({knob.decode(): leg, "kernel": kt["score_kernel"], "other_queries": args.other, "ms_per_1024_query_batch_synchronous": round(ms, 3),
                      "scoring_kernels_ms_per_batch": round(kt["score_busy_ms"] / max(1, kt["launches"]), 4), "queries_per_s": round(B / ms * 1e3)}))
