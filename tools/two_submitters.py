#!/usr/bin/env python3
"""One snapshot, one or two host threads submitting C2 batches (each thread its own stream and result buffers):
does a second submitter add throughput?  (`query(&self)` is re-entrant in the reference,
src/query.rs:21-27; here submissions of one snapshot take the engine's mutex for the enqueue, the GPU side keeps
three batches in flight.)  usage: python tools/two_submitters.py [--steps 200]"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import probly_search_amd as psa  # noqa: E402
from probly_search_amd import dist as psd, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
args = ap.parse_args()
os.environ["PS_ROW_CACHE_MB"] = "0"
cfg = dict(synth.CONFIGS["C2"])
corpus = synth.Corpus(**cfg)
F, K, B = cfg["fields"], cfg["top_k"], 1024
snap = synth.fill(psa.Index(F), corpus).snapshot(device=0)
sc = psa.bm25.new()
packed = [synth.pack_queries(corpus.queries(B, cfg["q_terms"], salt=s)) for s in range(32)]


def submitter(n, salt, go):
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    buf = torch.zeros(psd.block_bytes(B, K) // 8, dtype=torch.int64, device="cuda")
    base = buf.data_ptr()
    go.wait()
    for i in range(n):
        text, offs = packed[(i + salt) % len(packed)]
        snap.query_batch_device_flat(text, offs, sc, [1.0] * F, K, base, base + 8 * B * K, base + 16 * B * K, stream=st.cuda_stream)
    st.synchronize()


def run(n_threads, total):
    go = threading.Event()
    ts = [threading.Thread(target=submitter, args=(total // n_threads, 7 * t, go)) for t in range(n_threads)]
    for t in ts:
        t.start()
    time.sleep(0.2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go.set()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return {"threads": n_threads, "batches": total, "queries_per_s": B * total / wall, "ms_per_batch": wall / total * 1e3}


run(1, 20)  # warm
out = [run(1, args.steps), run(2, args.steps), run(1, args.steps), run(2, args.steps)]
print(json.dumps({"config": "C2, 1024-query bm25 batches, top-10, device-planned, rows rebuilt per batch", "legs": out}))
