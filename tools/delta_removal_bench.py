#!/usr/bin/env python3
"""Scoring-kernel time of C2 batches on a snapshot that carries delta removals (alive bitmap consulted per posting)
next to the same snapshot before the removals.  A/B two library builds with PS_SO=<path>.
usage: python tools/delta_removal_bench.py [--steps 30] [--remove-every 100]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (device buffers, streams)

import probly_search_amd as psa  # noqa: E402
from probly_search_amd import dist as psd, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--remove-every", type=int, default=100)
args = ap.parse_args()
os.environ["PS_ROW_CACHE_MB"] = "0"
cfg = dict(synth.CONFIGS["C2"])
corpus = synth.Corpus(**cfg)
F, K, B = cfg["fields"], cfg["top_k"], 1024
idx = synth.fill(psa.Index(F), corpus)
snap = idx.snapshot(device=0, headroom_pct=10)
sc = psa.bm25.new()
packed = [synth.pack_queries(corpus.queries(B, cfg["q_terms"], salt=s)) for s in range(args.steps + 3)]
buf = torch.zeros(psd.block_bytes(B, K) // 8, dtype=torch.int64, device="cuda")
base = buf.data_ptr()
st = torch.cuda.Stream()


def run(tag):
    for i, (text, offs) in enumerate(packed):
        if i == 3:
            st.synchronize()
            snap.kernel_breakdown(reset=True)
        snap.query_batch_device_flat(text, offs, sc, [1.0] * F, K, base, base + 8 * B * K, base + 16 * B * K, stream=st.cuda_stream)
    st.synchronize()
    kt = snap.kernel_breakdown(reset=True)
    return {"leg": tag, "kernel": kt["score_kernel"], "kernel_avg_ms": kt["score_ms"] / max(1, kt["launches"]), "rows_avg_ms": kt["rows_ms"] / max(1, kt["launches"]), "launches": int(kt["launches"])}


out = [run("no removals (alive bitmap absent)")]
for k in range(0, cfg["n_docs"], args.remove_every):
    idx.remove_document(k)
up = snap.update()
leg = run("after %d removals (delta: alive bitmap)" % len(range(0, cfg["n_docs"], args.remove_every)))
leg["update"] = up
out.append(leg)
print(json.dumps({"lib": psa.lib_path(), "legs": out}))
