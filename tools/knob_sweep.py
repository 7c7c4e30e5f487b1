#!/usr/bin/env python3
"""Kernel-level A/B of run-time knobs on one resident snapshot: one index build, then one leg per knob setting
(ps_set_option: engines re-read their knobs at the next batch).  Prints the scoring kernel's HIP-event average, the
whole step (wall clock around the submitted steps) and the kernels' work counters per launch.
usage: python tools/knob_sweep.py --config C3 [--steps 40] KNOB=v1,v2,... [KNOB2=...]   (legs = cross product)"""
import argparse
import itertools
import json
import os
import sys

os.environ.setdefault("PS_EXPERIMENT_KNOBS", "1")  # (ps_set_option takes the engine's experiment knobs too)
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (device buffers, streams)

import probly_search_amd as psa  # noqa: E402
from probly_search_amd import dist as psd, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--n-docs", type=int, default=0)
ap.add_argument("knobs", nargs="*")
args = ap.parse_args()
os.environ.setdefault("PS_ROW_CACHE_MB", "0")
cfg = dict(synth.CONFIGS[args.config])
if args.n_docs:
    cfg["n_docs"] = args.n_docs
corpus = synth.Corpus(**cfg)
F, K, B = cfg["fields"], cfg["top_k"], 1024
idx = synth.fill(psa.Index(F), corpus)
snap = idx.snapshot(device=0)
sc = psa.zero_to_one.new() if cfg.get("scorer") == "zero_to_one" else psa.bm25.new()
packed = [synth.pack_queries(corpus.queries(B, cfg["q_terms"], salt=s)) for s in range(args.steps + 3)]
buf = torch.zeros(psd.block_bytes(B, K) // 8, dtype=torch.int64, device="cuda")
base = buf.data_ptr()
st = torch.cuda.Stream()
L = psa.load()


def run(tag):
    t0 = 0.0
    for i, (text, offs) in enumerate(packed):
        if i == 3:
            st.synchronize()
            snap.kernel_breakdown(reset=True)
            snap.work_counters(reset=True)
            t0 = time.perf_counter()
        snap.query_batch_device_flat(text, offs, sc, [1.0] * F, K, base, base + 8 * B * K, base + 16 * B * K, stream=st.cuda_stream)
    st.synchronize()
    step_ms = (time.perf_counter() - t0) * 1e3 / args.steps
    kt = snap.kernel_breakdown(reset=True)
    w = snap.work_counters(reset=True)
    n = max(1, int(kt["launches"]))
    keep = ("items", "items_run", "postings_scanned", "postings_reached_lookups", "lookups_row", "lookups_cell", "lookups_probe", "lookup_hits",
            "offers", "bytes_touched", "rows_built", "rows_used")
    return {"leg": tag, "kernel": kt["score_kernel"], "kernel_avg_ms": round(kt["score_ms"] / n, 4), "kernel_busy_ms": round(kt["score_busy_ms"] / n, 4), "step_ms": round(step_ms, 4),
            "rows_avg_ms": round(kt["rows_ms"] / n, 4), "per_launch": {k: round(w[k] / n) for k in keep}}


names = [k.split("=")[0] for k in args.knobs]
values = [[int(v) for v in k.split("=")[1].split(",")] for k in args.knobs]
for combo in itertools.product(*values) if names else [()]:
    for n, v in zip(names, combo):
        L.ps_set_option(n.encode(), v)
    print(json.dumps(run(" ".join("%s=%d" % nv for nv in zip(names, combo)) or "defaults")), flush=True)
