#!/usr/bin/env python3
"""p50 latency of one synchronous top-k query (ps_snapshot_query) on C1 (50k docs) and C2 (1M docs),
with the latency-path pieces toggled (PS_ZERO_COPY) for an A/B in one process; parity of every
answer against the zero-copy-off answer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import probly_search_amd as psa
from probly_search_amd import synth

for name in sys.argv[1:] or ["C1", "C2"]:
    cfg = dict(synth.CONFIGS[name])
    F = cfg["fields"]
    c = synth.Corpus(**cfg)
    idx = synth.fill(psa.Index(F), c)
    qs = c.queries(300, cfg["q_terms"])
    sc = psa.bm25.new()
    boosts = [1.0] * F
    ref = None
    for zc in ("0", "1"):
        os.environ["PS_ZERO_COPY"] = zc  # knobs are read once per engine: a fresh snapshot per mode
        snap = idx.snapshot(device=0)
        for q in qs[:20]:
            snap.query(q, sc, None, boosts, top_k=10)
        ts, out, eng = [], [], []
        for q in qs:
            t = time.perf_counter()
            r = snap.query(q, sc, None, boosts, top_k=10)
            ts.append((time.perf_counter() - t) * 1e6)
            eng.append(snap.last_stats()["total_ms"] * 1e3)
            out.append([(x.key, x.score) for x in r])
        if ref is None:
            ref = out
        print("%s zero_copy=%s: p50 %.1f us  p10 %.1f  p90 %.1f  (inside ps_snapshot_query, plan + GPU: p50 %.1f us)  identical=%s" % (
            name, zc, np.percentile(ts, 50), np.percentile(ts, 10), np.percentile(ts, 90), np.percentile(eng, 50),
            out == ref), flush=True)
