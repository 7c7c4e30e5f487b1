#!/usr/bin/env python3
"""Latency of the reference-shaped call (every match, sorted) on C2: Index::query semantics."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import probly_search_amd as psa
from probly_search_amd import synth
cfg = dict(synth.CONFIGS["C2"])
c = synth.Corpus(**cfg); idx = synth.fill(psa.Index(2), c); snap = idx.snapshot(device=0)
qs = c.queries(32, 3)
for q in qs[:3]:
    snap.query(q, psa.bm25.new(), None, [1.0, 1.0])
ts, ns = [], []
for q in qs:
    t = time.perf_counter(); r = snap.query(q, psa.bm25.new(), None, [1.0, 1.0]); ts.append((time.perf_counter() - t) * 1e3); ns.append(len(r))
print("full-result single query ms (python objects incl.): p50 %.2f max %.2f; mean results %d; last engine total_ms %.3f" % (
    sorted(ts)[len(ts) // 2], max(ts), sum(ns) / len(ns), snap.last_stats()["total_ms"]))
t = time.perf_counter(); k, s, o = snap.query_batch_arrays(qs, psa.bm25.new(), None, [1.0, 1.0], 0)
print("batch of 32 full-result queries -> numpy: %.2f ms for %d results" % ((time.perf_counter() - t) * 1e3, len(k)))
