#!/usr/bin/env python3
"""Where a full-result (top_k = 0) batch spends its time: C2, 24 queries, library stats + wall clock."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import probly_search_amd as psa
from probly_search_amd import synth
cfg = dict(synth.CONFIGS["C2"])
corpus = synth.Corpus(**cfg)
snap = synth.fill(psa.Index(2), corpus).snapshot(device=0)
qs = corpus.queries(24, 3)
sc = psa.bm25.new()
for i in range(3):
    t0 = time.perf_counter()
    k, s, o = snap.query_batch_arrays(qs, sc, None, [1.0, 1.0], 0)
    t = time.perf_counter() - t0
    st = snap.last_stats()
    print("call %d: wall %.1f ms, library total %.1f ms (plan %.2f, kernels %.2f), results %d" % (i, t * 1e3, st["total_ms"], st["plan_ms"], st["kernel_ms"], o[-1]))
