#!/usr/bin/env python3
"""How the oracle's all-cores leg scales with the thread count on this host (both container legs; profiles/r06_cpu_baseline_scaling.txt):
flat throughput from N threads on with per-query time growing linearly = a CPU quota, not contention."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probly_search_amd import synth
from oracle import oracle as orc
cfg = dict(synth.CONFIGS["C2"])
corpus = synth.Corpus(**cfg)
t0 = time.time()
o = synth.fill(orc.Index(2), corpus)
print("index built in %.1f s; MALLOC env: %s" % (time.time() - t0, {k: v for k, v in os.environ.items() if k.startswith("MALLOC")}), flush=True)
qs = corpus.queries(1024, 3, salt=5)
for flat in (False, True):
    w1, s1, _, _ = o.bench_queries(qs[:16], orc.bm25(), [1.0, 1.0], threads=1, flat=flat)
    alone = s1.mean()
    print("flat" if flat else "literal", "1 thread: %.3f s/query" % alone, flush=True)
    for th in (8, 32, 64, 128, 256):
        n = min(1024, th * 3)
        w, s, _, _ = o.bench_queries(qs[:n], orc.bm25(), [1.0, 1.0], threads=th, flat=flat)
        print("  threads %3d: %6.1f q/s, speedup %.1f, per-query slowdown %.1f (mean %.2f s)" % (th, n / w, (n / w) * alone, s.mean() / alone, s.mean()), flush=True)
