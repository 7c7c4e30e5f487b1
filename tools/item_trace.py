#!/usr/bin/env python3
"""Reads a PS_ITEM_TRACE dump (profiling build, tools/build_variant.sh trace -DPS_ITEM_TRACE): per item start / end
(100 MHz s_memrealtime), trips, rank, postings scanned / reached -> where a k_daat launch spends its time."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
run = a[:, 1] > 0
r = a[run]
t0 = r[:, 0].min()
st = (r[:, 0] - t0) / 100.0  # us
en = (r[:, 1] - t0) / 100.0
trips = (r[:, 2] & 0xFFFFFFFF).astype(np.int64)
rank = (r[:, 2] >> 32).astype(np.int64)
scanned = (r[:, 3] & 0xFFFFFFFF).astype(np.int64)
dur = en - st
print("items in launch %d, ran (not skipped whole) %d, span %.1f us" % (len(a), len(r), en.max()))
print("duration per item us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
ok = trips > 0
print("us per trip (items with trips): mean %.2f p50 %.2f p90 %.2f" % ((dur[ok] / trips[ok]).mean(), *np.percentile(dur[ok] / trips[ok], [50, 90])))
print("trips per item: mean %.1f max %d; total trips %d; wave-seconds %.0f us over %.0f us span = %.0f waves busy on average" % (
    trips.mean(), trips.max(), trips.sum(), dur.sum(), en.max(), dur.sum() / en.max()))
for rk in range(0, min(4, rank.max() + 1)):
    m = rank == rk
    if m.any():
        print("rank %d: %d items, start p50 %.0f p99 %.0f us, end p50 %.0f max %.0f, scanned %d, trips %d, zero-scan items %d" % (
            rk, m.sum(), *np.percentile(st[m], [50, 99]), np.percentile(en[m], 50), en[m].max(), scanned[m].sum(), trips[m].sum(), (scanned[m] == 0).sum()))
# concurrency over time
edges = np.linspace(0, en.max(), 21)
for lo, hi in zip(edges[:-1], edges[1:]):
    act = ((st < hi) & (en > lo)).sum()
    started = ((st >= lo) & (st < hi)).sum()
    print("%6.0f-%6.0f us: %5d items active, %5d started" % (lo, hi, act, started))
