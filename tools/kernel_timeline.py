#!/usr/bin/env python3
"""Timeline of the last dispatches of a rocprofv3 --kernel-trace run (rocpd sqlite): start offset, duration,
gap to the previous kernel's end, stream.  usage: kernel_timeline.py <results.db> [n_last]
       kernel_timeline.py <results.db> --summary [kernel substring, default the serving scoring kernels "Lb0E"]
--summary: the region from the first to the last dispatch whose name holds the substring (bench.py's timed steps run the serving
instantiations, template argument WC = false): per stream the share of the region its kernels were executing, per kernel its share
and average duration, and the union of the scoring kernels' intervals - what overlaps what, and which stream paces the pipeline."""
import sqlite3
import sys


def find(c, prefix):
    for (n,) in c.execute("select name from sqlite_master where type='table'"):
        if n.startswith(prefix):
            return n
    return None


def summary(c, needle):
    kd, ks = find(c, "rocpd_kernel_dispatch"), find(c, "rocpd_info_kernel_symbol")
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    sid = "d.stream_id" if "stream_id" in cols else ("d.queue_id" if "queue_id" in cols else "0")
    rows = list(c.execute("select s.kernel_name, d.start, d.end, %s from %s d join %s s on d.kernel_id=s.id order by d.start" % (sid, kd, ks)))
    score = [r for r in rows if ("k_daat" in r[0] or "k_score" in r[0] or "k_z21" in r[0]) and needle in r[0]]
    if not score:
        sys.exit("no scoring kernel whose name holds %r" % needle)
    t0, t1 = score[0][1], max(r[2] for r in score)
    reg = [r for r in rows if r[1] >= t0 and r[2] <= t1]
    span = (t1 - t0) / 1e3
    print("region %.0f us, %d scoring launches, %.1f us per batch" % (span, len(score), span / len(score)))
    by_stream = {}
    for name, st, en, q in reg:
        by_stream.setdefault(q, []).append((name, st, en))

    def union(iv):
        tot, cur_s, cur_e = 0.0, None, None
        for st, en in sorted(iv):
            if cur_e is None or st > cur_e:
                if cur_e is not None:
                    tot += cur_e - cur_s
                cur_s, cur_e = st, en
            else:
                cur_e = max(cur_e, en)
        return tot + (cur_e - cur_s if cur_e is not None else 0.0)

    for q, ks_ in sorted(by_stream.items(), key=lambda kv: -union([(a, b) for _, a, b in kv[1]])):
        print("stream %s busy %.3f" % (q, union([(a, b) for _, a, b in ks_]) / 1e3 / span))
        per = {}
        for name, st, en in ks_:
            per.setdefault(name, []).append(en - st)
        for name, ds in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            print("     %-62s %.3f  avg %.1f us" % (name[:62], sum(ds) / 1e3 / span, sum(ds) / len(ds) / 1e3))
    print("union of the scoring kernels' intervals: %.3f of the region" % (union([(r[1], r[2]) for r in score]) / 1e3 / span))


def main():
    c = sqlite3.connect(sys.argv[1])
    if len(sys.argv) > 2 and sys.argv[2] == "--summary":
        return summary(c, sys.argv[3] if len(sys.argv) > 3 else "Lb0E")
    n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    kd, ks = find(c, "rocpd_kernel_dispatch"), find(c, "rocpd_info_kernel_symbol")
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    sid = "d.stream_id" if "stream_id" in cols else ("d.queue_id" if "queue_id" in cols else "0")
    rows = list(c.execute("select s.kernel_name, d.start, d.end, %s, d.grid_size_x*d.grid_size_y*d.grid_size_z from %s d join %s s "
                          "on d.kernel_id=s.id order by d.start" % (sid, kd, ks)))
    rows = rows[-n_last:]
    t0 = rows[0][1]
    prev_end = None
    print("%10s %9s %9s %6s %10s  %s" % ("start_us", "dur_us", "gap_us", "strm", "grid", "kernel"))
    for name, st, en, q, grid in rows:
        gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
        print("%10.1f %9.1f %9.1f %6s %10d  %s" % ((st - t0) / 1e3, (en - st) / 1e3, gap, q, grid, name[:70]))
        prev_end = max(prev_end or en, en)


if __name__ == "__main__":
    main()
