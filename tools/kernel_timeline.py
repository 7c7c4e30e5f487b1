#!/usr/bin/env python3
"""Timeline of the last dispatches of a rocprofv3 --kernel-trace run (rocpd sqlite): start offset, duration,
gap to the previous kernel's end, stream.  usage: kernel_timeline.py <results.db> [n_last]"""
import sqlite3
import sys


def find(c, prefix):
    for (n,) in c.execute("select name from sqlite_master where type='table'"):
        if n.startswith(prefix):
            return n
    return None


def main():
    c = sqlite3.connect(sys.argv[1])
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
