#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd sqlite output (kernel trace and/or PMC passes) as text.

usage: rocpd_summary.py <results.db> [min_grid]   — PMC values are summed over dispatches whose
grid is >= min_grid work-items (default 0) and printed per kernel with per-dispatch averages.
"""
import sqlite3
import sys


def find(c, prefix):
    for (n,) in c.execute("select name from sqlite_master where type='table'"):
        if n.startswith(prefix):
            return n
    return None


def main():
    db = sys.argv[1]
    min_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    c = sqlite3.connect(db)
    kd, ks = find(c, "rocpd_kernel_dispatch"), find(c, "rocpd_info_kernel_symbol")
    print("# %s (dispatches with grid >= %d work-items)" % (db, min_grid))
    print("%-64s %6s %12s %12s %12s %12s %5s %5s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_us",
                                                          "vgpr", "sgpr", "lds"))
    q = ("select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3, "
         "sum(d.end-d.start)/1e3, s.arch_vgpr_count, s.sgpr_count, max(d.group_segment_size) "
         "from %s d join %s s on d.kernel_id=s.id where d.grid_size_x*d.grid_size_y*d.grid_size_z >= %d "
         "group by s.kernel_name order by 6 desc" % (kd, ks, min_grid))
    for r in c.execute(q):
        print("%-64s %6d %12.2f %12.2f %12.2f %12.2f %5s %5s %7s" % ((r[0][:64],) + tuple(r[1:])))
    pe, pi = find(c, "rocpd_pmc_event"), find(c, "rocpd_info_pmc")
    if pe and c.execute("select count(*) from %s" % pe).fetchone()[0]:
        print("\n# PMC counters, summed over instances; per-dispatch average")
        q = ("select s.kernel_name, i.name, count(distinct d.id), sum(p.value) from %s p join %s i on p.pmc_id=i.id "
             "join %s d on p.event_id=d.event_id join %s s on d.kernel_id=s.id "
             "where d.grid_size_x*d.grid_size_y*d.grid_size_z >= %d group by s.kernel_name, i.name order by 1, 2"
             % (pe, pi, kd, ks, min_grid))
        for name, ctr, n, tot in c.execute(q):
            print("%-48s %-24s dispatches=%-4d avg=%.6g" % (name[:48], ctr, n, tot / max(1, n)))


if __name__ == "__main__":
    main()
