#!/usr/bin/env python3
"""Counter passes of tools/fetch_size_calibration.hip -> counter value per known byte, per pattern.

usage: fetch_size_calibration.py <dir with known.jsonl and p_*/p_results.db>
Dispatches named cal_* are taken in launch order and matched with known.jsonl line by line.  Writes
<dir>/calibration.json: per pattern the counters, FETCH_SIZE (KiB -> B) over useful bytes, over the bytes of
the distinct 64-byte sectors and of the distinct 128-byte lines the pattern touches."""
import glob, json, os, sqlite3, sys


def table(c, prefix):
    for (n,) in c.execute("select name from sqlite_master where type='table'"):
        if n.startswith(prefix):
            return n


def per_dispatch(db):
    c = sqlite3.connect(db)
    kd, ks, pe, pi = (table(c, p) for p in ("rocpd_kernel_dispatch", "rocpd_info_kernel_symbol", "rocpd_pmc_event", "rocpd_info_pmc"))
    q = ("select d.id, s.kernel_name, i.name, sum(p.value) from %s p join %s i on p.pmc_id=i.id join %s d on p.event_id=d.event_id "
         "join %s s on d.kernel_id=s.id group by d.id, i.name order by d.start, d.id" % (pe, pi, kd, ks))
    rows, order = {}, []
    for did, kname, ctr, val in c.execute(q):
        if "cal_" not in kname:
            continue
        if did not in rows:
            rows[did] = {}
            order.append(did)
        rows[did][ctr] = val
    return [rows[d] for d in order]


def main():
    d = sys.argv[1]
    known = [json.loads(l) for l in open(os.path.join(d, "known.jsonl")) if l.startswith("{")]
    for db in sorted(glob.glob(os.path.join(d, "p_*", "**", "*_results.db"), recursive=True)):
        vals = per_dispatch(db)
        if len(vals) != len(known):
            print("skip %s: %d dispatches for %d patterns" % (db, len(vals), len(known)))
            continue
        for k, v in zip(known, vals):
            k.setdefault("counters", {}).update(v)
    print("%-26s %12s %9s | %8s %8s %8s | %8s %8s" % ("pattern", "useful MB", "GB/s", "F/useful", "F/64B", "F/128B", "W/useful", "miss*64/F"))
    for k in known:
        c = k.get("counters", {})
        f = c.get("FETCH_SIZE", 0) * 1024.0
        w = c.get("WRITE_SIZE", 0) * 1024.0
        k["fetch_over_useful"] = f / k["useful_bytes"]
        k["fetch_over_64B_lines"] = f / k["bytes_as_64B_lines"]
        k["fetch_over_128B_lines"] = f / k["bytes_as_128B_lines"]
        k["write_over_useful"] = w / k["useful_bytes"]
        miss = c.get("TCC_MISS_sum", 0) * 64.0
        print("%-26s %12.1f %9.1f | %8.3f %8.3f %8.3f | %8.3f %8.3f" % (k["kernel"], k["useful_bytes"] / 1e6, k["useful_GBps"], k["fetch_over_useful"],
              k["fetch_over_64B_lines"], k["fetch_over_128B_lines"], k["write_over_useful"], miss / f if f else 0))
    json.dump(known, open(os.path.join(d, "calibration.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
