#!/usr/bin/env python3
"""PMC passes -> fraction of each hardware resource the dominant scoring kernel uses.

usage: derive_roofline.py <dir with kt.txt pmc_*.txt> <config tag> <git head> [bench args...]
Writes <dir>/roofline_<tag>.json (copy to profiles/: bench.py reads profiles/roofline_<config>.json and
prices `roofline` against it when kernel symbol, config, scorer and row mode match) and prints the
derivation.  Everything is per launch of the dominant kernel (the one with the largest total time
in the kernel trace), averaged over the dispatches of the counter passes.

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed
over waves; GRBM_GUI_ACTIVE counts shader clocks the chip was busy, summed over the 8 XCD instances;
FETCH_SIZE / WRITE_SIZE are KiB of fabric requests tallied at 64 bytes each.  Calibrated on this chip
(tools/fetch_size_calibration.sh, profiles/r04_fetch_size_calibration.txt): FETCH_SIZE x 1024 / 64 is the number
of L2 -> fabric read requests (TCC_EA0_RDREQ); a wide coalesced stream issues one request per 128-byte line
(counter = half the bytes: the guide's x 2), a scattered 4 / 8 / 16-byte load that misses L2 issues one request
per load (counter = 64 bytes per load whatever its width; whether 64 or 128 bytes cross the fabric for it the
counters do not say).  What the fabric sustains is a REQUEST rate: 41 G/s for the 128-byte requests of a stream
(= 5.3 TB/s), 43 G/s for scattered loads over 8 GiB, 56 G/s when the target fits the Infinity Cache (hits there
are counted, and cost nearly a full slot); only L2 hits (239 G/s) are free of it.  So beside
hbm = 2 x FETCH_SIZE + WRITE_SIZE (bytes if every request moved a 128-byte line: exact for streams, an upper
bound for scattered loads) the derivation reports `fabric_requests` against the measured 43 G/s ceiling - the
roofline that binds a kernel of scattered lookups.  Other peaks: HBM 8 TB/s (spec), L2 34.5 TB/s of 128-byte
lines, LDS 256 B/clk/CU, one VALU issue per SIMD per clock.
"""
import hashlib
import json
import os
import re
import subprocess
import sys

N_CU, N_SIMD, N_XCD, CLOCK_HZ = 256, 1024, 8, 2.4e9
HBM_PEAK, L2_PEAK, LDS_BPC = 8.0e12, 34.5e12, 256
FABRIC_REQ_PEAK = 43.0e9  # scattered L2-missing loads per second this chip sustained (profiles/r04_fetch_size_calibration.txt)
KERNEL_SOURCES = ("ps_kernels_common.hpp", "ps_kernels_score.hpp", "ps_kernels_daat.hpp", "ps_kernels_plan.hpp", "ps_prep_kernels.hpp", "ps_z21_daat.hpp")


def kernel_source_hash(root):
    """sha256 over the device-code headers: bench.py refuses a derivation made for other kernel sources."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(root, "probly-search_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def demangle(name):
    name = re.sub(r"\.kd$", "", name.strip())
    try:
        out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        out = name
    out = re.sub(r"^void ", "", out)
    return re.sub(r"\(ps::KParams\)$", "", out)


def parse(path):
    """-> ({kernel: {calls, avg_us}}, {kernel: {counter: per-dispatch avg}})"""
    kern, pmc = {}, {}
    if not os.path.exists(path):
        return kern, pmc
    for line in open(path):
        m = re.match(r"(\S+)\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+(\S+)\s+(\S+)\s+(\S+)", line)
        if m and not line.startswith("kernel"):
            kern[m.group(1)] = {"calls": int(m.group(2)), "avg_us": float(m.group(3)), "total_us": float(m.group(6)),
                                "vgpr": m.group(7), "sgpr": m.group(8), "lds": m.group(9)}
        m = re.match(r"(\S+)\s+(\S+)\s+dispatches=(\d+)\s+avg=([0-9.e+-]+)", line)
        if m:
            pmc.setdefault(m.group(1), {})[m.group(2)] = float(m.group(4))
    return kern, pmc


def main():
    d, tag, head = sys.argv[1], sys.argv[2], sys.argv[3]
    args = sys.argv[4:]
    kt, _ = parse(os.path.join(d, "kt.txt"))
    score = {k: v for k, v in kt.items() if re.search(r"k_daat|k_score|k_z21", k)}
    if not score:
        sys.exit("no scoring kernel in the kernel trace")
    dom = max(score, key=lambda k: score[k]["total_us"])
    # bench.py times the headline with the serving instantiation (no work counters) and prices the roofline with the
    # counting one: take the symbol the traced run's bench line names, if the trace holds it
    try:
        line = [l for l in open(os.path.join(d, "kt.bench.json")) if l.startswith("{")][-1]
        want = json.loads(line)["roofline"]["kernel"]
        named = [k for k in score if demangle(k) == want]
        if named:
            dom = named[0]
    except Exception:  # noqa: BLE001
        pass
    key = dom[:48]  # rocpd_summary truncates names in the PMC section
    ctr = {}
    for f in sorted(os.listdir(d)):
        if f.startswith("pmc_") and f.endswith(".txt"):
            _, pmc = parse(os.path.join(d, f))
            for k, v in pmc.items():
                if k == key or dom.startswith(k):
                    ctr.update(v)
    t = score[dom]["avg_us"] * 1e-6
    g = lambda n: ctr.get(n)
    cycles = g("GRBM_GUI_ACTIVE") / N_XCD if g("GRBM_GUI_ACTIVE") else t * CLOCK_HZ  # busy shader clocks of the launch
    clock = cycles / t
    res, notes = {}, []
    if g("FETCH_SIZE") is not None:
        rd, wr = 2 * g("FETCH_SIZE") * 1024, (g("WRITE_SIZE") or 0) * 1024
        res["hbm"] = {"per_launch": rd + wr, "unit": "B", "scale": 1e9, "peak": HBM_PEAK / 1e9, "rate_unit": "GB/s"}
        notes.append("hbm = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> B): bytes if every read request moved a 128-byte line")
        res["fabric_requests"] = {"per_launch": g("FETCH_SIZE") * 1024 / 64, "unit": "read requests", "scale": 1e9, "peak": FABRIC_REQ_PEAK / 1e9,
                                  "rate_unit": "G requests/s"}
        notes.append("fabric_requests = FETCH_SIZE x 1024 / 64 (= TCC_EA0_RDREQ) against the 43 G/s this chip sustained for scattered "
                     "L2-missing loads (calibration); a pure 128-byte stream tops out at 41 G/s = 5.3 TB/s")
    if g("TCC_REQ_sum") is not None:
        res["l2"] = {"per_launch": g("TCC_REQ_sum") * 128, "unit": "B (128-B line requests)", "scale": 1e9, "peak": L2_PEAK / 1e9,
                     "rate_unit": "GB/s"}
        notes.append("l2 = TCC_REQ_sum x 128 B; hit rate %.3f" % (g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
    if g("SQ_ACTIVE_INST_VALU") is not None:
        # quad-cycles of VALU issue summed over waves -> SIMD-cycles; one issue per SIMD per clock at best
        res["valu_issue"] = {"per_launch": g("SQ_ACTIVE_INST_VALU") * 4, "unit": "SIMD-cycles", "scale": 1e9,
                             "peak": N_SIMD * clock / 1e9, "rate_unit": "G SIMD-cycles/s"}
        notes.append("valu_issue = SQ_ACTIVE_INST_VALU x 4 against %d SIMDs x %.2f GHz (measured clock)" % (N_SIMD, clock / 1e9))
    if g("SQ_ACTIVE_INST_LDS") is not None:
        res["lds_issue"] = {"per_launch": g("SQ_ACTIVE_INST_LDS") * 4, "unit": "CU-cycles", "scale": 1e9,
                            "peak": N_CU * clock / 1e9, "rate_unit": "G CU-cycles/s"}
        notes.append("lds_issue = SQ_ACTIVE_INST_LDS x 4 against %d CUs" % N_CU)
    if g("SQ_ACTIVE_INST_VMEM"):  # (reads 0 on this rocprofv3 / gfx950: then it is left out)
        res["vmem_issue"] = {"per_launch": g("SQ_ACTIVE_INST_VMEM") * 4, "unit": "CU-cycles", "scale": 1e9,
                             "peak": N_CU * clock / 1e9, "rate_unit": "G CU-cycles/s"}
        notes.append("vmem_issue = SQ_ACTIVE_INST_VMEM x 4 against %d CUs (one vector-memory issue port per CU)" % N_CU)
    wave = {}
    if g("SQ_WAVE_CYCLES"):
        wc = g("SQ_WAVE_CYCLES")
        wave = {"stalled_on_waitcnt": (g("SQ_WAIT_ANY") or 0) / wc, "issue_stalled": (g("SQ_WAIT_INST_ANY") or 0) / wc,
                "issuing": (g("SQ_ACTIVE_INST_ANY") or 0) / wc,
                "resident_waves_per_simd": wc * 4 / (N_SIMD * cycles),
                "note": "fractions of wave-cycles (SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES)"}
    fr = {k: (v["per_launch"] / t) / v["scale"] / v["peak"] for k, v in res.items()}
    scorer = "bm25"
    cfg = tag.split("_")[0]
    if "--scorer" in args:
        scorer = args[args.index("--scorer") + 1]
    elif cfg == "C3":
        scorer = "zero_to_one"
    work = None  # the kernel's own work counters, as the traced run's bench line reports them (per launch)
    try:
        line = [l for l in open(os.path.join(d, "kt.bench.json")) if l.startswith("{")][-1]
        rl = json.loads(line)["roofline"]
        work = {"units_processed": rl.get("units_processed"), "bytes_touched": rl.get("bytes_touched"),
                "kernel_avg_ms_live": rl.get("kernel_avg_ms")}
    except Exception:  # noqa: BLE001
        pass
    # the roofline from the traces alone: bytes the kernel counted itself (the traced run's bench line) / the trace's average
    # duration - overlapped pass (two scoring queues: individual durations overlap, the fraction understates the chip) and
    # serialised pass (PS_SCORE_ALT=0: duration = cost)
    from_trace = {}
    for name in ("kt", "kt_serial"):
        try:
            ktx, _ = parse(os.path.join(d, name + ".txt"))
            line = [l for l in open(os.path.join(d, name + ".bench.json")) if l.startswith("{")][-1]
            rlx = json.loads(line)["roofline"]
            sym = [k for k in ktx if demangle(k) == rlx["kernel"]]
            if sym:
                avg = ktx[sym[0]]["avg_us"]
                from_trace[name] = {"kernel": rlx["kernel"], "calls": ktx[sym[0]]["calls"], "avg_us_in_trace": avg,
                                    "bytes_touched_per_launch": rlx["bytes_touched"],
                                    "frac_of_8TBps": rlx["bytes_touched"] / (avg * 1e-6) / HBM_PEAK,
                                    "bench_line": {k: rlx.get(k) for k in ("frac", "frac_overlapped", "frac_serial", "kernel_avg_ms",
                                                                           "kernel_busy_avg_ms", "kernel_serial_avg_ms")}}
        except Exception as e:  # noqa: BLE001
            from_trace[name] = {"error": str(e)}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {"config": cfg, "roofline_from_traces": from_trace, "scorer": scorer, "head": head, "kernel_sources_sha16": kernel_source_hash(root), "kernel": demangle(dom), "kernel_symbol": dom, "work_counters": work,
           "resident_rows": "--resident-rows" in args, "bench_args": args,
           "kernel_avg_us_in_trace": score[dom]["avg_us"], "registers": {k: score[dom][k] for k in ("vgpr", "sgpr", "lds")},
           "measured_clock_GHz": clock / 1e9, "counters_per_launch": ctr,
           "hbm_bytes_per_launch": res.get("hbm", {}).get("per_launch"), "resources": res,
           "fractions_in_profiled_run": fr, "wave_cycles": wave, "notes": notes,
           "binding": (max(fr, key=fr.get) if fr else None),
           "reading": "the largest fraction names the unit nearest its peak; when every fraction is small and most "
                      "wave-cycles are stalled on s_waitcnt, the kernel is bound by dependent memory latency, not by a "
                      "throughput peak - then `binding` still names the busiest unit and `wave_cycles` says why it is idle"}
    json.dump(out, open(os.path.join(d, "roofline_%s.json" % cfg), "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("kernel", "roofline_from_traces", "kernel_avg_us_in_trace", "measured_clock_GHz", "fractions_in_profiled_run",
                                           "wave_cycles", "binding", "hbm_bytes_per_launch")}, indent=1))
    print("all kernels in the trace:")
    for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_us"]):
        print("  %-70s calls=%-4d avg_us=%9.2f" % (demangle(k)[:70], v["calls"], v["avg_us"]))


if __name__ == "__main__":
    main()
