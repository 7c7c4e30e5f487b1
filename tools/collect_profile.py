#!/usr/bin/env python3
"""gpurun_out/prof_<cfg>/ (tools/profile_bench.sh) -> profiles/r<round>_<cfg>_rocprof_summary.txt (kernel
trace + every PMC pass, one text file) and profiles/roofline_<cfg>.json (the derivation bench.py reads).
usage: collect_profile.py <round tag, e.g. r02> <cfg> [<cfg> ...]"""
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
for cfg in sys.argv[2:]:
    d = os.path.join(root, "gpurun_out", "prof_" + cfg)
    out = os.path.join(root, "profiles", "%s_%s_rocprof_summary.txt" % (tag, cfg))
    with open(out, "w") as f:
        f.write("# %s %s: rocprofv3 passes of bench.py on one MI355X, summarised by tools/rocpd_summary.py; one pass per "
                "section (counters never share a run with the kernel trace)\n" % (tag, cfg))
        f.write("# " + open(os.path.join(d, "command.txt")).read())
        for name in ("kt", "kt_serial", "pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq1", "pmc_sq2", "pmc_sq3"):
            p = os.path.join(d, name + ".txt")
            if os.path.exists(p):
                f.write("\n## pass %s\n" % name)
                f.write(open(p).read())
            b = os.path.join(d, name + ".bench.json")
            if name in ("kt", "kt_serial") and os.path.exists(b):
                f.write("\n## bench.py line printed by the traced run (pass %s)\n" % name + open(b).read())
        f.write("\n## derivation (tools/derive_roofline.py)\n" + open(os.path.join(d, "derive.log")).read())
    shutil.copy(os.path.join(d, "roofline_%s.json" % cfg), os.path.join(root, "profiles", "roofline_%s.json" % cfg))
    print("wrote", out)
