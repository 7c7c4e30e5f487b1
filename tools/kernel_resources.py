#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in ps_engine.hip (hipcc -Rpass-analysis=kernel-resource-usage).
K1 must stay <= 128 VGPRs (4 waves per SIMD: two 8-wave workgroups per CU) and use no scratch."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "probly-search_amd", "csrc")
extra = sys.argv[1:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                      "-c", "ps_engine.hip", "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra,
                     cwd=src, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (Function Name: (\S+)|\s+(\w[\w \[\]/]*): (\d+))", line)
    if not m:
        continue
    if m.group(2):
        cur = m.group(2)
        rows[cur] = {}
    elif cur:
        rows[cur][m.group(3).strip()] = int(m.group(4))
print("%-62s %5s %5s %7s %5s %5s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "sspill"))
for k, r in rows.items():
    name = k.replace("_ZN2ps7", "").replace("EEEvNS_7KParamsE", "")
    flag = "  <-- !" if ("k_score" in k and (r.get("VGPRs", 0) > 128 and "Lb1ELi" not in k or r.get("ScratchSize [bytes/lane]", 0))) else ""
    print("%-62s %5d %5d %7d %5d %5d%s" % (name[:62], r.get("VGPRs", 0), r.get("TotalSGPRs", 0),
                                            r.get("ScratchSize [bytes/lane]", 0), r.get("Occupancy [waves/SIMD]", 0),
                                            r.get("SGPRs Spill", 0), flag))
