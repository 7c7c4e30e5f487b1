"""Register / scratch budget of the hot kernels, checked at build time (no GPU): hipcc cross-compiles
tests/kernel_tu/hot_kernels.hip - explicit instantiations of the kernels the BASELINE configs run - with
-Rpass-analysis=kernel-resource-usage, and every kernel must stay inside its budget.  The scoring kernels of the
document-at-a-time family are latency bound and scale with resident waves (capped at 3 waves per
SIMD they lose 20 %): <= 128 VGPRs, i.e. 4 waves per SIMD, and no scratch is the contract; the ceilings on SGPR
spills and on the known exceptions are the values at this head, so a regression fails here instead of surfacing as
a slower bench line."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

# kernel (demangled prefix) -> (max VGPRs, min waves per SIMD, max scratch bytes per lane, max SGPR spills)
BUDGET = {
    "ps::k_daat_small<2, true, 4>": (128, 4, 0, 165),
    "ps::k_daat_small<2, false, 4>": (128, 4, 0, 125),
    "ps::k_daat_small<1, false, 4>": (128, 4, 0, 125),
    # queries of <= 3 lists (BASELINE configs 2 and 4): two other lists of wave-uniform state instead of three
    "ps::k_daat_small<2, true, 3>": (128, 4, 0, 165),
    "ps::k_daat_small<2, false, 3>": (128, 4, 0, 125),
    "ps::k_daat<2, true>": (128, 4, 0, 105),
    "ps::k_daat<2, false>": (128, 4, 0, 105),
    # (36 bytes of frame are reserved for k_daat_z but its ISA holds no scratch instruction)
    # (its scan holds ~95 VGPRs since the first level moved behind the reach queue: 5 waves per SIMD for the serving
    # instantiation; the wave-uniform state - two sets of field-length limits, three tie levels - spills more SGPRs)
    "ps::k_daat_z<2, true, 4>": (104, 4, 36, 300),   # (the counting instantiation; 85 VGPRs)
    "ps::k_daat_z<2, false, 4>": (104, 5, 36, 210),  # (+6 spills in round 6: the doc-ordered filter word carries a shift beside its base)
    "ps::k_daat_z<1, false, 4>": (96, 5, 36, 195),
    # queries of five to eight records (round 5): twice the per-list state - 15 KB of LDS queues per two-wave workgroup (its bound
    # table is recomputed per threshold change instead of tabulated: 4 waves per SIMD) and the wave-uniform per-list words of 7
    # other lists in scalar registers (spilled to VGPR lanes); no scratch
    "ps::k_daat_z<2, false, 8>": (128, 4, 0, 560),
    "ps::k_score<0, 2, false, false, 8>": (128, 4, 0, 115),
    "ps::k_score<1, 2, false, false, 8>": (128, 4, 0, 125),
    # known debt, frozen: the single-field latency kernel of C1
    "ps::k_score<0, 1, false, false, 8>": (130, 3, 0, 95),
    "ps::k_score<0, 1, false, false, 4>": (130, 3, 0, 95),
    # (a thread per query, 16 one-wave workgroups per batch: occupancy is not what bounds it; the 320 bytes are the frame of
    # prep_query_general, out of line, for plans of > 4 entries)
    # (round 6: + the primed threshold per entry - k_list_kth table reads, a fourth array of per-entry doubles in the register arm)
    # (round 6, second half: a small query requests the tables of all its entries together - 7 doubles x 4 entries in flight -
    # and keeps its entries' facts in registers for the entry loop: 102 VGPRs; 16 one-wave workgroups per batch, occupancy is moot)
    "ps::k_prep_query": (104, 4, 384, 0),
    "ps::k_zprep_query<4>": (48, 8, 0, 0),
    "ps::k_zprep_query<8>": (64, 8, 0, 0),
    "ps::k_zprep_items": (48, 8, 0, 0),
    "ps::k_prep_items": (32, 8, 0, 0),
    "ps::k_merge_items": (40, 8, 0, 0),
}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_hot_kernels_stay_inside_their_register_budget():
    src = os.path.join(ROOT, "tests", "kernel_tu", "hot_kernels.hip")
    err = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c", src, "-o", os.devnull,
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900).stderr
    rows, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"remark: (Function Name: (\S+)|\s+(\w[\w \[\]/]*): (\d+))", line)
        if not m:
            continue
        if m.group(2):
            cur = m.group(2)
            rows[cur] = {}
        elif cur:
            rows[cur][m.group(3).strip()] = int(m.group(4))
    assert rows, err[-2000:]
    names = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.split("\n")
    seen = {}
    for mangled, dem in zip(rows, names):
        dem = re.sub(r"^void ", "", dem)
        seen[re.sub(r"\(.*$", "", dem)] = rows[mangled]
    bad = []
    for name, (vgpr, occ, scratch, spill) in BUDGET.items():
        assert name in seen, (name, sorted(seen))
        r = seen[name]
        got = (r.get("VGPRs", 0), r.get("Occupancy [waves/SIMD]", 0), r.get("ScratchSize [bytes/lane]", 0), r.get("SGPRs Spill", 0))
        if got[0] > vgpr or got[1] < occ or got[2] > scratch or got[3] > spill:
            bad.append((name, "VGPRs %d (<= %d), waves/SIMD %d (>= %d), scratch %d (<= %d), SGPR spills %d (<= %d)" % (
                got[0], vgpr, got[1], occ, got[2], scratch, got[3], spill)))
    assert not bad, bad
