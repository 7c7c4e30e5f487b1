"""The C-ABI library loads (no GPU needed) and exports every symbol include/*.h declares."""
import ctypes
import os
import re

import probly_search_amd as psa
from probly_search_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "probly_search_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", src))
    names.discard("ps_tokenizer_fn")
    return names


def test_header_symbols_are_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 25
    L = ctypes.CDLL(psa.lib_path())
    for n in sorted(names):
        assert hasattr(L, n), "library does not export %s" % n
    assert names == set(_lib.SYMBOLS), (names ^ set(_lib.SYMBOLS))


def test_struct_sizes_match_header(tmp_path):
    """sizeof of every struct that crosses the ABI, as a C compiler sees the header, against the ctypes
    mirrors the tests and the bench use."""
    import os
    import subprocess
    pairs = [("ps_plan_entry", _lib.PlanEntry), ("ps_result", _lib.Result), ("ps_scorer_desc", _lib.ScorerDesc),
             ("ps_kernel_times", _lib.KernelTimes), ("ps_score_callbacks", _lib.ScoreCallbacks), ("ps_term_data", _lib.TermData),
             ("ps_batch_stats", _lib.BatchStats), ("ps_work_counters", _lib.WorkCounters), ("ps_snapshot_info", _lib.SnapshotInfo),
             ("ps_update_stats", _lib.UpdateStats), ("ps_field_data", _lib.FieldData), ("ps_host_csr", _lib.HostCsr)]
    src = tmp_path / "sizes.c"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src.write_text('#include <stdio.h>\n#include "probly_search_amd.h"\nint main(void) {\n' +
                   "".join('  printf("%%zu\\n", sizeof(%s));\n' % c for c, _ in pairs) + "  return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    for (cname, ct), n in zip(pairs, sizes):
        assert ctypes.sizeof(ct) == n, (cname, ctypes.sizeof(ct), n)
    assert ctypes.sizeof(_lib.PlanEntry) == 56 and ctypes.sizeof(_lib.Result) == 16


def test_queries_fail_loudly_without_device_snapshot():
    idx = psa.Index(1)
    idx.add_field_values(0, ["a b c"])
    snap = idx.snapshot(device=-1)
    try:
        snap.query("a", psa.bm25.new(), None, [1.0])
    except psa.PsError as e:
        assert e.status == _lib.PS_ENODEVICE
    else:
        raise AssertionError("host-only snapshot must refuse to score (no CPU fallback)")


def test_set_option_takes_the_documented_runtime_options_only(monkeypatch):
    """ps_set_option: the short list of run-time options in the header; the engine's several dozen experiment knobs stay
    environment variables (and go through this call only under PS_EXPERIMENT_KNOBS=1, which the A/B tools set)."""
    import probly_search_amd as psa
    from probly_search_amd import _lib
    L = psa.load()
    monkeypatch.delenv("PS_EXPERIMENT_KNOBS", raising=False)
    header = open(os.path.join(ROOT, "include", "probly_search_amd.h")).read()
    for name in (b"PS_DAAT", b"PS_DAAT_PRIME", b"PS_DEVICE_PLAN", b"PS_ROW_CACHE_MB", b"PS_WORK_COUNTERS", b"PS_SCORE_ALT", b"PS_DCTX"):
        assert name.decode() in header
        assert L.ps_get_option(name, None) in (0, 1)
    assert L.ps_set_option(b"PS_DAAT_PRIME", 1) == _lib.PS_OK
    assert L.ps_set_option(b"PS_DAAT_UMQ", 3) == _lib.PS_EINVAL and b"not a run-time option" in L.ps_last_error()
    assert L.ps_set_option(b"NOT_A_KNOB", 1) == _lib.PS_EINVAL
    monkeypatch.setenv("PS_EXPERIMENT_KNOBS", "1")
    assert L.ps_set_option(b"PS_DAAT_SAMPLE_DIV", 24) == _lib.PS_OK
    assert L.ps_abi_version() == _lib.PS_ABI_VERSION
