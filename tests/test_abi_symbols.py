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


def test_struct_sizes_match_header():
    assert ctypes.sizeof(_lib.PlanEntry) == 56
    assert ctypes.sizeof(_lib.Result) == 16
    assert ctypes.sizeof(_lib.ScorerDesc) == 32
    assert ctypes.sizeof(_lib.KernelTimes) == 120
    assert ctypes.sizeof(_lib.ScoreCallbacks) == 40
    assert ctypes.sizeof(_lib.TermData) == 48
    assert ctypes.sizeof(_lib.BatchStats) == 96


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
