"""The drop-in boundary is a C ABI: a plain C99 translation unit (tests/c_abi/readme_example.c,
the reference README's example) compiles against include/probly_search_amd.h, links the library
and gets the oracle's answer - or, without a HIP device, PS_ENODEVICE (never a CPU fallback)."""
import os
import subprocess

import pytest

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "probly-search_amd", "csrc")


def _build(tmp_path):
    exe = str(tmp_path / "readme_example")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "readme_example.c"), "-o", exe,
                    "-L", CSRC, "-lprobly_search_amd", "-Wl,-rpath," + CSRC], check=True)
    return exe


def _expected():
    o = orc.Index(2)
    o.add_document(0, [["abc"], ["dfg"]])
    o.add_document(1, [["dfgh"], ["abcd"]])
    return o.query("abc", orc.bm25(), [1.0, 1.0])


def _check_results(stdout):
    import struct
    got = [(int(l.split()[1]), l.split()[2]) for l in stdout.splitlines() if l.startswith("result ")]
    exp = [(k, "%016x" % struct.unpack("<Q", struct.pack("<d", s))[0]) for k, s in _expected()]
    assert got == exp and len(got) == 2


def test_c99_caller_builds_and_fails_loudly_without_a_device(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True)
    assert "docs 2 nodes 9" in r.stdout
    if r.returncode == 0:  # a HIP device is present after all
        _check_results(r.stdout)
    else:
        assert r.returncode == 5 and "no CPU scoring fallback" in r.stdout  # PS_ENODEVICE


@pytest.mark.gpu
def test_c99_caller_gets_the_oracle_answer(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    _check_results(r.stdout)


def test_c99_caller_of_the_key_table(tmp_path):
    """tests/c_abi/keytable.c: `Index<T>` keys that are not u64 (src/index.rs:19-33) through ps_keytable_* - host only."""
    exe = str(tmp_path / "keytable")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "keytable.c"), "-o", exe,
                    "-L", CSRC, "-lprobly_search_amd", "-Wl,-rpath," + CSRC], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_boost_cone_dominates_every_boost_vector(tmp_path):
    """tests/c_abi/bounds_check.cpp: the host side of the two-field joint bound (csrc/ps_bounds.hpp) - for 200 000 boost vectors
    from subnormal to 1e300 the two-direction combination dominates the vector componentwise, and the interpolated bound
    dominates b . v on random point sets.  Plain C++, no device."""
    exe = str(tmp_path / "bounds_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", os.path.join(ROOT, "tests", "c_abi", "bounds_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
