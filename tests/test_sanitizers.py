"""SURVEY section 5's sanitizer counterpart for the 7 k lines of host C++ behind the C ABI (raw offsets, an mmap loader):
  * a corrupted-file fuzz of ps_snapshot_load against the normal library;
  * the host-logic CPU tests + the same fuzz in a child process against an -fsanitize=address,undefined build of
    ps_index.cpp / ps_snapshot.cpp / ps_keytable.cpp / ps_capi.cpp (csrc/Makefile `asan`), device = -1 snapshots.
The oracle's own sanitized build is tests/test_oracle_golden.py::test_oracle_under_address_and_ub_sanitizers."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "probly-search_amd", "csrc")
HOST_TESTS = ["test_host_logic.py", "test_keytable.py", "test_delta_snapshot.py", "test_host_callbacks.py",
              "test_reference_style.py", "test_device_planner.py", "test_integration_layout.py"]


def test_snapshot_load_fuzz(tmp_path):
    from fuzz_snapshot import fuzz
    total = {"rejected": 0, "accepted": 0}
    for seed in (1, 2):
        st = fuzz(str(tmp_path), n=250, seed=seed)
        total["rejected"] += st["rejected"]
        total["accepted"] += st["accepted"]
    assert total["rejected"] > 300, total  # most damage must be caught; the rest left a valid file and planned cleanly


def test_host_library_under_asan_ubsan(tmp_path):
    subprocess.check_call(["make", "-s", "-C", CSRC])       # the HIP objects the sanitized library links against
    subprocess.check_call(["make", "-s", "-C", CSRC, "asan"])
    so = os.path.join(CSRC, "alt", "libprobly_search_amd_asan.so")
    # python is not linked against libstdc++: without it in the preload ASan's __cxa_throw interceptor finds no real function
    preload = " ".join(subprocess.check_output(["g++", "-print-file-name=" + n], text=True).strip() for n in ("libasan.so", "libstdc++.so.6"))
    env = dict(os.environ, PS_SO=so, LD_PRELOAD=preload, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=67", PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] +
                       [os.path.join(ROOT, "tests", t) for t in HOST_TESTS], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stdout + r.stderr and "runtime error:" not in r.stdout + r.stderr, (r.stdout[-3000:], r.stderr[-3000:])
    code = ("import sys, json; from fuzz_snapshot import fuzz; import probly_search_amd as psa; "
            "assert psa.lib_path().endswith('_asan.so'), psa.lib_path(); "
            "print(json.dumps([fuzz(%r, n=200, seed=s) for s in (3, 4)]))" % str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-3000:]
