"""bench.py --gpus N: the flag is honoured (self-spawned ranks or a loud failure), never ignored."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "PS_BENCH_DEBUG_ONE_GPU"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          env=e, timeout=timeout)


def test_more_gpus_than_devices_fails_loudly():
    import probly_search_amd as psa
    have = psa.load().ps_device_count()
    r = _run(["--gpus", str(max(2, have + 1)), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "HIP device(s) are visible" in (r.stderr + r.stdout)


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_two_ranks_debug_one_gpu_end_to_end():
    """Self-spawned 2-rank run on the box's single GPU (debug transport): the N>1 path of bench.py -
    snapshot file shared between ranks, sharded batch, library-side all-gather - runs end to end."""
    r = _run(["--gpus", "2", "--n-docs", "20000", "--batch", "64", "--steps", "3", "--warmup", "1",
              "--no-cpu-baseline", "--no-single-latency"], env={"PS_BENCH_DEBUG_ONE_GPU": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["value"] > 0
    assert "DEBUG" in d["data"]
    # N > 1 also reports BASELINE config 4 (one 8192-query batch split over the ranks) and which RCCL was resolved
    assert d["config4"].get("queries_per_s", 0) > 0 and "8192-query" in d["config4"]["workload"] and "rccl_path" in d, d["config4"]


@pytest.mark.gpu
def test_eight_ranks_debug_one_gpu_end_to_end():
    """The launch shape of the driver's 8-GPU scaling run - `bench.py --gpus 8`, eight self-spawned ranks, one
    snapshot file mmap-loaded by seven of them, 8 x 1024-query shards of C4's 8192-query batch, the library-side
    exchange of the top-k blocks every step - on the box's single GPU through the debug transport, so that the
    first real 8-GPU run does not fail on plumbing.  (Not a measurement.)"""
    r = _run(["--gpus", "8", "--config", "C4", "--n-docs", "30000", "--steps", "3", "--warmup", "1",
              "--no-cpu-baseline", "--no-single-latency"], env={"PS_BENCH_DEBUG_ONE_GPU": "1"}, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 8192 and d["value"] > 0 and d["scaling"] == "weak"
    assert "DEBUG" in d["data"]


def test_effective_cpus_reads_the_containers_quota(tmp_path, monkeypatch):
    """bench.py sizes the oracle's all-cores legs from the CPUs the process can really use: the GPU boxes show 256 hardware threads
    under a cgroup quota of 16 (profiles/r06_cpu_baseline_scaling.txt)."""
    import builtins
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    n, src = b.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    real_open = builtins.open

    def fake_open(path, *a, **kw):
        if path == "/sys/fs/cgroup/cpu.max":
            f = tmp_path / "cpu.max"
            f.write_text("200000 100000\n")
            return real_open(str(f), *a, **kw)
        return real_open(path, *a, **kw)

    monkeypatch.setattr(builtins, "open", fake_open)
    if (os.cpu_count() or 1) > 2:
        assert b.effective_cpus() == (2, "cgroup cpu quota 2.00")
