"""The multi-GPU exchange lives behind the C ABI (ps_comm_*, ps_snapshot_query_batch_allgather_flat):
  * CPU: the entry points exist, bad arguments are refused, the block layout is the documented one;
  * GPU: a C program forks 2 ranks that share the box's single GPU (PS_COMM_TRANSPORT=hostshm, the
    debugging transport - RCCL refuses two ranks on one device) and each must print the oracle's
    answer for the WHOLE batch; a world-of-1 communicator goes through real RCCL
    (ncclCommInitRank + ncclAllGather) via the Python thin caller."""
import os
import struct
import subprocess

import pytest

import probly_search_amd as psa
from probly_search_amd import _lib, dist as psd, synth
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "probly-search_amd", "csrc")


def test_block_layout_and_argument_checks():
    L = _lib.load()
    assert L.ps_topk_block_bytes(3, 10) == 3 * 10 * 16 + 16
    assert L.ps_topk_block_bytes(1024, 10) == 1024 * 10 * 16 + 4096
    assert L.ps_comm_world_size(None) == 1 and L.ps_comm_rank(None) == 0
    import ctypes as C
    h = C.c_void_p()
    assert L.ps_comm_init_rank(None, 2, 0, 0, C.byref(h)) == _lib.PS_EINVAL
    assert L.ps_comm_init_rank(b"\0" * 128, 2, 5, 0, C.byref(h)) == _lib.PS_EINVAL


def test_id_exchange_through_the_launcher_store_needs_no_process_group(tmp_path):
    """bench.py --gpus N (dist.Comm.from_env_store): the 128-byte communicator id goes from rank 0 to the others through the
    launcher's TCP store - here rank 0 hosts it, as without torchrun's agent store - and no torch.distributed process group
    (no second RCCL communicator, no gloo) is ever initialised.  Two processes, no GPU."""
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from probly_search_amd import dist as psd\n"
            "import torch.distributed as td\n"
            "uid, store = psd.Comm.exchange_id_via_store(2, int(sys.argv[1]))\n"
            "store.add('seen', 1)\n"
            "import time\n"
            "while int(store.add('seen', 0)) < 2: time.sleep(0.01)\n"
            "print('UID=' + uid.hex() + ' PG=' + str(td.is_initialized()))\n" % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PS_COMM_TRANSPORT="hostshm")
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in (0, 1)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines = [[l for l in o.splitlines() if l.startswith("UID=")][-1] for o, _ in outs]
    assert lines[0] == lines[1] and lines[0].endswith("PG=False") and len(lines[0]) > 4 + 2 * 128 - 1, lines
    import ctypes as C
    assert _lib.load().ps_comm_all_gather(None, None, None, 8, None) == _lib.PS_EINVAL


def _corpus_file(tmp_path, n_docs=300, n_queries=7):
    cfg = dict(synth.CONFIGS["C2"], n_docs=n_docs, vocab=120)
    corpus = synth.Corpus(**cfg)
    o = orc.Index(2)
    lines = []
    for keys, text, offsets in corpus.chunks(n_docs):
        raw = text.tobytes()
        for i, k in enumerate(keys):
            f0 = raw[int(offsets[2 * i]):int(offsets[2 * i + 1])].decode()
            f1 = raw[int(offsets[2 * i + 1]):int(offsets[2 * i + 2])].decode()
            lines.append("D %d\t%s\t%s" % (k, f0, f1))
            o.add_document(int(k), [[f0], [f1]])
    queries = corpus.queries(n_queries, 3)
    lines += ["Q " + q for q in queries]
    path = str(tmp_path / "corpus.txt")
    open(path, "w").write("\n".join(lines) + "\n")
    return path, o, queries


def _build(tmp_path):
    exe = str(tmp_path / "two_ranks")
    subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                    "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "c_abi", "two_ranks.c"), "-o", exe,
                    "-L", CSRC, "-lprobly_search_amd", "-L", "/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_two_rank_c_driver_builds(tmp_path):
    _build(tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_two_ranks_one_gpu_through_the_abi(tmp_path, world):
    path, o, queries = _corpus_file(tmp_path)
    top_k = 5
    env = dict(os.environ, PS_COMM_TRANSPORT="hostshm")
    r = subprocess.run([_build(tmp_path), path, str(world), str(top_k)], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    exp = []
    for q in queries:
        res = o.query(q, orc.bm25(), [1.0, 1.0])[:top_k]
        exp.append(["%d:%016x" % (k, struct.unpack("<Q", struct.pack("<d", s))[0]) for k, s in res])
    for rank in range(world):
        got = {}
        for line in r.stdout.splitlines():
            p = line.split()
            if p[:2] == ["rank", str(rank)]:
                got[int(p[3])] = p[5:]
        assert [got.get(i) for i in range(len(queries))] == exp, (rank, r.stdout)


@pytest.mark.gpu
def test_world_of_one_goes_through_rccl(monkeypatch):
    """ncclGetUniqueId / ncclCommInitRank / ncclAllGather really run (PS_COMM_FORCE_COLLECTIVE makes
    the 1-rank call issue the collective it would otherwise skip)."""
    monkeypatch.delenv("PS_COMM_TRANSPORT", raising=False)
    monkeypatch.setenv("PS_COMM_FORCE_COLLECTIVE", "1")
    cfg = dict(synth.CONFIGS["C2"], n_docs=3000, vocab=400)
    corpus = synth.Corpus(**cfg)
    p, o = synth.fill(psa.Index(2), corpus), synth.fill(orc.Index(2), corpus)
    snap = p.snapshot(device=0, tile_docs=256)
    comm = psd.Comm.init_rank(psd.Comm.unique_id(), 1, 0, 0)
    queries = corpus.queries(9, 3)
    got = psd.query_batch_sharded(snap, queries, psa.bm25.new(), [1.0, 1.0], 10, comm)
    exp = [o.query(q, orc.bm25(), [1.0, 1.0])[:10] for q in queries]
    assert got == exp
    # the caller's own small exchanges go through the same communicator (bench.py's barrier / max over ranks)
    assert comm.all_gather_bytes(b"abcdefgh-123") == [b"abcdefgh-123"]
    comm.barrier()
    assert comm.max_f64(2.5) == 2.5 and comm.min_i64(-7) == -7 and comm.broadcast_bytes(b"xyz", size=16) == b"xyz".ljust(16, b"\0")
    comm.free()
    assert psd.query_batch_sharded(snap, queries, psa.bm25.new(), [1.0, 1.0], 10, None) == exp


def _rccl_path_in_subprocess(preload_torch):
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n" % ROOT +
            ("import torch\n" if preload_torch else "") +
            "import probly_search_amd as psa\n"
            "p = psa.load().ps_comm_rccl_path()\n"
            "print('RCCL_PATH=' + (p.decode() if p else 'NONE'))\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return [l for l in out.stdout.splitlines() if l.startswith("RCCL_PATH=")][-1][len("RCCL_PATH="):]


def test_one_rccl_per_process():
    """A host application may run torch's nccl backend (torch/lib/librccl.so) beside the library's own communicator in one
    process (bench.py itself no longer does): the library must pick up the instance that is already mapped instead of loading /opt/rocm's beside it;
    a process without torch gets the system library.  No GPU needed: only the loader is exercised."""
    import os
    with_torch = _rccl_path_in_subprocess(True)
    assert "librccl.so" in with_torch
    mapped_by_torch = os.path.join("torch", "lib")
    # (a torch build that links the system RCCL maps /opt/rocm's: either way it is the mapped one)
    assert mapped_by_torch in with_torch or with_torch.startswith("/opt/rocm"), with_torch
    alone = _rccl_path_in_subprocess(False)
    assert "librccl.so" in alone and mapped_by_torch not in alone, alone
