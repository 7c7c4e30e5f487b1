"""Multi-GPU plumbing for the query path: replicated corpus, query batch sharded across ranks,
per-rank top-k blocks all-gathered by the library's own RCCL communicator (ps_comm_*: ncclAllGather over xGMI).
`all_gather_topk` is the same exchange written against torch.distributed for hosts that already own a process group
("gloo" in the CPU tests).

The reference has no distributed mode at all; what makes this sharding natural is that queries
are independent — each `Index::query` call owns its `scores` / `visited` maps and the index is
read-only under `&self` (src/query.rs:21-37).  No collective is needed while scoring; the only
exchange is the final all-gather of (B/G) x K x {u64 key, f64 score} blocks, and only when every
rank needs every query's results.

The collective of the product path lives BEHIND the C ABI (ps_comm_*,
ps_snapshot_query_batch_allgather_flat -> ncclAllGather in librccl); `Comm` / `query_batch_sharded`
below are thin callers.  `all_gather_topk` is the torch.distributed form of the same exchange, kept
for hosts that already own a process group (and for the gloo plumbing test on CPU).
"""
import ctypes as C

import numpy as np

from . import _lib


def shard_bounds(n, world, rank):
    """Contiguous shard [lo, hi) of n queries for `rank` of `world` (sizes differ by at most 1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pad_topk(results, top_k):
    """list[list[(key, score)]] -> (keys i64[B*K], scores f64[B*K], counts i32[B]) numpy blocks in
    the device layout of ps_snapshot_query_batch_device (unused slots: key = -1, score = 0)."""
    B = len(results)
    keys = np.full(B * top_k, -1, dtype=np.int64)
    scores = np.zeros(B * top_k, dtype=np.float64)
    counts = np.zeros(B, dtype=np.int32)
    for i, res in enumerate(results):
        res = list(res)[:top_k]
        counts[i] = len(res)
        for k, r in enumerate(res):
            key, score = tuple(r)
            keys[i * top_k + k] = np.int64(np.uint64(key).astype(np.int64))
            scores[i * top_k + k] = score
    return keys, scores, counts


def unpack_topk(keys, scores, counts, top_k):
    """Inverse of pad_topk for host tensors/arrays -> list[list[(key, score)]]."""
    keys = np.asarray(keys).reshape(-1, top_k)
    scores = np.asarray(scores).reshape(-1, top_k)
    counts = np.asarray(counts)
    out = []
    for i in range(len(counts)):
        out.append([(int(np.int64(keys[i, k]).astype(np.uint64)), float(scores[i, k])) for k in range(int(counts[i]))])
    return out


def all_gather_topk(keys, scores, counts, shard_sizes, top_k, group=None):
    """All-gather per-rank top-k blocks (torch tensors, device or host) into global-batch order.

    Ranks may own shards of different sizes (shard_bounds); blocks are padded to the largest
    shard for the collective and trimmed afterwards.  Returns (keys, scores, counts) tensors for
    the whole batch on the same device."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    mx = max(shard_sizes)

    def padded(t, per_query, fill):
        want = mx * per_query
        if t.numel() == want:
            return t.contiguous()
        p = torch.full((want,), fill, dtype=t.dtype, device=t.device)
        p[:t.numel()] = t
        return p

    outs = []
    for t, per_query, fill in ((keys, top_k, -1), (scores, top_k, 0.0), (counts, 1, 0)):
        src = padded(t, per_query, fill)
        dst = torch.empty(world * src.numel(), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(dst, src, group=group)
        parts = [dst[r * src.numel(): r * src.numel() + shard_sizes[r] * per_query] for r in range(world)]
        outs.append(torch.cat(parts))
    return tuple(outs)


class Comm:
    """ps_comm: the RCCL communicator behind the C ABI (include/probly_search_amd.h, ps_comm_*).
    One process per GPU.  The 128-byte id is made by rank 0 and shipped by whatever channel the
    host application has; `from_torch_distributed` uses an initialised torch.distributed group of
    any backend for that single broadcast (the data-path collective itself is ncclAllGather inside
    the library, not torch)."""

    def __init__(self, handle, world, rank, device):
        self._L = _lib.load()
        self._h = handle
        self.world, self.rank, self.device = world, rank, device

    @staticmethod
    def unique_id():
        L = _lib.load()
        buf = C.create_string_buffer(_lib.PS_COMM_ID_BYTES)
        _lib.check(L.ps_comm_get_unique_id(buf))
        return buf.raw

    @classmethod
    def init_rank(cls, uid, world, rank, device):
        L = _lib.load()
        h = C.c_void_p()
        _lib.check(L.ps_comm_init_rank(C.c_char_p(uid), world, rank, device, C.byref(h)))
        return cls(h, world, rank, device)

    @classmethod
    def from_torch_distributed(cls, device, group=None):
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls.init_rank(box[0], world, rank, device)

    @classmethod
    def from_env_store(cls, device, world=None, rank=None, timeout_s=300):
        """The launch contract of `python -m torch.distributed.run` (WORLD_SIZE / RANK / MASTER_ADDR / MASTER_PORT) WITHOUT a
        torch.distributed process group: the 128-byte id travels through the launcher's key-value store (a TCPStore client
        of the agent's store under torchrun, TORCHELASTIC_USE_AGENT_STORE; otherwise rank 0 hosts one on MASTER_PORT), and
        everything else the ranks exchange - barriers, the max of a timing - goes through the library's own communicator
        (`all_gather_bytes`).  One RCCL communicator per process: the library's; no "nccl" group beside it, no gloo either."""
        import os
        world = int(os.environ["WORLD_SIZE"]) if world is None else world
        rank = int(os.environ["RANK"]) if rank is None else rank
        uid, store = cls.exchange_id_via_store(world, rank, timeout_s)
        comm = cls.init_rank(uid, world, rank, device)
        comm._store = store  # rank 0's server must outlive the slowest reader
        comm.barrier()
        return comm

    @classmethod
    def exchange_id_via_store(cls, world, rank, timeout_s=300):
        """Rank 0 makes the id and publishes it in the launcher's store; every rank returns (id, store).  No device involved."""
        import datetime
        import os
        from torch.distributed import TCPStore
        addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"])
        agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() in ("1", "true")
        store = TCPStore(addr, port, world, is_master=(rank == 0 and not agent), timeout=datetime.timedelta(seconds=timeout_s),
                         wait_for_workers=False)
        # (a key per launch attempt: a restarted group must not read the id of the one before it)
        key = "ps_comm_id/%s/%s" % (os.environ.get("TORCHELASTIC_RUN_ID", "run"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
        if rank == 0:
            store.set(key, cls.unique_id())
        return bytes(store.get(key)), store  # (get blocks until rank 0 has set the key)

    def all_gather_bytes(self, payload):
        """Every rank contributes the same number of bytes; returns the `world` payloads in rank order (host bytes).  Through the
        library's communicator (ncclAllGather / the debugging transport), blocking."""
        n = len(payload)
        send, recv = _DeviceBuffer(max(16, n)), _DeviceBuffer(max(16, n) * self.world)
        send.from_host(payload)
        _lib.check(self._L.ps_comm_all_gather(self._h, send.ptr, recv.ptr, max(16, n), None))
        raw = recv.to_host()
        return [raw[r * max(16, n): r * max(16, n) + n] for r in range(self.world)]

    def barrier(self):
        self.all_gather_bytes(b"\0" * 8)

    def max_f64(self, x):
        import struct
        return max(struct.unpack("<d", b)[0] for b in self.all_gather_bytes(struct.pack("<d", float(x))))

    def min_i64(self, x):
        import struct
        return min(struct.unpack("<q", b)[0] for b in self.all_gather_bytes(struct.pack("<q", int(x))))

    def broadcast_bytes(self, payload, src=0, size=None):
        """`payload` of rank `src` on every rank (all ranks pass a payload of the same length, or `size`)."""
        n = len(payload) if size is None else size
        mine = payload if self.rank == src else b"\0" * n
        return self.all_gather_bytes(mine.ljust(n, b"\0"))[src]

    def free(self):
        if self._h:
            self._L.ps_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def block_bytes(n_queries, top_k):
    return _lib.load().ps_topk_block_bytes(n_queries, top_k)


def unpack_blocks(raw, world, n_queries, top_k, shard_sizes):
    """bytes of `world` top-k blocks (ps_topk_block_bytes each) -> list[list[(key, score)]] in
    global-batch order, dropping each shard's padding queries."""
    bb = block_bytes(n_queries, top_k)
    out = []
    for r in range(world):
        blk = raw[r * bb:(r + 1) * bb]
        nk = n_queries * top_k
        keys = np.frombuffer(blk, dtype=np.int64, count=nk, offset=0)
        scores = np.frombuffer(blk, dtype=np.float64, count=nk, offset=nk * 8)
        counts = np.frombuffer(blk, dtype=np.uint32, count=n_queries, offset=nk * 16)
        out.extend(unpack_topk(keys, scores, counts[:shard_sizes[r]], top_k)[:shard_sizes[r]])
    return out


def query_batch_sharded(snapshot, queries, score_calculator, fields_boost, top_k, comm=None):
    """Scores this rank's contiguous shard of `queries` on its GPU (replicated snapshot) and
    all-gathers the top-k blocks through the C ABI (ps_snapshot_query_batch_allgather_flat ->
    ncclAllGather): every rank returns the results of the whole batch.  comm=None: one rank."""
    from . import synth
    from .index import _boosts, _scorer_desc
    L = _lib.load()
    world, rank = (comm.world, comm.rank) if comm is not None else (1, 0)
    sizes = [shard_bounds(len(queries), world, r)[1] - shard_bounds(len(queries), world, r)[0] for r in range(world)]
    per = max(sizes) if sizes else 0
    lo, hi = shard_bounds(len(queries), world, rank)
    mine = list(queries[lo:hi]) + [""] * (per - (hi - lo))  # equal blocks: pad with empty queries
    if per == 0:
        return []
    text, offsets = synth.pack_queries(mine)
    bb = block_bytes(per, top_k)
    local = _DeviceBuffer(bb)
    gathered = _DeviceBuffer(bb * world)
    desc = _scorer_desc(score_calculator)
    b, nb = _boosts(fields_boost)
    # stream NULL: the library scores on the snapshot's own stream, gathers on the communicator's,
    # and returns when both are done
    _lib.check(L.ps_snapshot_query_batch_allgather_flat(snapshot._h, comm._h if comm is not None else None, C.byref(desc),
                                                        text.ctypes.data, offsets.ctypes.data, per, b, nb, None, None,
                                                        top_k, local.ptr, gathered.ptr, None))
    raw = gathered.to_host()
    return unpack_blocks(raw, world, per, top_k, sizes)


class _DeviceBuffer:
    """hipMalloc'd scratch through the HIP runtime the library itself uses (no torch needed)."""
    _hip = None

    @classmethod
    def hip(cls):
        if cls._hip is None:
            _lib.load()
            for name in ("libamdhip64.so.7", "libamdhip64.so"):
                try:
                    cls._hip = C.CDLL(name)
                    break
                except OSError:
                    continue
            if cls._hip is None:
                raise _lib.LibraryNotBuilt("libamdhip64 not found")
            cls._hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            cls._hip.hipFree.argtypes = [C.c_void_p]
            cls._hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        return cls._hip

    def __init__(self, nbytes):
        self.nbytes = nbytes
        p = C.c_void_p()
        if self.hip().hipMalloc(C.byref(p), max(nbytes, 16)) != 0:
            raise MemoryError("hipMalloc(%d) failed" % nbytes)
        self.ptr = p

    def from_host(self, payload):
        if self.hip().hipMemcpy(self.ptr, C.c_char_p(bytes(payload)), len(payload), 1) != 0:  # hipMemcpyHostToDevice
            raise RuntimeError("hipMemcpy H2D failed")

    def to_host(self):
        buf = C.create_string_buffer(self.nbytes)
        if self.hip().hipMemcpy(buf, self.ptr, self.nbytes, 2) != 0:  # hipMemcpyDeviceToHost
            raise RuntimeError("hipMemcpy D2H failed")
        return buf.raw

    def __del__(self):
        try:
            if self.ptr:
                self.hip().hipFree(self.ptr)
                self.ptr = None
        except Exception:
            pass
