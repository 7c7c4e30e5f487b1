"""Multi-GPU plumbing for the query path: replicated corpus, query batch sharded across ranks,
per-rank top-k blocks all-gathered with torch.distributed (backend "nccl" == RCCL over xGMI on
ROCm; "gloo" in the CPU tests).

The reference has no distributed mode at all; what makes this sharding natural is that queries
are independent — each `Index::query` call owns its `scores` / `visited` maps and the index is
read-only under `&self` (src/query.rs:21-37).  No collective is needed while scoring; the only
exchange is the final all-gather of (B/G) x K x {u64 key, f64 score} blocks, and only when every
rank needs every query's results.
"""
import numpy as np


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


def query_batch_sharded(snapshot, queries, score_calculator, fields_boost, top_k, device, group=None):
    """Scores this rank's contiguous shard of `queries` on its GPU (replicated snapshot) and
    all-gathers the top-k blocks: every rank returns the results of the whole batch."""
    import torch
    import torch.distributed as dist
    world, rank = (dist.get_world_size(group), dist.get_rank(group)) if dist.is_initialized() else (1, 0)
    sizes = [shard_bounds(len(queries), world, r)[1] - shard_bounds(len(queries), world, r)[0] for r in range(world)]
    lo, hi = shard_bounds(len(queries), world, rank)
    n = hi - lo
    dev = torch.device("cuda", device)
    dk = torch.full((max(n, 1) * top_k,), -1, dtype=torch.int64, device=dev)
    ds = torch.zeros(max(n, 1) * top_k, dtype=torch.float64, device=dev)
    dc = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    if n:
        snapshot.query_batch_device(queries[lo:hi], score_calculator, None, fields_boost, top_k, dk.data_ptr(),
                                    ds.data_ptr(), dc.data_ptr(), stream=stream.cuda_stream)
    if world == 1:
        stream.synchronize()
        return unpack_topk(dk.cpu().numpy(), ds.cpu().numpy(), dc.cpu().numpy()[:n], top_k)
    gk, gs, gc = all_gather_topk(dk[:n * top_k], ds[:n * top_k], dc[:n], sizes, top_k, group)
    return unpack_topk(gk.cpu().numpy(), gs.cpu().numpy(), gc.cpu().numpy(), top_k)
