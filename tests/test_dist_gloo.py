"""N>1 path on CPU: world_size-2 gloo processes shard a query batch, each rank produces its
shard's top-k blocks (the oracle stands in for the GPU kernel here — this test is about the
sharding / all-gather plumbing), all-gather, and every rank must hold the unsharded answer."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_queries, top_k, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import probly_search_amd as psa  # noqa: F401  (package import must work without a GPU)
        from probly_search_amd import dist as psd, synth
        from oracle import oracle as orc
        cfg = dict(synth.CONFIGS["C2"], n_docs=2000, vocab=300)
        corpus = synth.Corpus(**cfg)
        o = synth.fill(orc.Index(2), corpus)  # replicated corpus: every rank builds the same index
        queries = corpus.queries(n_queries, 3)
        lo, hi = psd.shard_bounds(n_queries, world, rank)
        local = [o.query(q, orc.bm25(), [1.0, 1.0])[:top_k] for q in queries[lo:hi]]
        k, s, c = psd.pad_topk(local, top_k)
        sizes = [psd.shard_bounds(n_queries, world, r)[1] - psd.shard_bounds(n_queries, world, r)[0]
                 for r in range(world)]
        gk, gs, gc = psd.all_gather_topk(torch.from_numpy(k), torch.from_numpy(s), torch.from_numpy(c), sizes, top_k)
        got = psd.unpack_topk(gk.numpy(), gs.numpy(), gc.numpy(), top_k)
        exp = [o.query(q, orc.bm25(), [1.0, 1.0])[:top_k] for q in queries]
        ret[rank] = got == exp
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_queries", [8, 7])  # even and ragged shards
def test_sharded_batch_allgather_world2(n_queries):
    world, top_k = 2, 5
    port = 29500 + (os.getpid() % 2000) + n_queries
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, n_queries, top_k, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}


def test_shard_bounds_cover_batch_exactly():
    sys.path.insert(0, ROOT)
    from probly_search_amd import dist as psd
    for n in (0, 1, 7, 1024, 8191):
        for world in (1, 2, 3, 8):
            spans = [psd.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
