#!/usr/bin/env python3
"""bench.py — queries/sec of the BM25 / zero-to-one query-scoring hot path on MI355X.

A "step" = one pass of the hot path (host plan -> K1/K2 posting accumulate -> K3 top-k merge)
over one batch of synthetic queries; the corpus snapshot is already resident in HBM when the
timed region starts.  N=1 workload: BASELINE.json configs[1] (C2: 1M docs, 2 fields, 1024-query
BM25 batch, top-10).  N>1: the corpus is replicated, every rank scores its own 1024-query shard
of an N*1024 global batch (weak scaling) and the per-rank top-k blocks are all-gathered over
RCCL (torch.distributed backend "nccl"), inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` for the
dominant kernel (k_bm25 / k_z21; HIP-event timed inside the library on the launch stream) and
`cpu_baseline` (the reference-faithful C++ restatement in oracle/, timed on this box's host
cores on a bounded sample of the same batch; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--resident-rows", action="store_true",
                    help="keep dense score rows resident across steps (the library's default); without it "
                         "every step rebuilds its rows inside the timed region (the conservative, reported number)")
    ap.add_argument("--no-single-latency", action="store_true",
                    help="skip the synchronous single-query latency loop (profiling runs: only batch launches)")
    ap.add_argument("--config", default="C2", help="C1..C5 (SURVEY.md App. C); C2 is the headline config")
    ap.add_argument("--n-docs", type=int, default=0, help="override the config's corpus size (debug only)")
    ap.add_argument("--batch", type=int, default=0, help="override queries per rank per step")
    ap.add_argument("--tile-docs", type=int, default=0)
    ap.add_argument("--cpu-queries", type=int, default=24, help="bounded CPU-baseline sample (queries)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scorer", default="", help="override: bm25 | zero_to_one")
    args = ap.parse_args()

    import numpy as np
    import torch
    import probly_search_amd as psa
    from probly_search_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # debugging aid for 1-GPU boxes: every rank on device 0, collectives over gloo (exercises the
    # N>1 code path end to end; the numbers it prints are not a multi-GPU measurement)
    debug_1gpu = os.environ.get("PS_BENCH_DEBUG_ONE_GPU") == "1"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if debug_1gpu:
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world
    dev = local_rank if (world > 1 and not debug_1gpu) else 0
    torch.cuda.set_device(dev)

    # Dense score rows of hot lists depend on the index and the scorer parameters only, and the library
    # keeps them resident across batches.  The headline number does NOT lean on that: unless asked,
    # the row slab is disabled so each timed step builds the rows it uses (K0b inside the timed region).
    if not args.resident_rows:
        os.environ["PS_ROW_CACHE_MB"] = "0"
    cfg = dict(synth.CONFIGS[args.config])
    if args.n_docs:
        cfg["n_docs"] = args.n_docs
    if args.scorer:
        cfg["scorer"] = args.scorer
    # per-rank batch: the config's batch at N=1; C4's 8192-query batch is 1024 per GPU at 8 GPUs
    B = args.batch or min(cfg["batch"], 1024)
    K = cfg["top_k"]
    F = cfg["fields"]
    boosts = [1.0] * F
    scorer = psa.bm25.new() if cfg["scorer"] == "bm25" else psa.zero_to_one.new()

    t0 = time.time()
    corpus = synth.Corpus(**cfg)
    index = psa.Index(F)
    t_index = 0.0  # inside the library (tokenise + trie + postings); the rest of the loop is synthetic text generation
    for keys, text, offsets in corpus.chunks(100_000):
        ta = time.time()
        index.add_documents_flat(keys, text, offsets)
        t_index += time.time() - ta
    t_generate = time.time() - t0 - t_index
    t0 = time.time()
    # zero_to_one keeps F accumulator planes per tile in LDS: a smaller tile keeps occupancy up
    tile_docs = args.tile_docs or (512 if cfg["scorer"] == "zero_to_one" else 0)
    snap = index.snapshot(device=dev, tile_docs=tile_docs)
    t_snap = time.time() - t0
    info = snap.info()

    # global batch of step s = queries(world*B, salt=s); this rank scores the contiguous shard rank*B..
    def shard(step):
        q = corpus.queries(world * B, cfg["q_terms"], salt=step)
        return q[rank * B:(rank + 1) * B]

    n_total = args.warmup + args.steps
    batches = [shard(s) for s in range(n_total)]
    # the batch as it reaches the C ABI: one contiguous UTF-8 buffer + offsets (no per-query
    # marshalling inside the timed region; a Rust/C caller would hand over exactly this)
    packed = [synth.pack_queries(b) for b in batches]

    # one device block per rank: [B*K keys u64 | B*K scores f64 | B counts (u32, in 8-byte slots)],
    # so the multi-GPU exchange is a single all-gather.  Two blocks alternate: the all-gather of
    # step s runs on RCCL's stream while the GPU already scores step s+1 into the other block.
    n_blk = 2 if world > 1 else 1
    blocks = [torch.zeros(2 * B * K + B, dtype=torch.int64, device="cuda") for _ in range(n_blk)]
    if world > 1:
        gathered = [torch.zeros(world * blocks[0].numel(), dtype=torch.int64, device="cuda") for _ in range(n_blk)]
    works = [None] * n_blk
    # a real (non-null) stream: the library then only enqueues and returns, so the host plans
    # batch s+1 while the GPU scores batch s; torch/RCCL work is ordered against the same stream
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)

    def step(batch, i):
        text, offsets = batch
        slot = i % n_blk
        if works[slot] is not None:  # the block's previous all-gather must have read it
            works[slot].wait()
            works[slot] = None
        block = blocks[slot]
        base = block.data_ptr()
        snap.query_batch_device_flat(text, offsets, scorer, boosts, K, base, base + 8 * B * K, base + 16 * B * K,
                                     stream=stream.cuda_stream)
        if world > 1:  # top-k all-gather over xGMI only when the batch spans >1 GPU
            works[slot] = dist.all_gather_into_tensor(gathered[slot], block, async_op=True)

    def fence():
        for slot in range(n_blk):
            if works[slot] is not None:
                works[slot].wait()
                works[slot] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        step(packed[s], s)
    fence()
    snap.kernel_times(reset=True)
    postings = 0
    layout_bytes = 0
    dense_rows = 0
    dense_built = 0
    plan_ms = 0.0
    lat = []
    t_start = time.perf_counter()
    for s in range(args.warmup, n_total):
        ts = time.perf_counter()
        step(packed[s], s)
        st = snap.last_stats()
        postings += st["postings_visited"]
        layout_bytes += st["layout_bytes"]
        dense_rows += st["dense_rows"]
        dense_built += st["dense_rows_built"]
        plan_ms += st["plan_ms"]
        lat.append(time.perf_counter() - ts)
    fence()
    elapsed = time.perf_counter() - t_start
    k_total_ms, k_launches = snap.kernel_times(reset=False)
    if world > 1:
        # the exchange really happened: this rank's slice of the last gathered buffer is its own block
        last = (n_total - 1) % n_blk
        n_el = blocks[last].numel()
        assert torch.equal(gathered[last][rank * n_el:(rank + 1) * n_el], blocks[last]), "all-gather mismatch"
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        steps = args.steps
        qps = world * B * steps / elapsed
        alg_bytes_launch = (postings / max(1, steps)) * (4 + 8 * F) + B * K * 16
        k_avg_ms = k_total_ms / max(1, k_launches)
        achieved_alg = alg_bytes_launch / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
        # bytes of the layout the kernels actually streamed (dense score rows are an 8 B/document
        # stream, narrower than 20 B/posting: SURVEY 8d says to price against what is really read)
        layout_bytes_launch = layout_bytes / max(1, steps)
        achieved = layout_bytes_launch / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
        # latency views: p50 of host-side step submission, and of a synchronous single query
        single = []
        pool = [] if args.no_single_latency else [q for b in batches for q in b][:220]
        while pool and len(pool) < 220:
            pool += pool
        for q in pool[:20]:  # warm the synchronous path (its stream, result buffer, clocks)
            snap.query(q, scorer, None, boosts, top_k=K)
        for q in pool[20:220]:
            ts = time.perf_counter()
            snap.query(q, scorer, None, boosts, top_k=K)
            single.append(time.perf_counter() - ts)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
        if os.path.exists(tpath):  # PMC pass of this same command, committed under profiles/
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "queries/sec, %s over %d-doc/%d-field index (top-%d, %d-query batches)" % (
                cfg["scorer"], cfg["n_docs"], F, K, B),
            "value": qps, "unit": "queries/s", "n_gpus": n_gpus, "steps": steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" + (" [DEBUG: all ranks on one GPU]" if debug_1gpu else ""),
            "config": {"workload": "%s: %d synthetic docs, %d fields, Zipf(s=%.1f) over %d stems x %d variants, "
                                   "%d-query %s batch per GPU, %d terms/query, top-%d" % (
                                       args.config, cfg["n_docs"], F, cfg["zipf_s"], cfg["vocab"], cfg["variants"], B,
                                       cfg["scorer"], cfg["q_terms"], K),
                       "global_batch": world * B, "parallelism": "replicated corpus, query batch sharded x%d" % world,
                       "tile_docs": info["tile_docs"], "postings": info["n_postings"], "pointers": info["n_pointers"]},
            "p50_single_query_ms": float(np.median(single) * 1e3) if single else None,
            "p50_batch_submit_ms": float(np.median(lat) * 1e3),
            "host_plan_ms_per_step": plan_ms / steps,
            "postings_per_step": postings / steps,
            "index_build_s": t_index, "corpus_generation_s": t_generate, "snapshot_s": t_snap, "hbm_resident_bytes": info["device_bytes"],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_bm25" if cfg["scorer"] == "bm25" else "k_z21",
                         "kernel_avg_ms": k_avg_ms, "launches": int(k_launches),
                         "bytes_per_launch": layout_bytes_launch,
                         "basis": "bytes of the layout actually streamed by the timed kernels (20 B postings, "
                                  "8 B/doc dense score rows incl. building the non-resident ones, 16 B results)",
                         "algorithmic_bytes_per_launch": alg_bytes_launch,
                         "achieved_algorithmic": achieved_alg, "frac_algorithmic": achieved_alg / HBM_PEAK_GBS,
                         "dense_rows_per_launch": dense_rows / max(1, steps),
                         "dense_rows_built_per_launch": dense_built / max(1, steps),
                         "rows_resident_across_steps": bool(args.resident_rows)},
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = [q for b in batches[args.warmup:] for q in b][:args.cpu_queries if B > 1 else 1000]
            result["cpu_baseline"] = cpu_baseline(cfg, corpus, sample, boosts, snap, scorer, K)
    fence()
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(cfg, corpus, sample, boosts, snap, scorer, K):
    """Times the oracle (reference-faithful single-threaded C++ restatement) on the first n_sample
    queries of the first timed batch, 1 thread (the reference's execution model) and all cores
    (one query per thread over the shared read-only index); cross-checks the GPU top-k on them."""
    from oracle import oracle as orc
    from probly_search_amd import synth
    t0 = time.time()
    o = synth.fill(orc.Index(cfg["fields"]), corpus)
    t_build = time.time() - t0
    osc = orc.bm25() if cfg["scorer"] == "bm25" else orc.zero_to_one()
    wall1, secs1, nres, top = o.bench_queries(sample, osc, boosts, threads=1, top_k=K)
    cores = os.cpu_count() or 1
    wallN, secsN, _, _ = o.bench_queries(sample, osc, boosts, threads=cores, top_k=0)
    got = snap.query_batch(sample, scorer, None, boosts, top_k=K)
    mism = sum(1 for g, e in zip(got, top) if [(r.key, r.score) for r in g] != e)
    import numpy as np
    return {"value": len(sample) / wall1, "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": "first %d queries of the timed batches, full-result Index::query per query, "
                      "oracle/probly_oracle.cpp (-O2), single thread" % len(sample),
            "p50_query_ms": float(np.median(secs1) * 1e3),
            "all_cores": {"value": len(sample) / wallN, "cores": cores},
            "mean_results_per_query": float(np.mean(nres)), "oracle_index_build_s": t_build,
            "gpu_topk_mismatches_vs_oracle": mism}


if __name__ == "__main__":
    main()
