"""Experiment: R engine replicas of one index on ONE GPU, batches alternated over R streams
(each replica has its own per-batch buffers, so consecutive batches overlap on the device)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PS_ROW_CACHE_MB"] = "0"
import numpy as np, torch
import probly_search_amd as psa
from probly_search_amd import dist as psd, synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfgname = sys.argv[2] if len(sys.argv) > 2 else "C2"
steps = 40
cfg = dict(synth.CONFIGS[cfgname])
corpus = synth.Corpus(**cfg)
index = synth.fill(psa.Index(cfg["fields"]), corpus)
snaps = [index.snapshot(device=0) for _ in range(R)]
B, K = 1024, 10
packed = [synth.pack_queries(corpus.queries(B, cfg["q_terms"], salt=s)) for s in range(steps + 5)]
bb = psd.block_bytes(B, K)
blocks = [torch.zeros(bb // 8, dtype=torch.int64, device="cuda") for _ in range(R)]
streams = [torch.cuda.Stream() for _ in range(R)]
sc = psa.bm25.new()
def step(i):
    r = i % R
    t, o = packed[i]
    base = blocks[r].data_ptr()
    snaps[r].query_batch_device_flat(t, o, sc, [1.0, 1.0], K, base, base + 8 * B * K, base + 16 * B * K, stream=streams[r].cuda_stream)
for i in range(5): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(5, steps + 5): step(i)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("replicas", R, cfgname, "qps", round(B * steps / el), "ms/step", round(el / steps * 1e3, 3))
