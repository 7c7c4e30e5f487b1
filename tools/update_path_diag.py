#!/usr/bin/env python3
"""Where a ps_snapshot_update cycle's time goes (C2): step time, scoring kernel and work counters per launch on a static snapshot, after
1000 removals, after 1000 additions, and with the split / priming / sample phase switched off (profiles/r06_update_path.txt)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PS_ROW_CACHE_MB", "0")
import numpy as np, torch
import probly_search_amd as psa
from probly_search_amd import dist as psd, synth
cfg = dict(synth.CONFIGS["C2"]); F, K, B = 2, 10, 1024
corpus = synth.Corpus(**cfg)
idx = synth.fill(psa.Index(F), corpus)
snap = idx.snapshot(device=0, headroom_pct=5)
sc = psa.bm25.new()
packed = [synth.pack_queries(corpus.queries(B, 3, salt=500 + s)) for s in range(24)]
buf = torch.zeros(psd.block_bytes(B, K) // 8, dtype=torch.int64, device="cuda"); base = buf.data_ptr(); st = torch.cuda.Stream()
def run(tag, n=20):
    for i in range(3): snap.query_batch_device_flat(*packed[i], sc, [1.0]*F, K, base, base + 8*B*K, base + 16*B*K, stream=st.cuda_stream)
    st.synchronize(); snap.kernel_breakdown(reset=True); snap.work_counters(reset=True)
    t0 = time.perf_counter()
    for i in range(n): snap.query_batch_device_flat(*packed[i % 24], sc, [1.0]*F, K, base, base + 8*B*K, base + 16*B*K, stream=st.cuda_stream)
    st.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    kt = snap.kernel_breakdown(reset=True); w = snap.work_counters(reset=True); nl = max(1, int(kt["launches"]))
    print(tag, "step %.3f ms" % dt, kt["score_kernel"], "busy %.3f" % (kt["score_busy_ms"]/nl), {k: round(w[k]/nl) for k in ("items","items_run","postings_scanned","postings_reached_lookups","lookups_row","lookups_cell","lookups_probe")}, flush=True)
run("static")
n_upd = 1000
rng = np.random.default_rng(7); victims = rng.choice(cfg["n_docs"], size=4*n_upd, replace=False)
for k in victims[:n_upd]: idx.remove_document(int(k))
print(snap.update()["mode"]); run("after 1000 removals")
fresh = synth.Corpus(**dict(cfg, n_docs=2*n_upd, seed=cfg["seed"] ^ 0xABCDEF)); ch = list(fresh.chunks(n_upd))
keys, text, offs = ch[0]; idx.add_documents_flat(keys + np.uint64(cfg["n_docs"]), text, offs)
print(snap.update()["mode"]); run("after 1000 additions")
for knob, v in ((b"PS_DAAT_SPLIT", 0), (b"PS_DAAT_PRIME", 0)):
    psa.load().ps_set_option(knob, v); run("  with %s=%d" % (knob.decode(), v)); psa.load().ps_set_option(knob, 1)
os.environ["PS_EXPERIMENT_KNOBS"] = "1"
psa.load().ps_set_option(b"PS_DAAT_SAMPLE_DIV", 0); run("  with PS_DAAT_SAMPLE_DIV=0"); psa.load().ps_set_option(b"PS_DAAT_SAMPLE_DIV", 24)
