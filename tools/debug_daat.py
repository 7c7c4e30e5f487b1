"""GPU debugging aid: a small BM25 top-k batch through K1d with stage traces, compared with the oracle."""
import faulthandler
import os
import sys

faulthandler.dump_traceback_later(40, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import probly_search_amd as psa  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from probly_search_amd import synth  # noqa: E402

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 9
cfg = dict(synth.CONFIGS["C2"], n_docs=n_docs, vocab=400)
corpus = synth.Corpus(**cfg)
p, o = synth.fill(psa.Index(2), corpus), synth.fill(orc.Index(2), corpus)
snap = p.snapshot(device=0, tile_docs=256)
queries = corpus.queries(B, 3)
print("snapshot ok", snap.info(), flush=True)
got = snap.query_batch(queries, psa.bm25.new(), None, [1.0, 1.0], top_k=10)
print("batch ok", snap.kernel_breakdown(), flush=True)
bad = 0
for q, g in zip(queries, got):
    exp = o.query(q, orc.bm25(), [1.0, 1.0])[:10]
    if [(r.key, r.score) for r in g] != exp:
        bad += 1
        print("MISMATCH", q, [(r.key, r.score) for r in g][:3], exp[:3])
print("mismatches", bad, "of", B)
