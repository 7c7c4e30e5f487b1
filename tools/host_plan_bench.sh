#!/bin/bash
# Builds the C2 index on the host, saves its snapshot, and times the batch planner at several pool sizes.
set -e
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
mkdir -p /tmp/ps_plan_bench
cd $R
python - <<'PY'
import sys, time
import probly_search_amd as psa
from probly_search_amd import synth
cfg = dict(synth.CONFIGS["C2"])
corpus = synth.Corpus(**cfg)
idx = psa.Index(cfg["fields"])
synth.fill(idx, corpus)
snap = idx.snapshot(device=-1, tile_docs=1024)
snap.save("/tmp/ps_plan_bench/c2.snap")
with open("/tmp/ps_plan_bench/queries.txt", "w") as f:
    for s in range(64):
        for q in corpus.queries(1024, cfg["q_terms"], salt=s):
            f.write(q + "\n")
PY
g++ -O2 -std=c++17 -pthread -I$R/include $R/tools/host_plan_bench.cpp -o /tmp/ps_plan_bench/hpb -L$R/probly-search_amd/csrc -lprobly_search_amd \
    -Wl,-rpath,$R/probly-search_amd/csrc -Wl,-rpath,/opt/rocm/lib
for t in 1 16 32; do for g in 300; do echo "threads=$t gap=$g"; /tmp/ps_plan_bench/hpb $t $g | tail -2; done; done
