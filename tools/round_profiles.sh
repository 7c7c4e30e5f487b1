#!/bin/bash
# The round's evidence set, on the GPU box (one gpurun call): rocprofv3 kernel trace (overlapped + serialised) and PMC passes per
# BASELINE config (tools/profile_bench.sh -> gpurun_out/prof_<cfg>/), the streaming kernel's profile (PS_DAAT=0), and the bench lines.
#   usage: tools/round_profiles.sh <round tag, e.g. r06> <git head>
# Afterwards, here: python tools/collect_profile.py <tag> C2 C3 C4 C5 C2stream; copy gpurun_out/<tag>final/bench_*.json into profiles/.
TAG=$1; HEAD=$2
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for C in C2 C3 C5 C4; do
  bash tools/profile_bench.sh $C $HEAD --config $C > gpurun_out/prof_$C.log 2>&1
done
PS_DAAT=0 bash tools/profile_bench.sh C2stream $HEAD --config C2 > gpurun_out/prof_C2stream.log 2>&1
mkdir -p gpurun_out/${TAG}final
python bench.py --gpus 1 --steps 25 --warmup 5 > gpurun_out/${TAG}final/bench_C2_driver_style.json 2> gpurun_out/${TAG}final/bench_C2_driver_style.err
for C in C2 C3 C5 C1 C4; do
  python bench.py --config $C --steps 100 --warmup 10 > gpurun_out/${TAG}final/bench_$C.json 2> gpurun_out/${TAG}final/bench_$C.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/${TAG}final/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        cb=d.get('cpu_baseline') or {}
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), r['kernel'], 'avg', round(r['kernel_avg_ms'],4), 'busy', round(r.get('kernel_busy_avg_ms',0),4), 'serial', round(r.get('kernel_serial_avg_ms') or 0,4),
              'frac', round(r['frac'],3), 'frac_serial', round(r.get('frac_serial') or 0,3), 'cpu', cb.get('value'), (cb.get('flat') or {}).get('value'), (cb.get('all_cores') or {}).get('value'), 'mism', cb.get('gpu_topk_mismatches_vs_oracle'),
              'upd', (d.get('live_index_updates') or {}).get('ratio'))
    except Exception as e:
        print(f,'ERR',e)
PY
