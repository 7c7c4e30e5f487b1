# round-5 experiment 3 (GPU box): contexts in the rotation under PS_SCORE_ALT=1, the sample-first item phase, a kernel timeline
mkdir -p gpurun_out/r05e3
B="python bench.py --steps 100 --no-cpu-baseline --no-streaming-leg --no-bulk-index --no-alternating-boosts-leg --no-single-latency"
for C in C2 C3 C5 C4; do
  for D in 5 6 8; do
    PS_SCORE_ALT=1 PS_DCTX=$D $B --config $C > gpurun_out/r05e3/bench_${C}_alt1_dctx$D.json 2> gpurun_out/r05e3/bench_${C}_alt1_dctx$D.err
  done
done
for C in C5 C2 C4; do
  python tools/knob_sweep.py --config $C --steps 60 PS_DAAT_SAMPLE_DIV=0,4,8,16,32 > gpurun_out/r05e3/sample_$C.jsonl 2> gpurun_out/r05e3/sample_$C.err
done
PS_DAAT_SAMPLE_DIV=8 PS_SCORE_ALT=1 PS_DCTX=8 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05e3/parity.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PS_SCORE_ALT=1 PS_DCTX=8 timeout 300 rocprofv3 --kernel-trace -d gpurun_out/r05e3/kt -o kt -- $B --config C2 --steps 30 > gpurun_out/r05e3/kt.bench.json 2> gpurun_out/r05e3/kt.err
python tools/kernel_timeline.py gpurun_out/r05e3/kt/kt_results.db 400 > gpurun_out/r05e3/timeline_c2_alt1.txt 2>&1
rm -rf gpurun_out/r05e3/kt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05e3/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), r['kernel'], round(r['kernel_avg_ms'],4), round(r['kernel_individual_avg_ms'],4), round(r['frac'],3))
    except Exception as e:
        print(f, 'ERR', e)
for f in sorted(glob.glob('gpurun_out/r05e3/sample_*.jsonl')):
    for l in open(f):
        d=json.loads(l); p=d['per_launch']
        print(f.split('/')[-1], d['leg'], d['kernel_avg_ms'], d['kernel_busy_ms'], d['step_ms'], p['items'], p['items_run'], p['postings_scanned'])
PY
cat gpurun_out/r05e3/parity.log
