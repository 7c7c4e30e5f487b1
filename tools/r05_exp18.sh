B="python bench.py --steps 100 --no-cpu-baseline --no-streaming-leg --no-bulk-index --no-alternating-boosts-leg --no-single-latency"
mkdir -p gpurun_out/r05e18
for C in C2 C3; do
  $B --config $C > gpurun_out/r05e18/${C}_p1.json 2>/dev/null
  PS_PREP_STREAMS=2 $B --config $C > gpurun_out/r05e18/${C}_p2.json 2>/dev/null
  $B --config $C > gpurun_out/r05e18/${C}_p1_b.json 2>/dev/null
  PS_PREP_STREAMS=2 $B --config $C > gpurun_out/r05e18/${C}_p2_b.json 2>/dev/null
done
PS_PREP_STREAMS=2 PS_DCTX=8 $B --config C2 --plan-ahead-depth 3 > gpurun_out/r05e18/C2_p2_d3_x8.json 2>/dev/null
PS_PREP_STREAMS=2 $B --config C5 > gpurun_out/r05e18/C5_p2.json 2>/dev/null
PS_PREP_STREAMS=2 python -m pytest tests/test_gpu_gate_edges.py tests/test_device_planner.py tests/test_gpu_z21_daat.py -x -q -m gpu 2>&1 | tail -n 2
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05e18/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), 'busy', round(r['kernel_avg_ms'],4), 'submit', round(d['p50_batch_submit_ms'],3))
PY
