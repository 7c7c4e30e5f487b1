B="python bench.py --steps 100 --no-cpu-baseline --no-streaming-leg --no-bulk-index --no-alternating-boosts-leg --no-single-latency"
mkdir -p gpurun_out/r05e15
python -m pytest tests/test_gpu_gate_edges.py tests/test_device_planner.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -n 3
for C in C2 C4; do
  PS_DAAT_SMALL_NL=0 $B --config $C > gpurun_out/r05e15/${C}_nl4.json 2>/dev/null
  $B --config $C > gpurun_out/r05e15/${C}_nl3.json 2>/dev/null
  PS_SO=$PWD/probly-search_amd/csrc/alt/libw5.so $B --config $C > gpurun_out/r05e15/${C}_nl3_w5.json 2>/dev/null
  $B --config $C > gpurun_out/r05e15/${C}_nl3_b.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05e15/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), r['kernel'], 'busy', round(r['kernel_avg_ms'],4), 'submit', round(d['p50_batch_submit_ms'],3))
PY
