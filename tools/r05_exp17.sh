B="python bench.py --steps 100 --no-cpu-baseline --no-streaming-leg --no-bulk-index --no-alternating-boosts-leg --no-single-latency --config C3"
mkdir -p gpurun_out/r05e17
python -m pytest tests/test_gpu_z21_daat.py -x -q -m gpu 2>&1 | tail -n 2
PS_DAAT_SMALL_NL=0 $B > gpurun_out/r05e17/zn4.json 2>/dev/null
$B > gpurun_out/r05e17/zn3.json 2>/dev/null
PS_DAAT_SMALL_NL=0 $B > gpurun_out/r05e17/zn4_b.json 2>/dev/null
$B > gpurun_out/r05e17/zn3_b.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05e17/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), r['kernel'], 'busy', round(r['kernel_avg_ms'],4))
PY
