B="python bench.py --steps 100 --no-cpu-baseline --no-streaming-leg --no-bulk-index --no-alternating-boosts-leg --no-single-latency --config C2"
mkdir -p gpurun_out/r05e14
$B > gpurun_out/r05e14/a.json 2>/dev/null
PS_SO=$PWD/probly-search_amd/csrc/alt/libno2.so $B > gpurun_out/r05e14/no2.json 2>/dev/null
$B > gpurun_out/r05e14/b.json 2>/dev/null
PS_SO=$PWD/probly-search_amd/csrc/alt/libno2.so $B > gpurun_out/r05e14/no2b.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05e14/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), 'busy', round(r['kernel_avg_ms'],4))
PY
