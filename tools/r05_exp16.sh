mkdir -p gpurun_out/r05e16
B="python bench.py --steps 100 --no-cpu-baseline --no-streaming-leg --no-bulk-index --no-alternating-boosts-leg --no-single-latency"
for D in 1 2 3; do for X in 5 8; do
  PS_DCTX=$X $B --plan-ahead-depth $D > gpurun_out/r05e16/c2_d${D}_x$X.json 2>/dev/null
done; done
PS_DCTX=8 $B --plan-ahead-depth 3 --resident-rows > gpurun_out/r05e16/c2_d3_x8_resident.json 2>/dev/null
PS_SCORE_ALT=0 $B > gpurun_out/r05e16/c2_alt0.json 2>/dev/null
PS_SCORE_ALT=4 PS_DCTX=8 $B --plan-ahead-depth 3 > gpurun_out/r05e16/c2_alt4_d3_x8.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05e16/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), 'busy', round(r['kernel_avg_ms'],4), 'submit', round(d['p50_batch_submit_ms'],3), 'hostwait', round(d['host_plan_ms_per_step'],3))
PY
