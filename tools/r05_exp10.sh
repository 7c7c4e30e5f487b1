mkdir -p gpurun_out/r05e10
B="python bench.py --steps 100 --no-cpu-baseline --no-streaming-leg --no-bulk-index --no-alternating-boosts-leg --no-single-latency"
for D in 1 2 3; do for X in 5 6 8; do
  PS_DCTX=$X $B --plan-ahead-depth $D > gpurun_out/r05e10/c2_d${D}_x$X.json 2>gpurun_out/r05e10/c2_d${D}_x$X.err
done; done
for C in C3 C4 C5; do PS_DCTX=6 $B --config $C --plan-ahead-depth 2 > gpurun_out/r05e10/${C}_d2_x6.json 2>/dev/null; done
python -m pytest tests/test_gpu_z21_daat.py tests/test_device_planner.py -x -q -m gpu -k "flight or ahead or announced or planned" 2>&1 | tail -n 3
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05e10/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), round(r['kernel_avg_ms'],4), round(r['kernel_individual_avg_ms'],4), round(r['frac'],3), 'submit', round(d['p50_batch_submit_ms'],3), 'hostwait', round(d['host_plan_ms_per_step'],3))
    except Exception as e:
        print(f,'ERR',e)
PY
