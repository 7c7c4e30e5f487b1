mkdir -p gpurun_out/r05e2
for C in C2 C5 C3 C4; do
  for A in 0 1 4; do
    PS_SCORE_ALT=$A python bench.py --config $C --steps 100 --no-cpu-baseline --no-streaming-leg --no-bulk-index --no-alternating-boosts-leg --no-single-latency > gpurun_out/r05e2/bench_${C}_alt$A.json 2> gpurun_out/r05e2/bench_${C}_alt$A.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05e2/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), r['kernel'], round(r['kernel_avg_ms'],4), round(r['kernel_individual_avg_ms'],4), round(r['frac'],3))
    except Exception as e:
        print(f, 'ERR', e)
PY
