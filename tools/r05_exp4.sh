mkdir -p gpurun_out/r05e4
B="python bench.py --steps 100 --no-cpu-baseline --no-streaming-leg --no-bulk-index --no-alternating-boosts-leg --no-single-latency"
for S in 0 12 16 24; do
  for A in 0 1; do
    PS_SCORE_ALT=$A PS_DAAT_SAMPLE_DIV=$S $B --config C5 > gpurun_out/r05e4/bench_C5_alt${A}_s$S.json 2> gpurun_out/r05e4/bench_C5_alt${A}_s$S.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05e4/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), r['kernel'], round(r['kernel_avg_ms'],4), round(r['kernel_individual_avg_ms'],4), round(r['frac'],3), round(r['units_processed']['postings_scanned']))
    except Exception as e:
        print(f, 'ERR', e)
PY
