mkdir -p gpurun_out/r05final
for C in C2 C3 C5 C1 C4; do
  python bench.py --config $C > gpurun_out/r05final/bench_$C.json 2> gpurun_out/r05final/bench_$C.err
done
python bench.py --config C2 --host-plan --no-cpu-baseline --no-bulk-index > gpurun_out/r05final/bench_C2_host_plan.json 2>/dev/null
PS_SCORE_ALT=0 python bench.py --config C2 --no-cpu-baseline --no-bulk-index --no-streaming-leg > gpurun_out/r05final/bench_C2_alt0.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05final/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        cb=d.get('cpu_baseline') or {}
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), r['kernel'], round(r['kernel_avg_ms'],4), round(r['kernel_individual_avg_ms'],4), 'frac', round(r['frac'],3), 'req', round((r.get('request_roofline') or {}).get('frac',0),3), 'traffic', r.get('traffic'), 'cpu', cb.get('value'), (cb.get('all_cores') or {}).get('value'), 'mism', cb.get('gpu_topk_mismatches_vs_oracle'))
    except Exception as e:
        print(f,'ERR',e)
PY
