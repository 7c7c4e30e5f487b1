mkdir -p gpurun_out/r05e7
python -m pytest tests/test_gpu_gate_edges.py -x -q -m gpu -k "mixed or fresh" 2>&1 | tail -n 4 > gpurun_out/r05e7/t1.log
for o in 32 1024; do python tools/mixed_batch_bench.py --other $o --scorer bm25 > gpurun_out/r05e7/mixed_bm25_$o.jsonl 2>/dev/null; done
python tools/mixed_batch_bench.py --other 1024 > gpurun_out/r05e7/mixed_z_1024.jsonl 2>/dev/null
B="python bench.py --steps 100 --no-cpu-baseline --no-streaming-leg --no-bulk-index --no-alternating-boosts-leg --no-single-latency"
$B --config C3 --q-terms 5 > gpurun_out/r05e7/bench_c3_q5.json 2>gpurun_out/r05e7/bench_c3_q5.err
PS_DAAT_Z=0 $B --config C3 --q-terms 5 --steps 30 > gpurun_out/r05e7/bench_c3_q5_stream.json 2>/dev/null
$B --config C2 --q-terms 5 > gpurun_out/r05e7/bench_c2_q5.json 2>/dev/null
for C in C2 C4; do
  $B --config $C > gpurun_out/r05e7/bench_${C}_base.json 2>/dev/null
  PS_SO=$PWD/probly-search_amd/csrc/alt/libstep2.so $B --config $C > gpurun_out/r05e7/bench_${C}_step2.json 2>gpurun_out/r05e7/bench_${C}_step2.err
  $B --config $C > gpurun_out/r05e7/bench_${C}_base2.json 2>/dev/null
done
cat gpurun_out/r05e7/t1.log gpurun_out/r05e7/mixed_*.jsonl
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05e7/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        u=r['units_processed']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), r['kernel'], round(r['kernel_avg_ms'],4), round(r['kernel_individual_avg_ms'],4), round(r['frac'],3), {k:round(v) for k,v in u.items() if k.startswith('lookups') or k=='postings_scanned'})
    except Exception as e:
        print(f, 'ERR', e)
PY
