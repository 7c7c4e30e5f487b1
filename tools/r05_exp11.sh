mkdir -p gpurun_out/r05e11
B="python bench.py --steps 100 --no-cpu-baseline --no-streaming-leg --no-bulk-index --no-alternating-boosts-leg --no-single-latency"
python -m pytest tests/test_gpu_gate_edges.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -n 4 > gpurun_out/r05e11/t1.log
for C in C2 C4; do
  $B --config $C > gpurun_out/r05e11/${C}_words_a.json 2>/dev/null
  PS_SO=$PWD/probly-search_amd/csrc/alt/libplane.so $B --config $C > gpurun_out/r05e11/${C}_plane.json 2>/dev/null
  $B --config $C > gpurun_out/r05e11/${C}_words_b.json 2>/dev/null
done
cat gpurun_out/r05e11/t1.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05e11/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), 'devonly', round(d['ms_per_step_device_only'],4), 'busy', round(r['kernel_avg_ms'],4), round(r['kernel_individual_avg_ms'],4))
    except Exception as e:
        print(f,'ERR',e)
PY
