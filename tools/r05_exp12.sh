python -m pytest tests/test_gpu_z21_daat.py -x -q -m gpu 2>&1 | tail -n 3
B="python bench.py --steps 100 --no-cpu-baseline --no-streaming-leg --no-bulk-index --no-alternating-boosts-leg --no-single-latency"
$B --config C3 --q-terms 5 > gpurun_out/r05_c3_q5_nobtab.json 2>/dev/null
$B --config C3 --q-terms 8 > gpurun_out/r05_c3_q8.json 2>/dev/null
python - <<'PY'
import json
for f in ["gpurun_out/r05_c3_q5_nobtab.json","gpurun_out/r05_c3_q8.json"]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, round(d["value"]), round(d["ms_per_step"],4), r["kernel"], round(r["kernel_avg_ms"],4))
PY
