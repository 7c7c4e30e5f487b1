cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05tl
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/r05tl/kt -o kt -- python bench.py --config C2 --steps 60 --no-cpu-baseline --no-streaming-leg --no-bulk-index --no-alternating-boosts-leg --no-single-latency > gpurun_out/r05tl/bench.json 2> gpurun_out/r05tl/err.txt
python tools/kernel_timeline.py gpurun_out/r05tl/kt/kt_results.db 4000 > gpurun_out/r05tl/timeline.txt 2>&1
rm -rf gpurun_out/r05tl/kt
wc -l gpurun_out/r05tl/timeline.txt
