#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of bench.py (each pass its own
# run and its own timeout; counters never share a run with a trace domain), text summaries into
# gpurun_out/prof_<tag>/, then tools/derive_roofline.py turns them into the derivation JSON that
# bench.py prices its roofline against (copy both into profiles/ to have them judged).
#   usage: tools/profile_bench.sh <tag> <git head> [bench args...]
set -u
TAG=$1; shift
HEAD=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
BENCH="python bench.py --no-cpu-baseline --no-single-latency --no-bulk-index --no-streaming-leg --no-alternating-boosts-leg --no-update-leg $*"
echo "rocprofv3 <pass> -- $BENCH --steps 6 --warmup 2" > $OUT/command.txt
run() {  # name, rocprof args...
  local name=$1; shift
  timeout 300 rocprofv3 "$@" -d $OUT/$name -o $name -- $BENCH --steps 6 --warmup 2 > $OUT/$name.bench.json 2> $OUT/$name.err
  echo "$name rc=$?"
  python tools/rocpd_summary.py $OUT/$name/${name}_results.db 100000 > $OUT/$name.txt 2>&1
  rm -rf $OUT/$name
}
run kt --kernel-trace --stats
# the same kernel trace with scoring serialised (one scoring queue): a kernel's duration in THIS trace is what it costs the chip,
# so bytes touched / avg_us here is a roofline fraction a reader can recompute from the summary alone (derive.log, frac_serial_from_trace)
PS_SCORE_ALT=0 run kt_serial --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run pmc_sq2 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run pmc_sq3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
python tools/derive_roofline.py $OUT "$TAG" "$HEAD" $* > $OUT/derive.log 2>&1
cat $OUT/derive.log | tail -30
