#!/bin/bash
# Runs on the GPU box: tools/fetch_size_calibration.hip under rocprofv3 PMC passes (each its own run),
# then tools/fetch_size_calibration.py divides the counters by the known byte counts.
#   usage: tools/fetch_size_calibration.sh [GiB of array] [loads per random pattern]
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/fetch_cal
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/fetch_size_calibration.hip -o $OUT/fetch_cal || exit 1
$OUT/fetch_cal "$@" > $OUT/known.jsonl || exit 1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  name=$(echo $pass | tr ' ' '+')
  timeout 300 rocprofv3 --pmc $pass -d $OUT/p_$name -o p -- $OUT/fetch_cal "$@" > $OUT/p_$name.out 2> $OUT/p_$name.err
  echo "$name rc=$?"
done
python tools/fetch_size_calibration.py $OUT | tee $OUT/calibration.txt
rm -f $OUT/fetch_cal
