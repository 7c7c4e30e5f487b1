# round-5 profile set (GPU box): rocprofv3 kernel trace + PMC passes per config -> gpurun_out/prof_<tag>/ (tools/profile_bench.sh)
for C in C2 C3 C5 C4; do
  bash tools/profile_bench.sh $C 7b720e8 --config $C > gpurun_out/prof_$C.log 2>&1
done
