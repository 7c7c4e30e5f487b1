#!/bin/bash
# A/B builds of the library with other build-time knobs: tools/build_variant.sh <name> <extra hipcc flags...>
# -> probly-search_amd/csrc/alt/lib<name>.so (select it with PS_SO=<path>; *.so is git-ignored but travels with gpurun).
set -e
NAME=$1; shift
cd "$(dirname "$0")/../probly-search_amd/csrc"
mkdir -p alt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result "$@" -c ps_engine.hip -o alt/ps_engine_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o alt/lib$NAME.so ps_index.o ps_snapshot.o ps_capi.o ps_keytable.o alt/ps_engine_$NAME.o ps_sort.o ps_comm.o ps_build.o -pthread -ldl -lrt
rm -f alt/ps_engine_$NAME.o
echo built alt/lib$NAME.so
