#!/bin/bash
# rocprofv3 kernel stats of the bench with the product library and with a variant library, back to back on one box;
# usage: bash tools/gpu_ab_variant.sh <variant-name> [bench args]
V=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_base -o base -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer "$@" > $R/gpurun_out/ab_base.log 2>&1 || exit 1
export GNM_HIP_LIB=$R/graph-neural-mapping_amd/lib/variants/$V.so
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$V -o $V -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer "$@" > $R/gpurun_out/ab_$V.log 2>&1 || exit 1
echo ok
