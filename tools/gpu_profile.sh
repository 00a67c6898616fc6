#!/bin/bash
# bench + rocprofv3 kernel trace; usage: bash tools/gpu_profile.sh <tag>
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.log 2>&1; echo "bench exit $?"
tail -1 gpurun_out/bench_$TAG.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/prof_$TAG.log 2>&1; echo "prof exit $?"
