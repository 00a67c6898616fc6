#!/bin/bash
# rocprofv3 kernel stats of the bench under an environment setting; usage: bash tools/gpu_prof_env.sh <tag> [VAR=val ...] [-- bench args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
while [ $# -gt 0 ] && [ "$1" != "--" ]; do export "$1"; shift; done
[ "$1" == "--" ] && shift
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pe_$TAG -o $TAG -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer "$@" > $R/gpurun_out/pe_$TAG.log 2>&1 || exit 1
echo ok $TAG
