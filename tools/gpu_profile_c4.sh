#!/bin/bash
# kernel trace of the sparse-regime configuration (BASELINE configs[3]); usage: bash tools/gpu_profile_c4.sh <tag>
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profc4_$TAG -- python3 $R/bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/profc4_$TAG.log 2>&1; echo "prof exit $?"
tail -1 $R/gpurun_out/profc4_$TAG.log | cut -c1-200
