#!/bin/bash
# kernel trace of the hipGraph-replayed step at 512 graphs (the per-GPU share of configs[2]): how much of the
# step is kernel time and how much is gaps between the ~90 launches.  usage: bash tools/gpu_profile_replay.sh <tag>
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profr_$TAG -- python3 $R/bench.py --batch 512 --graph on --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/profr_$TAG.log 2>&1; echo "prof exit $?"
tail -1 $R/gpurun_out/profr_$TAG.log | cut -c1-260
