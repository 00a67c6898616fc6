#!/bin/bash
# kernel trace of the default bench run (gpurun -- 'bash tools/gpu_trace_bench.sh <tag>'): per-kernel stats CSV and the
# bench line under gpurun_out/, to be copied into profiles/<tag>_kernel_stats.csv / <tag>_bench.json
TAG=${1:-x}
set -o pipefail
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o $TAG --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1; echo "exit $?"
tail -1 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.log | cut -c1-200
cd $GRAFT_REPO_ROOT && timeout -k 10 300 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-300
