#!/bin/bash
# first GPU pass: smoke, bench, kernel trace
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench1.log 2>&1; echo "bench exit $?" >> gpurun_out/bench1.log
tail -5 gpurun_out/bench1.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof1 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/prof1.log 2>&1; echo "prof exit $?" >> $R/gpurun_out/prof1.log
tail -3 $R/gpurun_out/prof1.log
find $R/gpurun_out/prof1 -name "*stats*" | head
