#!/bin/bash
# HBM traffic of ONE whole training step (bench.py's `roofline_step`): separate rocprofv3 --pmc passes for FETCH_SIZE and
# WRITE_SIZE over the default bench command (replayed steps), counters only with --kernel-trace as MI355X_MICROARCH.md
# prescribes.  usage: bash tools/pmc_step.sh <tag>; then python tools/pmc_step_summary.py gpurun_out/pmc_step_<tag> <out.md>
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_step_$TAG
mkdir -p $OUT
run() { name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timer > $OUT/$name.log 2>&1
  echo "$name exit $?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
