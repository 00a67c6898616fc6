#!/bin/bash
# PMC passes over a short bench run, summarised for the Linear kernels (MFMA busy cycles, HBM traffic).
# counters only with --kernel-trace, separate passes (MI355X_MICROARCH.md).  usage: bash tools/pmc_lin.sh <tag>
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_lin_$TAG
mkdir -p $OUT
run() { name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer > $OUT/$name.log 2>&1
  echo "$name exit $?"; }
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run grbm GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
