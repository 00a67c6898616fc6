#!/bin/bash
# HBM-traffic and SQ counter passes on the aggregation micro-benchmark (separate --pmc passes,
# counters only with --kernel-trace, as MI355X_MICROARCH.md prescribes).  usage: bash tools/pmc_agg.sh <tag>
TAG=${1:-x}
EXTRA=${2:-}      # modes, e.g. mfused
ARGS=${3:-}       # extra bench_agg.py arguments, e.g. "--knn --F 128 --batch 256 --pool 256"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_agg_$TAG
mkdir -p $OUT
run() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/bench_agg.py --iters 10 --modes ${EXTRA:-plain} $ARGS > $OUT/$name.log 2>&1
  echo "$name exit $?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES SQ_LDS_UNALIGNED_STALL
run grbm GRBM_GUI_ACTIVE
