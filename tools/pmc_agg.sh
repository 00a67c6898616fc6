#!/bin/bash
# PMC passes on the aggregation micro-benchmark (counters only: no trace domains besides kernel-trace)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_agg
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
run() { # name, counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/bench_agg.py --iters 10 > $OUT/$name.log 2>&1
  echo "$name exit $?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
find $OUT -name "*counter_collection.csv" | head
