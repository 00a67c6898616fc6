#!/bin/bash
# evidence run on the GPU box (gpurun -- 'bash tools/gpu_evidence.sh'): PMC passes on the three matrix-core
# aggregation forms (summarise with tools/pmc_agg_summary.py -> profiles/ + profiles/agg_traffic.json; clear
# gpurun_out/pmc_agg_* first, the summary takes medians over every CSV it finds), then a kernel-trace of bench.py
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in mplain mfused mbwdstats; do
  echo "== pmc $m"; bash tools/pmc_agg.sh r02_$m $m 2>&1 | tail -5
done
echo "== rocprof stats of bench"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_evidence -o evidence --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 > $GRAFT_REPO_ROOT/gpurun_out/evidence_prof_bench.log 2>&1; echo "exit $?"
tail -1 $GRAFT_REPO_ROOT/gpurun_out/evidence_prof_bench.log | cut -c1-200
