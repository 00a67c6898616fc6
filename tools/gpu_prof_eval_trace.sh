#!/bin/bash
# rocprofv3 kernel stats of the one-graph-per-forward evaluation loop (tools/time_eval.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pe_eval -o eval -- python3 $R/tools/time_eval.py > $R/gpurun_out/pe_eval.log 2>&1 || exit 1
echo ok
