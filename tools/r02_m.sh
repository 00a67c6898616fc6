#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== kernel + parity tests"; timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model_parity.py -x -q > gpurun_out/r02m_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r02m_pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --time-all-kernels --no-cpu-baseline --graph off > gpurun_out/r02m_bench.log 2>&1; tail -1 gpurun_out/r02m_bench.log | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
print(round(j['value']), j['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02m -o r02m --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph off > $GRAFT_REPO_ROOT/gpurun_out/r02m_prof.log 2>&1; grep "disc_" $GRAFT_REPO_ROOT/gpurun_out/prof_r02m/r02m_kernel_stats.csv | cut -c1-140
