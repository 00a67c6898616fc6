#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== aggm + kernel tests"; timeout -k 10 900 python -m pytest tests/test_gpu_aggm.py tests/test_gpu_kernels.py tests/test_gpu_model_parity.py -x -q > gpurun_out/r02m_pytest.log 2>&1; rc=$?; tail -4 gpurun_out/r02m_pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --time-all-kernels --no-cpu-baseline --graph off > gpurun_out/r02m_bench.log 2>&1; tail -1 gpurun_out/r02m_bench.log | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
print(round(j['value']), j['ms_per_step']); print({k:v for k,v in j['kernel_ms'].items() if 'F7' in k or 'agg' in k})"
