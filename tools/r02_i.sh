#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== aggm + kernel tests"; timeout -k 10 900 python -m pytest tests/test_gpu_aggm.py tests/test_gpu_kernels.py -x -q > gpurun_out/r02i_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r02i_pytest.log; [ $rc -ne 0 ] && exit $rc
echo "== all gpu tests"; timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02i_pytest_all.log 2>&1; rc=$?; tail -5 gpurun_out/r02i_pytest_all.log; [ $rc -ne 0 ] && exit $rc
echo "== bench"; timeout -k 10 600 python bench.py --steps 20 --warmup 5 --time-all-kernels --no-cpu-baseline > gpurun_out/r02i_bench.log 2>&1; tail -1 gpurun_out/r02i_bench.log
