#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=graph-neural-mapping_amd/lib/variants
echo "== kernel tests"; timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q > gpurun_out/r02m_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r02m_pytest.log; [ $rc -ne 0 ] && exit $rc
echo "== lin A/B vs HEAD"; timeout -k 10 300 python tools/bench_lin.py --modes bwd,bwd_first --ab $V/r02base.so 2>&1 | tee gpurun_out/r02m_lin.log
