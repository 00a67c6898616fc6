#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=graph-neural-mapping_amd/lib/variants
echo "== lin A/B + check"; timeout -k 10 300 python tools/bench_lin.py --modes fwd,bwd,bwd_first --check --ab $V/r02lin0.so 2>&1 | tee gpurun_out/r02h_lin.log || exit 1
echo "== kernel tests"; timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model_parity.py -x -q > gpurun_out/r02h_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r02h_pytest.log; [ $rc -ne 0 ] && exit $rc
exit 0
