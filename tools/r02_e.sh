#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=graph-neural-mapping_amd/lib/variants
echo "== kernel tests"; timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q > gpurun_out/r02e_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r02e_pytest.log; [ $rc -ne 0 ] && exit $rc
echo "== paired A/B (product vs r01)"; timeout -k 10 600 python tools/bench_agg.py --modes plain,fused,bwdstats --iters 60 --ab $V/r01.so --check 2>&1 | grep -E "AGG|spot" | tee gpurun_out/r02e_agg.log
echo "== timeline"; for m in plain bwdstats; do GNM_HIP_LIB=$V/tuning.so timeout -k 10 300 python tools/agg_timeline.py --mode $m 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02e_timeline.log; done
echo "== parity tests"; timeout -k 10 900 python -m pytest tests/test_gpu_model_parity.py tests/test_gpu_fuzz_parity.py tests/test_gpu_eval_replay.py -x -q > gpurun_out/r02e_pytest2.log 2>&1; rc=$?; tail -3 gpurun_out/r02e_pytest2.log; exit $rc
