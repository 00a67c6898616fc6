#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "linear or split or recomputing" > gpurun_out/rz_tests.log 2>&1 || { tail -30 gpurun_out/rz_tests.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_model_parity.py -x -q -m gpu >> gpurun_out/rz_tests.log 2>&1 || { tail -30 gpurun_out/rz_tests.log; exit 1; }
grep -E "passed|failed" gpurun_out/rz_tests.log
bash tools/gpu_prof_env.sh rz16 || exit 1
bash tools/gpu_prof_env.sh norz GNM_NO_RZ=1 || exit 1
cd $R
export GNM_HIP_LIB=$R/graph-neural-mapping_amd/lib/variants/lintune.so
for k in rz rz_first bwd; do timeout -k 10 120 python tools/lin_timeline.py --kernel $k > gpurun_out/tl_$k.txt 2>&1 || exit 1; done
