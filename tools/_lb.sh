cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "linear or lin" > gpurun_out/split_tests.log 2>&1; tail -5 gpurun_out/split_tests.log
timeout -k 10 200 python tools/bench_lin.py --modes bwd_first,fwd 2>&1 | grep -v amdgpu | tail -2
GNM_LIN_NO_SPLIT=1 timeout -k 10 200 python tools/bench_lin.py --modes bwd_first 2>&1 | grep -v amdgpu | tail -1
