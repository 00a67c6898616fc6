#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_train_replay.py tests/test_gpu_training_loop.py tests/test_gpu_eval_replay.py -x -q -m gpu > gpurun_out/r03g_pytest.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r03g_pytest.log
timeout -k 10 400 python tools/time_train_loop.py 32 > gpurun_out/r03g_train_loop.log 2>&1; echo "exit $?"; grep "B=" gpurun_out/r03g_train_loop.log
timeout -k 10 400 python tools/time_train_loop.py 8 >> gpurun_out/r03g_train_loop.log 2>&1; echo "exit $?"; grep "B=8" gpurun_out/r03g_train_loop.log
