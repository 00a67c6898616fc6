#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_eval_fused.py -x -q -m gpu > gpurun_out/r03i_pytest_fused.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -5 gpurun_out/r03i_pytest_fused.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/time_eval.py > gpurun_out/r03i_time_eval.log 2>&1; echo "exit $?"; grep eval_fused gpurun_out/r03i_time_eval.log
timeout -k 10 300 python tools/prof_train_loop.py > gpurun_out/r03i_prof_train_loop.log 2>&1; echo "exit $?"; head -40 gpurun_out/r03i_prof_train_loop.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03i_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03i_pytest.log
