#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_train_replay.py tests/test_gpu_training_loop.py tests/test_gpu_eval_replay.py -x -q -m gpu > gpurun_out/r03h_pytest.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -5 gpurun_out/r03h_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/time_train_loop.py 32 > gpurun_out/r03h_train_loop.log 2>&1; rc=$?; echo "exit $rc"; grep "B=" gpurun_out/r03h_train_loop.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/time_train_loop.py 8 >> gpurun_out/r03h_train_loop.log 2>&1; echo "exit $?"; grep "B=8" gpurun_out/r03h_train_loop.log
timeout -k 10 600 python -m pytest tests/test_gpu_aggm.py -x -q -m gpu > gpurun_out/r03h_pytest_aggm.log 2>&1; rc=$?; echo "pytest aggm exit $rc"; tail -3 gpurun_out/r03h_pytest_aggm.log
[ $rc -eq 0 ] || exit 1
GNM_AGGP_MIN_UNITS=1 GNM_AGGM_FORM=2 timeout -k 10 200 python tools/bench_agg.py --modes mplain,mfused,mbwdstats --iters 30 > gpurun_out/r03h_agg_q.log 2>&1; echo "exit $?"; grep AGG gpurun_out/r03h_agg_q.log
timeout -k 10 200 python tools/bench_agg.py --modes mplain,mfused,mbwdstats --iters 30 > gpurun_out/r03h_agg_perunit.log 2>&1; echo "exit $?"; grep AGG gpurun_out/r03h_agg_perunit.log
