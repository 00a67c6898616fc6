#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_aggm.py -x -q -m gpu > gpurun_out/r03c_pytest_aggm.log 2>&1; rc=$?; echo "pytest aggm exit $rc"; tail -3 gpurun_out/r03c_pytest_aggm.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/bench_agg.py --modes mplain,mfused,mbwdstats --iters 30 > gpurun_out/r03c_agg_persist.log 2>&1; echo "exit $?"; grep AGG gpurun_out/r03c_agg_persist.log
GNM_AGGM_NO_PERSIST=1 timeout -k 10 200 python tools/bench_agg.py --modes mplain,mfused,mbwdstats --iters 30 > gpurun_out/r03c_agg_perunit.log 2>&1; echo "exit $?"; grep AGG gpurun_out/r03c_agg_perunit.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_aggm.py > gpurun_out/r03c_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03c_pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err; echo "bench exit $?"; cut -c1-250 gpurun_out/r03c_bench.json
GNM_NO_DISC_UNIT=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03c_bench_nounit.json 2> gpurun_out/r03c_bench_nounit.err; echo "bench exit $?"; cut -c1-250 gpurun_out/r03c_bench_nounit.json
