#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03f_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03f_pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03f_bench.json 2> gpurun_out/r03f_bench.err; echo "bench exit $?"; cut -c1-250 gpurun_out/r03f_bench.json
GNM_NO_DISC_UNIT=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03f_bench_nounit.json 2> gpurun_out/r03f_bench_nounit.err; echo "bench exit $?"; cut -c1-250 gpurun_out/r03f_bench_nounit.json
timeout -k 10 200 python tools/prof_step_ops.py 1024 > gpurun_out/r03f_step_ops.log 2>&1; echo "ops exit $?"; head -70 gpurun_out/r03f_step_ops.log
