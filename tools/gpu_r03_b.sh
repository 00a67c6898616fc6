#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03b_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03b_pytest.log
timeout -k 10 200 python tools/diag_cpu.py > gpurun_out/r03b_diag_cpu.log 2>&1; echo "diag exit $?"; cat gpurun_out/r03b_diag_cpu.log
timeout -k 10 300 python bench.py > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err; echo "bench exit $?"; cut -c1-300 gpurun_out/r03b_bench.json
GNM_NO_DISC_UNIT=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03b_bench_nounit.json 2> gpurun_out/r03b_bench_nounit.err; echo "bench exit $?"; cut -c1-300 gpurun_out/r03b_bench_nounit.json
