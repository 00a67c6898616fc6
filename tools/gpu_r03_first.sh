#!/bin/bash
# round-3 first GPU call: GPU tests, default bench, config-4 evidence (bench line, kernel trace, PMC traffic)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03a_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03a_pytest.log
timeout -k 10 300 python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err; echo "bench exit $?"; cut -c1-300 gpurun_out/r03a_bench.json
timeout -k 10 300 python bench.py --config c4 > gpurun_out/r03a_c4_bench.json 2> gpurun_out/r03a_c4_bench.err; echo "c4 exit $?"; cut -c1-300 gpurun_out/r03a_c4_bench.json
bash tools/gpu_profile_c4.sh r03a
rm -rf gpurun_out/pmc_agg_r03_c4
bash tools/pmc_agg.sh r03_c4 plain "--knn --F 128 --batch 256 --pool 256" 2>&1 | tail -6
