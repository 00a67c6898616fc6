#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=graph-neural-mapping_amd/lib/variants
echo "== aggm tests (7 waves)"; GNM_HIP_LIB=$V/w7.so timeout -k 10 900 python -m pytest tests/test_gpu_aggm.py -x -q > gpurun_out/r02i_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r02i_pytest.log; [ $rc -ne 0 ] && exit $rc
echo "== bench"; timeout -k 10 300 python tools/bench_agg.py --modes mplain,mfused,mbwdstats --ab $V/w7.so 2>&1 | tee gpurun_out/r02i_bench.log
