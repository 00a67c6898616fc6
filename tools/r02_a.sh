#!/bin/bash
# round 2, GPU pass A: parity suite, aggregation A/B (product vs round-1 kernels vs ablations), bench at pool 4096 / 1024
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=graph-neural-mapping_amd/lib/variants
echo "== pytest -m gpu"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r02a_pytest.log
[ $rc -ne 0 ] && exit $rc
echo "== agg product"; timeout -k 10 300 python tools/bench_agg.py --modes plain,fused,bwdstats,bwd,phasea --check 2>&1 | grep -E "AGG|spot" | tee gpurun_out/r02a_agg.log
timeout -k 10 300 python tools/bench_agg.py --modes plain,fused,bwdstats --cold 2>&1 | grep AGG | tee -a gpurun_out/r02a_agg.log
echo "== agg r01"; GNM_HIP_LIB=$V/r01.so timeout -k 10 300 python tools/bench_agg.py --modes plain,fused,bwdstats 2>&1 | grep AGG | tee -a gpurun_out/r02a_agg.log
GNM_HIP_LIB=$V/r01.so timeout -k 10 300 python tools/bench_agg.py --modes plain,fused,bwdstats --cold 2>&1 | grep AGG | tee -a gpurun_out/r02a_agg.log
echo "== ablations (tuning build)"
for d in 0 1 2 4 8 3 7; do
  GNM_AGG16_DEBUG=$d GNM_HIP_LIB=$V/tuning.so timeout -k 10 300 python tools/bench_agg.py --modes plain,bwdstats --tag dbg$d 2>&1 | grep AGG | tee -a gpurun_out/r02a_agg.log
done
echo "== bench pool 4096"; timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02a_bench4096.log 2>&1; tail -1 gpurun_out/r02a_bench4096.log | cut -c1-1500
echo "== bench pool 1024"; timeout -k 10 600 python bench.py --steps 20 --warmup 5 --pool 1024 --no-cpu-baseline > gpurun_out/r02a_bench1024.log 2>&1; tail -1 gpurun_out/r02a_bench1024.log | cut -c1-900
