#!/bin/bash
# round 2, GPU pass B: aggregation A/B after the id-ring rewrite, then the affected tests
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=graph-neural-mapping_amd/lib/variants
echo "== agg product"; timeout -k 10 300 python tools/bench_agg.py --modes plain,fused,bwdstats,bwd --check 2>&1 | grep -E "AGG|spot" | tee gpurun_out/r02b_agg.log
timeout -k 10 300 python tools/bench_agg.py --modes plain,fused,bwdstats --cold 2>&1 | grep AGG | tee -a gpurun_out/r02b_agg.log
echo "== agg r01"; GNM_HIP_LIB=$V/r01.so timeout -k 10 300 python tools/bench_agg.py --modes plain,fused,bwdstats 2>&1 | grep AGG | tee -a gpurun_out/r02b_agg.log
echo "== ablations"
for d in 1 2 8; do
  GNM_AGG16_DEBUG=$d GNM_HIP_LIB=$V/tuning.so timeout -k 10 300 python tools/bench_agg.py --modes plain --tag dbg$d 2>&1 | grep AGG | tee -a gpurun_out/r02b_agg.log
done
echo "== bench"; timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02b_bench.log 2>&1; tail -1 gpurun_out/r02b_bench.log | cut -c1-400; tail -1 gpurun_out/r02b_bench.log | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['kernel_ms'], j['roofline']['frac'], j['roofline'].get('backward'))"
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model_parity.py tests/test_gpu_fuzz_parity.py tests/test_gpu_eval_replay.py -x -q > gpurun_out/r02b_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r02b_pytest.log; exit $rc
