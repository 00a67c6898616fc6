#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=graph-neural-mapping_amd/lib/variants
echo "== kernel tests"; timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q > gpurun_out/r02g_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r02g_pytest.log; [ $rc -ne 0 ] && exit $rc
echo "== c4 bench (new)"; timeout -k 10 600 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline --time-all-kernels > gpurun_out/r02g_c4.log 2>&1; tail -1 gpurun_out/r02g_c4.log | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), j['ms_per_step'], j['roofline']); print({k:v for k,v in j['kernel_ms'].items() if 'agg' in k})"

echo "== parity tests"; timeout -k 10 900 python -m pytest tests/test_gpu_model_parity.py tests/test_gpu_fuzz_parity.py -x -q > gpurun_out/r02g_pytest2.log 2>&1; rc=$?; tail -3 gpurun_out/r02g_pytest2.log; exit $rc
