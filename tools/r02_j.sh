#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== kernel tests"; timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q > gpurun_out/r02j_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r02j_pytest.log; [ $rc -ne 0 ] && exit $rc
echo "== bench"; timeout -k 10 600 python bench.py --steps 20 --warmup 5 --time-all-kernels --no-cpu-baseline > gpurun_out/r02j_bench.log 2>&1; tail -1 gpurun_out/r02j_bench.log | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
print(round(j['value']), j['ms_per_step'], 'host', j['host_enqueue_ms_per_step'])
for k,(c,ms) in sorted(j['kernel_ms'].items(), key=lambda kv:-kv[1][0]*kv[1][1]):
    print('  %-40s %4d x %.4f ms = %.3f ms/step' % (k, c, ms, c*ms/20))
print(j['roofline_mlp'])
"
