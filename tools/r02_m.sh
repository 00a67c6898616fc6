#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bench_launch.py tests/test_host_logic.py -x -q 2>&1 | tail -3
for i in 1 2; do
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r02m_bench$i.log 2>&1; tail -1 gpurun_out/r02m_bench$i.log | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['ms_per_step'],3), 'host', round(j['host_enqueue_ms_per_step'],3), j['launch_mode'], j['kernel_ms'], round(j['roofline']['frac'],3))"
done
timeout -k 10 600 python bench.py --no-cpu-baseline --graph off > gpurun_out/r02m_bench3.log 2>&1; tail -1 gpurun_out/r02m_bench3.log | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('eager', round(j['value']), round(j['ms_per_step'],3), 'host', round(j['host_enqueue_ms_per_step'],3))"
