#!/bin/bash
# sweep the persistent-grid caps of the Linear kernels; prints the two timed kernels per setting
cd $GRAFT_REPO_ROOT
for g in 512 768 1024 1536 2048 3200; do
  GNM_LIN_GRID=$g timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --time-all-kernels > gpurun_out/sw.log 2>&1
  tail -1 gpurun_out/sw.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['kernel_ms']; print('LIN_GRID', $g, 'lin_fwd', d['lin_fwd_K64_H64'][1])"
done
for g in 256 512 768 1024 1536; do
  GNM_LINBWD_GRID=$g timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --time-all-kernels > gpurun_out/sw.log 2>&1
  tail -1 gpurun_out/sw.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['kernel_ms']; print('LINBWD_GRID', $g, 'linbwd', d['linbwd_K64_H64'][1])"
done
