#!/bin/bash
# persistent-grid caps of the Linear kernels at the 512-graph (per-GPU) batch, replayed step time
cd $GRAFT_REPO_ROOT
for g in 768 512 384 256; do
  GNM_LIN_GRID=$g timeout -k 10 200 python bench.py --batch 512 --graph on --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('LIN_GRID', $g, 'ms/step', round(j['ms_per_step'],4))"
done
for g in 512 384 256 192; do
  GNM_LINBWD_GRID=$g timeout -k 10 200 python bench.py --batch 512 --graph on --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('LINBWD_GRID', $g, 'ms/step', round(j['ms_per_step'],4))"
done
