#!/bin/bash
# in-step effect of the aggregation workgroup size (GNM_AGG16_THREADS tuning knob)
cd $GRAFT_REPO_ROOT
for t in 1024 896 768 640 512; do
  GNM_AGG16_THREADS=$t timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --time-all-kernels > gpurun_out/sw.log 2>&1
  tail -1 gpurun_out/sw.log | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); d=j['kernel_ms']; print('THREADS', $t, 'agg_fwd', d['agg_fwd_F64'][1], 'agg_bwd', d['agg_bwd_F64'][1], 'ms/step', round(j['ms_per_step'],3))"
done
