#!/bin/bash
# configs[3] with the sliced gather's widest slice capped (GNM_AGG_SLICE_MAX): narrower slices = more workgroups per CU
cd $GRAFT_REPO_ROOT
for w in 32 16 8; do
  GNM_AGG_SLICE_MAX=$w timeout -k 10 200 python bench.py --config c4 --no-cpu-baseline > gpurun_out/c4_slice_$w.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/c4_slice_$w.json").read().strip().splitlines()[-1])
print("slice $w", round(d["value"]), d["ms_per_step"], d.get("kernel_ms"))
PY
done
