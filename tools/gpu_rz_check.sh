#!/bin/bash
# correctness then timing of the Z-recomputing Linear backward: kernel test, model parity suites, bench A/B
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "recomputing or linear_backward" > gpurun_out/rz_tests.log 2>&1 || { tail -30 gpurun_out/rz_tests.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_model_parity.py tests/test_gpu_training_loop.py -x -q -m gpu >> gpurun_out/rz_tests.log 2>&1 || { tail -30 gpurun_out/rz_tests.log; exit 1; }
grep -E "passed|failed" gpurun_out/rz_tests.log
for v in "rz16:" "rz32:GNM_LINBWD_WG16=0" "norz:GNM_NO_RZ=1"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/rz_bench_$name.json 2> gpurun_out/rz_bench_$name.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/rz_bench_$name.json").read().strip().splitlines()[-1])
print("$name", round(d["value"]), d["ms_per_step"], d.get("kernel_ms"))
PY
done
