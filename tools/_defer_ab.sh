cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model_parity.py tests/test_gpu_eval_replay.py tests/test_gpu_training_loop.py -x -q > gpurun_out/defer_tests.log 2>&1 && \
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('defer', round(d['value']), d['ms_per_step'])" >> gpurun_out/defer_ab.log
  GNM_NO_DEFER_REDUCE=1 timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('nodefer', round(d['value']), d['ms_per_step'])" >> gpurun_out/defer_ab.log
done
tail -3 gpurun_out/defer_tests.log; cat gpurun_out/defer_ab.log
