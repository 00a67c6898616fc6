#!/bin/bash
# Round-3 evidence run on the GPU box (gpurun -- 'bash tools/gpu_r03_evidence.sh'): all GPU tests, the default bench line,
# configs[3], the forced single-rank RCCL line, PMC passes on the three matrix-core aggregation forms and on the sliced
# gather of configs[3] (summarise with tools/pmc_agg_summary.py -> profiles/ + profiles/agg_traffic.json), a
# rocprofv3 kernel trace of the bench, and the caller-side timings (reference-shaped training loop, per-graph evaluation).
set -o pipefail
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r03z_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03z_pytest.log
grep -E "^FAILED|^E  " gpurun_out/r03z_pytest.log | cut -c1-300 | head -20
timeout -k 10 300 python bench.py > gpurun_out/r03z_bench.json 2> gpurun_out/r03z_bench.err; echo "bench exit $?"; cut -c1-250 gpurun_out/r03z_bench.json
timeout -k 10 300 python bench.py --config c4 > gpurun_out/r03z_c4_bench.json 2> gpurun_out/r03z_c4_bench.err; echo "c4 exit $?"; cut -c1-200 gpurun_out/r03z_c4_bench.json
GNM_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --batch 512 --no-cpu-baseline > gpurun_out/r03z_bench_rccl1.json 2> gpurun_out/r03z_bench_rccl1.err; echo "rccl exit $?"; cut -c1-200 gpurun_out/r03z_bench_rccl1.json
timeout -k 10 300 python tools/time_train_loop.py 32 > gpurun_out/r03z_train_loop.log 2>&1; echo "exit $?"; grep "B=" gpurun_out/r03z_train_loop.log
timeout -k 10 300 python tools/time_eval.py > gpurun_out/r03z_time_eval.log 2>&1; echo "exit $?"; grep eval_fused gpurun_out/r03z_time_eval.log
rm -rf gpurun_out/pmc_agg_r03_m* gpurun_out/pmc_agg_r03_c4
for m in mplain mfused mbwdstats; do
  echo "== pmc $m"; bash tools/pmc_agg.sh r03_$m $m 2>&1 | tail -5
done
bash tools/pmc_agg.sh r03_c4 plain "--knn --F 128 --batch 256 --pool 256" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r03z
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r03z -o r03z --output-format csv -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r03z_prof_bench.log 2>&1; echo "trace exit $?"
