#!/bin/bash
# Final round-3 evidence (gpurun -- 'bash tools/gpu_r03_final.sh'): all GPU tests, the default bench line, configs[3], the
# forced single-rank RCCL line, caller-side timings, rocprofv3 kernel traces of both benches.  (The PMC passes of
# tools/gpu_r03_evidence.sh are not repeated: csrc/agg.hip and csrc/aggm.hip are unchanged since -- bench.py checks their hash.)
set -o pipefail
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r03f_pytest.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/r03f_pytest.log
timeout -k 10 300 python bench.py > gpurun_out/r03f_bench.json 2> gpurun_out/r03f_bench.err; echo "bench exit $?"; cut -c1-250 gpurun_out/r03f_bench.json
timeout -k 10 300 python bench.py --config c4 > gpurun_out/r03f_c4_bench.json 2> gpurun_out/r03f_c4_bench.err; echo "c4 exit $?"; cut -c1-200 gpurun_out/r03f_c4_bench.json
GNM_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --batch 512 --no-cpu-baseline > gpurun_out/r03f_bench_rccl1.json 2> gpurun_out/r03f_bench_rccl1.err; echo "rccl exit $?"; cut -c1-200 gpurun_out/r03f_bench_rccl1.json
timeout -k 10 300 python tools/time_train_loop.py 32 > gpurun_out/r03f_train_loop.log 2>&1; echo "exit $?"; grep "B=" gpurun_out/r03f_train_loop.log
timeout -k 10 300 python tools/time_eval.py > gpurun_out/r03f_time_eval.log 2>&1; echo "exit $?"; grep eval_fused gpurun_out/r03f_time_eval.log
timeout -k 10 200 python tools/bench_sgemm.py > gpurun_out/r03f_sgemm.log 2>&1; grep "B =" gpurun_out/r03f_sgemm.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r03f $R/gpurun_out/prof_r03f_c4
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r03f -o r03f --output-format csv -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r03f_prof_bench.log 2>&1; echo "trace exit $?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r03f_c4 -o r03f_c4 --output-format csv -- python3 $R/bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r03f_c4_prof_bench.log 2>&1; echo "c4 trace exit $?"
