#!/bin/bash
# Evidence run of a round on the GPU box:  gpurun --timeout 1200 -- 'bash tools/gpu_evidence.sh r04'
# All GPU tests; the default bench line, configs[3], the forced single-rank RCCL line; PMC passes on the three matrix-core
# aggregation forms and on the sliced gather of configs[3] (-> tools/pmc_agg_summary.py -> profiles/agg_traffic.json), on the
# big kernels of a step (tools/pmc_lin_summary.py) and on one whole step (tools/pmc_step_summary.py ->
# profiles/step_traffic.json); rocprofv3 kernel traces of both benches; in-kernel timelines; caller-side timings.
# Everything lands in gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards (tools/collect_evidence.py).
T=${1:-rXX}
set -o pipefail
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/${T}_pytest.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/${T}_pytest.log
timeout -k 10 300 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench exit $?"; cut -c1-220 gpurun_out/${T}_bench.json
timeout -k 10 300 python bench.py --config c4 > gpurun_out/${T}_c4_bench.json 2> gpurun_out/${T}_c4_bench.err; echo "c4 exit $?"; cut -c1-200 gpurun_out/${T}_c4_bench.json
GNM_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --batch 512 --no-cpu-baseline > gpurun_out/${T}_bench_rccl_world1.json 2> gpurun_out/${T}_bench_rccl1.err; echo "rccl exit $?"; cut -c1-200 gpurun_out/${T}_bench_rccl_world1.json
echo "# python tools/time_train_loop.py 32 ; python tools/time_eval.py" > gpurun_out/${T}_caller_side_timings.txt
timeout -k 10 300 python tools/time_train_loop.py 32 > gpurun_out/${T}_train_loop.log 2>&1; echo "exit $?"; grep "B=" gpurun_out/${T}_train_loop.log | tee -a gpurun_out/${T}_caller_side_timings.txt
timeout -k 10 300 python tools/time_eval.py > gpurun_out/${T}_time_eval.log 2>&1; echo "exit $?"; grep -i "eval" gpurun_out/${T}_time_eval.log | tail -6 | tee -a gpurun_out/${T}_caller_side_timings.txt
rm -rf gpurun_out/pmc_agg_${T}_* gpurun_out/pmc_lin_${T} gpurun_out/pmc_step_${T}
for m in mplain mfused mbwdstats; do
  echo "== pmc $m"; bash tools/pmc_agg.sh ${T}_$m $m 2>&1 | tail -5
done
bash tools/pmc_agg.sh ${T}_c4 plain "--knn --F 128 --batch 256 --pool 256" 2>&1 | tail -5
bash tools/pmc_agg.sh ${T}_c4f fused "--knn --F 128 --batch 256 --pool 256" 2>&1 | tail -5
bash tools/pmc_agg.sh ${T}_c4b bwdstats "--knn --F 128 --batch 256 --pool 256" 2>&1 | tail -5
bash tools/pmc_lin.sh ${T} 2>&1 | tail -4
bash tools/pmc_step.sh ${T} 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${T} $R/gpurun_out/prof_${T}_c4
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${T} -o ${T} --output-format csv -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $R/gpurun_out/${T}_prof_bench.log 2>&1; echo "trace exit $?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${T}_c4 -o ${T}_c4 --output-format csv -- python3 $R/bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${T}_c4_prof_bench.log 2>&1; echo "c4 trace exit $?"
cd $R
python tools/step_sequence.py gpurun_out/prof_${T} > gpurun_out/${T}_step_sequence.txt 2>&1
python tools/step_sequence.py gpurun_out/prof_${T}_c4 > gpurun_out/${T}_c4_step_sequence.txt 2>&1
export GNM_HIP_LIB=graph-neural-mapping_amd/lib/variants/tuning.so
if [ -f $GNM_HIP_LIB ]; then
  for m in plain bwdstats; do timeout -k 10 120 python tools/aggm_timeline.py --mode $m > gpurun_out/${T}_aggm_timeline_$m.log 2>&1; done
  timeout -k 10 200 python tools/agg_timeline.py --config c4 > gpurun_out/${T}_c4_timeline.log 2>&1
fi
unset GNM_HIP_LIB
bash tools/bench_matrix.sh > gpurun_out/${T}_config_matrix.txt 2>&1; tail -3 gpurun_out/${T}_config_matrix.txt
echo done
