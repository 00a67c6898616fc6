#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in plain fused bwdstats; do
GNM_HIP_LIB=graph-neural-mapping_amd/lib/variants/tuning.so timeout -k 10 200 python tools/aggp_timeline.py --mode $m > gpurun_out/r03e_aggp_timeline_$m.log 2>&1; echo "exit $?"; cat gpurun_out/r03e_aggp_timeline_$m.log
done
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03e_bench.json 2> gpurun_out/r03e_bench.err; echo "bench exit $?"; cut -c1-250 gpurun_out/r03e_bench.json
GNM_NO_DISC_UNIT=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03e_bench_nounit.json 2> gpurun_out/r03e_bench_nounit.err; echo "bench exit $?"; cut -c1-250 gpurun_out/r03e_bench_nounit.json
GNM_AGGM_NO_PERSIST=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03e_bench_nopersist.json 2> gpurun_out/r03e_bench_nopersist.err; echo "bench exit $?"; cut -c1-250 gpurun_out/r03e_bench_nopersist.json
