#!/bin/bash
# Short evidence refresh after a kernel change: PMC bytes of one step (-> profiles/step_traffic.json), both bench lines,
# kernel traces and step sequences of both configurations.  usage: gpurun -- bash tools/gpu_evidence_short.sh <tag>
T=${1:-r04d}
R=$GRAFT_REPO_ROOT
cd $R
bash tools/pmc_step.sh ${T} 2>&1 | tail -2
python tools/pmc_step_summary.py gpurun_out/pmc_step_${T} profiles/${T}_step_traffic.md 1024 | tail -1
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench $?"
python bench.py --config c4 > gpurun_out/${T}_c4_bench.json 2> gpurun_out/${T}_c4_bench.err; echo "c4 $?"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${T} $R/gpurun_out/prof_${T}_c4
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${T} -o ${T} --output-format csv -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $R/gpurun_out/${T}_prof_bench.log 2>&1; echo "trace $?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${T}_c4 -o ${T}_c4 --output-format csv -- python3 $R/bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${T}_c4_prof_bench.log 2>&1; echo "c4 trace $?"
cd $R
python tools/step_sequence.py gpurun_out/prof_${T} > gpurun_out/${T}_step_sequence.txt 2>&1
python tools/step_sequence.py gpurun_out/prof_${T}_c4 > gpurun_out/${T}_c4_step_sequence.txt 2>&1
cp profiles/step_traffic.json gpurun_out/${T}_step_traffic.json; cp profiles/${T}_step_traffic.md gpurun_out/
echo done
