#!/bin/bash
# round 2, GPU pass C: paired A/B of the aggregation builds, in-kernel timeline, eval host profile
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=graph-neural-mapping_amd/lib/variants
echo "== paired A/B (product vs r01)"; timeout -k 10 600 python tools/bench_agg.py --modes plain,fused,bwdstats --iters 60 --ab $V/r01.so 2>&1 | grep -E "AGG" | tee gpurun_out/r02c_agg.log
echo "== timeline"; for m in plain fused bwdstats; do GNM_HIP_LIB=$V/tuning.so timeout -k 10 300 python tools/agg_timeline.py --mode $m 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02c_timeline.log; done
echo "== eval host profile"; timeout -k 10 300 python tools/prof_eval.py 2>&1 | grep -v amdgpu.ids | head -60 | tee gpurun_out/r02c_prof_eval.log
