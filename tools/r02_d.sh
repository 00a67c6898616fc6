#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=graph-neural-mapping_amd/lib/variants
echo "== timeline"; for m in plain fused bwdstats; do GNM_HIP_LIB=$V/tuning.so timeout -k 10 300 python tools/agg_timeline.py --mode $m 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02d_timeline.log; done
echo "== eval host profile"; timeout -k 10 300 python tools/prof_eval.py 2>&1 | grep -v amdgpu.ids | head -45 | tee gpurun_out/r02d_prof_eval.log
