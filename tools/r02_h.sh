#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=graph-neural-mapping_amd/lib/variants
for k in fwd bwd bwd_first; do
GNM_HIP_LIB=$V/lintune2.so timeout -k 10 300 python tools/lin_timeline.py --kernel $k 2>&1 | tee -a gpurun_out/r02h_timeline.log || exit 1
done
