#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
V=graph-neural-mapping_amd/lib/variants
for g in 256; do
echo "== GNM_LINBWD_GRID=$g (variant) vs product default grid"; timeout -k 10 300 python tools/bench_lin.py --modes bwd_first 2>&1 | grep LIN; GNM_HIP_LIB=$V/lbw1.so GNM_LINBWD_GRID=$g timeout -k 10 300 python tools/bench_lin.py --modes bwd_first 2>&1 | grep LIN
done
