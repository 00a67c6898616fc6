#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for d in 0.04 0.08 0.12 0.16 0.20; do
  echo "== density $d"; timeout -k 10 300 python tools/bench_agg.py --density $d --modes plain,mplain,bwdstats,mbwdstats --iters 20 2>&1 | grep AGG | cut -c1-100 | tee -a gpurun_out/r02j_density.log
done
for n in 100 200; do
  echo "== n=$n density 0.3"; timeout -k 10 300 python tools/bench_agg.py --density 0.3 --nodes $n --modes plain,mplain --iters 20 2>&1 | grep AGG | cut -c1-100 | tee -a gpurun_out/r02j_density.log
done
