#!/bin/bash
# the "other configurations" table of DESIGN.md section 6; usage: bash tools/bench_matrix.sh <tag>
TAG=${1:-x}
cd $GRAFT_REPO_ROOT
out=gpurun_out/matrix_$TAG.txt
: > $out
run() {
  echo "== $* ${GNM_DENSE_FILL:+(GNM_DENSE_FILL=$GNM_DENSE_FILL: CSR gather only)}" >> $out
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'graphs/s', round(d['ms_per_step'],3), 'ms/step', d['launch_mode'])" >> $out
}
run --batch 512 --graph on
run --graph on
run --batch 256 --neighbor-pooling sum --graph-pooling sum
run --batch 256 --neighbor-pooling sum --graph-pooling average
run --batch 256 --neighbor-pooling average --graph-pooling sum
run --batch 256 --neighbor-pooling average --graph-pooling average
run --batch 256 --graph on
run --no-learn-eps
run --keep-pct 100 --batch 512
run --config c4
run --batch 32
run --batch 32 --graph on
run --agg0-cache
run --graph off
GNM_DENSE_FILL=2 run --graph off
run --batch 256 --pool 256 --graph off
run --batch 256 --pool 256 --graph off --neighbor-pooling max
cat $out
