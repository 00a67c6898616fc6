#!/bin/bash
# rebuild norm.hip with different finalize slice widths and time the 512-graph replayed step
# (edits norm.hip in place: the original value is put back at the end, also when a run is interrupted)
cd $GRAFT_REPO_ROOT
ORIG=$(grep -o "static constexpr int kFinCols = [0-9]*;" graph-neural-mapping_amd/csrc/norm.hip | grep -o "[0-9]*" | head -1)
restore() { sed -i "s/static constexpr int kFinCols = [0-9]*;/static constexpr int kFinCols = $ORIG;/" graph-neural-mapping_amd/csrc/norm.hip; python graph-neural-mapping_amd/gnm/_build.py > /dev/null 2>&1; }
trap restore EXIT
for c in 16 8 4 2; do
  sed -i "s/static constexpr int kFinCols = [0-9]*;/static constexpr int kFinCols = $c;/" graph-neural-mapping_amd/csrc/norm.hip
  python graph-neural-mapping_amd/gnm/_build.py > /dev/null 2>&1
  for rep in 1 2; do
  timeout -k 10 200 python bench.py --batch 512 --graph on --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('kFinCols', $c, 'ms/step', round(j['ms_per_step'],4))"
  done
done
