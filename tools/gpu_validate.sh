#!/bin/bash
# what the driver runs at round end, in one call: all GPU tests, smoke(), the default bench line
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== all gpu tests"; timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/validate_pytest_all.log 2>&1; rc=$?; tail -4 gpurun_out/validate_pytest_all.log; [ $rc -ne 0 ] && exit $rc
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
echo "== bench (default)"; timeout -k 10 600 python bench.py > gpurun_out/validate_bench.log 2>&1; rc=$?; tail -1 gpurun_out/validate_bench.log | cut -c1-400; exit $rc
