#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in el_noa el_nob el_noab; do
  GNM_HIP_LIB=$R/graph-neural-mapping_amd/lib/variants/$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pe_$v -o $v -- python3 $R/tools/host_eval_breakdown.py layers > $R/gpurun_out/pe_$v.log 2>&1 || exit 1
  grep "gnm_eval_layer_kernel" $R/gpurun_out/pe_$v/${v}_kernel_stats.csv | cut -d, -f1-4
done
