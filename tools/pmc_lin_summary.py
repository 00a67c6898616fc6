#!/usr/bin/env python3
"""Summarise tools/pmc_lin.sh (PMC passes over a short bench run) into a markdown table of per-launch medians for the
big kernels of the step: MFMA busy cycles, HBM traffic, waits.
usage: python tools/pmc_lin_summary.py gpurun_out/pmc_lin_TAG profiles/rNN_pmc_step.md"""
import csv
import glob
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
KERNELS = ["gnm_lin_split_kernel<2>", "gnm_linear_bwd_rz_kernel<false, true>", "gnm_linear_bwd_rzn_kernel",
           "gnm_linear_bwd_rz_kernel<true, true>", "gnm_lin_split128_kernel<false>", "gnm_lin_split128_kernel<true>",
           "gnm_wgrad_split128_kernel", "gnm_small_gemm_kernel",
           "gnm_lin_stream_kernel<64, 2>", "gnm_linear_bwd_pipe_kernel<2, 2, true>",
           "gnm_linear_bwd_pipe_kernel<2, 2, false>", "gnm_linear_bwd_fused_kernel<2, 2, true, false, true, true>",
           "gnm_linear_bwd_fused_kernel<2, 2, true, false, true, false>",
           "gnm_aggm_kernel<false, false, false>", "gnm_aggm_kernel<true, false, false>", "gnm_aggm_kernel<false, false, true>",
           "gnm_disc_score_kernel<16", "gnm_disc_du_kernel"]
vals = {}
for f in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        for k in KERNELS:
            if k in row["Kernel_Name"]:
                vals.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
med = lambda v: sorted(v)[len(v) // 2]
cols = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU",
        "FETCH_SIZE", "WRITE_SIZE"]
lines = ["# Big kernels of a training step -- PMC counters (medians per launch)", "",
         "Command: `bash tools/pmc_lin.sh <tag>` = four separate `rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py "
         "--steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer` passes (SQ group, GRBM, FETCH_SIZE, WRITE_SIZE); "
         "configs[1]: N = 409,600 rows, K = H = 64, 1024 graphs.", "",
         "| kernel | " + " | ".join(cols) + " | MFMA busy | HBM MB (2 x FETCH + WRITE) |", "|---|" + "---|" * (len(cols) + 2)]
for k in KERNELS:
    if k not in vals:
        continue
    m = {c: med(v) for c, v in vals[k].items()}
    mf = ""
    if m.get("SQ_VALU_MFMA_BUSY_CYCLES") and m.get("GRBM_GUI_ACTIVE"):
        mf = "%.1f %%" % (100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024))
    hbm = ""
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        hbm = "%.0f" % ((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024 / 1e6)
    lines.append("| `%s` | " % k + " | ".join("%.4g" % m[c] if c in m else "-" for c in cols) + " | %s | %s |" % (mf, hbm))
lines += ["", "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); under the profiler launches run "
          "~10-15 % longer than un-profiled.  FETCH_SIZE / WRITE_SIZE are in KB; HBM traffic uses the gfx950 correction of "
          "MI355X_MICROARCH.md (2 x FETCH_SIZE for wide coalesced read streams).", ""]
open(dst, "w").write("\n".join(lines))
print("\n".join(lines))
