#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_agg.sh (aggregation micro-benchmark) into
  * a markdown table (profiles/<round>_<tag>_agg_pmc.md) and
  * the entry of profiles/agg_traffic.json that bench.py reads for `roofline.traffic`,
    stamped with the sha256 of the csrc/agg.hip the counters were taken on.

usage: python tools/pmc_agg_summary.py <gpurun_out/pmc_agg_TAG> <variant> <kernel-name-substring> <out.md> [graphs] [source]
  variant: plain | fused_bnrelu | backward_stats | sliced_n1000_F128 | sliced_fused_n1000_F128 | sliced_bwdstats_n1000_F128 |
           mfma_plain | mfma_fused_bnrelu | mfma_backward_stats
  source:  the csrc file the kernel lives in (default agg.hip; aggm.hip for the mfma_* variants) -- its sha256 is
           recorded with the entry, and bench.py drops the entry once that file changes
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE in KB units (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports
half of a wide coalesced read stream; WRITE_SIZE is exact for 16-B-per-lane stores), scaled to a 1024-graph launch.
"""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def medians(pmc_dir, kernel_sub):
    vals, dur = {}, []
    for f in glob.glob(os.path.join(pmc_dir, "*", "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if kernel_sub in row["Kernel_Name"]:
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
    med = lambda v: sorted(v)[len(v) // 2]
    return {k: med(v) for k, v in vals.items()}, (med(dur) if dur else None), {k: len(v) for k, v in vals.items()}


def main():
    pmc_dir, variant, ksub, out_md = sys.argv[1:5]
    graphs = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
    src_name = sys.argv[6] if len(sys.argv) > 6 else ("aggm.hip" if variant.startswith("mfma_") else "agg.hip")
    m, dur_us, cnt = medians(pmc_dir, ksub)
    if "FETCH_SIZE" not in m or "WRITE_SIZE" not in m:
        raise SystemExit("no FETCH_SIZE / WRITE_SIZE rows for kernel '%s' under %s" % (ksub, pmc_dir))
    fetch_b, write_b = m["FETCH_SIZE"] * 1024.0, m["WRITE_SIZE"] * 1024.0
    hbm = 2.0 * fetch_b + write_b
    agg = os.path.join(ROOT, "graph-neural-mapping_amd", "csrc", src_name)
    sha = hashlib.sha256(open(agg, "rb").read()).hexdigest()
    lines = ["# %s -- PMC counters, variant `%s`" % (ksub, variant), "",
             "Source: `%s` (separate `rocprofv3 --kernel-trace --pmc` passes, medians over the profiled launches; "
             "%d graphs per launch; csrc/%s sha256 %s)." % (os.path.relpath(pmc_dir, ROOT), graphs, src_name, sha[:16]), "",
             "| counter | per launch (median) | launches |", "|---|---|---|"]
    for k in sorted(m):
        lines.append("| %s | %.5g | %d |" % (k, m[k], cnt[k]))
    lines += ["", "Kernel duration under the profiler (median): %.1f us." % dur_us if dur_us else "",
              "", "HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE = 2 x %.1f MB + %.1f MB = **%.1f MB**."
              % (fetch_b / 1e6, write_b / 1e6, hbm / 1e6)]
    if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_ANY" in m:
        lines.append("Waves parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES): %.1f %%." % (100 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]))
    if "SQ_LDS_IDX_ACTIVE" in m and "SQ_BUSY_CYCLES" in m:
        lines.append("LDS array busy (SQ_LDS_IDX_ACTIVE / (SQ_BUSY_CYCLES x 4... see DESIGN.md)): IDX_ACTIVE %.4g, "
                     "BANK_CONFLICT %.4g." % (m["SQ_LDS_IDX_ACTIVE"], m.get("SQ_LDS_BANK_CONFLICT", 0.0)))
    os.makedirs(os.path.dirname(os.path.abspath(out_md)), exist_ok=True)
    open(out_md, "w").write("\n".join(lines) + "\n")
    tj_path = os.path.join(ROOT, "profiles", "agg_traffic.json")
    tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
    tj.pop("agg_hip_sha256", None)               # (round 1 kept one hash for the whole file)
    tj[variant] = {"hbm_bytes_per_launch": hbm * 1024.0 / graphs, "fetch_size_bytes_raw": fetch_b,
                   "write_size_bytes": write_b, "graphs_in_profiled_launch": graphs, "kernel": ksub,
                   "kernel_source": src_name, "sha256": sha, "source": os.path.relpath(out_md, ROOT)}
    json.dump(tj, open(tj_path, "w"), indent=1)
    print(open(out_md).read())


if __name__ == "__main__":
    main()
