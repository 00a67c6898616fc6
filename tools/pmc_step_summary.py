#!/usr/bin/env python3
"""Sum the HBM traffic of ONE replayed training step from the two passes of tools/pmc_step.sh and write
  * profiles/step_traffic.json -- what bench.py's `roofline_step` reads (stamped with the sha256 of csrc/: the entry is
    dropped once any kernel source changes), and
  * a markdown table (per kernel: launches per step, MB per step).
A step = the dispatches between two consecutive gnm_bce_kernel launches in the middle of the run (the loss kernel runs
once per step).  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB units; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half
of a wide coalesced read stream -- narrow or strided streams are over-corrected by this, so the total is an upper bound).
usage: python tools/pmc_step_summary.py gpurun_out/pmc_step_TAG profiles/rNN_step_traffic.md [graphs_per_step]"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_step(pass_dir, counter):
    f = glob.glob(os.path.join(pass_dir, "**", "*_counter_collection.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), r["Kernel_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter]
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "gnm_bce_kernel" in r[1]]
    if len(marks) < 4:
        raise SystemExit("fewer than 4 steps in %s" % f)
    k = len(marks) // 2
    return rows[marks[k]:marks[k + 1]]


def csrc_sha256(csrc_dir):
    h = hashlib.sha256()
    for f in sorted(os.listdir(csrc_dir)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc_dir, f), "rb").read())
    return h.hexdigest()


def main():
    src, out_md = sys.argv[1], sys.argv[2]
    graphs = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    fetch = one_step(os.path.join(src, "fetch"), "FETCH_SIZE")
    write = one_step(os.path.join(src, "write"), "WRITE_SIZE")
    per = OrderedDict()
    for _, nm, v in fetch:
        e = per.setdefault(nm[:70], [0, 0.0, 0.0])
        e[0] += 1
        e[1] += v * 1024.0
    for _, nm, v in write:
        e = per.setdefault(nm[:70], [0, 0.0, 0.0])
        e[2] += v * 1024.0
    tot_f = sum(e[1] for e in per.values())
    tot_w = sum(e[2] for e in per.values())
    hbm = 2.0 * tot_f + tot_w
    kernels = {nm: {"launches": e[0], "hbm_MB": round((2 * e[1] + e[2]) / 1e6, 2)} for nm, e in per.items()}
    lines = ["# HBM traffic of one training step (configs[1], %d graphs), PMC" % graphs, "",
             "Source: `%s` (tools/pmc_step.sh: separate `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` passes over "
             "`bench.py --steps 6 --warmup 3`; the dispatches of one replayed step in the middle of the run)." % os.path.relpath(src, ROOT),
             "", "| kernel | launches / step | 2 x FETCH + WRITE, MB / step |", "|---|---|---|"]
    for nm, e in sorted(per.items(), key=lambda kv: -(2 * kv[1][1] + kv[1][2])):
        lines.append("| `%s` | %d | %.1f |" % (nm, e[0], (2 * e[1] + e[2]) / 1e6))
    lines += ["", "Total: %d launches, 2 x %.1f MB + %.1f MB = **%.1f MB per step**." % (len(fetch), tot_f / 1e6, tot_w / 1e6, hbm / 1e6), ""]
    open(out_md, "w").write("\n".join(lines))
    ent = {"note": "HBM bytes of one replayed training step = sum over its kernels of 2 x FETCH_SIZE + WRITE_SIZE "
                   "(tools/pmc_step.sh, tools/pmc_step_summary.py); bench.py drops it when csrc/ changes",
           "hbm_bytes_per_step": hbm, "graphs_per_step": graphs, "launches_per_step": len(fetch),
           "csrc_sha256": csrc_sha256(os.path.join(ROOT, "graph-neural-mapping_amd", "csrc")),
           "source": os.path.relpath(out_md, ROOT), "kernels": kernels}
    json.dump(ent, open(os.path.join(ROOT, "profiles", "step_traffic.json"), "w"), indent=1)
    print("\n".join(lines[-3:]))


if __name__ == "__main__":
    main()
