#!/usr/bin/env python3
"""The launch sequence of ONE training step from a rocprofv3 kernel trace of bench.py (replayed steps):
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timer
    python tools/step_sequence.py <dir>
Prints every kernel of the median step in order with its duration and the gap to its predecessor, then totals:
kernel time, gap time, launches; grouped per kernel name."""
import csv
import glob
import os
import sys
from collections import OrderedDict

d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# a step begins at the kernel that zeroes the gradients ... robust marker: the aggregation of the input layer (one per step)
marks = [i for i, r in enumerate(rows) if "gnm_loss_finish" in r[2] or "gnm_bce_kernel" in r[2]]
starts = [i for i, r in enumerate(rows) if "gnm_bce_kernel" in r[2]]
if len(starts) < 4:
    raise SystemExit("no steps found")
# step k = kernels between consecutive bce launches (rotated: a step's own order is recovered up to rotation)
k = len(starts) // 2
seg = rows[starts[k]:starts[k + 1]]
tot_k = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0] + 0
prev_end = rows[starts[k] - 1][1]
print("%-4s %-86s %9s %8s" % ("#", "kernel", "dur us", "gap us"))
gaps = 0
per = OrderedDict()
for i, (s, e, nm) in enumerate(seg):
    gap = (s - prev_end) / 1e3
    gaps += max(gap, 0)
    prev_end = e
    short = nm[:86]
    print("%-4d %-86s %9.1f %8.1f" % (i, short, (e - s) / 1e3, gap))
    c = per.setdefault(nm[:60], [0, 0.0])
    c[0] += 1
    c[1] += (e - s) / 1e3
print("\nlaunches %d   kernel time %.1f us   gaps %.1f us   span %.1f us" % (len(seg), tot_k / 1e3, gaps, (rows[starts[k + 1]][0] - rows[starts[k]][0]) / 1e3))
print("\nper kernel (sorted by time):")
for nm, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("  %-60s x%-3d %8.1f us" % (nm, c, t))
