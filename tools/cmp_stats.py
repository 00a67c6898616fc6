#!/usr/bin/env python3
"""Side-by-side average durations of two rocprofv3 kernel_stats.csv files: python tools/cmp_stats.py a.csv b.csv [min_us]"""
import csv, sys
def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(p))}
a, b = load(sys.argv[1]), load(sys.argv[2])
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[0] * a.get(k, (0, 0))[1])):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    if max(ta, tb) < lo: continue
    print("%-62s %5d %8.1f | %5d %8.1f  %+6.1f" % (k[:62], ca, ta, cb, tb, tb - ta))
