#!/usr/bin/env python3
"""Per-MSM (or per-step) kernel time table from a rocprofv3 --kernel-trace --stats directory.  usage: kstats.py <dir> <launches of k_accumulate ...>"""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
acc = [r for r in rows if "k_accumulate" in r["Name"]]
n = sum(int(r["Calls"]) for r in acc) or 1
tot = 0.0
for r in rows:
    name = r["Name"].split("(")[0].replace("void ", "").replace("zkhip::", "")
    if name in ("k_table_build", "k_fixed_base_mul", "k_bases_to_dev"):
        continue
    per = float(r["TotalDurationNs"]) / 1e3 / n
    tot += per
    print("  %-30s calls %5s  avg %9.1f us  min %9.1f us   per MSM %8.1f us" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, per))
print("  sum of kernel durations per MSM: %.1f us (%d MSMs)" % (tot, n))
