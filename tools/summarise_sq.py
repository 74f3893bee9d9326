#!/usr/bin/env python3
"""gpurun_out/prof_sq/p*/…counter_collection.csv (tools/collect_sq_counters.sh) -> profiles/<tag>_sq_counters_k_accumulate.csv"""
import csv, glob, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
acc = defaultdict(list)
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "prof_sq", "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_accumulate<1>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(ROOT, "profiles", f"{tag}_sq_counters_k_accumulate.csv"), "w") as f:
    f.write("# rocprofv3 --pmc passes (tools/collect_sq_counters.sh: three separate runs of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --serial`)\n")
    f.write("# kernel zkhip::k_accumulate<1>, 2^20 terms x 19 table levels; averages per dispatch.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles.\n")
    f.write("# counter, dispatches, average\n")
    for k in sorted(acc):
        f.write("%s,%d,%.0f\n" % (k, len(acc[k]), sum(acc[k]) / len(acc[k])))
        print(k, len(acc[k]), "%.4g" % (sum(acc[k]) / len(acc[k])))
