#!/bin/bash
# SQ / memory counters of k_sum_lds (the first levels of the row / column trees of the bucket reduction) in a 2^20-term MSM, one MSM at a
# time, per launch size.  Separate --pmc passes, no trace domains next to them; one kernel-trace pass for durations.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-sum}
O=$ROOT/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --serial --steps 2 --warmup 1 --no-secondary --no-cpu-baseline"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o sq -- $CMD > $O/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $O/p$i.log; }
done
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o kt -- $CMD > $O/trace.log 2>&1
python3 - "$O" "$TAG" <<'EOF'
import csv, glob, os, sys
from collections import defaultdict
o, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(o, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for kn in ("k_sum_lds", "k_accumulate<1>", "k_fixup("):
            if kn in r["Kernel_Name"]:
                acc[(kn, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for f in glob.glob(os.path.join(o, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for kn in ("k_sum_lds", "k_accumulate<1>", "k_fixup("):
            if kn in r["Kernel_Name"]:
                dur[(kn, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
with open(os.path.join(o, "summary.csv"), "w") as f:
    f.write("# 2^20-term MSM, one at a time (tools/collect_sum_counters.sh): kernel, grid size (lanes), counter, dispatches, average per dispatch\n")
    for key in sorted(acc):
        for k in sorted(acc[key]):
            v = acc[key][k]
            f.write("%s,%s,%s,%d,%.0f\n" % (key[0], key[1], k, len(v), sum(v) / len(v)))
    for key in sorted(dur):
        v = dur[key]
        f.write("%s,%s,duration_us_trace,%d,%.1f\n" % (key[0], key[1], len(v), sum(v) / len(v)))
print(open(os.path.join(o, "summary.csv")).read())
EOF
