#!/bin/bash
# SQ / LDS counters of k_ntt_pass at 2^20: separate --pmc passes, no trace domains next to them; one kernel-trace
# pass for durations.  Output: gpurun_out/prof_<tag>/summary.csv (copied to profiles/ by hand).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ntt}
O=$ROOT/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/ntt_measure.py 20"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o sq -- $CMD > $O/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $O/p$i.log; }
done
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o kt -- $CMD > $O/trace.log 2>&1
python3 - "$O" "$TAG" <<'PYEOF'
import csv, glob, os, sys
from collections import defaultdict
o, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(o, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ntt_pass" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0].replace("void zkhip::", ""), r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for f in glob.glob(os.path.join(o, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ntt_pass" in r["Kernel_Name"]:
            dur[(r["Kernel_Name"].split("(")[0].replace("void zkhip::", ""), r.get("Grid_Size", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
with open(os.path.join(o, "summary.csv"), "w") as f:
    f.write("# NTT passes alone at 2^20 (tools/collect_ntt_counters.sh -> tools/ntt_measure.py 20): kernel, grid size (work-items: x3 = the batch of three), counter, dispatches, average per dispatch\n")
    for key in sorted(acc):
        for k in sorted(acc[key]):
            v = acc[key][k]
            f.write("%s,%s,%s,%d,%.0f\n" % (key[0], key[1], k, len(v), sum(v) / len(v)))
    for key in sorted(dur):
        v = dur[key]
        f.write("%s,%s,duration_us_trace,%d,%.1f\n" % (key[0], key[1], len(v), sum(v) / len(v)))
print(open(os.path.join(o, "summary.csv")).read())
print(open(os.path.join(o, "trace.log")).read()[-3000:])
PYEOF
find $O -name "*kernel_trace.csv" -delete 2>/dev/null; find $O -name "*.db" -delete 2>/dev/null
