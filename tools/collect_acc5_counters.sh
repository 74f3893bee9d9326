#!/bin/bash
# SQ / memory counters of k_accumulate<5> on the WRAPPING key (44,183 constraints), one proof at a time: what
# profiles/r03_sq_counters_k_accumulate.csv holds for k_accumulate<1> at 2^20.  Separate --pmc passes, no trace
# domains next to them.  usage: tools/collect_acc5_counters.sh <kind:window> [tag]     e.g. naf:16 / win:16
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-naf:0}
TAG=${2:-$(echo $CFG | tr ':' '_')}
O=$ROOT/gpurun_out/prof_acc5_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_STALL" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o sq -- python3 $ROOT/tools/acc_probe.py --grid $CFG --proofs 3 --no-dump > $O/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $O/p$i.log; }
done
python3 - "$O" "$TAG" <<'EOF'
import csv, glob, os, sys
from collections import defaultdict
o, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob(os.path.join(o, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_accumulate<5>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(o, "summary.csv"), "w") as f:
    f.write("# k_accumulate<5>, wrapping key %s, one proof at a time (tools/collect_acc5_counters.sh): counter, dispatches, average per dispatch\n" % tag)
    for k in sorted(acc):
        f.write("%s,%d,%.0f\n" % (k, len(acc[k]), sum(acc[k]) / len(acc[k])))
        print(k, len(acc[k]), "%.5g" % (sum(acc[k]) / len(acc[k])))
EOF
