#!/bin/bash
# SQ counters of the batched-affine level kernel and the XYZZ accumulation next to it (separate --pmc passes): tools/aff_sweep.py with LEVELS set.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof_sq_aff
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_INSTS_SMEM"; do
  i=$((i+1))
  LEVELS=${LEVELS:-1} timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o sq -- python3 $ROOT/tools/aff_sweep.py > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_affine_level" in k or "k_accumulate" in k:
            acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s n=%d avg=%.4g" % (c, len(v), sum(v) / len(v)))
PY
