#!/bin/bash
# SQ counters of the accumulation kernel (separate --pmc passes, no trace domains next to them): python bench.py --serial.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof_sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o sq -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --serial > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
done
ls $O
