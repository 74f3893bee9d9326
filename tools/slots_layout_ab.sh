#!/bin/bash
# Round 6: the slot array of a launch sequence as an array of structures (the tree, ec_mem.cuh ZK_SLOTS_AOS=1)
# against the limb-major array of rounds 1-5 (build/libzkhip_limb.so: msm.hip compiled -DZK_SLOTS_AOS=0), interleaved on one box.
#   step 1 (here, no GPU):   bash tools/slots_layout_ab.sh build
#   step 2 (GPU box):        bash tools/slots_layout_ab.sh run      -> gpurun_out/r06f/
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DZK_MUL_INLINE=1 -fPIC -DZK_SLOTS_AOS=0 -c $ROOT/zecale_amd/csrc/msm.hip -o $ROOT/build/msm_limb.o
  objs=""
  for o in ntt qap zkhip_api witness aggregator witness_tape pipeline multi_device; do objs="$objs $ROOT/build/$o.o"; done
  hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $ROOT/build/libzkhip_limb.so $ROOT/build/msm_limb.o $objs
  ls -la $ROOT/build/libzkhip_limb.so
  exit 0
fi
OUT=$ROOT/gpurun_out/${OUTDIR:-r06f}
VARIANTS=${VARIANTS:-"aos limb"}      # aos: the tree; limb: build/libzkhip_limb.so
setlib() { if [ $1 = aos ]; then unset ZKHIP_LIB; else export ZKHIP_LIB=$ROOT/build/libzkhip_$1.so; fi; }
mkdir -p $OUT
cd $ROOT
msm_line() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$1', '%.3f Mscalar/s' % d['value'], 'k_accumulate<1> alone %.3f ms' % r['kernel_ms'], 'mad peak this run %.2f G/s' % r['fq_mul_peak_this_run_g_per_s'])"; }
val_line() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], d['unit'], 'ms_per_step', d['ms_per_step'], 'verifies', d.get('last_proof_verifies'))"; }
for rep in 1 2 3; do
  for v in $VARIANTS; do
    setlib $v
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | msm_line "msm stream  $v" >> $OUT/ab.txt
  done
done
for rep in 1 2; do
  for v in $VARIANTS; do
    setlib $v
    python3 bench.py --workload prover --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | val_line "prover 2^20  $v" >> $OUT/ab.txt
    python3 bench.py --workload aggregator --steps 800 --warmup 80 --no-cpu-baseline 2>/dev/null | val_line "wrapping    $v" >> $OUT/ab.txt
    python3 bench.py --workload aggregator --serial --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | val_line "wrapping one proof at a time $v" >> $OUT/ab.txt
    python3 bench.py --serial --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | msm_line "msm one at a time $v" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in $VARIANTS; do
  setlib $v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -o t -- python3 $ROOT/bench.py --serial --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/trace_$v.log 2>&1
  find $OUT/trace_$v -name "*kernel_trace.csv" -delete; find $OUT/trace_$v -name "*.db" -delete
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${v}_$c -o p -- python3 $ROOT/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_${v}_$c.log 2>&1
    python3 - <<PY >> $OUT/pmc.txt
import csv, glob, collections
rows = [r for f in glob.glob("$OUT/pmc_${v}_$c/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
acc = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void zkhip::", "")
    if k.startswith(("k_accumulate", "k_fixup", "k_sum", "k_slots")):
        acc[k].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("$v $c %-22s %4d launches, avg %10.0f KiB per launch" % (k, len(v), sum(v) / len(v)))
PY
    find $OUT/pmc_${v}_$c -name "*.csv" -size +1M -delete
  done
done
cat $OUT/pmc.txt
