#!/bin/bash
# what would an array-of-structures slot layout buy k_accumulate?  Prices the WRITE side before anybody rewrites the
# readers: build/libzkhip_aos.so is the tree's library with msm.hip compiled -DZK_EXP_AOS_CLOSE (a run's 105 words go to 432 contiguous
# bytes instead of 105 rows of the limb-major array; readers unchanged: its results are WRONG, it is a measurement build).
#   step 1 (here, no GPU):   bash tools/aos_close_experiment.sh build
#   step 2 (GPU box):        bash tools/aos_close_experiment.sh run      -> gpurun_out/r06e/
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DZK_MUL_INLINE=1 -fPIC -DZK_EXP_AOS_CLOSE -c $ROOT/zecale_amd/csrc/msm.hip -o $ROOT/build/msm_aos.o
  objs=""
  for o in ntt qap zkhip_api witness aggregator witness_tape pipeline multi_device; do objs="$objs $ROOT/build/$o.o"; done
  hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $ROOT/build/libzkhip_aos.so $ROOT/build/msm_aos.o $objs
  ls -la $ROOT/build/libzkhip_aos.so
  exit 0
fi
OUT=$ROOT/gpurun_out/r06e
mkdir -p $OUT
cd $ROOT
line() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$1', 'k_accumulate<1> alone %.3f ms' % r['kernel_ms'], 'ms_per_step %.3f' % d['ms_per_step'], 'mad peak this run %.2f G/s' % r['fq_mul_peak_this_run_g_per_s'])"; }
for rep in 1 2 3; do
  python3 bench.py --serial --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | line "limb-major (the tree) " >> $OUT/ab.txt
  ZKHIP_LIB=$ROOT/build/libzkhip_aos.so python3 bench.py --serial --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | line "AoS close (experiment)" >> $OUT/ab.txt
done
cat $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in tree aos; do
  if [ $v = aos ]; then export ZKHIP_LIB=$ROOT/build/libzkhip_aos.so; else unset ZKHIP_LIB; fi
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${v}_$c -o p -- python3 $ROOT/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_${v}_$c.log 2>&1
    python3 - <<PY >> $OUT/pmc.txt
import csv, glob
rows = [r for f in glob.glob("$OUT/pmc_${v}_$c/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
v = [float(r["Counter_Value"]) for r in rows if r["Kernel_Name"].startswith("void zkhip::k_accumulate<1>")]
print("$v $c k_accumulate<1>: %d launches, avg %.0f KiB per launch" % (len(v), sum(v) / max(1, len(v))))
PY
    find $OUT/pmc_${v}_$c -name "*.csv" -size +2M -delete
  done
done
cat $OUT/pmc.txt
