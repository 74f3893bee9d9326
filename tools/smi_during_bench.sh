#!/bin/bash
# As tools/smi_during_stream.sh, over the driver's whole command (twice): every stream of the secondaries writes its start / end time and
# rate (ZKHIP_BENCH_MARKS), rocm-smi is sampled twice a second beside it.  Output: gpurun_out/smi_bench/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/smi_bench
mkdir -p $OUT
cd $ROOT
( while [ ! -f $OUT/stop ]; do echo "t=$(date +%s.%N)"; rocm-smi -P -c -t --json 2>/dev/null | tr -d "\n"; echo; sleep 0.5; done > $OUT/smi.log ) &
SMI=$!
export ZKHIP_BENCH_MARKS=$OUT/marks.log
for k in 1 2; do
  timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench$k.log 2>&1
done
touch $OUT/stop
wait $SMI
rm -f $OUT/stop
cat $OUT/marks.log
