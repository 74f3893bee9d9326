#!/bin/bash
# Clocks, power and temperature of the GPU (rocm-smi, twice a second) while the wrapping stream runs: does the rate's run-to-run
# spread (390-400 against 420-440 proofs/s inside the full bench, round 5) follow the chip's clock?  Output: gpurun_out/smi/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/smi
mkdir -p $OUT
cd $ROOT
( for i in $(seq 1 90); do echo "t=$(date +%s.%N)"; rocm-smi -P -c -t --json 2>/dev/null | tr -d "\n"; echo; sleep 0.5; done > $OUT/smi.log ) &
SMI=$!
for k in 1 2 3; do
  echo "start$k=$(date +%s.%N)" >> $OUT/marks.log
  timeout -k 10 200 python3 bench.py --workload aggregator --steps 4000 --warmup 200 --no-cpu-baseline > $OUT/wrap$k.log 2>&1
  echo "end$k=$(date +%s.%N)" >> $OUT/marks.log
done
wait $SMI
grep -o '"value": [0-9.]*' $OUT/wrap*.log
