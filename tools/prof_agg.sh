#!/bin/bash
# Kernel trace of the wrapping prover: the stream (steady state) and one proof at a time.   usage: bash tools/prof_agg.sh <tag>
set -e
TAG=${1:-cur}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/agg_stream -o agg -- python3 $ROOT/bench.py --workload aggregator --steps 400 --warmup 40 --no-cpu-baseline > $OUT/agg_stream.log 2>&1
echo stream done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/agg_serial -o agg -- python3 $ROOT/bench.py --workload aggregator --serial --steps 30 --warmup 3 --no-cpu-baseline > $OUT/agg_serial.log 2>&1
echo serial done
tail -n 1 $OUT/agg_stream.log | cut -c1-300
tail -n 1 $OUT/agg_serial.log | cut -c1-1500
