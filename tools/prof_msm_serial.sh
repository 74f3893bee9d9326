#!/bin/bash
# Per-kernel breakdown of ONE 2^20 G1 MSM at a time (bench.py --serial) and of the streamed default, under rocprofv3 --kernel-trace --stats.
# usage (through gpurun, from the repo root): bash tools/prof_msm_serial.sh <tag>
set -e
TAG=${1:-cur}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o msm -- python3 $ROOT/bench.py --serial --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/serial.log 2>&1
echo serial done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stream -o msm -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/stream.log 2>&1
echo stream done
tail -n 1 $OUT/serial.log | cut -c1-400
tail -n 1 $OUT/stream.log | cut -c1-400
