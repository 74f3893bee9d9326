#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun from the repo root):
#   kernel traces (--kernel-trace --stats) of the three bench workloads, PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs,
#   no trace domains next to --pmc) of the default bench command and of the calibration kernels.
# Output: gpurun_out/prof/<name>/...csv ; tools/summarise_profiles.py turns them into profiles/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 500 rocprofv3 "$@" > $OUT/$name.log 2>&1; echo "$name done"; }
run msm_trace        --kernel-trace --stats --output-format csv -d $OUT/msm_trace -o msm -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline
run prover_trace     --kernel-trace --stats --output-format csv -d $OUT/prover_trace -o prover -- python3 $ROOT/bench.py --workload prover --steps 5 --warmup 1 --no-cpu-baseline
run agg_trace        --kernel-trace --stats --output-format csv -d $OUT/agg_trace -o agg -- python3 $ROOT/bench.py --workload aggregator --steps 40 --warmup 4 --no-cpu-baseline
run agg_serial_trace --kernel-trace --stats --output-format csv -d $OUT/agg_serial_trace -o agg_serial -- python3 $ROOT/bench.py --workload aggregator --serial --steps 20 --warmup 3 --no-cpu-baseline
run msm_fetch        --pmc FETCH_SIZE --output-format csv -d $OUT/msm_fetch -o msm -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline
run msm_write        --pmc WRITE_SIZE --output-format csv -d $OUT/msm_write -o msm -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline
run calib_fetch      --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -o calib -- $ROOT/build/fetch_calib
run calib_write      --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -o calib -- $ROOT/build/fetch_calib
ls -R $OUT | grep -c csv
