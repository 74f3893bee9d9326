#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun from the repo root):
#   kernel traces (--kernel-trace --stats) of the DRIVER'S EXACT COMMAND (python3 bench.py --gpus 1 --steps 20 --warmup 5: the MSM
#   stream plus the secondaries of the default line), of one MSM at a time (--serial: every kernel alone on the chip - the
#   solo side of the solo / overlapped split), of the prover and aggregator workloads; PMC passes (FETCH_SIZE, WRITE_SIZE: separate
#   runs, no trace domains next to --pmc) of the serial MSM command and of the calibration kernels.
# Output: gpurun_out/prof/<name>/...csv ; tools/summarise_profiles.py <tag> turns them into profiles/<tag>_*.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
[ -x $ROOT/build/fetch_calib ] || hipcc --offload-arch=gfx950 -O3 -o $ROOT/build/fetch_calib $ROOT/tools/ubench/fetch_calib.hip
cd /tmp && export TMPDIR=/tmp
# (the per-dispatch traces of the full bench are tens of MiB each and gpurun copies back at most 64 MiB: only the --stats summaries and
#  the counter tables are kept)
run() { name=$1; shift; timeout -k 10 500 rocprofv3 "$@" > $OUT/$name.log 2>&1; find $OUT/$name -name "*kernel_trace.csv" -delete 2>/dev/null; find $OUT/$name -name "*.db" -delete 2>/dev/null; echo "$name done"; }
run msm_trace        --kernel-trace --stats --output-format csv -d $OUT/msm_trace -o msm -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5
run msm_only_trace   --kernel-trace --stats --output-format csv -d $OUT/msm_only_trace -o msm -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary
run msm_serial_trace --kernel-trace --stats --output-format csv -d $OUT/msm_serial_trace -o msm -- python3 $ROOT/bench.py --serial --steps 10 --warmup 2 --no-cpu-baseline --no-secondary
run prover_trace     --kernel-trace --stats --output-format csv -d $OUT/prover_trace -o prover -- python3 $ROOT/bench.py --workload prover --steps 5 --warmup 1 --no-cpu-baseline
run agg_trace        --kernel-trace --stats --output-format csv -d $OUT/agg_trace -o agg -- python3 $ROOT/bench.py --workload aggregator --steps 400 --warmup 40 --no-cpu-baseline
run agg_serial_trace --kernel-trace --stats --output-format csv -d $OUT/agg_serial_trace -o agg_serial -- python3 $ROOT/bench.py --workload aggregator --serial --steps 20 --warmup 3 --no-cpu-baseline
run msm_fetch        --pmc FETCH_SIZE --output-format csv -d $OUT/msm_fetch -o msm -- python3 $ROOT/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
run msm_write        --pmc WRITE_SIZE --output-format csv -d $OUT/msm_write -o msm -- python3 $ROOT/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
run calib_fetch      --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -o calib -- $ROOT/build/fetch_calib
run calib_write      --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -o calib -- $ROOT/build/fetch_calib
run prover_serial_trace --kernel-trace --stats --output-format csv -d $OUT/prover_serial_trace -o prover_serial -- python3 $ROOT/bench.py --workload prover --serial --steps 4 --warmup 1 --no-cpu-baseline
ls -R $OUT | grep -c csv
