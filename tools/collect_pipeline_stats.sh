#!/bin/bash
# Host side of the streaming prover, before / after round 5, on ONE box: the round-4 tree (exported and built under
# build/r04_tree by `git archive 1b01b24 | tar -x -C build/r04_tree` + its own build()) and this tree run the same stream -
# `bench.py --workload aggregator --steps 2000 --warmup 100` - with ZKHIP_PIPELINE_STATS=1 (per-proof times inside the pipeline, printed
# when it is freed) and ZKHIP_BENCH_THREADS=1 (CPU seconds per thread name).  Output: gpurun_out/pipeline_stats/{r04,r05}*.log ;
# tools/summarise_pipeline_stats.py turns them into profiles/r05_pipeline_host_side.txt.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pipeline_stats
mkdir -p $OUT
export ZKHIP_PIPELINE_STATS=1 ZKHIP_BENCH_THREADS=1
cd $ROOT
timeout -k 10 300 python3 bench.py --workload aggregator --steps 2000 --warmup 100 --no-cpu-baseline > $OUT/r05_host.log 2>&1
timeout -k 10 300 python3 bench.py --workload aggregator --steps 2000 --warmup 300 --no-cpu-baseline --gpu-witness --witness-workers 4 > $OUT/r05_gpu.log 2>&1
timeout -k 10 300 python3 bench.py --workload aggregator --steps 2000 --warmup 100 --no-cpu-baseline --no-app-cache > $OUT/r05_host_nocache.log 2>&1
if [ -d $ROOT/build/r04_tree ]; then
  cd $ROOT/build/r04_tree
  timeout -k 10 300 python3 bench.py --workload aggregator --steps 2000 --warmup 100 --no-cpu-baseline > $OUT/r04_host.log 2>&1
  timeout -k 10 300 python3 bench.py --workload aggregator --steps 2000 --warmup 300 --no-cpu-baseline --gpu-witness --witness-workers 4 > $OUT/r04_gpu.log 2>&1
fi
grep -c '"value"' $OUT/*.log
