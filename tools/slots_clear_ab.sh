#!/bin/bash
# ZZ-only clear kernel (ZKHIP_SLOTS_CLEAR=0) against one memset of the whole slot array (=1), interleaved on one box.  -> gpurun_out/r06g/
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r06g
mkdir -p $OUT
cd $ROOT
msm_line() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$1', '%.3f Mscalar/s' % d['value'], 'ms_per_step %.3f' % d['ms_per_step'], 'k_accumulate<1> alone %.3f ms' % r['kernel_ms'], 'mad peak this run %.2f G/s' % r['fq_mul_peak_this_run_g_per_s'])"; }
val_line() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], d['unit'], 'ms_per_step', d['ms_per_step'], 'verifies', d.get('last_proof_verifies'))"; }
for rep in 1 2 3; do
  for v in 0 1; do
    export ZKHIP_SLOTS_CLEAR=$v
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | msm_line "msm stream  clear=$v" >> $OUT/ab.txt
    python3 bench.py --serial --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | msm_line "msm serial  clear=$v" >> $OUT/ab.txt
  done
done
for rep in 1 2; do
  for v in 0 1; do
    export ZKHIP_SLOTS_CLEAR=$v
    python3 bench.py --workload prover --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | val_line "prover 2^20  clear=$v" >> $OUT/ab.txt
    python3 bench.py --workload aggregator --steps 800 --warmup 80 --no-cpu-baseline 2>/dev/null | val_line "wrapping    clear=$v" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
