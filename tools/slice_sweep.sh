# slice length of k_accumulate (ZKHIP_SLICE_TARGET -> S = entries / (131072 x fills)) in the default stream (eight MSMs in flight)
for st in 48 64 80 96 128 48 64; do
  echo -n "SLICE_TARGET=$st  "
  ZKHIP_SLICE_TARGET=$st python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_alone'])"
done
