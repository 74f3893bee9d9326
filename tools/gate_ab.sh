# A/B of the accumulation gate of the MSM stream (ZKHIP_MSM_GATE): eight MSMs in flight, the driver's step counts
for g in 1 0 1 0 1 0; do
  echo -n "GATE=$g  "
  ZKHIP_MSM_GATE=$g python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel_ms'], r.get('kernel_ms_mean_of_overlapping_launches'), r['kernel_ms_alone'])"
done
