# A/B of the accumulation gate of the MSM stream (ZKHIP_MSM_GATE), three runs each
for g in 1 0 1 0 1 0; do
  echo -n "GATE=$g  "
  ZKHIP_MSM_GATE=$g python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('kernel_ms_alone'))"
done
