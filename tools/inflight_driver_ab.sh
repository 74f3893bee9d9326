for ns in 2 4 6 8 4 6 8; do
  echo -n "INFLIGHT=$ns steps 20/5  "
  ZKHIP_BENCH_INFLIGHT=$ns python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], d['ms_per_step'], r.get('kernel_active_ms_per_step'))"
done
