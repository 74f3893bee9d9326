# MSMs in flight in the bench's stream (zkhip_msm_submit / collect slots): throughput at 2^20 terms
for ns in 2 4 6 8 3 5 4 8; do
  echo -n "INFLIGHT=$ns  "
  ZKHIP_BENCH_INFLIGHT=$ns python3 bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel_ms'], r.get('kernel_active_ms_per_step'), d.get('msm_in_flight'), d.get('host_cores_busy'))"
done
