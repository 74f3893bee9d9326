for wb in 16 8 4 16 8; do
  echo -n "WIT_BATCH=$wb  "
  ZKHIP_WIT_BATCH=$wb python3 bench.py --workload aggregator --gpu-witness --steps 600 --warmup 100 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('host_cores_busy'), d.get('last_proof_verifies'))"
done
echo -n "host witness  "; python3 bench.py --workload aggregator --steps 600 --warmup 100 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('host_cores_busy'))"
