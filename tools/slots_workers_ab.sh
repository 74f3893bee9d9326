python3 bench.py --workload aggregator --steps 300 --warmup 50 --no-cpu-baseline > /dev/null 2>&1
for cfg in "24 8" "32 8" "24 12" "32 12" "28 10" "24 8"; do
  set -- $cfg
  echo -n "gpu_slots=$1 witness_workers=$2  "
  python3 bench.py --workload aggregator --gpu-slots $1 --witness-workers $2 --steps 600 --warmup 100 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('host_cores_busy'))"
done
