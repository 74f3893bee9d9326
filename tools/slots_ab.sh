# prover slots of the streaming wrapping prover (bench.py --workload aggregator --gpu-slots N), alternating to see the run-to-run noise
python3 bench.py --workload aggregator --steps 300 --warmup 50 --no-cpu-baseline > /dev/null 2>&1     # (a first run: clocks, page cache)
for gs in 14 24 14 24 20 14 24; do
  echo -n "gpu_slots=$gs  "
  python3 bench.py --workload aggregator --gpu-slots $gs --steps 600 --warmup 100 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('host_cores_busy'))"
done
