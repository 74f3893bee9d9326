for qb in 65536 1024 8192 65536 1024; do
  echo -n "QUAD_BELOW=$qb  "
  ZKHIP_QUAD_BELOW=$qb python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('stream', d['value'], d['ms_per_step'])"
done
