for tl in 1024 2048 4096 512; do
  echo -n "SORT_TILE=$tl  "
  ZKHIP_SORT_TILE=$tl python3 bench.py --serial --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('serial', d['value'], d['ms_per_step'], end='   ')"
  ZKHIP_SORT_TILE=$tl python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('stream', d['value'], d['ms_per_step'])"
done
