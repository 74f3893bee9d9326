for pp in 1024 512 2048 1024 512; do
  echo -n "SORT_PARTS=$pp  "
  ZKHIP_SORT_PARTS=$pp python3 bench.py --serial --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('serial', d['value'], d['ms_per_step'], end='   ')"
  ZKHIP_SORT_PARTS=$pp python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('stream', d['value'], d['ms_per_step'])"
done
