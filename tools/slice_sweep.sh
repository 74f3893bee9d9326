for st in 48 64 96 150 200; do
  for mode in "--serial" ""; do
    echo "SLICE_TARGET=$st mode=$mode"
    ZKHIP_SLICE_TARGET=$st python3 bench.py $mode --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('kernel_ms_alone'))"
  done
done
