for np in 2 3 4 2 3; do
  echo -n "PROVERS=$np  "
  ZKHIP_BENCH_PROVERS=$np python3 bench.py --workload prover --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('proofs_in_flight'), d.get('last_proof_verifies'))"
done
