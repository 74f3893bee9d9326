#!/bin/bash
# N = 2 REHEARSAL on a one-GPU box: two ranks started by bench.py's own launcher share device 0, partial sums cross the process group
# over gloo (RCCL refuses two ranks on one device).  Everything but RCCL itself runs as it would on two GPUs: the partition, the
# kernels, the streaming exchange, the key-partitioned prover (the last proof is VERIFIED), the replicas.  Not a scaling measurement.
export ZKHIP_BENCH_SHARE_GPU=1 ZKHIP_BENCH_BACKEND=gloo
set -e
echo "--- msm, two ranks"
python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print({k: d[k] for k in ('value','n_gpus','ms_per_step','scaling','rehearsal')}, d['config']['parallelism'])"
echo "--- key-partitioned prover, 2^18 constraints, two ranks"
python3 bench.py --gpus 2 --workload prover --log-n 18 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print({k: d.get(k) for k in ('value','n_gpus','ms_per_step','scaling','last_proof_verifies')}, d['config']['parallelism'])"
echo "--- replicas of the wrapping prover, two ranks"
python3 bench.py --gpus 2 --workload aggregator --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print({k: d.get(k) for k in ('value','n_gpus','ms_per_step','last_proof_verifies')}, d['config']['parallelism'])"
