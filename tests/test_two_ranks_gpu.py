"""N = 2 on the one-GPU test box: two ranks started by bench.py's own launcher share device 0 and exchange their partial sums over
gloo (RCCL refuses two ranks on one device).  Everything but RCCL itself runs as it would on two GPUs - the partition of the key, the
HIP kernels of each rank, the exchange, the finish on every rank - and the proof that comes out is verified by the host pairing check:
BASELINE configs[3] (MSM point-partitioned, partial sums combined) in a real multi-process job."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ZKHIP_BENCH_SHARE_GPU="1", ZKHIP_BENCH_BACKEND="gloo", ZKHIP_BENCH_PARTITION_LOG="16")      # (the N > 1 legs at a rehearsal size)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def test_key_partitioned_prover_across_two_processes():
    line = _bench("--gpus", "2", "--workload", "prover", "--log-n", "16", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["last_proof_verifies"] is True
    assert "rehearsal" in line                       # the line must not pass for a two-GPU measurement


def test_point_partitioned_msm_across_two_processes():
    line = _bench("--gpus", "2", "--log-n", "16", "--steps", "4", "--warmup", "1", "--no-cpu-baseline")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["terms_per_gpu"] == 1 << 16
    assert line["combined_result_matches_closed_form"] is True        # sum over BOTH ranks' slices = (sum s_i k_i) G
    # the driver's N > 1 command also measures BASELINE configs[3] and [4]: one proof per step over the key cut two
    # ways (verified on every rank), and one streaming prover per rank on the nine-input circuit
    part, rep = line["prover_2_22_partitioned"], line["wrapping_replicas"]
    assert part["n_gpus"] == 2 and part["scaling"] == "strong" and part["last_proof_verifies_on_every_rank"] is True and part["value"] > 0
    assert part["constraints"] == (1 << 16) - 8 and "exchange_and_additions" in part["phase_ms_slowest_rank"]
    assert rep["n_gpus"] == 2 and rep["scaling"] == "weak" and rep["last_proof_verifies_on_every_rank"] is True and rep["proofs_per_rank"] == 576
