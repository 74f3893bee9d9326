"""N > 1 paths on CPU.  (1) world_size-2 and -3 gloo jobs exercising partition + all-gather + combine (zecale_amd/dist.py), as
bench.py --gpus N does with RCCL: the point-partitioned MSM with its streaming exchange (BASELINE configs[1] at N > 1) and the
key-partitioned prover zdist.prove_distributed (configs[3]).  (2) bench.py's own launcher: `python bench.py --gpus N` (the driver's
command) must start N ranks that rendezvous - checked with --dry-launch, which needs no GPU - and must refuse a world size that
differs from --gpus."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(world, port, script, *args):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script, *args]
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_msm_gloo(world):
    p = _torchrun(world, 29600 + world, os.path.join(ROOT, "tests", "dist_worker.py"), "msm")
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.parametrize("world", [2, 3])
def test_key_partitioned_prover_gloo(world):
    """zdist.prove_distributed in a real multi-process job: every rank proves over its slice of the key (the C oracle stands in
    for the partial MSMs), the partial sums cross the process group, every rank finishes the same proof = the golden proof /
    the oracle's whole-key proof."""
    p = _torchrun(world, 29610 + world, os.path.join(ROOT, "tests", "dist_worker.py"), "prover")
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.parametrize("world", [2, 4])
def test_bench_multi_gpu_legs_gloo(world):
    """bench.py's two N > 1 legs - `prover_2_22_partitioned` (BASELINE configs[3]) and `wrapping_replicas` (configs[4]) - in real
    multi-process jobs with stand-ins for the kernels: slices by finite terms (dist.py = zkhip_key_partition), the exchange, the finish
    on every rank = the oracle's whole-key proof, and the legs' aggregation over ranks."""
    p = _torchrun(world, 29640 + world, os.path.join(ROOT, "tests", "dist_worker.py"), "legs")
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]


def _json_line(text):
    for line in text.splitlines():
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in: " + text[-1000:])


@pytest.mark.parametrize("world", [2, 4])
def test_bench_launches_its_own_ranks(world):
    """The driver's command, `python bench.py --gpus N`, with no launcher around it: bench.py starts the N ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dry-launch"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = _json_line(p.stdout)
    assert line["n_gpus"] == world and line["ranks_seen"] == list(range(world))
    assert line["spawned_by_bench"] and line["partial_sums_combined"]


def test_bench_under_torchrun_and_world_size_mismatch():
    """The launcher form of the contract (torch.distributed.run ... bench.py --gpus N) still works, and a world size that differs
    from --gpus is refused instead of being reported as a smaller job."""
    ok = _torchrun(2, 29631, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch")
    assert ok.returncode == 0, ok.stdout[-2000:] + ok.stderr[-2000:]
    line = _json_line(ok.stdout)
    assert line["n_gpus"] == 2 and not line["spawned_by_bench"]
    bad = _torchrun(2, 29632, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-launch")
    assert bad.returncode != 0
    assert "refusing to run" in bad.stderr


def test_bench_without_a_gpu_fails_in_every_rank():
    """No silent N = 1 run and no CPU fallback: on a box without a GPU both ranks of `bench.py --gpus 2` stop at 'needs a GPU'."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stderr.count("needs a GPU") == 2, p.stderr[-2000:]
    assert "{" not in p.stdout
