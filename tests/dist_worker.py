"""Worker for tests/test_dist_cpu.py: one rank of a world_size-N gloo job.  Each rank computes the
partial MSM of its slice (with the CPU oracle standing in for the HIP kernel - this test covers the
partition + exchange + combine logic, not the kernel) and all ranks combine."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import pyref as R  # noqa: E402
from tests.helpers import aff_limbs, random_fr_canonical  # noqa: E402
from zecale_amd import dist as zdist  # noqa: E402
from zecale_amd import zkhip  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 1000 + 7
    g = aff_limbs(R.G1_GEN)
    d = O.jac_to_affine(O.scalar_mul(g, random_fr_canonical(1, 1)[0]))
    bases = O.point_progression(g, d, n)
    scal = random_fr_canonical(2, n)
    lo, hi = zdist.partition(n, world, rank)
    part = O.msm(bases[lo:hi], scal[lo:hi], chunks=1)
    parts2 = np.stack([part, O.msm(bases[lo:hi], scal[lo:hi][::-1].copy(), chunks=1)])   # two MSMs in one exchange
    total = zdist.combine_partial_sums(part)
    total2 = zdist.combine_partial_sums(parts2)
    # the non-blocking form used by the streaming bench: two exchanges in flight, read in order
    h1 = zdist.combine_partial_sums_async(part)
    h2 = zdist.combine_partial_sums_async(parts2)
    async_ok = bool((h1.result() == total).all() and (h2.result() == total2).all())
    full = O.msm(bases, scal, chunks=2)
    ok = async_ok and (zkhip.jac_to_affine(total) == O.jac_to_affine(full)).all() and (zkhip.jac_to_affine(total2[0]) == O.jac_to_affine(full)).all()
    # every rank must hold identical limbs
    import torch
    t = torch.from_numpy(total.view(np.int64).copy())
    ref = t.clone()
    dist.broadcast(ref, 0)
    ok = ok and bool((t == ref).all())
    # partition covers everything exactly once
    cover = sum(zdist.partition(n, world, r)[1] - zdist.partition(n, world, r)[0] for r in range(world))
    ok = ok and cover == n
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
