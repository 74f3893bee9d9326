"""Multi-GPU exchange for the point-partitioned MSM: every rank owns a slice of the base
points and of the scalar vector, computes its partial sum with the HIP MSM, and the ranks exchange
ONLY the partial sums (one Jacobian point, 288 bytes, per MSM).  Group addition is not a reduction
operator RCCL knows, so the "all-reduce" is an all-gather of the 288-byte points followed by the same
rank-ordered additions on every rank (exact arithmetic: every rank ends with identical limbs).
One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" for CPU tests)."""
import numpy as np
import torch
import torch.distributed as dist

from . import zkhip


def partition(n, world, rank):
    """Contiguous slice [lo, hi) of n items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def combine_partial_sums(part_jac, group=None, device=None):
    """part_jac: this rank's partial sum(s), uint64 array of shape (36,) or (k, 36).  Returns the sum over
    all ranks with the same shape.  Collective: every rank must call it."""
    p = np.ascontiguousarray(part_jac, dtype=np.uint64)
    shape = p.shape
    p2 = p.reshape(-1, 36)
    if not dist.is_initialized():
        return p.copy()
    world = dist.get_world_size(group)
    mine = torch.from_numpy(p2.view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    parts = [g.cpu().numpy().view(np.uint64) for g in gathered]
    out = parts[0].copy()
    for q in parts[1:]:
        for k in range(out.shape[0]):
            out[k] = zkhip.jac_add(out[k], q[k])
    return out.reshape(shape)


class PendingSum:
    """An exchange of partial sums in flight (combine_partial_sums_async).  result() waits for the all-gather and adds the
    ranks' points in rank order."""

    def __init__(self, shape, gathered, work, local):
        self._shape, self._gathered, self._work, self._local = shape, gathered, work, local

    def result(self):
        if self._gathered is None:
            return self._local
        if self._work is not None:
            self._work.wait()
        parts = [g.cpu().numpy().view(np.uint64) for g in self._gathered]
        out = parts[0].copy()
        for q in parts[1:]:
            for k in range(out.shape[0]):
                out[k] = zkhip.jac_add(out[k], q[k])
        return out.reshape(self._shape)


def combine_partial_sums_async(part_jac, group=None, device=None):
    """Start the exchange of this rank's partial sum(s) and return a PendingSum.  While an MSM accumulates, the chip has no free
    compute unit for the (tiny) all-gather kernel: a streaming caller starts the exchange of MSM i-1 when it collects it and reads
    the result one step later, so the collective never stalls the MSM stream.  Collective: every rank must call it, in the same order."""
    p = np.ascontiguousarray(part_jac, dtype=np.uint64)
    shape = p.shape
    if not dist.is_initialized():
        return PendingSum(shape, None, None, p.copy())
    world = dist.get_world_size(group)
    mine = torch.from_numpy(p.reshape(-1, 36).view(np.int64).copy())
    if device is not None:
        mine = mine.to(device, non_blocking=True)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    work = dist.all_gather(gathered, mine, group=group, async_op=True)
    return PendingSum(shape, gathered, work, None)


def key_slices(n_vars, n_primary, domain_size, world, rank):
    """Ranges of the A/B, H and L queries owned by `rank` (contiguous, sizes differ by at most one)."""
    return (partition(n_vars, world, rank), partition(domain_size - 1, world, rank), partition(n_vars - n_primary - 1, world, rank))


def cuts_by_weight(w, parts):
    """cuts[0 .. parts] of len(w) items: cut k is the first index at which the running weight reaches ceil(k W / parts) - contiguous
    slices of equal WEIGHT (multi_device.cpp cuts_by_weight is the same rule)."""
    w = np.asarray(w, dtype=np.uint64)
    run = np.concatenate([[0], np.cumsum(w)])
    total = int(run[-1])
    cuts = [0]
    for k in range(1, parts):
        want = -(-total * k // parts)
        cuts.append(int(np.searchsorted(run, want, side="left")))
    cuts.append(len(w))
    return cuts


def key_slices_by_finite_terms(pk, n_vars, n_primary, domain_size, world, rank):
    """Ranges of the A/B, H and L queries owned by `rank`, cut so that every rank gets the same number of FINITE bases (a base at
    infinity - a third of a real key's B query - produces no bucket entry): what zkhip_key_partition computes for the one-process
    form.  pk: dict of host arrays A, B2, B1, H, L (n x 24 limbs, all-zero = infinity)."""
    fin = lambda a: np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 24).any(axis=1).astype(np.uint64)
    wa = fin(pk["A"]) + fin(pk["B2"]) + fin(pk["B1"])
    assert len(wa) == n_vars and len(pk["H"]) == domain_size - 1 and len(pk["L"]) == n_vars - n_primary - 1
    ac, hc, lc = cuts_by_weight(wa, world), cuts_by_weight(fin(pk["H"]), world), cuts_by_weight(fin(pk["L"]), world)
    return (ac[rank], ac[rank + 1]), (hc[rank], hc[rank + 1]), (lc[rank], lc[rank + 1])


def prove_distributed(crs_slice, r1cs, pk_consts, z, r, s, group=None, device=None, backend=None):
    """One Groth16 proof with the proving key partitioned over the ranks of `group`: every rank runs the (cheap, replicated)
    QAP map and the five MSMs over its slice, the 5 x 288-byte partial sums are all-gathered and added in rank order on
    every rank, and every rank finishes the same proof.  Collective.
    backend: the object that provides groth16_prove_partial / groth16_finish (default: the HIP library; the gloo tests pass a
    stand-in for the partial sums so that the partition + exchange + finish logic runs without a GPU)."""
    be = zkhip if backend is None else backend
    sums = be.groth16_prove_partial(crs_slice, r1cs, z)
    total = combine_partial_sums(sums, group=group, device=device)
    return be.groth16_finish(pk_consts, total, r, s)
