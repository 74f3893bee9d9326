"""Worker for tests/test_dist_cpu.py: one rank of a world_size-N gloo job.  The CPU oracle stands in for the HIP kernels (this
test covers the partition + exchange + combine + finish logic of zecale_amd/dist.py, not the kernels):
  * the point-partitioned MSM (BASELINE configs[1] at N > 1): blocking and streaming exchange, several exchanges in flight,
    read in submission order as bench.py's MSM stream does;
  * the key-partitioned Groth16 prover (BASELINE configs[3]): zdist.prove_distributed over key_slices, against the whole-key
    proof of the oracle and the golden proof."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import pyref as R  # noqa: E402
from tests.helpers import (aff_limbs, csr_from_rows, fr_array, fr_limbs, golden, h2i, make_r1cs, pt_from_json,  # noqa: E402
                           random_fr_canonical, random_fr_uniform)
from zecale_amd import dist as zdist  # noqa: E402
from zecale_amd import zkhip  # noqa: E402


class KeySlice:
    """What a rank holds of a proving key: its ranges of the five query vectors (the shape of zkhip.Crs.upload_slice)."""

    def __init__(self, pk, m, l, d, world, rank):
        self.ranges = zdist.key_slices(m, l, d, world, rank)
        (a0, a1), (h0, h1), (l0, l1) = self.ranges
        self.A, self.B2, self.B1 = pk["A"][a0:a1], pk["B2"][a0:a1], pk["B1"][a0:a1]
        self.H, self.L = pk["H"][h0:h1], pk["L"][l0:l1]


class OracleBackend:
    """groth16_prove_partial by the C oracle (the stand-in for zkhip_groth16_prove_partial), groth16_finish by the product's
    host tail (zkhip_groth16_finish runs on the host: no GPU needed)."""

    def __init__(self, csr, n, l):
        self.csr, self.n, self.l = csr, n, l

    def groth16_prove_partial(self, ks, r1cs, z):
        (a0, a1), (h0, h1), (l0, l1) = ks.ranges
        h = O.qap_h(*self.csr, z, self.n, self.l)                # replicated on every rank, like the QAP map on the GPU
        za, zl = z[a0:a1], z[self.l + 1 + l0:self.l + 1 + l1]
        msm = lambda b, s_: O.msm(b, s_, chunks=1) if len(b) else O.aff_to_jac(np.zeros(24, dtype=np.uint64))
        return np.stack([msm(ks.A, za), msm(ks.B2, za), msm(ks.B1, za), msm(ks.H, h[h0:h1]), msm(ks.L, zl)])

    groth16_finish = staticmethod(zkhip.groth16_finish)


def cpu_key(A, B, C, m, l, trapdoor):
    """Proving key of a small system with known toxic waste: exponents from oracle/pyref, group elements by the C oracle."""
    tau, alpha, beta, delta = trapdoor
    st = R.groth16_setup_scalars(A, B, C, m, l, tau, alpha, beta, delta)
    d = st["d"]
    dinv = pow(delta, -1, R.R_MOD)
    g1, g2 = aff_limbs(R.G1_GEN), aff_limbs(R.G2_GEN)
    fb = lambda g, xs: (np.array([O.jac_to_affine(O.scalar_mul(g, fr_limbs(x % R.R_MOD))) for x in xs], dtype=np.uint64).reshape(-1, 24)
                        if len(xs) else np.zeros((0, 24), dtype=np.uint64))
    hs, t = [], st["Zt"] * dinv % R.R_MOD
    for _ in range(d - 1):
        hs.append(t)
        t = t * tau % R.R_MOD
    ls = [(beta * st["At"][i] + alpha * st["Bt"][i] + st["Ct"][i]) * dinv % R.R_MOD for i in range(l + 1, m)]
    pk = dict(alpha_g1=fb(g1, [alpha])[0], beta_g1=fb(g1, [beta])[0], beta_g2=fb(g2, [beta])[0], delta_g1=fb(g1, [delta])[0],
              delta_g2=fb(g2, [delta])[0], A=fb(g1, st["At"]), B2=fb(g2, st["Bt"]), B1=fb(g1, st["Bt"]), H=fb(g1, hs), L=fb(g1, ls))
    return pk, d


def same_on_all_ranks(arr):
    t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).copy())
    ref = t.clone()
    dist.broadcast(ref, 0)
    return bool((t == ref).all())


def check_msm(rank, world):
    n = 1000 + 7
    g = aff_limbs(R.G1_GEN)
    d = O.jac_to_affine(O.scalar_mul(g, random_fr_canonical(1, 1)[0]))
    bases = O.point_progression(g, d, n)
    scal = fr_array_mont(random_fr_uniform(2, n))                 # uniform in [0, r): the top bit is exercised
    lo, hi = zdist.partition(n, world, rank)
    part = O.msm(bases[lo:hi], scal[lo:hi], chunks=1)
    parts2 = np.stack([part, O.msm(bases[lo:hi], scal[lo:hi][::-1].copy(), chunks=1)])   # two MSMs in one exchange
    total = zdist.combine_partial_sums(part)
    total2 = zdist.combine_partial_sums(parts2)
    # the streaming form of bench.py's MSM loop: the exchange of MSM i-1 is started when it is collected and read one step
    # later, so several exchanges are in flight and must come back in submission order
    stream = [part, parts2, parts2[1], part]
    pending, got = [], []
    for p in stream:
        if pending:
            got.append(pending.pop(0).result())
        pending.append(zdist.combine_partial_sums_async(p))
    while pending:
        got.append(pending.pop(0).result())
    async_ok = (bool((got[0] == total).all()) and bool((got[1] == total2).all()) and bool((got[2] == total2[1]).all())
                and bool((got[3] == total).all()))
    full = O.msm(bases, scal, chunks=2)
    ok = async_ok and (zkhip.jac_to_affine(total) == O.jac_to_affine(full)).all() and (zkhip.jac_to_affine(total2[0]) == O.jac_to_affine(full)).all()
    ok = ok and same_on_all_ranks(total)                        # every rank must hold identical limbs
    cover = sum(zdist.partition(n, world, r)[1] - zdist.partition(n, world, r)[0] for r in range(world))
    return bool(ok) and cover == n                              # the partition covers everything exactly once


def fr_array_mont(canon):
    """canonical limbs -> Montgomery limbs (the ABI's scalar form)."""
    return fr_array([R.limbs_to_int(row) for row in canon])


def check_prover(rank, world):
    # (1) the golden instance: 7 variables - slices of two or three points, an empty slice for some ranks at world 3
    g = golden("groth16_small.json")
    pts = lambda L: np.array([aff_limbs(pt_from_json(p)) for p in L]).reshape(-1, 24)
    pk = {k: (aff_limbs(pt_from_json(v)) if k in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2") else pts(v))
          for k, v in g["pk"].items()}
    csr = tuple(csr_from_rows(g[k]) for k in "ABC")
    z = fr_array([h2i(x) for x in g["z"]])
    m, l, d = len(g["z"]), g["n_primary"], 1 << g["log_d"]
    r, s = fr_limbs(h2i(g["r"])), fr_limbs(h2i(g["s"]))
    be = OracleBackend(csr, len(csr[0][0]) - 1, l)
    proof = zdist.prove_distributed(KeySlice(pk, m, l, d, world, rank), None, pk, z, r, s, backend=be)
    exp = np.concatenate([aff_limbs(pt_from_json(g["proof"][k])) for k in "abc"])
    ok = bool((proof == exp).all()) and same_on_all_ranks(proof)
    # (2) a 150-constraint system with its own key: uneven slices, compared with the oracle's whole-key proof
    n, l2 = 150, 4
    A, B, C, zi = make_r1cs(5, n, l2, n, 0.3)
    m2 = len(zi)
    pk2, d2 = cpu_key(A, B, C, m2, l2, (0x1234567, 0x2345678, 0x3456789, 0x456789a))
    csr2 = tuple(csr_from_rows(x) for x in (A, B, C))
    z2 = fr_array(zi)
    r2, s2 = fr_array_mont(random_fr_uniform(7, 1))[0], fr_array_mont(random_fr_uniform(8, 1))[0]
    be2 = OracleBackend(csr2, n, l2)
    proof2 = zdist.prove_distributed(KeySlice(pk2, m2, l2, d2, world, rank), None, pk2, z2, r2, s2, backend=be2)
    whole = O.groth16_prove(pk2, z2, l2, O.qap_h(*csr2, z2, n, l2), r2, s2, chunks=1)
    ok = ok and bool((proof2 == whole).all()) and same_on_all_ranks(proof2)
    # the slices of all ranks tile the three query ranges
    for which, total in ((0, m2), (1, d2 - 1), (2, m2 - l2 - 1)):
        sl = [zdist.key_slices(m2, l2, d2, world, rk)[which] for rk in range(world)]
        ok = ok and sl[0][0] == 0 and sl[-1][1] == total and all(sl[i][1] == sl[i + 1][0] for i in range(world - 1))
    return ok


def check_bench_legs(rank, world):
    """bench.py's N > 1 legs under gloo: partitioned_prover_leg - BASELINE configs[3]: one proof per step over a key
    cut by FINITE terms, the exchange of zecale_amd/dist.py, every rank finishing the same proof - with the C oracle standing in for
    the partial MSMs, and replicas_leg - configs[4]: independent streams, no collective on the data path - with a stub stream; the
    legs' own aggregation (slowest rank's time and phases, all ranks' proofs, every rank's verification) is what is checked."""
    import json
    import time
    import bench
    n, l = 150, 4
    A, B, C, zi = make_r1cs(5, n, l, n, 0.3)
    m = len(zi)
    pk, d = cpu_key(A, B, C, m, l, (0x1234567, 0x2345678, 0x3456789, 0x456789a))
    pk["B2"][::3] = 0; pk["B1"][::3] = 0                                # a sparse B query, as a real key has (the proof then differs from
    csr = tuple(csr_from_rows(x) for x in (A, B, C))                    # the trapdoor's: it is compared with the oracle's whole-key proof)
    z = fr_array(zi)
    r_, s_ = fr_array_mont(random_fr_uniform(7, 1))[0], fr_array_mont(random_fr_uniform(8, 1))[0]
    whole = O.groth16_prove(pk, z, l, O.qap_h(*csr, z, n, l), r_, s_, chunks=1)
    ranges = zdist.key_slices_by_finite_terms(pk, m, l, d, world, rank)
    # the slices tile the queries, carry equal finite terms (+-1 per cut) and agree with the library's own rule (zkhip_key_partition)
    ac, hc, lc = zkhip.key_partition(pk, m, l, d, world)
    ok = ranges == ((ac[rank], ac[rank + 1]), (hc[rank], hc[rank + 1]), (lc[rank], lc[rank + 1])) and ac[0] == 0 and ac[-1] == m and hc[-1] == d - 1 and lc[-1] == m - l - 1
    fin = lambda a: int(np.asarray(a).reshape(-1, 24).any(axis=1).sum())
    w_all = [fin(pk["A"][ac[k]:ac[k + 1]]) + fin(pk["B2"][ac[k]:ac[k + 1]]) + fin(pk["B1"][ac[k]:ac[k + 1]]) for k in range(world)]
    ok = ok and max(w_all) - min(w_all) <= 6                         # (an index weighs up to 3: a cut lands within 3 of its target)
    if not ok:
        print('slices:', ranges, ac, hc, lc, w_all, flush=True)

    class Slice:
        pass
    ks = Slice()
    ks.ranges = ranges
    (a0, a1), (h0, h1), (l0, l1) = ranges
    ks.A, ks.B2, ks.B1, ks.H, ks.L = pk["A"][a0:a1], pk["B2"][a0:a1], pk["B1"][a0:a1], pk["H"][h0:h1], pk["L"][l0:l1]
    be = OracleBackend(csr, n, l)

    class Prover:
        def prove(self):
            t = time.time()
            sums = be.groth16_prove_partial(ks, None, z)
            self.ms = (time.time() - t) * 1e3
            return zkhip.groth16_finish(pk, zdist.combine_partial_sums(sums), r_, s_)

        def verify(self, proof):
            return bool((proof == whole).all())

        def phases(self):
            return {"partial_sums": self.ms}

        def info(self):
            return {"constraints": n}
    barrier = dist.barrier

    def red(x, op):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=op)
        return float(t.item())
    rmax, rsum = (lambda x: red(x, dist.ReduceOp.MAX)), (lambda x: red(x, dist.ReduceOp.SUM))
    leg = bench.partitioned_prover_leg(world, rank, Prover(), 2, 1, barrier, rmax, 8)

    class Stream:
        def run(self, k_warm, k_timed):
            time.sleep(0.01 * (rank + 1))                              # rank r takes (r + 1) x 10 ms: the slowest rank sets the rate
            return 0.01 * (rank + 1), 0.02 * (rank + 1), True
    rep = bench.replicas_leg(world, rank, Stream(), 3, 16, barrier, rmax, rsum, "stub")
    if rank == 0:
        json.dumps(leg); json.dumps(rep)                               # both must serialise into the bench line
        ok = ok and leg["n_gpus"] == world and leg["scaling"] == "strong" and leg["last_proof_verifies_on_every_rank"] is True and leg["value"] > 0
        ok = ok and "partial_sums" in leg["phase_ms_slowest_rank"] and leg["steps"] == 2
        ok = ok and rep["n_gpus"] == world and rep["scaling"] == "weak" and rep["proofs_per_rank"] == 48
        ok = ok and abs(rep["value"] - 48 * world / (0.01 * world)) < 1e-3 * rep["value"] and abs(rep["host_cores_busy_per_rank_max"] - 2.0) < 0.01
    else:
        ok = ok and leg is None and rep is None
    return bool(ok)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    which = sys.argv[1] if len(sys.argv) > 1 else "msm"
    ok = check_msm(rank, world) if which == "msm" else check_bench_legs(rank, world) if which == "legs" else check_prover(rank, world)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
