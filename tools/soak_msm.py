"""Randomised soak of the MSM against the CPU oracle (test infrastructure: run by hand on the GPU box, like the tests).
Adversarial mixes: repeated points (doubling inside buckets), P / -P pairs (cancellation to infinity), bases at infinity,
scalars 0 / 1 / r-1 / equal scalars (heavy buckets), random sizes, plain and table-backed base sets with random windows,
sub-ranges, both kinds of window table (one level per window / every bit position with NAF scalars).  Usage: python tools/soak_msm.py [cases]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from oracle import pyref as R
from tests.helpers import aff_limbs, random_fr_canonical
from zecale_amd import zkhip

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
zkhip.init(0)
O.load()
rng = np.random.default_rng(2024)
g = aff_limbs(R.G1_GEN)
pool = zkhip.fixed_base_mul(g, random_fr_canonical(5, 4096), montgomery=False)      # distinct points to draw from
rm1 = np.array(R.int_to_limbs(R.R_MOD - 1, 6), dtype=np.uint64)
bad, t0 = 0, time.time()
for case in range(cases):
    n = int(rng.choice([1, 2, 3, 17, 255, 256, 257, 1000, 4097, 12000, 30000]))
    idx = rng.integers(0, 4096 if rng.random() < 0.5 else 8, n)                        # few distinct points: many equal-x additions
    bases = pool[idx].copy()
    neg = rng.random(n) < 0.2
    for i in np.nonzero(neg)[0]:                                                       # -P: negate y (Montgomery limbs of q - y)
        bases[i, 12:] = O.f_op("sub", 0, np.zeros(12, dtype=np.uint64), bases[i, 12:])
    bases[rng.random(n) < 0.05] = 0                                                    # infinity
    scal = random_fr_canonical(1000 + case, n)
    sel = rng.random(n)
    scal[sel < 0.1] = 0
    scal[(sel >= 0.1) & (sel < 0.25), :] = 0
    scal[(sel >= 0.1) & (sel < 0.25), 0] = 1
    scal[(sel >= 0.25) & (sel < 0.3)] = rm1
    scal[(sel >= 0.3) & (sel < 0.4)] = scal[0]
    scal_m = np.array([O.f_op("from_canonical", 1, s) for s in scal])                  # the oracle takes Montgomery scalars
    exp = O.jac_to_affine(O.msm(bases, scal_m))
    b = zkhip.Bases.upload(bases)
    mode = rng.integers(0, 3)
    if mode == 0:
        zkhip.set_msm_window(int(rng.choice([0, 4, 7, 11, 14])))
    else:
        zkhip.set_table_naf(int(rng.integers(0, 2)))                                       # either kind of window table
        b.precompute(int(rng.choice([0, 4, 9, 13, 17, 20, 22])))
        zkhip.set_table_naf(-1)
    got = zkhip.jac_to_affine(b.msm(scal, montgomery=False))
    ok = (got == exp).all()
    if ok and n > 10:                                                                   # a sub-range too
        off = int(rng.integers(0, n // 2)); ln = int(rng.integers(1, n - off))
        ok = (zkhip.jac_to_affine(b.msm(scal[:ln], offset=off, montgomery=False)) == O.jac_to_affine(O.msm(bases[off:off + ln], scal_m[:ln]))).all()
    zkhip.set_msm_window(0)
    b.free()
    if not ok:
        bad += 1
        print("MISMATCH case", case, "n", n, "mode", mode, flush=True)
    if case % 25 == 24:
        print(f"{case + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"soak: {cases} MSMs against the oracle, mismatches: {bad}")
sys.exit(1 if bad else 0)
