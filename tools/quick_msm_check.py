"""Quick GPU sanity check of the MSM against the pure-Python oracle (small n)."""
import random, sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import pyref as R
from zecale_amd import zkhip

def aff_limbs(P):
    if P is None: return [0]*24
    return R.int_to_limbs(R.to_mont(P[0], R.Q_MOD, 12), 12) + R.int_to_limbs(R.to_mont(P[1], R.Q_MOD, 12), 12)
def from_aff(a):
    x = R.from_mont(R.limbs_to_int(a[:12]), R.Q_MOD, 12); y = R.from_mont(R.limbs_to_int(a[12:]), R.Q_MOD, 12)
    return None if (x == 0 and y == 0) else (x, y)

random.seed(5)
zkhip.init(0)
for n, c in ((1, 8), (2, 8), (33, 8), (200, 8), (200, 5), (300, 10), (1000, 12)):
    D = R.ec_mul(random.randrange(R.R_MOD), R.G1_GEN)
    P = R.ec_mul(random.randrange(R.R_MOD), R.G1_GEN)
    pts = []
    for i in range(n):
        pts.append(P); P = R.ec_add(P, D)
    sc = [random.randrange(R.R_MOD) for _ in range(n)]
    if n >= 33:
        sc[0] = 0; sc[1] = 1; sc[2] = R.R_MOD - 1; pts[5] = pts[4]; pts[7] = R.ec_neg(pts[6]); sc[7] = sc[6]; pts[9] = None
    zkhip.set_msm_window(c)
    bases = np.array([aff_limbs(p) for p in pts], dtype=np.uint64)
    scal = np.array([R.int_to_limbs(R.to_mont(s, R.R_MOD, 6), 6) for s in sc], dtype=np.uint64)
    t = time.time()
    out = zkhip.msm_raw(bases, scal)
    dt = time.time() - t
    got = from_aff(zkhip.jac_to_affine(out))
    exp = R.msm_pippenger(sc, [p for p in pts], c=8) if n > 40 else R.msm_naive(sc, pts)
    print(n, c, "OK" if got == exp else "MISMATCH", "%.3fs" % dt, "acc_ms=%.3f" % zkhip.last_accumulate_ms(), flush=True)
    assert got == exp
print("all ok")
