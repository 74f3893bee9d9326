"""Conversions between the golden files' canonical hex integers and ABI limb arrays."""
import json
import os

import numpy as np

from oracle import pyref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def h2i(s):
    return int(s, 16)


def fq_limbs(x):
    return np.array(R.int_to_limbs(R.to_mont(x, R.Q_MOD, 12), 12), dtype=np.uint64)


def fr_limbs(x):
    return np.array(R.int_to_limbs(R.to_mont(x, R.R_MOD, 6), 6), dtype=np.uint64)


def fq_int(a):
    return R.from_mont(R.limbs_to_int(a), R.Q_MOD, 12)


def fr_int(a):
    return R.from_mont(R.limbs_to_int(a), R.R_MOD, 6)


def pt_from_json(p):
    return None if p is None else (h2i(p[0]), h2i(p[1]))


def aff_limbs(P):
    if P is None:
        return np.zeros(24, dtype=np.uint64)
    return np.concatenate([fq_limbs(P[0]), fq_limbs(P[1])])


def aff_point(a):
    x, y = fq_int(a[:12]), fq_int(a[12:24])
    return None if (x == 0 and y == 0) else (x, y)


def fr_array(ints):
    return np.array([R.int_to_limbs(R.to_mont(x, R.R_MOD, 6), 6) for x in ints], dtype=np.uint64).reshape(-1, 6)


def fr_ints(arr):
    return [fr_int(r) for r in np.asarray(arr).reshape(-1, 6)]


def splitmix64(seed, n):
    """n pseudo-random u64 (numpy, vectorised)."""
    x = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)).astype(np.uint64)
    x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


def random_fr_canonical(seed, n):
    """n x 6 limbs of CANONICAL scalars in [0, 2^376) (< r): top limb masked to 56 bits."""
    a = splitmix64(seed, n * 6).reshape(n, 6)
    a[:, 5] &= np.uint64((1 << 56) - 1)
    return a


_R_LIMBS = [(R.R_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]


def random_fr_uniform(seed, n):
    """n x 6 limbs of CANONICAL scalars UNIFORM in [0, r) (bench.py draws its scalars the same way): 377-bit draws
    from a splitmix64 stream, redrawn while >= r (16 % are).  Unlike random_fr_canonical the top bit of r's range is exercised."""
    a = splitmix64(seed, n * 6).reshape(n, 6)
    a[:, 5] &= np.uint64((1 << 57) - 1)
    rnd = 0
    while True:
        ge = np.zeros(n, dtype=bool)          # lexicographic a >= r from the top limb down
        undecided = np.ones(n, dtype=bool)
        for k in range(5, -1, -1):
            rk = np.uint64(_R_LIMBS[k])
            ge |= undecided & (a[:, k] > rk)
            undecided &= a[:, k] == rk
        ge |= undecided
        bad = np.nonzero(ge)[0]
        if bad.size == 0:
            return a
        rnd += 1
        fresh = splitmix64(seed ^ (0xD1B54A32D192ED03 * rnd & 0xFFFFFFFFFFFFFFFF), bad.size * 6).reshape(bad.size, 6)
        fresh[:, 5] &= np.uint64((1 << 57) - 1)
        a[bad] = fresh


# ---------------------------------------------------------------- synthetic R1CS / CRS (tests + bench)
def make_r1cs(seed, n_constraints, n_primary, n_aux, bool_frac=0.0):
    """Satisfiable-by-construction R1CS in Python ints: rows are lists of (var, coeff).
    Constraint j: (sum a z)(sum b z) = c1 z_k + c0 with c0 solved for.  bool_frac of the auxiliary
    variables are 0/1 (boolean-heavy witness)."""
    import random
    rng = random.Random(seed)
    m = 1 + n_primary + n_aux
    z = [1] + [rng.randrange(R.R_MOD) for _ in range(m - 1)]
    for i in range(1 + n_primary, m):
        if rng.random() < bool_frac:
            z[i] = rng.randrange(2)
    A, B, C = [], [], []
    for _ in range(n_constraints):
        ra = [(rng.randrange(m), rng.randrange(1, 1 << 30)) for _ in range(rng.randrange(1, 4))]
        rb = [(rng.randrange(m), rng.randrange(1, R.R_MOD)) for _ in range(rng.randrange(1, 3))]
        va, vb = R.r1cs_eval_row(ra, z), R.r1cs_eval_row(rb, z)
        k = rng.randrange(1, m)
        c1 = rng.randrange(1, R.R_MOD)
        c0 = (va * vb - c1 * z[k]) % R.R_MOD
        A.append(ra); B.append(rb); C.append([(k, c1), (0, c0)])
    return A, B, C, z


def csr_from_rows(rows):
    rp, col, val = [0], [], []
    for row in rows:
        for i, c in row:
            col.append(i)
            val.append(h2i(c) if isinstance(c, str) else c)
        rp.append(len(col))
    return (np.array(rp, dtype=np.uint32), np.array(col, dtype=np.uint32), fr_array(val))


def crs_from_trapdoor(zk, A, B, C, n_vars, n_primary, tau, alpha, beta, delta, domain=None):
    """Proving key with known toxic waste: exponents from oracle/pyref (big ints), group elements by the
    product's fixed-base kernel (checked against the oracle in test_msm_gpu).  domain: None = the reference's forced power of two,
    R.STEP = libfqfft's unforced step domain, an int = that size (oracle/pyref.py qap_domain_size)."""
    st = R.groth16_setup_scalars(A, B, C, n_vars, n_primary, tau, alpha, beta, delta, domain)
    d = st["d"]
    dinv = pow(delta, -1, R.R_MOD)
    g1, g2 = aff_limbs(R.G1_GEN), aff_limbs(R.G2_GEN)
    can = lambda xs: np.array([R.int_to_limbs(x % R.R_MOD, 6) for x in xs], dtype=np.uint64).reshape(-1, 6)
    fb = lambda g, xs: zk.fixed_base_mul(g, can(xs), montgomery=False) if len(xs) else np.zeros((0, 24), dtype=np.uint64)
    hs, t = [], st["Zt"] * dinv % R.R_MOD
    for _ in range(d - 1):
        hs.append(t)
        t = t * tau % R.R_MOD
    ls = [(beta * st["At"][i] + alpha * st["Bt"][i] + st["Ct"][i]) * dinv % R.R_MOD for i in range(n_primary + 1, n_vars)]
    abc = [(beta * st["At"][i] + alpha * st["Bt"][i] + st["Ct"][i]) % R.R_MOD for i in range(n_primary + 1)]
    vk = dict(alpha=fb(g1, [alpha])[0], beta=fb(g2, [beta])[0], delta=fb(g2, [delta])[0], ABC=fb(g1, abc))
    pk = dict(vk=vk, alpha_g1=fb(g1, [alpha])[0], beta_g1=fb(g1, [beta])[0], beta_g2=fb(g2, [beta])[0],
              delta_g1=fb(g1, [delta])[0], delta_g2=fb(g2, [delta])[0],
              A=fb(g1, st["At"]), B2=fb(g2, st["Bt"]), B1=fb(g1, st["Bt"]), H=fb(g1, hs), L=fb(g1, ls))
    return pk, d
