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
