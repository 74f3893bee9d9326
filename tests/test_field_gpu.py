"""The DEVICE build's multiplier bodies (fp29.cuh -> fp29_chain.cuh: every column one dependent chain of v_mad_u64_u32 in asm
blocks) on raw limbs, through the C ABI's test hook, against big integers: the counterpart of tests/test_fp29_host.py (which checks
the C++ bodies g++ compiles).  The cases are the ones that decide whether a 64-bit column accumulator holds: every limb at 2^29 - 1,
the top limb at the lazy bound 2^10 p, mixtures of full and random limbs, zeros and ones."""
import random

import numpy as np
import pytest

from oracle import pyref as R

pytestmark = pytest.mark.gpu


def _val(limbs):
    return sum(int(v) << (29 * i) for i, v in enumerate(limbs))


def _cases(p, nl, rng, n_random):
    top_max = (p << 10) >> (29 * (nl - 1))              # top limb of values below 2^10 p
    full = [(1 << 29) - 1] * (nl - 1)
    Rdev = 1 << (29 * nl)
    cases = [[full + [top_max - 1]] * 4, [[0] * nl] * 4, [[1] + [0] * (nl - 1)] * 4,
             [full + [top_max - 1], [0] * nl, [1] + [0] * (nl - 1), full + [0]]]
    for _ in range(n_random):
        cases.append([[rng.choice([(1 << 29) - 1, 0, rng.randrange(1 << 29), rng.randrange(1 << 29)]) for _ in range(nl - 1)] + [rng.randrange(top_max)]
                      for _ in range(4)])
    out = []
    for a, b, c, d in cases:
        a, b, c, d = list(a), list(b), list(c), list(d)
        # the bodies' contract: a b + c d (and a a) below 2^10 R p
        while _val(a) * _val(b) + _val(c) * _val(d) >= (Rdev * p) << 10 or _val(a) * _val(a) >= (Rdev * p) << 10:
            a[-1] >>= 1; c[-1] >>= 1
        out.append((a, b, c, d))
    return out


@pytest.mark.parametrize("field,p,nl", [(0, R.Q_MOD, 27), (1, R.R_MOD, 14)])
def test_device_multiplier_bodies_on_raw_limbs(zk, field, p, nl):
    rng = random.Random(29 + field)
    cases = _cases(p, nl, rng, 600)
    got = zk.field_selftest(field, np.array(cases, dtype=np.uint32))
    Rdev = 1 << (29 * nl)
    rinv = pow(Rdev, -1, p)
    for (a, b, c, d), (m, q, m2) in zip(cases, got):
        va, vb, vc, vd = _val(a), _val(b), _val(c), _val(d)
        for limbs, want in ((m, va * vb), (q, va * va), (m2, va * vb + vc * vd)):
            v = _val(limbs)
            assert all(int(x) < (1 << 29) for x in limbs[:-1])              # normalised limbs
            assert v % p == want * rinv % p                                  # the Montgomery product
            assert v < want // Rdev + p + 1                                  # the bound every caller relies on: < x / R + p
