"""Parity of the HIP NTT (C ABI zkhip_ntt) with the golden vectors and the CPU oracle: bit-exact,
all four modes (FFT, iFFT, cosetFFT, icosetFFT), sizes 2^0 .. 2^20, round trips at full size."""
import numpy as np
import pytest

from tests.helpers import fr_array, fr_ints, golden, h2i, random_fr_canonical, random_fr_uniform

pytestmark = pytest.mark.gpu
MODES = [(False, False, "fft"), (True, False, "ifft"), (False, True, "coset_fft"), (True, True, "icoset_fft")]


def test_golden_vectors(zk):
    for v in golden("ntt_vectors.json"):
        a = fr_array([h2i(x) for x in v["input"]])
        for inv, coset, key in MODES:
            assert fr_ints(zk.ntt(a, v["log_d"], inverse=inv, coset=coset)) == [h2i(x) for x in v[key]], (v["log_d"], key)


@pytest.mark.parametrize("log_d", [0, 1, 2, 3, 6, 9, 10, 11, 12, 13, 15, 16, 17, 18, 19])
def test_vs_oracle(zk, oracle_lib, log_d):
    O = oracle_lib
    a = random_fr_uniform(300 + log_d, 1 << log_d)    # uniform in [0, r): every Montgomery residue the ABI allows
    for inv, coset, _ in MODES:
        assert (zk.ntt(a, log_d, inverse=inv, coset=coset) == O.ntt(a, log_d, inverse=inv, coset=coset)).all(), (log_d, inv, coset)


def test_full_size_round_trip_and_oracle(zk, oracle_lib):
    """2^20 (BASELINE size): iFFT(FFT(a)) = a, icosetFFT(cosetFFT(a)) = a, and FFT vs the oracle."""
    log_d = 20
    a = random_fr_uniform(77, 1 << log_d)
    f = zk.ntt(a, log_d)
    assert (zk.ntt(f, log_d, inverse=True) == a).all()
    c = zk.ntt(a, log_d, coset=True)
    assert (zk.ntt(c, log_d, inverse=True, coset=True) == a).all()
    assert (f == oracle_lib.ntt(a, log_d)).all()


def test_log_d_21(zk, oracle_lib):
    """an odd size above 2^18 (2^11 x 2^10: unequal factors, two row transforms per workgroup): round trip and the oracle"""
    log_d = 21
    a = random_fr_canonical(80, 1 << log_d)
    f = zk.ntt(a, log_d, inverse=True, coset=True)
    assert (f == oracle_lib.ntt(a, log_d, inverse=True, coset=True)).all()
    assert (zk.ntt(f, log_d, coset=True) == a).all()


def test_log_d_22(zk):
    log_d = 22
    a = random_fr_canonical(79, 1 << log_d)
    f = zk.ntt(a, log_d, coset=True)
    assert (zk.ntt(f, log_d, inverse=True, coset=True) == a).all()
