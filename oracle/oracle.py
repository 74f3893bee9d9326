"""ctypes binding of the C test oracle (oracle/bw6_oracle.c).  TEST INFRASTRUCTURE ONLY:
import from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, nowhere else."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbw6oracle.so")
_lib = None
_u64p = ctypes.POINTER(ctypes.c_uint64)
_u32p = ctypes.POINTER(ctypes.c_uint32)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libbw6oracle.so"])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.oracle_init()
    return _lib


def _p(a):
    return a.ctypes.data_as(_u64p)


def _p32(a):
    return a.ctypes.data_as(_u32p)


def _arr(x, n):
    return np.ascontiguousarray(x, dtype=np.uint64).reshape(n)


def f_op(name, which, a, b=None):
    n = 6 if which else 12
    out = np.zeros(n, dtype=np.uint64)
    fn = getattr(load(), "oracle_f_" + name)
    if b is None:
        fn(which, _p(_arr(a, n)), _p(out))
    else:
        fn(which, _p(_arr(a, n)), _p(_arr(b, n)), _p(out))
    return out


def jac_add(a, b):
    out = np.zeros(36, dtype=np.uint64)
    load().oracle_jac_add(_p(_arr(a, 36)), _p(_arr(b, 36)), _p(out))
    return out


def jac_dbl(a):
    out = np.zeros(36, dtype=np.uint64)
    load().oracle_jac_dbl(_p(_arr(a, 36)), _p(out))
    return out


def jac_to_affine(a):
    out = np.zeros(24, dtype=np.uint64)
    load().oracle_jac_to_affine(_p(_arr(a, 36)), _p(out))
    return out


def aff_to_jac(a):
    out = np.zeros(36, dtype=np.uint64)
    load().oracle_aff_to_jac(_p(_arr(a, 24)), _p(out))
    return out


def scalar_mul(aff, k_mont):
    out = np.zeros(36, dtype=np.uint64)
    load().oracle_scalar_mul(_p(_arr(aff, 24)), _p(_arr(k_mont, 6)), _p(out))
    return out


def on_curve(aff, g2=False):
    return bool(load().oracle_on_curve(_p(_arr(aff, 24)), int(g2)))


def point_progression(p0_aff, d_aff, n):
    out = np.zeros((n, 24), dtype=np.uint64)
    load().oracle_point_progression(_p(_arr(p0_aff, 24)), _p(_arr(d_aff, 24)), ctypes.c_size_t(n), _p(out))
    return out


def set_threads(n):
    load().oracle_set_threads(int(n))


def max_threads():
    return int(load().oracle_max_threads())


def msm(bases_aff, scalars_mont, chunks=None, with_mixed=True):
    b = np.ascontiguousarray(bases_aff, dtype=np.uint64).reshape(-1, 24)
    s = np.ascontiguousarray(scalars_mont, dtype=np.uint64).reshape(-1, 6)
    assert b.shape[0] == s.shape[0]
    out = np.zeros(36, dtype=np.uint64)
    if chunks is None:
        chunks = max_threads()
    load().oracle_msm(_p(b), _p(s), ctypes.c_size_t(b.shape[0]), int(chunks), int(with_mixed), _p(out))
    return out


def ntt(a, log_d, inverse=False, coset=False):
    x = np.array(a, dtype=np.uint64).reshape(-1, 6).copy()
    assert x.shape[0] == 1 << log_d
    load().oracle_ntt(_p(x), int(log_d), int(inverse), int(coset))
    return x


def qap_log_d(n, n_primary):
    return int(load().oracle_qap_log_d(ctypes.c_size_t(n), ctypes.c_size_t(n_primary)))


STEP = -1          # `domain` argument: libfqfft's unforced get_evaluation_domain (None / 0 = the reference's forced power of two)


def _dom_arg(domain):
    return ctypes.c_size_t(0 if not domain else (2 ** (8 * ctypes.sizeof(ctypes.c_size_t)) - 1 if domain == STEP else int(domain)))


def domain_size(min_size):
    """The reference's evaluation domain for min_size points: libzeth's groth16_snark forces a power of two."""
    fn = load().oracle_forced_domain_size
    fn.restype = ctypes.c_size_t
    return int(fn(ctypes.c_size_t(min_size)))


def step_domain_size(min_size):
    """Size of the domain libfqfft picks for min_size points when NOT forced (a power of two, or 2^k + 2^r: step_radix2_domain)."""
    fn = load().oracle_step_domain_size
    fn.restype = ctypes.c_size_t
    return int(fn(ctypes.c_size_t(min_size)))


def qap_domain_size(n, n_primary, domain=None):
    """domain: None = forced power of two (the reference), STEP = libfqfft unforced, else an explicit valid size."""
    fn = load().oracle_qap_domain_size_ex
    fn.restype = ctypes.c_size_t
    d = int(fn(ctypes.c_size_t(n), ctypes.c_size_t(n_primary), _dom_arg(domain)))
    assert d, "not a usable evaluation domain for this system: %r" % (domain,)
    return d


def domain_fft(a, inverse=False, coset=False):
    """FFT / iFFT / cosetFFT / icosetFFT over the domain of len(a) points (a power of two or 2^k + 2^r)."""
    x = np.array(a, dtype=np.uint64).reshape(-1, 6).copy()
    assert step_domain_size(x.shape[0]) == x.shape[0]
    load().oracle_domain_fft(_p(x), ctypes.c_size_t(x.shape[0]), int(inverse), int(coset))
    return x


def qap_h(A, B, C, z, n, n_primary, domain=None):
    """A, B, C: (row_ptr u32[n+1], col u32[nnz], val u64[nnz,6]) CSR triples; z: [m,6].  domain: as qap_domain_size - the proving
    key's domain_size when a key is at hand."""
    d = qap_domain_size(n, n_primary, domain)
    h = np.zeros((d, 6), dtype=np.uint64)
    args = []
    for (rp, col, val) in (A, B, C):
        args += [_p32(np.ascontiguousarray(rp, dtype=np.uint32)), _p32(np.ascontiguousarray(col, dtype=np.uint32)),
                 _p(np.ascontiguousarray(val, dtype=np.uint64))]
    zz = np.ascontiguousarray(z, dtype=np.uint64)
    fn = load().oracle_qap_h_ex
    fn.restype = ctypes.c_size_t
    got = int(fn(*args, _p(zz), ctypes.c_size_t(n), ctypes.c_size_t(n_primary), ctypes.c_size_t(d), _p(h)))
    assert got == d
    return h


def r1cs_first_unsatisfied(A, B, C, z):
    """Index of the first violated constraint of <A,z><B,z> = <C,z>, or -1."""
    n = len(A[0]) - 1
    args = []
    for (rp, col, val) in (A, B, C):
        args += [_p32(np.ascontiguousarray(rp, dtype=np.uint32)), _p32(np.ascontiguousarray(col, dtype=np.uint32)),
                 _p(np.ascontiguousarray(val, dtype=np.uint64))]
    zz = np.ascontiguousarray(z, dtype=np.uint64)
    fn = load().oracle_r1cs_first_unsatisfied
    fn.restype = ctypes.c_long
    return int(fn(*args, _p(zz), ctypes.c_size_t(n)))


def groth16_prove(pk, z, n_primary, h, r_m, s_m, chunks=None):
    zz = np.ascontiguousarray(z, dtype=np.uint64).reshape(-1, 6)
    hh = np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, 6)
    out = np.zeros(72, dtype=np.uint64)
    if chunks is None:
        chunks = max_threads()
    c = lambda a: _p(np.ascontiguousarray(a, dtype=np.uint64))
    load().oracle_groth16_prove(c(pk["alpha_g1"]), c(pk["beta_g1"]), c(pk["beta_g2"]), c(pk["delta_g1"]), c(pk["delta_g2"]),
                                c(pk["A"]), c(pk["B2"]), c(pk["B1"]), c(pk["H"]), c(pk["L"]),
                                _p(zz), ctypes.c_size_t(zz.shape[0]), ctypes.c_size_t(n_primary), _p(hh),
                                ctypes.c_size_t(hh.shape[0]), _p(_arr(r_m, 6)), _p(_arr(s_m, 6)), int(chunks), _p(out))
    return out
