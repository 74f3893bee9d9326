"""ctypes binding of the C ABI in include/zkhip.h (libzkhip.so, built in-tree by
__graft_entry__.build()).  Plain pointers and sizes only; numpy arrays carry the limbs.
There is no CPU implementation behind these calls: without the HIP library or a gfx950
device they raise."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzkhip.so")

EXPORTS = [
    "zkhip_init", "zkhip_shutdown", "zkhip_strerror", "zkhip_last_error", "zkhip_set_msm_window",
    "zkhip_bases_upload", "zkhip_bases_upload_dev", "zkhip_bases_len", "zkhip_bases_free",
    "zkhip_msm", "zkhip_msm_dev", "zkhip_msm_raw", "zkhip_last_accumulate_ms",
    "zkhip_fixed_base_mul", "zkhip_fixed_base_mul_dev", "zkhip_ntt", "zkhip_ntt_dev",
    "zkhip_jac_to_affine", "zkhip_jac_add",
]


class ZkhipError(RuntimeError):
    pass


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ZkhipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(the HIP library is the only compute path; there is no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    c_u64p = ctypes.POINTER(ctypes.c_uint64)
    lib.zkhip_init.argtypes = [ctypes.c_int]
    lib.zkhip_strerror.restype = ctypes.c_char_p
    lib.zkhip_strerror.argtypes = [ctypes.c_int]
    lib.zkhip_last_error.restype = ctypes.c_char_p
    lib.zkhip_set_msm_window.argtypes = [ctypes.c_int]
    lib.zkhip_bases_upload.argtypes = [c_u64p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    lib.zkhip_bases_upload_dev.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    lib.zkhip_bases_len.restype = ctypes.c_size_t
    lib.zkhip_bases_len.argtypes = [ctypes.c_void_p]
    lib.zkhip_bases_free.argtypes = [ctypes.c_void_p]
    lib.zkhip_msm.argtypes = [ctypes.c_void_p, ctypes.c_size_t, c_u64p, ctypes.c_size_t, ctypes.c_int, c_u64p]
    lib.zkhip_msm_dev.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, c_u64p]
    lib.zkhip_msm_raw.argtypes = [c_u64p, c_u64p, ctypes.c_size_t, ctypes.c_int, c_u64p]
    lib.zkhip_fixed_base_mul.argtypes = [c_u64p, c_u64p, ctypes.c_size_t, ctypes.c_int, c_u64p]
    lib.zkhip_fixed_base_mul_dev.argtypes = [c_u64p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    lib.zkhip_ntt.argtypes = [c_u64p, ctypes.c_uint, ctypes.c_int, ctypes.c_int]
    lib.zkhip_ntt_dev.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int]
    lib.zkhip_last_accumulate_ms.restype = ctypes.c_float
    lib.zkhip_jac_to_affine.argtypes = [c_u64p, c_u64p]
    lib.zkhip_jac_add.argtypes = [c_u64p, c_u64p, c_u64p]
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        lib = load()
        raise ZkhipError(f"zkhip error {rc} ({lib.zkhip_strerror(rc).decode()}): {lib.zkhip_last_error().decode()}")


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def init(device=0):
    _check(load().zkhip_init(device))


def set_msm_window(c):
    _check(load().zkhip_set_msm_window(c))


class Bases:
    """A base-point set resident in HBM (the proving key's query vectors)."""

    def __init__(self, handle):
        self.handle = handle

    @classmethod
    def upload(cls, bases_affine):
        a = np.ascontiguousarray(bases_affine, dtype=np.uint64).reshape(-1, 24)
        h = ctypes.c_void_p()
        _check(load().zkhip_bases_upload(_p(a), a.shape[0], ctypes.byref(h)))
        return cls(h)

    @classmethod
    def upload_dev(cls, dev_ptr, n):
        h = ctypes.c_void_p()
        _check(load().zkhip_bases_upload_dev(ctypes.c_void_p(dev_ptr), n, ctypes.byref(h)))
        return cls(h)

    def __len__(self):
        return load().zkhip_bases_len(self.handle)

    def free(self):
        if self.handle:
            load().zkhip_bases_free(self.handle)
            self.handle = None

    def msm(self, scalars, offset=0, montgomery=True):
        s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 6)
        out = np.zeros(36, dtype=np.uint64)
        _check(load().zkhip_msm(self.handle, offset, _p(s), s.shape[0], int(montgomery), _p(out)))
        return out

    def msm_dev(self, dev_ptr, n, offset=0, montgomery=True):
        out = np.zeros(36, dtype=np.uint64)
        _check(load().zkhip_msm_dev(self.handle, offset, ctypes.c_void_p(dev_ptr), n, int(montgomery), _p(out)))
        return out


def msm_raw(bases_affine, scalars, montgomery=True):
    a = np.ascontiguousarray(bases_affine, dtype=np.uint64).reshape(-1, 24)
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 6)
    assert a.shape[0] == s.shape[0]
    out = np.zeros(36, dtype=np.uint64)
    _check(load().zkhip_msm_raw(_p(a), _p(s), a.shape[0], int(montgomery), _p(out)))
    return out


def fixed_base_mul(base_affine, scalars, montgomery=True):
    """out[i] = scalars[i] * base (affine, n x 24 limbs) - the batch exponentiation of Groth16 setup."""
    b = np.ascontiguousarray(base_affine, dtype=np.uint64).reshape(24)
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 6)
    out = np.zeros((s.shape[0], 24), dtype=np.uint64)
    _check(load().zkhip_fixed_base_mul(_p(b), _p(s), s.shape[0], int(montgomery), _p(out)))
    return out


def fixed_base_mul_dev(base_affine, d_scalars_ptr, n, d_out_ptr, montgomery=True):
    b = np.ascontiguousarray(base_affine, dtype=np.uint64).reshape(24)
    _check(load().zkhip_fixed_base_mul_dev(_p(b), ctypes.c_void_p(d_scalars_ptr), n, int(montgomery), ctypes.c_void_p(d_out_ptr)))


def ntt(data, log_d, inverse=False, coset=False):
    """FFT / iFFT / cosetFFT / icosetFFT of 2^log_d Fr elements (n x 6 limbs); returns a new array."""
    a = np.array(data, dtype=np.uint64).reshape(-1, 6).copy()
    assert a.shape[0] == 1 << log_d
    _check(load().zkhip_ntt(_p(a), log_d, int(inverse), int(coset)))
    return a


def ntt_dev(dev_ptr, log_d, inverse=False, coset=False):
    _check(load().zkhip_ntt_dev(ctypes.c_void_p(dev_ptr), log_d, int(inverse), int(coset)))


def jac_to_affine(jac):
    j = np.ascontiguousarray(jac, dtype=np.uint64)
    out = np.zeros(24, dtype=np.uint64)
    _check(load().zkhip_jac_to_affine(_p(j), _p(out)))
    return out


def jac_add(a, b):
    out = np.zeros(36, dtype=np.uint64)
    _check(load().zkhip_jac_add(_p(np.ascontiguousarray(a, dtype=np.uint64)), _p(np.ascontiguousarray(b, dtype=np.uint64)), _p(out)))
    return out


def last_accumulate_ms():
    return float(load().zkhip_last_accumulate_ms())
