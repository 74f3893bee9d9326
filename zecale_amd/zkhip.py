"""ctypes binding of the C ABI in include/zkhip.h (libzkhip.so, built in-tree by
__graft_entry__.build()).  Plain pointers and sizes only; numpy arrays carry the limbs.
There is no CPU implementation behind these calls: without the HIP library or a gfx950
device they raise."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZKHIP_LIB: another build of the same library (A/B builds, the mutation run of tools/mutation_abc5.sh); default: the in-tree one
LIB_PATH = os.environ.get("ZKHIP_LIB") or os.path.join(_HERE, "libzkhip.so")

EXPORTS = [
    "zkhip_init", "zkhip_set_device", "zkhip_get_device", "zkhip_device_count", "zkhip_fr_random", "zkhip_shutdown", "zkhip_strerror", "zkhip_last_error", "zkhip_set_msm_window", "zkhip_set_affine_levels",
    "zkhip_bases_upload", "zkhip_bases_upload_dev", "zkhip_bases_len", "zkhip_bases_free",
    "zkhip_bases_precompute", "zkhip_bases_table_window", "zkhip_set_crs_precompute", "zkhip_crs_table_window", "zkhip_set_batch_msms",
    "zkhip_msm", "zkhip_msm_dev", "zkhip_msm_raw", "zkhip_msm_submit", "zkhip_msm_collect",
    "zkhip_device_alloc", "zkhip_device_free", "zkhip_device_copy_in", "zkhip_last_accumulate_ms", "zkhip_last_accumulate_entries", "zkhip_prover_last_accumulate_entries",
    "zkhip_prover_set_streaming", "zkhip_set_table_naf", "zkhip_fixed_base_mul", "zkhip_fixed_base_mul_dev", "zkhip_ntt", "zkhip_ntt_dev",
    "zkhip_r1cs_upload", "zkhip_r1cs_upload_ex", "zkhip_r1cs_set_domain", "zkhip_r1cs_free", "zkhip_r1cs_log_domain", "zkhip_r1cs_domain_size",
    "zkhip_domain_size", "zkhip_step_domain_size", "zkhip_domain_is_valid", "zkhip_groth16_setup_ex", "zkhip_dispatcher_outstanding", "zkhip_r1cs_is_satisfied", "zkhip_qap_h",
    "zkhip_crs_upload", "zkhip_crs_free", "zkhip_groth16_prove", "zkhip_last_prove_timings", "zkhip_groth16_verify",
    "zkhip_crs_upload_slice", "zkhip_groth16_prove_partial", "zkhip_groth16_finish",
    "zkhip_bls12_377_groth16_verify", "zkhip_aggregator_new", "zkhip_aggregator_free", "zkhip_aggregator_num_constraints",
    "zkhip_aggregator_num_variables", "zkhip_aggregator_num_primary_inputs", "zkhip_aggregator_get_r1cs",
    "zkhip_aggregator_witness", "zkhip_aggregator_witness_gpu", "zkhip_gpu_witness_new", "zkhip_gpu_witness_new_batched", "zkhip_gpu_witness_run", "zkhip_gpu_witness_run_batched", "zkhip_gpu_witness_free",
    "zkhip_gpu_witness_stats", "zkhip_prover_prove_dev", "zkhip_aggregator_check_inputs", "zkhip_aggregator_vk_hash", "zkhip_aggregator_num_proofs", "zkhip_aggregator_inputs_per_proof",
    "zkhip_prover_new", "zkhip_prover_create_streams", "zkhip_prover_prove", "zkhip_prover_timings", "zkhip_prover_free", "zkhip_prover_last_accumulate_ms",
    "zkhip_aggregator_pipeline_new", "zkhip_aggregator_pipeline_new_ex", "zkhip_crs_device", "zkhip_aggregator_pipeline_submit", "zkhip_aggregator_pipeline_wait", "zkhip_aggregator_pipeline_free",
    "zkhip_groth16_setup", "zkhip_groth16_setup_slice", "zkhip_keypair_crs_desc", "zkhip_keypair_vk", "zkhip_keypair_free", "zkhip_keypair_write", "zkhip_keypair_read",
    "zkhip_jac_to_affine", "zkhip_jac_add", "zkhip_to_canonical",
    "zkhip_last_accumulate_interval", "zkhip_crs_upload_ex", "zkhip_crs_upload_slice_ex", "zkhip_bases_precompute_ex", "zkhip_crs_table_kind", "zkhip_crs_finite_terms",
    "zkhip_bases_set_window", "zkhip_reset_time_base", "zkhip_measure_fq_mul_rate", "zkhip_internal_field_selftest", "zkhip_internal_tail_selftest", "zkhip_last_prove_split", "zkhip_set_prove_split", "zkhip_host_alloc", "zkhip_host_free",
    "zkhip_msm_stream_new", "zkhip_msm_stream_submit", "zkhip_msm_stream_submit_host", "zkhip_msm_stream_collect", "zkhip_msm_stream_last_accumulate_ms",
    "zkhip_msm_stream_last_accumulate_interval", "zkhip_msm_stream_free", "zkhip_prover_new_slice", "zkhip_prover_prove_partial",
    "zkhip_dispatcher_new", "zkhip_dispatcher_size", "zkhip_dispatcher_submit", "zkhip_dispatcher_wait", "zkhip_dispatcher_stats", "zkhip_dispatcher_free",
    "zkhip_multi_prover_new", "zkhip_multi_prover_size", "zkhip_multi_prover_prove", "zkhip_multi_prover_timings", "zkhip_multi_prover_free",
    "zkhip_aggregator_app_new", "zkhip_aggregator_app_free", "zkhip_aggregator_app_num_constants", "zkhip_aggregator_app_constants", "zkhip_aggregator_app_mask",
    "zkhip_aggregator_witness_app", "zkhip_groth16_prove_app", "zkhip_prover_prove_app", "zkhip_prover_prove_app_dev", "zkhip_gpu_witness_run_batched_app",
    "zkhip_aggregator_pipeline_register_app", "zkhip_aggregator_pipeline_app_hits", "zkhip_dispatcher_register_app", "zkhip_device_copy_out", "zkhip_measure_ntt", "zkhip_key_partition", "zkhip_prover_timings_chained",
]


class ZkhipError(RuntimeError):
    """A failed C ABI call.  `code` is the library's return code (include/zkhip.h: ZKHIP_ERR_ARG -1, _NO_DEVICE -2, _HIP -3, _STATE -4)."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


c_u64p_t = ctypes.POINTER(ctypes.c_uint64)


class R1csDesc(ctypes.Structure):
    _fields_ = [("n_constraints", ctypes.c_size_t), ("n_vars", ctypes.c_size_t), ("n_primary", ctypes.c_size_t)] + [
        (f"{m}_{k}", ctypes.c_void_p) for m in "abc" for k in ("row_ptr", "col", "val")]


class KeyOpts(ctypes.Structure):
    """zkhip_key_opts: the options of a proving key / base set, carried by the handle (include/zkhip.h)."""
    _fields_ = [("precompute", ctypes.c_int), ("table_naf", ctypes.c_int), ("window", ctypes.c_int), ("batch_msms", ctypes.c_int)]


def key_opts(precompute=None, table_naf=None, window=0, batch_msms=None):
    """None = the process-wide default (the deprecated zkhip_set_* switches), True / False = this key's own choice."""
    t = lambda v: -1 if v is None else int(bool(v))
    return KeyOpts(t(precompute), t(table_naf), int(window or 0), t(batch_msms))


class CrsDesc(ctypes.Structure):
    _fields_ = [("n_vars", ctypes.c_size_t), ("n_primary", ctypes.c_size_t), ("domain_size", ctypes.c_size_t)] + [
        (k, ctypes.c_void_p) for k in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2",
                                       "a_query", "b_g2_query", "b_g1_query", "h_query", "l_query")]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ZkhipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(the HIP library is the only compute path; there is no CPU fallback)")
    # several prover instances run on their own streams: more hardware queues than the runtime's default 4
    # (only effective if the HIP runtime has not been initialised yet in this process)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    lib = ctypes.CDLL(LIB_PATH)
    c_u64p = ctypes.POINTER(ctypes.c_uint64)
    lib.zkhip_init.argtypes = [ctypes.c_int]
    lib.zkhip_set_device.argtypes = [ctypes.c_int]
    lib.zkhip_strerror.restype = ctypes.c_char_p
    lib.zkhip_strerror.argtypes = [ctypes.c_int]
    lib.zkhip_last_error.restype = ctypes.c_char_p
    lib.zkhip_set_msm_window.argtypes = [ctypes.c_int]
    lib.zkhip_bases_upload.argtypes = [c_u64p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    lib.zkhip_bases_upload_dev.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    lib.zkhip_bases_len.restype = ctypes.c_size_t
    lib.zkhip_bases_len.argtypes = [ctypes.c_void_p]
    lib.zkhip_bases_free.argtypes = [ctypes.c_void_p]
    lib.zkhip_bases_precompute.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.zkhip_bases_table_window.argtypes = [ctypes.c_void_p]
    lib.zkhip_set_crs_precompute.argtypes = [ctypes.c_int]
    lib.zkhip_crs_table_window.argtypes = [ctypes.c_void_p]
    lib.zkhip_msm.argtypes = [ctypes.c_void_p, ctypes.c_size_t, c_u64p, ctypes.c_size_t, ctypes.c_int, c_u64p]
    lib.zkhip_msm_dev.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, c_u64p]
    lib.zkhip_msm_submit.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
    lib.zkhip_msm_collect.argtypes = [ctypes.c_int, c_u64p]
    lib.zkhip_device_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    lib.zkhip_device_free.argtypes = [ctypes.c_void_p]
    lib.zkhip_device_copy_in.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.zkhip_msm_raw.argtypes = [c_u64p, c_u64p, ctypes.c_size_t, ctypes.c_int, c_u64p]
    lib.zkhip_fixed_base_mul.argtypes = [c_u64p, c_u64p, ctypes.c_size_t, ctypes.c_int, c_u64p]
    lib.zkhip_fixed_base_mul_dev.argtypes = [c_u64p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    lib.zkhip_ntt.argtypes = [c_u64p, ctypes.c_uint, ctypes.c_int, ctypes.c_int]
    lib.zkhip_ntt_dev.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int]
    lib.zkhip_r1cs_upload.argtypes = [ctypes.POINTER(R1csDesc), ctypes.POINTER(ctypes.c_void_p)]
    lib.zkhip_r1cs_upload_ex.argtypes = [ctypes.POINTER(R1csDesc), ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    lib.zkhip_r1cs_set_domain.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.zkhip_step_domain_size.argtypes = [ctypes.c_size_t]
    lib.zkhip_step_domain_size.restype = ctypes.c_size_t
    lib.zkhip_domain_is_valid.argtypes = [ctypes.c_size_t]
    lib.zkhip_r1cs_free.argtypes = [ctypes.c_void_p]
    lib.zkhip_r1cs_log_domain.argtypes = [ctypes.c_void_p]
    lib.zkhip_r1cs_log_domain.restype = ctypes.c_uint
    lib.zkhip_r1cs_domain_size.argtypes = [ctypes.c_void_p]
    lib.zkhip_r1cs_domain_size.restype = ctypes.c_size_t
    lib.zkhip_domain_size.argtypes = [ctypes.c_size_t]
    lib.zkhip_domain_size.restype = ctypes.c_size_t
    lib.zkhip_r1cs_is_satisfied.argtypes = [ctypes.c_void_p, c_u64p, ctypes.POINTER(ctypes.c_int)]
    lib.zkhip_qap_h.argtypes = [ctypes.c_void_p, c_u64p, c_u64p]
    lib.zkhip_crs_upload.argtypes = [ctypes.POINTER(CrsDesc), ctypes.POINTER(ctypes.c_void_p)]
    lib.zkhip_crs_free.argtypes = [ctypes.c_void_p]
    lib.zkhip_crs_upload_ex.argtypes = [ctypes.POINTER(CrsDesc), ctypes.POINTER(KeyOpts), ctypes.POINTER(ctypes.c_void_p)]
    lib.zkhip_crs_upload_slice_ex.argtypes = [ctypes.POINTER(CrsDesc)] + [ctypes.c_size_t] * 6 + [ctypes.POINTER(KeyOpts), ctypes.POINTER(ctypes.c_void_p)]
    lib.zkhip_crs_table_kind.argtypes = [ctypes.c_void_p]
    lib.zkhip_crs_finite_terms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    lib.zkhip_bases_precompute_ex.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.zkhip_groth16_prove.argtypes = [ctypes.c_void_p, ctypes.c_void_p, c_u64p, c_u64p, c_u64p, c_u64p]
    lib.zkhip_last_prove_timings.argtypes = [ctypes.POINTER(ctypes.c_double)]
    lib.zkhip_crs_upload_slice.argtypes = [ctypes.POINTER(CrsDesc)] + [ctypes.c_size_t] * 6 + [ctypes.POINTER(ctypes.c_void_p)]
    lib.zkhip_groth16_prove_partial.argtypes = [ctypes.c_void_p, ctypes.c_void_p, c_u64p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, c_u64p]
    lib.zkhip_groth16_finish.argtypes = [c_u64p] * 9
    lib.zkhip_groth16_verify.argtypes = [c_u64p, c_u64p, c_u64p, c_u64p, c_u64p, ctypes.c_size_t, c_u64p, ctypes.POINTER(ctypes.c_int)]
    lib.zkhip_last_accumulate_ms.restype = ctypes.c_float
    lib.zkhip_last_accumulate_entries.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    lib.zkhip_prover_last_accumulate_entries.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
    lib.zkhip_jac_to_affine.argtypes = [c_u64p, c_u64p]
    lib.zkhip_jac_add.argtypes = [c_u64p, c_u64p, c_u64p]
    lib.zkhip_keypair_write.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    lib.zkhip_keypair_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
    lib.zkhip_prover_new.argtypes = [ctypes.c_void_p, ctypes.POINTER(R1csDesc), ctypes.POINTER(ctypes.c_void_p)]
    lib.zkhip_prover_create_streams.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.zkhip_prover_set_streaming.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.zkhip_prover_prove.argtypes = [ctypes.c_void_p, c_u64p, c_u64p, c_u64p, c_u64p]
    lib.zkhip_prover_timings.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    lib.zkhip_prover_free.argtypes = [ctypes.c_void_p]
    lib.zkhip_prover_last_accumulate_ms.argtypes = [ctypes.c_void_p]
    lib.zkhip_prover_last_accumulate_ms.restype = ctypes.c_float
    vpp = ctypes.POINTER(ctypes.c_void_p)
    lib.zkhip_host_alloc.argtypes = [ctypes.c_size_t, vpp]
    lib.zkhip_host_free.argtypes = [ctypes.c_void_p]
    lib.zkhip_msm_stream_new.argtypes = [ctypes.c_void_p, ctypes.c_int, vpp]
    lib.zkhip_msm_stream_submit.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, c_u64p]
    lib.zkhip_msm_stream_submit_host.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, c_u64p]
    lib.zkhip_msm_stream_collect.argtypes = [ctypes.c_void_p, ctypes.c_uint64, c_u64p]
    lib.zkhip_msm_stream_last_accumulate_ms.argtypes = [ctypes.c_void_p]
    lib.zkhip_msm_stream_last_accumulate_ms.restype = ctypes.c_float
    lib.zkhip_msm_stream_last_accumulate_interval.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.zkhip_msm_stream_free.argtypes = [ctypes.c_void_p]
    lib.zkhip_prover_new_slice.argtypes = [ctypes.c_void_p, ctypes.POINTER(R1csDesc), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, vpp]
    lib.zkhip_prover_prove_partial.argtypes = [ctypes.c_void_p, c_u64p, c_u64p]
    lib.zkhip_dispatcher_new.argtypes = [ctypes.c_void_p, ctypes.POINTER(CrsDesc), ctypes.POINTER(KeyOpts), ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_uint, vpp]
    lib.zkhip_dispatcher_size.argtypes = [ctypes.c_void_p]
    lib.zkhip_dispatcher_submit.argtypes = [ctypes.c_void_p, c_u64p, c_u64p, c_u64p, c_u64p, c_u64p, c_u64p]
    lib.zkhip_dispatcher_wait.argtypes = [ctypes.c_void_p, ctypes.c_uint64, c_u64p, c_u64p]
    lib.zkhip_dispatcher_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    lib.zkhip_dispatcher_free.argtypes = [ctypes.c_void_p]
    lib.zkhip_multi_prover_new.argtypes = [ctypes.POINTER(CrsDesc), ctypes.POINTER(R1csDesc), ctypes.POINTER(KeyOpts), ctypes.POINTER(ctypes.c_int), ctypes.c_int, vpp]
    lib.zkhip_multi_prover_size.argtypes = [ctypes.c_void_p]
    lib.zkhip_multi_prover_prove.argtypes = [ctypes.c_void_p, c_u64p, c_u64p, c_u64p, c_u64p]
    lib.zkhip_multi_prover_timings.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    lib.zkhip_multi_prover_free.argtypes = [ctypes.c_void_p]
    lib.zkhip_aggregator_pipeline_new.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    lib.zkhip_aggregator_pipeline_submit.argtypes = [ctypes.c_void_p, c_u64p, c_u64p, c_u64p, c_u64p, c_u64p, ctypes.POINTER(ctypes.c_uint64)]
    lib.zkhip_aggregator_pipeline_wait.argtypes = [ctypes.c_void_p, ctypes.c_uint64, c_u64p, c_u64p]
    lib.zkhip_aggregator_pipeline_free.argtypes = [ctypes.c_void_p]
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        lib = load()
        raise ZkhipError(f"zkhip error {rc} ({lib.zkhip_strerror(rc).decode()}): {lib.zkhip_last_error().decode()}", rc)


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def init(device=0):
    _check(load().zkhip_init(device))


def set_device(device):
    """Library device of the calling thread for the entry points without a handle (handles carry their own device)."""
    _check(load().zkhip_set_device(device))


def get_device():
    return load().zkhip_get_device()


def device_count():
    return load().zkhip_device_count()


def fr_random():
    """One uniform element of Fr (6 Montgomery limbs) from the OS entropy source."""
    out = np.zeros(6, dtype=np.uint64)
    _check(load().zkhip_fr_random(_p(out)))
    return out


def set_msm_window(c):
    _check(load().zkhip_set_msm_window(c))


def set_affine_levels(levels):
    """Batched-affine levels in front of the XYZZ bucket accumulation: -1 automatic, 0 none, up to 4."""
    _check(load().zkhip_set_affine_levels(int(levels)))


def set_batch_msms(on):
    """Table-backed keys: run the five MSMs of a proof through one launch sequence (default) or one each."""
    _check(load().zkhip_set_batch_msms(int(bool(on))))


def set_table_naf(on):
    """window tables with every bit position + non-adjacent-form scalars (zkhip_set_table_naf): 1 on, 0 off, -1 environment"""
    _check(load().zkhip_set_table_naf(int(on)))


def set_crs_precompute(on):
    """Whether Crs uploads build window tables for the five query vectors (default: on)."""
    _check(load().zkhip_set_crs_precompute(int(bool(on))))


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

    def precompute(self, c=0, table_naf=None):
        """Build the window table (2^(c w) P_i for every window position): all later msm calls use it.
        table_naf: None = the process default, True = every bit position + NAF scalars, False = one level per window."""
        _check(load().zkhip_bases_precompute_ex(self.handle, c, -1 if table_naf is None else int(bool(table_naf))))
        return self

    @property
    def table_window(self):
        return load().zkhip_bases_table_window(self.handle)

    def set_window(self, c):
        """Plain base set: window of the MSMs over it (0 = by the number of terms); an option of this handle."""
        load().zkhip_bases_set_window.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _check(load().zkhip_bases_set_window(self.handle, int(c)))
        return self

    def free(self):
        if self.handle:
            load().zkhip_bases_free(self.handle)
            self.handle = None

    def msm(self, scalars, offset=0, montgomery=True):
        s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 6)
        out = np.zeros(36, dtype=np.uint64)
        _check(load().zkhip_msm(self.handle, offset, _p(s), s.shape[0], int(montgomery), _p(out)))
        return out

    def msm_submit(self, dev_ptr, n, slot, offset=0, montgomery=True):
        """Enqueue an MSM on `slot` (0..7) and return; msm_collect(slot) waits for the result."""
        _check(load().zkhip_msm_submit(self.handle, offset, ctypes.c_void_p(dev_ptr), n, int(montgomery), slot))

    def msm_dev(self, dev_ptr, n, offset=0, montgomery=True):
        out = np.zeros(36, dtype=np.uint64)
        _check(load().zkhip_msm_dev(self.handle, offset, ctypes.c_void_p(dev_ptr), n, int(montgomery), _p(out)))
        return out


class MsmStream:
    """A stream of MSMs over one resident base set behind a handle (zkhip_msm_stream_*): `depth` MSMs in flight on contexts the
    stream owns.  submit() returns a ticket, collect(ticket) the Jacobian sum (ticket 0 / None: the oldest in flight)."""

    def __init__(self, bases, depth=8):
        h = ctypes.c_void_p()
        _check(load().zkhip_msm_stream_new(bases.handle, depth, ctypes.byref(h)))
        self.handle, self.depth, self._bases = h, depth, bases

    def submit(self, dev_ptr, n, offset=0, montgomery=True):
        t = ctypes.c_uint64(0)
        _check(load().zkhip_msm_stream_submit(self.handle, offset, ctypes.c_void_p(dev_ptr), n, int(montgomery), ctypes.byref(t)))
        return t.value

    def submit_host(self, host_ptr, n, offset=0, montgomery=True):
        """host_ptr: address of n x 6 u64 in host memory (a numpy array's ctypes.data, or a PinnedBuffer's ptr); it must stay valid
        until the ticket is collected."""
        t = ctypes.c_uint64(0)
        _check(load().zkhip_msm_stream_submit_host(self.handle, offset, ctypes.c_void_p(host_ptr), n, int(montgomery), ctypes.byref(t)))
        return t.value

    def collect(self, ticket=None):
        out = np.zeros(36, dtype=np.uint64)
        _check(load().zkhip_msm_stream_collect(self.handle, int(ticket or 0), _p(out)))
        return out

    def last_accumulate_ms(self):
        return float(load().zkhip_msm_stream_last_accumulate_ms(self.handle))

    def last_accumulate_interval(self):
        t = (ctypes.c_float * 2)()
        _check(load().zkhip_msm_stream_last_accumulate_interval(self.handle, t))
        return float(t[0]), float(t[1])

    def free(self):
        if self.handle:
            load().zkhip_msm_stream_free(self.handle)
            self.handle = None


class PinnedBuffer:
    """Pinned host memory owned by the library (zkhip_host_alloc) holding a copy of `host_array`: the source of asynchronous uploads."""

    def __init__(self, host_array):
        a = np.ascontiguousarray(host_array)
        p = ctypes.c_void_p()
        _check(load().zkhip_host_alloc(a.nbytes, ctypes.byref(p)))
        self.ptr, self.nbytes = p.value, a.nbytes
        ctypes.memmove(self.ptr, a.ctypes.data, a.nbytes)

    def free(self):
        if self.ptr:
            load().zkhip_host_free(ctypes.c_void_p(self.ptr))
            self.ptr = None


class DeviceBuffer:
    """A host array copied into device memory owned by the library (for the *_dev / submit entry points)."""

    def __init__(self, host_array):
        a = np.ascontiguousarray(host_array)
        p = ctypes.c_void_p()
        _check(load().zkhip_device_alloc(a.nbytes, ctypes.byref(p)))
        self.ptr, self.nbytes = p.value, a.nbytes
        _check(load().zkhip_device_copy_in(ctypes.c_void_p(self.ptr), a.ctypes.data_as(ctypes.c_void_p), a.nbytes))

    def free(self):
        if self.ptr:
            load().zkhip_device_free(ctypes.c_void_p(self.ptr))
            self.ptr = None


def msm_collect(slot):
    out = np.zeros(36, dtype=np.uint64)
    _check(load().zkhip_msm_collect(slot, _p(out)))
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


def make_r1cs_desc(A, B, C, n_vars, n_primary):
    """zkhip_r1cs_desc over CSR triples (row_ptr u32[n+1], col u32[nnz], val u64[nnz, 6]).  Returns (desc, keep): `keep` holds
    the arrays the descriptor points into and must stay alive while the descriptor is used (e.g. for Prover(crs, desc))."""
    keep = []
    d = R1csDesc()
    d.n_constraints = len(A[0]) - 1
    d.n_vars, d.n_primary = n_vars, n_primary
    for name, (rp, col, val) in zip("abc", (A, B, C)):
        rp = np.ascontiguousarray(rp, dtype=np.uint32)
        col = np.ascontiguousarray(col, dtype=np.uint32)
        val = np.ascontiguousarray(val, dtype=np.uint64).reshape(-1, 6)
        assert len(rp) == d.n_constraints + 1 and len(col) == len(val) == int(rp[-1])
        keep += [rp, col, val]
        setattr(d, name + "_row_ptr", rp.ctypes.data)
        setattr(d, name + "_col", col.ctypes.data if len(col) else None)
        setattr(d, name + "_val", val.ctypes.data if len(val) else None)
    return d, keep


DOMAIN_DEFAULT = 0                # ZKHIP_DOMAIN_DEFAULT: the reference's forced power of two (libzeth passes force_pow_2_domain = true)
DOMAIN_STEP = (1 << (8 * ctypes.sizeof(ctypes.c_size_t))) - 1     # ZKHIP_DOMAIN_STEP: libfqfft's unforced choice (2^k or 2^k + 2^r)


def _domain_arg(domain):
    return ctypes.c_size_t(DOMAIN_DEFAULT if not domain else (DOMAIN_STEP if domain in ("step", -1, DOMAIN_STEP) else int(domain)))


class R1cs:
    """A constraint system resident in HBM.  A, B, C: CSR triples (row_ptr u32[n+1], col u32[nnz], val u64[nnz, 6]).
    domain: None = the reference's forced power-of-two evaluation domain; "step" = libfqfft's unforced choice; an int = that many
    points (what a proving key says).  zkhip_groth16_prove moves the handle to its key's domain by itself."""

    def __init__(self, A, B, C, n_vars, n_primary, domain=None):
        self._keep = []
        d = R1csDesc()
        d.n_constraints = len(A[0]) - 1
        d.n_vars, d.n_primary = n_vars, n_primary
        for name, (rp, col, val) in zip("abc", (A, B, C)):
            rp = np.ascontiguousarray(rp, dtype=np.uint32)
            col = np.ascontiguousarray(col, dtype=np.uint32)
            val = np.ascontiguousarray(val, dtype=np.uint64).reshape(-1, 6)
            assert len(rp) == d.n_constraints + 1 and len(col) == len(val) == int(rp[-1])
            self._keep += [rp, col, val]
            setattr(d, name + "_row_ptr", rp.ctypes.data)
            setattr(d, name + "_col", col.ctypes.data if len(col) else None)
            setattr(d, name + "_val", val.ctypes.data if len(val) else None)
        self.n_vars, self.n_primary, self.n_constraints = n_vars, n_primary, d.n_constraints
        h = ctypes.c_void_p()
        _check(load().zkhip_r1cs_upload_ex(ctypes.byref(d), _domain_arg(domain), ctypes.byref(h)))
        self.handle = h
        self._keep = []

    @property
    def log_d(self):
        return int(load().zkhip_r1cs_log_domain(self.handle))

    @property
    def domain_size(self):
        """Points of the handle's CURRENT domain: a power of two (default), or 2^k + 2^r (libfqfft's step_radix2_domain)."""
        return int(load().zkhip_r1cs_domain_size(self.handle))

    def set_domain(self, domain):
        _check(load().zkhip_r1cs_set_domain(self.handle, _domain_arg(domain)))

    def is_satisfied(self, z):
        zz = np.ascontiguousarray(z, dtype=np.uint64).reshape(self.n_vars, 6)
        ok = ctypes.c_int(0)
        _check(load().zkhip_r1cs_is_satisfied(self.handle, _p(zz), ctypes.byref(ok)))
        return bool(ok.value)

    def qap_h(self, z):
        zz = np.ascontiguousarray(z, dtype=np.uint64).reshape(self.n_vars, 6)
        h = np.zeros((self.domain_size, 6), dtype=np.uint64)
        _check(load().zkhip_qap_h(self.handle, _p(zz), _p(h)))
        return h

    def free(self):
        if self.handle:
            load().zkhip_r1cs_free(self.handle)
            self.handle = None


class Crs:
    """The Groth16 proving key resident in HBM.  pk: dict with alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2
    (24 limbs) and the query arrays A, B2, B1, H, L (n x 24 limbs)."""

    def __init__(self, pk, n_vars, n_primary, domain_size, opts=None):
        d = CrsDesc()
        d.n_vars, d.n_primary, d.domain_size = n_vars, n_primary, domain_size
        keep = []
        for field, key, rows in (("alpha_g1", "alpha_g1", 1), ("beta_g1", "beta_g1", 1), ("beta_g2", "beta_g2", 1),
                                 ("delta_g1", "delta_g1", 1), ("delta_g2", "delta_g2", 1), ("a_query", "A", n_vars),
                                 ("b_g2_query", "B2", n_vars), ("b_g1_query", "B1", n_vars), ("h_query", "H", domain_size - 1),
                                 ("l_query", "L", n_vars - n_primary - 1)):
            a = np.ascontiguousarray(pk[key], dtype=np.uint64).reshape(-1, 24)  # (extra keys such as "vk" are ignored)
            assert a.shape[0] == rows, (key, a.shape, rows)
            keep.append(a)
            setattr(d, field, a.ctypes.data if a.size else None)
        h = ctypes.c_void_p()
        _check(load().zkhip_crs_upload_ex(ctypes.byref(d), ctypes.byref(opts) if opts is not None else None, ctypes.byref(h)))
        self.handle = h

    @property
    def table_window(self):
        """Window size of the key's precomputed tables (0: plain key)."""
        return load().zkhip_crs_table_window(self.handle)

    @property
    def table_kind(self):
        """0: plain key, 1: one table level per window, 2: every bit position (scalars in non-adjacent form)."""
        return load().zkhip_crs_table_kind(self.handle)

    def finite_terms(self):
        """Bases of A, B-G2, B-G1, H, L that are not the point at infinity (the terms an MSM over the key can have)."""
        out = (ctypes.c_size_t * 5)()
        _check(load().zkhip_crs_finite_terms(self.handle, out))
        return [int(v) for v in out]

    def free(self):
        if self.handle:
            load().zkhip_crs_free(self.handle)
            self.handle = None

    @classmethod
    def upload_slice(cls, pk, n_vars, n_primary, domain_size, a_range, h_range, l_range, opts=None):
        """This rank's slice of the proving key: ranges are (lo, hi) into the A/B queries, the H query and the L query."""
        d = CrsDesc()
        d.n_vars, d.n_primary, d.domain_size = n_vars, n_primary, domain_size
        keep = []
        for field, key in (("alpha_g1", "alpha_g1"), ("beta_g1", "beta_g1"), ("beta_g2", "beta_g2"), ("delta_g1", "delta_g1"),
                           ("delta_g2", "delta_g2"), ("a_query", "A"), ("b_g2_query", "B2"), ("b_g1_query", "B1"), ("h_query", "H"),
                           ("l_query", "L")):
            a = np.ascontiguousarray(pk[key], dtype=np.uint64).reshape(-1, 24)
            keep.append(a)
            setattr(d, field, a.ctypes.data if a.size else None)
        self = cls.__new__(cls)
        h = ctypes.c_void_p()
        _check(load().zkhip_crs_upload_slice_ex(ctypes.byref(d), a_range[0], a_range[1] - a_range[0], h_range[0], h_range[1] - h_range[0],
                                                l_range[0], l_range[1] - l_range[0], ctypes.byref(opts) if opts is not None else None,
                                                ctypes.byref(h)))
        self.handle = h
        self.ranges = (a_range, h_range, l_range)
        return self


def key_partition(pk, n_vars, n_primary, domain_size, parts):
    """zkhip_key_partition on host arrays: (a_cuts, h_cuts, l_cuts), parts + 1 values each.  Host code, no device."""
    d = CrsDesc()
    d.n_vars, d.n_primary, d.domain_size = n_vars, n_primary, domain_size
    keep = []
    for field, key in (("a_query", "A"), ("b_g2_query", "B2"), ("b_g1_query", "B1"), ("h_query", "H"), ("l_query", "L")):
        a = np.ascontiguousarray(pk[key], dtype=np.uint64).reshape(-1, 24)
        keep.append(a)
        setattr(d, field, a.ctypes.data if len(a) else None)
    cuts = [(ctypes.c_size_t * (parts + 1))() for _ in range(3)]
    lib = load()
    lib.zkhip_key_partition.argtypes = [ctypes.POINTER(CrsDesc), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    _check(lib.zkhip_key_partition(ctypes.byref(d), parts, cuts[0], cuts[1], cuts[2]))
    return tuple([int(x) for x in c] for c in cuts)


def crs_from_slice_arrays(consts, slices, n_vars, n_primary, domain_size, a_range, h_range, l_range):
    """Like Crs.upload_slice, but from arrays that hold ONLY this rank's slices (A, B2, B1: a_hi - a_lo points; H; L):
    what a rank of a multi-GPU job keeps in host memory.  consts: alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2."""
    d = CrsDesc()
    d.n_vars, d.n_primary, d.domain_size = n_vars, n_primary, domain_size
    keep = []
    for field in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2"):
        a = np.ascontiguousarray(consts[field], dtype=np.uint64).reshape(24)
        keep.append(a)
        setattr(d, field, a.ctypes.data)
    for field, key, lo, n in (("a_query", "A", a_range[0], a_range[1] - a_range[0]), ("b_g2_query", "B2", a_range[0], a_range[1] - a_range[0]),
                              ("b_g1_query", "B1", a_range[0], a_range[1] - a_range[0]), ("h_query", "H", h_range[0], h_range[1] - h_range[0]),
                              ("l_query", "L", l_range[0], l_range[1] - l_range[0])):
        a = np.ascontiguousarray(slices[key], dtype=np.uint64).reshape(-1, 24)
        assert a.shape[0] == n, (key, a.shape, n)
        keep.append(a)
        # the descriptor addresses the FULL vector: element `lo` of it is the first element of this slice
        setattr(d, field, (a.ctypes.data - lo * 192) if n else None)
    c = Crs.__new__(Crs)
    h = ctypes.c_void_p()
    _check(load().zkhip_crs_upload_slice(ctypes.byref(d), a_range[0], a_range[1] - a_range[0], h_range[0], h_range[1] - h_range[0],
                                         l_range[0], l_range[1] - l_range[0], ctypes.byref(h)))
    c.handle = h
    c.ranges = (a_range, h_range, l_range)
    return c


def groth16_prove_partial(crs_slice, r1cs, z):
    """The five partial sums (5 x 36 limbs) of this rank's key slice."""
    zz = np.ascontiguousarray(z, dtype=np.uint64).reshape(r1cs.n_vars, 6)
    out = np.zeros((5, 36), dtype=np.uint64)
    (a0, _), (h0, _), (l0, _) = crs_slice.ranges
    _check(load().zkhip_groth16_prove_partial(crs_slice.handle, r1cs.handle, _p(zz), a0, h0, l0, _p(out)))
    return out


def groth16_finish(pk, sums, r, s):
    c = lambda a: _p(np.ascontiguousarray(a, dtype=np.uint64))
    out = np.zeros(72, dtype=np.uint64)
    _check(load().zkhip_groth16_finish(c(pk["alpha_g1"]), c(pk["beta_g1"]), c(pk["beta_g2"]), c(pk["delta_g1"]), c(pk["delta_g2"]),
                                       c(np.ascontiguousarray(sums, dtype=np.uint64).reshape(-1)), c(r), c(s), _p(out)))
    return out


def groth16_prove(crs, r1cs, z, r, s):
    """Proof (A in G1, B in G2, C in G1) as 72 limbs, affine.  r, s: Fr in Montgomery form (6 limbs)."""
    zz = np.ascontiguousarray(z, dtype=np.uint64).reshape(r1cs.n_vars, 6)
    out = np.zeros(72, dtype=np.uint64)
    _check(load().zkhip_groth16_prove(crs.handle, r1cs.handle, _p(zz), _p(np.ascontiguousarray(r, dtype=np.uint64)),
                                      _p(np.ascontiguousarray(s, dtype=np.uint64)), _p(out)))
    return out


def groth16_verify(vk, inputs, proof):
    """vk: dict alpha (G1), beta, delta (G2), ABC ((n+1) x 24 limbs); inputs: n x 6 limbs; proof: 72 limbs.
    Host code: works without a GPU."""
    c = lambda a: np.ascontiguousarray(a, dtype=np.uint64)
    abc = c(vk["ABC"]).reshape(-1, 24)
    inp = c(inputs).reshape(-1, 6)
    assert abc.shape[0] == inp.shape[0] + 1
    ok = ctypes.c_int(0)
    _check(load().zkhip_groth16_verify(_p(c(vk["alpha"])), _p(c(vk["beta"])), _p(c(vk["delta"])), _p(abc), _p(inp), inp.shape[0],
                                       _p(c(proof)), ctypes.byref(ok)))
    return bool(ok.value)


def bls12_377_groth16_verify(nested_vk, inputs, proof):
    """Nested (BLS12-377) Groth16 verification on the host.  nested_vk: 60 + 12 (k+1) limbs (alpha | beta | delta | ABC);
    inputs: k x 6 limbs; proof: 48 limbs (a | b | c)."""
    c = lambda a: np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
    vk, pr, inp = c(nested_vk), c(proof), c(inputs).reshape(-1, 6)
    assert vk.size == 60 + 12 * (inp.shape[0] + 1) and pr.size == 48
    ok = ctypes.c_int(0)
    _check(load().zkhip_bls12_377_groth16_verify(_p(vk[:12]), _p(vk[12:36]), _p(vk[36:60]), _p(vk[60:]), _p(inp), inp.shape[0],
                                                 _p(pr[:12]), _p(pr[12:36]), _p(pr[36:]), ctypes.byref(ok)))
    return bool(ok.value)


class AggregatorCircuit:
    """Mirror of libzecale::aggregator_circuit<wpp, wsnark, nverifier, NumProofs> (aggregator_circuit.hpp:32-114):
    builds the wrapping circuit on construction; `witness` performs the generate_r1cs_witness half of prove()."""

    def __init__(self, num_proofs=2, inputs_per_nested_proof=1):
        h = ctypes.c_void_p()
        _check(load().zkhip_aggregator_new(num_proofs, inputs_per_nested_proof, ctypes.byref(h)))
        self.handle = h
        self.num_proofs, self.inputs_per_nested_proof = num_proofs, inputs_per_nested_proof
        lib = load()
        self.num_constraints = lib.zkhip_aggregator_num_constraints(h)
        self.num_variables = lib.zkhip_aggregator_num_variables(h)

    def num_primary_inputs(self):
        return load().zkhip_aggregator_num_primary_inputs(self.handle)

    def get_constraint_system(self):
        """CSR triples (row_ptr, col, val) of A, B, C as numpy arrays (copies)."""
        d = R1csDesc()
        _check(load().zkhip_aggregator_get_r1cs(self.handle, ctypes.byref(d)))
        out = []
        for m in "abc":
            n = d.n_constraints
            rp = np.ctypeslib.as_array(ctypes.cast(getattr(d, m + "_row_ptr"), ctypes.POINTER(ctypes.c_uint32)), (n + 1,)).copy()
            nnz = int(rp[-1])
            col = np.ctypeslib.as_array(ctypes.cast(getattr(d, m + "_col"), ctypes.POINTER(ctypes.c_uint32)), (nnz,)).copy()
            val = np.ctypeslib.as_array(ctypes.cast(getattr(d, m + "_val"), ctypes.POINTER(ctypes.c_uint64)), (nnz, 6)).copy()
            out.append((rp, col, val))
        return out

    def witness(self, nested_vk, nested_proofs, nested_inputs):
        c = lambda a: np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
        vk, pr, inp = c(nested_vk), c(nested_proofs), c(nested_inputs)
        k = self.inputs_per_nested_proof
        assert vk.size == 60 + 12 * (k + 1) and pr.size == 48 * self.num_proofs and inp.size == 6 * k * self.num_proofs
        z = np.zeros((self.num_variables, 6), dtype=np.uint64)
        _check(load().zkhip_aggregator_witness(self.handle, _p(vk), _p(pr), _p(inp), _p(z)))
        return z

    def witness_gpu(self, nested_vk, nested_proofs, nested_inputs):
        """The same assignment computed on the GPU (zkhip_aggregator_witness_gpu): needs zkhip.init()."""
        c = lambda a: np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
        vk, pr, inp = c(nested_vk), c(nested_proofs), c(nested_inputs)
        k = self.inputs_per_nested_proof
        assert vk.size == 60 + 12 * (k + 1) and pr.size == 48 * self.num_proofs and inp.size == 6 * k * self.num_proofs
        z = np.zeros((self.num_variables, 6), dtype=np.uint64)
        _check(load().zkhip_aggregator_witness_gpu(self.handle, _p(vk), _p(pr), _p(inp), _p(z)))
        return z

    def gpu_witness_stats(self):
        out = (ctypes.c_size_t * 6)()
        _check(load().zkhip_gpu_witness_stats(self.handle, out))
        return dict(zip(["recorded", "positions", "levels", "multiplications", "inversions", "constants"], [int(x) for x in out]))

    def check_inputs(self, nested_vk, nested_proofs):
        """True iff every point of the nested key and proofs is on its curve (proof.is_well_formed() in the reference's stack)."""
        c = lambda a: np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
        ok = ctypes.c_int(0)
        _check(load().zkhip_aggregator_check_inputs(self.handle, _p(c(nested_vk)), _p(c(nested_proofs)), ctypes.byref(ok)))
        return bool(ok.value)

    def free(self):
        if self.handle:
            load().zkhip_aggregator_free(self.handle)
            self.handle = None


class AggregatorApp:
    """A registered application (zkhip_aggregator_app; RegisterApplication in the reference, aggregator_server.cpp:170-235): the
    constants of its nested key for one proving key - positions, values, the four cached points - computed once.  `witness` gives
    the MASKED assignment of a batch (zeros at those positions), `prove` the same proof as groth16_prove on the full one."""

    def __init__(self, agg, crs, nested_vk):
        vk = np.ascontiguousarray(nested_vk, dtype=np.uint64).reshape(-1)
        assert vk.size == 60 + 12 * (agg.inputs_per_nested_proof + 1)
        h = ctypes.c_void_p()
        lib = load()
        lib.zkhip_aggregator_app_new.argtypes = [ctypes.c_void_p, ctypes.c_void_p, c_u64p_t, ctypes.POINTER(ctypes.c_void_p)]
        _check(lib.zkhip_aggregator_app_new(agg.handle, crs.handle, _p(vk), ctypes.byref(h)))
        self.handle, self._agg, self._crs = h, agg, crs
        lib.zkhip_aggregator_app_num_constants.restype = ctypes.c_size_t
        lib.zkhip_aggregator_app_num_constants.argtypes = [ctypes.c_void_p]
        self.num_constants = int(lib.zkhip_aggregator_app_num_constants(h))

    def constants(self):
        """(positions u32[n], values u64[n, 6], vk_hash u64[6], points u64[4, 36])"""
        pos = np.zeros(self.num_constants, dtype=np.uint32)
        val = np.zeros((self.num_constants, 6), dtype=np.uint64)
        h, pts = np.zeros(6, dtype=np.uint64), np.zeros((4, 36), dtype=np.uint64)
        lib = load()
        lib.zkhip_aggregator_app_constants.argtypes = [ctypes.c_void_p, ctypes.c_void_p, c_u64p_t, c_u64p_t, c_u64p_t]
        _check(lib.zkhip_aggregator_app_constants(self.handle, pos.ctypes.data, _p(val), _p(h), _p(pts)))
        return pos, val, h, pts

    def mask(self, z):
        """A full assignment generated under this application's key, masked (copy)."""
        zz = np.ascontiguousarray(z, dtype=np.uint64).reshape(-1, 6).copy()
        lib = load()
        lib.zkhip_aggregator_app_mask.argtypes = [ctypes.c_void_p, c_u64p_t]
        _check(lib.zkhip_aggregator_app_mask(self.handle, _p(zz)))
        return zz

    def witness(self, nested_proofs, nested_inputs):
        c = lambda a: np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
        pr, inp = c(nested_proofs), c(nested_inputs)
        k = self._agg.inputs_per_nested_proof
        assert pr.size == 48 * self._agg.num_proofs and inp.size == 6 * k * self._agg.num_proofs
        z = np.zeros((self._agg.num_variables, 6), dtype=np.uint64)
        lib = load()
        lib.zkhip_aggregator_witness_app.argtypes = [ctypes.c_void_p, c_u64p_t, c_u64p_t, c_u64p_t]
        _check(lib.zkhip_aggregator_witness_app(self.handle, _p(pr), _p(inp), _p(z)))
        return z

    def witness_gpu(self, batches):
        """batches: list of (nested_proofs, nested_inputs) -> list of MASKED assignments generated on the GPU by the application's own
        program in ONE launch sequence (zkhip_gpu_witness_run_batched_app); a degenerate batch yields None."""
        lib = load()
        n, m, l = len(batches), self._agg.num_variables, self._agg.num_primary_inputs()
        c = lambda a: np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
        prs, ins = [c(b[0]) for b in batches], [c(b[1]) for b in batches]
        gw = ctypes.c_void_p()
        lib.zkhip_gpu_witness_new_batched.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        _check(lib.zkhip_gpu_witness_new_batched(self._agg.handle, n, ctypes.byref(gw)))
        dz = ctypes.c_void_p()
        _check(lib.zkhip_device_alloc(n * m * 48, ctypes.byref(dz)))
        try:
            arr = lambda xs: (ctypes.c_void_p * n)(*[x.ctypes.data for x in xs])
            prim = np.zeros((n, l, 6), dtype=np.uint64)
            deg = (ctypes.c_int * n)()
            lib.zkhip_gpu_witness_run_batched_app.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_u64p_t, ctypes.POINTER(ctypes.c_int)]
            _check(lib.zkhip_gpu_witness_run_batched_app(gw, self.handle, n, arr(prs), arr(ins), dz, _p(prim), deg))
            z = np.zeros((n, m, 6), dtype=np.uint64)
            lib.zkhip_device_copy_out.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
            _check(lib.zkhip_device_copy_out(z.ctypes.data, dz, n * m * 48))
            return [None if deg[i] else z[i] for i in range(n)], prim
        finally:
            lib.zkhip_device_free(dz)
            lib.zkhip_gpu_witness_free(gw)

    def prove(self, r1cs, z_masked, r, s):
        c = lambda a: np.ascontiguousarray(a, dtype=np.uint64)
        zz, r, s = c(z_masked).reshape(r1cs.n_vars, 6), c(r), c(s)
        out = np.zeros(72, dtype=np.uint64)
        lib = load()
        lib.zkhip_groth16_prove_app.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_u64p_t, c_u64p_t, c_u64p_t, c_u64p_t]
        _check(lib.zkhip_groth16_prove_app(self._crs.handle, r1cs.handle, self.handle, _p(zz), _p(r), _p(s), _p(out)))
        return out

    def free(self):
        if self.handle:
            load().zkhip_aggregator_app_free(self.handle)
            self.handle = None


class Prover:
    """A prover instance (zkhip_prover): own streams and work space, one proof in flight; several instances, one host
    thread each, keep several proofs in flight on one GPU.  `crs` must outlive the instance."""

    def __init__(self, crs, r1cs_desc):
        h = ctypes.c_void_p()
        _check(load().zkhip_prover_new(crs.handle, ctypes.byref(r1cs_desc), ctypes.byref(h)))
        self.handle, self._crs = h, crs

    def prove(self, z, r, s):
        c = lambda a: np.ascontiguousarray(a, dtype=np.uint64)
        z, r, s = c(z), c(r), c(s)
        out = np.zeros(72, dtype=np.uint64)
        _check(load().zkhip_prover_prove(self.handle, _p(z), _p(r), _p(s), _p(out)))
        return out

    def prove_app(self, app, z_masked, r, s):
        """The same proof from a MASKED assignment and the application's constants (zkhip_prover_prove_app)."""
        c = lambda a: np.ascontiguousarray(a, dtype=np.uint64)
        z, r, s = c(z_masked), c(r), c(s)
        out = np.zeros(72, dtype=np.uint64)
        lib = load()
        lib.zkhip_prover_prove_app.argtypes = [ctypes.c_void_p, ctypes.c_void_p, c_u64p_t, c_u64p_t, c_u64p_t, c_u64p_t]
        _check(lib.zkhip_prover_prove_app(self.handle, app.handle, _p(z), _p(r), _p(s), _p(out)))
        return out

    def prove_app_dev(self, app, d_z_masked, r, s):
        """zkhip_prover_prove_app_dev: the masked assignment already in DEVICE memory (an integer pointer; what the application's GPU
        program writes).  Precondition: it is masked (see include/zkhip.h) - not checked."""
        c = lambda a: np.ascontiguousarray(a, dtype=np.uint64)
        r, s = c(r), c(s)
        out = np.zeros(72, dtype=np.uint64)
        lib = load()
        lib.zkhip_prover_prove_app_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_u64p_t, c_u64p_t, c_u64p_t]
        _check(lib.zkhip_prover_prove_app_dev(self.handle, app.handle, ctypes.c_void_p(int(d_z_masked)), _p(r), _p(s), _p(out)))
        return out

    def create_streams(self, which):
        """zkhip_prover_create_streams: for a caller with several instances - which = 0 for all of them, then which = 1 for all"""
        _check(load().zkhip_prover_create_streams(self.handle, int(which)))

    def set_streaming(self, on=True):
        """this instance shares the GPU with others: total work over single-proof latency (zkhip_prover_set_streaming)"""
        _check(load().zkhip_prover_set_streaming(self.handle, int(bool(on))))

    def last_accumulate_ms(self):
        return float(load().zkhip_prover_last_accumulate_ms(self.handle))

    def last_accumulate_entries(self):
        out = ctypes.c_uint64(0)
        _check(load().zkhip_prover_last_accumulate_entries(self.handle, ctypes.byref(out)))
        return int(out.value)

    def timings(self):
        t = (ctypes.c_double * 8)()
        _check(load().zkhip_prover_timings(self.handle, t))
        out = dict(zip(["upload_z", "qap", "msm_A", "msm_B2", "msm_B1", "msm_H", "msm_L", "host_tail"], list(t)))
        out["chained"] = bool(load().zkhip_prover_timings_chained(self.handle))       # True: upload_z / qap are enqueueing times
        return out

    def free(self):
        if self.handle:
            load().zkhip_prover_free(self.handle)
            self.handle = None


class AggregatorPipeline:
    """Streaming aggregator_circuit::prove (zkhip_aggregator_pipeline_*): submit() returns a ticket at once, wait(ticket)
    returns (primary_inputs, proof).  Witness generation, the GPU prover and the host tail of successive batches overlap."""

    def __init__(self, agg, crs, gpu_slots=2, witness_workers=2, gpu_witness=False, app_cache=True, hybrid=False):
        """app_cache: keep a zkhip_aggregator_app per nested key seen and prove its batches from masked assignments (default; the
        proofs are the same either way).  hybrid (with gpu_witness): two host generators beside the GPU batchers."""
        h = ctypes.c_void_p()
        lib = load()
        lib.zkhip_aggregator_pipeline_new_ex.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.POINTER(ctypes.c_void_p)]
        flags = (1 if gpu_witness else 0) | (0 if app_cache else 2) | (4 if (hybrid and gpu_witness) else 0)
        _check(lib.zkhip_aggregator_pipeline_new_ex(agg.handle, crs.handle, gpu_slots, witness_workers, flags, ctypes.byref(h)))
        self.handle, self._agg, self._crs = h, agg, crs
        self.n_primary = agg.num_primary_inputs()

    def register_app(self, nested_vk):
        """RegisterApplication's part in the prover: the application's constants are computed now (zkhip_aggregator_pipeline_register_app)."""
        vk = np.ascontiguousarray(nested_vk, dtype=np.uint64).reshape(-1)
        lib = load()
        lib.zkhip_aggregator_pipeline_register_app.argtypes = [ctypes.c_void_p, c_u64p_t]
        _check(lib.zkhip_aggregator_pipeline_register_app(self.handle, _p(vk)))

    def app_hits(self):
        lib = load()
        lib.zkhip_aggregator_pipeline_app_hits.restype = ctypes.c_size_t
        lib.zkhip_aggregator_pipeline_app_hits.argtypes = [ctypes.c_void_p]
        return int(lib.zkhip_aggregator_pipeline_app_hits(self.handle))

    def submit(self, nested_vk, nested_proofs, nested_inputs, r, s):
        c = lambda a: np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
        vk, pr, inp, r, s = c(nested_vk), c(nested_proofs), c(nested_inputs), c(r), c(s)
        k, npf = self._agg.inputs_per_nested_proof, self._agg.num_proofs
        assert vk.size == 60 + 12 * (k + 1) and pr.size == 48 * npf and inp.size == 6 * k * npf and r.size == 6 and s.size == 6
        t = ctypes.c_uint64(0)
        _check(load().zkhip_aggregator_pipeline_submit(self.handle, _p(vk), _p(pr), _p(inp), _p(r), _p(s), ctypes.byref(t)))
        return t.value

    def wait(self, ticket):
        prim = np.zeros((self.n_primary, 6), dtype=np.uint64)
        proof = np.zeros(72, dtype=np.uint64)
        _check(load().zkhip_aggregator_pipeline_wait(self.handle, ticket, _p(prim), _p(proof)))
        return prim, proof

    def free(self):
        if self.handle:
            load().zkhip_aggregator_pipeline_free(self.handle)
            self.handle = None


def _int_list(devices):
    d = [int(x) for x in devices]
    return (ctypes.c_int * len(d))(*d), len(d)


class AggregatorDispatcher:
    """The GPUs of a node behind ONE streaming prover (zkhip_dispatcher_*): a resident copy of the key and a pipeline per entry of
    `devices` (an index may repeat: two contexts on one GPU), every batch goes to the entry with the fewest batches outstanding.
    Same submit / wait as AggregatorPipeline.  keypair: a Keypair (its proving half is uploaded to every entry)."""

    def __init__(self, agg, keypair, devices, opts=None, gpu_slots=32, witness_workers=10, gpu_witness=False, app_cache=True, hybrid=False):
        d = CrsDesc()
        _check(load().zkhip_keypair_crs_desc(keypair.handle, ctypes.byref(d)))
        arr, n = _int_list(devices)
        h = ctypes.c_void_p()
        _check(load().zkhip_dispatcher_new(agg.handle, ctypes.byref(d), ctypes.byref(opts) if opts is not None else None, arr, n,
                                           gpu_slots, witness_workers, (1 if gpu_witness else 0) | (0 if app_cache else 2) | (4 if (hybrid and gpu_witness) else 0),
                                           ctypes.byref(h)))
        self.handle, self._agg, self._kp = h, agg, keypair
        self.n_primary = agg.num_primary_inputs()
        self.size = n

    def submit(self, nested_vk, nested_proofs, nested_inputs, r, s):
        c = lambda a: np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
        vk, pr, inp, r, s = c(nested_vk), c(nested_proofs), c(nested_inputs), c(r), c(s)
        k, npf = self._agg.inputs_per_nested_proof, self._agg.num_proofs
        assert vk.size == 60 + 12 * (k + 1) and pr.size == 48 * npf and inp.size == 6 * k * npf and r.size == 6 and s.size == 6
        t = ctypes.c_uint64(0)
        _check(load().zkhip_dispatcher_submit(self.handle, _p(vk), _p(pr), _p(inp), _p(r), _p(s), ctypes.byref(t)))
        return t.value

    def wait(self, ticket):
        prim = np.zeros((self.n_primary, 6), dtype=np.uint64)
        proof = np.zeros(72, dtype=np.uint64)
        _check(load().zkhip_dispatcher_wait(self.handle, ticket, _p(prim), _p(proof)))
        return prim, proof

    def register_app(self, nested_vk):
        """RegisterApplication on every entry (zkhip_dispatcher_register_app)."""
        vk = np.ascontiguousarray(nested_vk, dtype=np.uint64).reshape(-1)
        lib = load()
        lib.zkhip_dispatcher_register_app.argtypes = [ctypes.c_void_p, c_u64p_t]
        _check(lib.zkhip_dispatcher_register_app(self.handle, _p(vk)))

    def stats(self):
        """Batches handed to each entry of the device list so far."""
        out = (ctypes.c_size_t * self.size)()
        _check(load().zkhip_dispatcher_stats(self.handle, out))
        return list(out)

    def outstanding(self):
        """Batches each entry still owes a collector (what submit's least-loaded routing looks at)."""
        out = (ctypes.c_size_t * self.size)()
        _check(load().zkhip_dispatcher_outstanding(self.handle, out))
        return list(out)

    def free(self):
        if self.handle:
            load().zkhip_dispatcher_free(self.handle)
            self.handle = None


class MultiProver:
    """One proof over a key partitioned across `devices` (zkhip_multi_prover_*): a slice and a prover instance per entry, host
    threads side by side, partial sums added on the host, one tail.  The proof equals the whole-key proof limb for limb."""

    def __init__(self, keypair, r1cs_desc, devices, opts=None):
        d = CrsDesc()
        _check(load().zkhip_keypair_crs_desc(keypair.handle, ctypes.byref(d)))
        arr, n = _int_list(devices)
        h = ctypes.c_void_p()
        _check(load().zkhip_multi_prover_new(ctypes.byref(d), ctypes.byref(r1cs_desc), ctypes.byref(opts) if opts is not None else None, arr, n, ctypes.byref(h)))
        self.handle, self._kp, self._desc, self.size = h, keypair, r1cs_desc, n

    def prove(self, z, r, s):
        z = np.ascontiguousarray(z, dtype=np.uint64).reshape(-1, 6)
        r, s = (np.ascontiguousarray(a, dtype=np.uint64).reshape(6) for a in (r, s))
        out = np.zeros(72, dtype=np.uint64)
        _check(load().zkhip_multi_prover_prove(self.handle, _p(z), _p(r), _p(s), _p(out)))
        return out

    def timings(self):
        t = (ctypes.c_double * 3)()
        _check(load().zkhip_multi_prover_timings(self.handle, t))
        return dict(zip(["slowest_slice", "host_additions", "host_tail"], list(t)))

    def free(self):
        if self.handle:
            load().zkhip_multi_prover_free(self.handle)
            self.handle = None


class Keypair:
    """Groth16 keypair from a trusted setup on the GPU (mirror of wsnark::generate_setup / keypair)."""

    def __init__(self, r1cs_desc, tau, alpha, beta, delta, domain=None):
        """domain: None = the forced power of two the reference's generate_setup uses; "step" = libfqfft's unforced choice (an
        option; not a reference deployment's key); an int = that many points."""
        h = ctypes.c_void_p()
        c = lambda a: _p(np.ascontiguousarray(a, dtype=np.uint64))
        lib = load()
        lib.zkhip_groth16_setup_ex.argtypes = [ctypes.POINTER(R1csDesc), c_u64p_t, c_u64p_t, c_u64p_t, c_u64p_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        _check(lib.zkhip_groth16_setup_ex(ctypes.byref(r1cs_desc), c(tau), c(alpha), c(beta), c(delta), _domain_arg(domain), ctypes.byref(h)))
        self.handle = h

    @classmethod
    def setup_slice(cls, r1cs_desc, tau, alpha, beta, delta, parts, part, opts=None, domain=None):
        """zkhip_groth16_setup_slice: this rank's share of a trusted setup -> (keypair WITHOUT queries: vk() and consts() work, Crs slice
        with .ranges).  Only the slice is multiplied out, on the device; nothing of the proving half visits the host."""
        lib = load()
        c = lambda a: _p(np.ascontiguousarray(a, dtype=np.uint64))
        lib.zkhip_groth16_setup_slice.argtypes = [ctypes.POINTER(R1csDesc), c_u64p_t, c_u64p_t, c_u64p_t, c_u64p_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_void_p)]
        hc, hk = ctypes.c_void_p(), ctypes.c_void_p()
        rng = (ctypes.c_size_t * 6)()
        _check(lib.zkhip_groth16_setup_slice(ctypes.byref(r1cs_desc), c(tau), c(alpha), c(beta), c(delta), _domain_arg(domain), int(parts), int(part),
                                             ctypes.byref(opts) if opts is not None else None, ctypes.byref(hc), rng, ctypes.byref(hk)))
        kp = cls.__new__(cls)
        kp.handle = hk
        crs = Crs.__new__(Crs)
        crs.handle = hc
        crs.ranges = ((int(rng[0]), int(rng[1])), (int(rng[2]), int(rng[3])), (int(rng[4]), int(rng[5])))
        return kp, crs

    @property
    def domain_size(self):
        d = CrsDesc()
        _check(load().zkhip_keypair_crs_desc(self.handle, ctypes.byref(d)))
        return int(d.domain_size)

    def write(self, path):
        """Mirror of wsnark::keypair_write_bytes (aggregator_server.cpp:88-94); the library's own container format."""
        _check(load().zkhip_keypair_write(self.handle, str(path).encode()))

    @classmethod
    def read(cls, path):
        """Mirror of wsnark::keypair_read_bytes (aggregator_server.cpp:77-86)."""
        h = ctypes.c_void_p()
        _check(load().zkhip_keypair_read(str(path).encode(), ctypes.byref(h)))
        kp = cls.__new__(cls)
        kp.handle = h
        return kp

    def upload_crs(self, opts=None):
        """opts: key_opts(...) - this key's own table / launch options (None: the process defaults)."""
        d = CrsDesc()
        _check(load().zkhip_keypair_crs_desc(self.handle, ctypes.byref(d)))
        if d.n_vars and not d.a_query:
            raise ZkhipError("this keypair holds no query vectors (Keypair.setup_slice): there is nothing to upload")
        crs = Crs.__new__(Crs)
        h = ctypes.c_void_p()
        _check(load().zkhip_crs_upload_ex(ctypes.byref(d), ctypes.byref(opts) if opts is not None else None, ctypes.byref(h)))
        crs.handle = h
        return crs

    def consts(self):
        """alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2 of the proving key (24 limbs each): what zkhip_groth16_finish takes."""
        d = CrsDesc()
        _check(load().zkhip_keypair_crs_desc(self.handle, ctypes.byref(d)))
        return {k: np.ctypeslib.as_array(ctypes.cast(getattr(d, k), c_u64p_t), (24,)).copy()
                for k in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2")}

    def pk_arrays(self):
        """Host copies of the proving half (dict in the layout Crs / the test oracle take) plus n_vars, n_primary, domain_size."""
        d = CrsDesc()
        _check(load().zkhip_keypair_crs_desc(self.handle, ctypes.byref(d)))
        m, l, dom = d.n_vars, d.n_primary, d.domain_size
        if m and not d.a_query:
            raise ZkhipError("this keypair holds no query vectors (Keypair.setup_slice: the slice lives on the device; vk() and consts() are what it keeps)")
        arr = lambda ptr, n: (np.ctypeslib.as_array(ctypes.cast(ptr, c_u64p_t), (n, 24)).copy() if n else np.zeros((0, 24), dtype=np.uint64))
        pk = {k: arr(getattr(d, k), 1).reshape(24) for k in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2")}
        pk.update(A=arr(d.a_query, m), B2=arr(d.b_g2_query, m), B1=arr(d.b_g1_query, m), H=arr(d.h_query, dom - 1), L=arr(d.l_query, m - l - 1))
        return pk, m, l, dom

    def vk(self):
        a, b, dl = (np.zeros(24, dtype=np.uint64) for _ in range(3))
        abc = c_u64p_t()
        n = load().zkhip_keypair_vk(self.handle, _p(a), _p(b), _p(dl), ctypes.byref(abc))
        return dict(alpha=a, beta=b, delta=dl, ABC=np.ctypeslib.as_array(abc, (n, 24)).copy())

    def free(self):
        if self.handle:
            load().zkhip_keypair_free(self.handle)
            self.handle = None


def r1cs_desc_from_aggregator(agg):
    d = R1csDesc()
    _check(load().zkhip_aggregator_get_r1cs(agg.handle, ctypes.byref(d)))
    return d


def r1cs_from_desc(desc, domain=None):
    """Upload a constraint system described by a zkhip_r1cs_desc (e.g. the aggregator circuit's).  domain: as R1cs."""
    r = R1cs.__new__(R1cs)
    h = ctypes.c_void_p()
    _check(load().zkhip_r1cs_upload_ex(ctypes.byref(desc), _domain_arg(domain), ctypes.byref(h)))
    r.handle, r._keep = h, []
    r.n_vars, r.n_primary, r.n_constraints = desc.n_vars, desc.n_primary, desc.n_constraints
    return r


def aggregator_vk_hash(nested_vk, inputs_per_nested_proof=1):
    vk = np.ascontiguousarray(nested_vk, dtype=np.uint64).reshape(-1)
    out = np.zeros(6, dtype=np.uint64)
    _check(load().zkhip_aggregator_vk_hash(_p(vk), inputs_per_nested_proof, _p(out)))
    return out


def domain_size(min_size):
    """Points of the reference's evaluation domain for min_size points: the forced power of two (host code)."""
    return int(load().zkhip_domain_size(min_size))


def step_domain_size(min_size):
    """Points of the domain libfqfft picks for min_size points when not forced: 2^k, or 2^k + 2^r (host code)."""
    return int(load().zkhip_step_domain_size(min_size))


def domain_is_valid(d):
    return bool(load().zkhip_domain_is_valid(d))


def last_prove_timings():
    t = (ctypes.c_double * 8)()
    _check(load().zkhip_last_prove_timings(t))
    return dict(zip(["upload_z", "qap", "msm_A", "msm_B2", "msm_B1", "msm_H", "msm_L", "host_tail"], list(t)))


def set_prove_split(mode):
    """0: a proof's five MSMs in one launch sequence (default); 1 / 2: the four MSMs over z beside the QAP map, H behind it (gated / not)"""
    _check(load().zkhip_set_prove_split(int(mode)))


def last_prove_split():
    """True: this thread's last plain proof ran as two launch sequences (the four MSMs over z beside the QAP map, H behind it)"""
    return bool(load().zkhip_last_prove_split())


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


def last_accumulate_entries():
    """Mixed additions of the k_accumulate launch zkhip_last_accumulate_ms timed (the non-zero digits it sorted)."""
    out = ctypes.c_uint64(0)
    _check(load().zkhip_last_accumulate_entries(ctypes.byref(out)))
    return int(out.value)


def field_selftest(field, limbs_in):
    """Test hook: the device's multiplier bodies on raw limbs.  field 0 = Fq (27 limbs), 1 = Fr (14); limbs_in: uint32 [n][4][NL] =
    cases of (a, b, c, d); returns uint32 [n][3][NL] = (a b / R, a^2 / R, (a b + c d) / R)."""
    nl = 27 if field == 0 else 14
    x = np.ascontiguousarray(limbs_in, dtype=np.uint32).reshape(-1, 4, nl)
    out = np.zeros((x.shape[0], 3, nl), dtype=np.uint32)
    lib = load()
    lib.zkhip_internal_field_selftest.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    _check(lib.zkhip_internal_field_selftest(int(field), x.ctypes.data, x.shape[0], out.ctypes.data))
    return out


def measure_fq_mul_rate():
    """Fq multiplications per second of the device, measured now (dependent fp_mul chains, two waves per SIMD)."""
    v = ctypes.c_double(0)
    _check(load().zkhip_measure_fq_mul_rate(ctypes.byref(v)))
    return float(v.value)


def measure_ntt(log_d, inverse=False, coset=False, batch=1, reps=10):
    """ms per size-2^log_d transform, the pass kernels alone (zkhip_measure_ntt: HIP events on the passes' stream)."""
    out = ctypes.c_double(0)
    lib = load()
    lib.zkhip_measure_ntt.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    _check(lib.zkhip_measure_ntt(log_d, int(inverse), int(coset), batch, reps, ctypes.byref(out)))
    return out.value


def reset_time_base():
    """Record the origin of the accumulate intervals again (float milliseconds lose resolution far from their origin)."""
    _check(load().zkhip_reset_time_base())


def last_accumulate_interval():
    """(begin, end) of the last collected MSM's accumulation launch, ms on the device's time base."""
    t = (ctypes.c_float * 2)()
    _check(load().zkhip_last_accumulate_interval(t))
    return float(t[0]), float(t[1])
