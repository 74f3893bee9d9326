"""The C-ABI library loads and exports every symbol include/zkhip.h declares (no compute calls:
this test runs without a GPU).  Also: the product refuses to run without a device."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "zkhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(zkhip_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from zecale_amd import zkhip
    lib = ctypes.CDLL(zkhip.LIB_PATH)
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/zkhip.h but not exported"
    assert sorted(zkhip.EXPORTS) == names, "zecale_amd/zkhip.py EXPORTS out of sync with include/zkhip.h"


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        return  # meaningful only on the CPU-only container
    from zecale_amd import zkhip
    lib = zkhip.load()
    rc = lib.zkhip_init(0)
    assert rc == -2, "zkhip_init must fail with ZKHIP_ERR_NO_DEVICE when there is no GPU"
    import numpy as np
    import pytest
    with pytest.raises(zkhip.ZkhipError):
        zkhip.msm_raw(np.zeros((1, 24), dtype=np.uint64), np.zeros((1, 6), dtype=np.uint64))


def test_prover_tail_abandoned_proofs_and_small_order_delta():
    """ADVICE r5 (medium + low), host only: (1) a proof that fails between the start of the tail's key-only scalar multiplications and
    their collection must not leave tasks behind that read the caller's stack or the key's tables - the TailPre waits for its tasks
    (the futures of a thread pool do not, unlike std::async's), and the tasks hold their scalars by value; 30 abandoned tails with
    the tables freed straight after (tools/sanitize/asan_host_tests.sh runs this file under the CPU AddressSanitizer build);
    (2) a delta of small order - (1, 0) has order 2 on y^2 = x^3 - 1 - gets no fixed-base table (one inversion for all multiples
    would meet a zero) and the tail's variable-base fallback gives g1 + r delta_1."""
    import ctypes
    import numpy as np
    from oracle import pyref as R
    from tests.helpers import aff_limbs
    from zecale_amd import zkhip
    lib = zkhip.load()
    lib.zkhip_internal_tail_selftest.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int]
    small = (1, 0)
    assert R.on_curve(small, R.G1_B) and R.ec_add(small, small) is None
    g1, g2, sm = (np.ascontiguousarray(aff_limbs(P)) for P in (R.G1_GEN, R.G2_GEN, small))
    rc = lib.zkhip_internal_tail_selftest(g1.ctypes.data, g2.ctypes.data, sm.ctypes.data, 30)
    assert rc == 0, lib.zkhip_last_error().decode()
