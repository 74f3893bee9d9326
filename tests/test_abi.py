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
