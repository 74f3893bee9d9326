import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _host_threads():
    """Cores this process may really use: the GPU box caps a one-GPU job with a cgroup quota (16 of 256 hardware threads); an OpenMP
    team of 256 on 16 cores makes the oracle's multi_exp several times slower."""
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else max(1, int(int(q) / int(per)))
    except Exception:
        pass
    avail = len(os.sched_getaffinity(0))
    return min(avail, quota) if quota else avail


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as O
    O.load()
    O.set_threads(_host_threads())
    return O


@pytest.fixture(scope="session")
def zk():
    """The product library through its C ABI; initialised on device 0.  Fails loudly without a GPU."""
    from zecale_amd import zkhip
    zkhip.init(0)
    return zkhip
