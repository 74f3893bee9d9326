"""The keypair container (zkhip_keypair_write / zkhip_keypair_read: the role of wsnark::keypair_{write,read}_bytes,
aggregator_server.cpp:77-94).  Host code: a file assembled by hand here is read back limb for limb; truncation, a flipped
bit and a wrong magic are refused.  The write -> read -> prove round trip is in tests/test_aggregator_gpu.py."""
import struct

import numpy as np
import pytest

from zecale_amd import zkhip


def _fnv1a(data, h=0xcbf29ce484222325):
    for b in data:
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def _file_bytes(m, l, d, seed=3):
    rng = np.random.default_rng(seed)
    sizes = [1, 1, 1, 1, 1, m, m, m, d - 1, m - l - 1, l + 1]
    parts = [rng.integers(0, 1 << 63, size=(k, 24), dtype=np.uint64) for k in sizes]
    body = b"ZKHIPKP1" + struct.pack("<7Q", m, l, d, 0, 0, 0, 0) + b"".join(p.tobytes() for p in parts)
    return body + struct.pack("<Q", _fnv1a(body)), parts


def test_read_hand_made_file(tmp_path):
    data, parts = _file_bytes(m=9, l=2, d=8)
    path = tmp_path / "kp.bin"
    path.write_bytes(data)
    kp = zkhip.Keypair.read(path)
    vk = kp.vk()
    assert (vk["alpha"] == parts[0][0]).all() and (vk["beta"] == parts[2][0]).all() and (vk["delta"] == parts[4][0]).all()
    assert (vk["ABC"] == parts[10]).all()
    out = tmp_path / "kp2.bin"
    kp.write(out)
    assert out.read_bytes() == data                 # write is the exact inverse of read
    kp.free()


@pytest.mark.parametrize("damage", ["truncate", "flip", "magic", "sizes"])
def test_damaged_files_are_refused(tmp_path, damage):
    data, _ = _file_bytes(m=9, l=2, d=8)
    b = bytearray(data)
    if damage == "truncate":
        b = b[:-200]
    elif damage == "flip":
        b[300] ^= 1
    elif damage == "magic":
        b[0] = ord("X")
    else:
        b[8:16] = struct.pack("<Q", 1)              # n_vars < n_primary + 1
    path = tmp_path / "bad.bin"
    path.write_bytes(bytes(b))
    with pytest.raises(zkhip.ZkhipError):
        zkhip.Keypair.read(path)
    with pytest.raises(zkhip.ZkhipError):
        zkhip.Keypair.read(tmp_path / "missing.bin")
