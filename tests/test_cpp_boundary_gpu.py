"""The C++ boundary executed on a device: tests/cpp/boundary_gpu.cpp drives include/aggregator_circuit_hip.hpp
like aggregator_server.cpp drives the reference class (trusted setup -> prove -> verify, then the stream), tests/cpp/multi_device.cpp
drives two prover contexts of one process through the C ABI.  The programs print JSON in the reference's encodings; this side parses
it with zecale_amd.encoding and re-verifies with the host pairing verifier."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import pyref as R
from tests.helpers import fr_int, fr_limbs
from tests.test_aggregator_host import nested_proof_limbs, nested_vk_limbs
from tests.test_oracle_pins import load_nested_fixtures

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, name):
    exe = tmp_path / name
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", name + ".cpp"),
                           "-o", str(exe), "-L", os.path.join(ROOT, "zecale_amd"), "-lzkhip", "-Wl,-rpath," + os.path.join(ROOT, "zecale_amd")])
    return exe


def _fixture_file(tmp_path):
    nvk, proofs = load_nested_fixtures()
    (p1, in1), (p2, in2) = proofs[0], proofs[1]
    blob = np.concatenate([nested_vk_limbs(nvk), nested_proof_limbs(p1), nested_proof_limbs(p2), fr_limbs(in1[0]), fr_limbs(in2[0])]).astype(np.uint64)
    path = tmp_path / "in.bin"
    blob.tofile(path)
    return path, nvk, (in1[0], in2[0])


def test_cpp_mirror_setup_prove_verify_and_stream(zk, tmp_path):
    from zecale_amd import encoding as E
    exe = _build(tmp_path, "boundary_gpu")
    path, nvk, (x1, x2) = _fixture_file(tmp_path)
    out = subprocess.run([str(exe), str(path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = dict((ln.split(" ", 1) + [""])[:2] for ln in out.stdout.splitlines() if ln and not ln.startswith("STREAM"))
    assert lines["ABC"] == "5 PRIMARY 4"                       # vk.ABC.size() == num_primary_inputs() + 1 (aggregator_server.cpp:490)
    assert lines["VERIFY"] == "1" and lines["REPROVE_DIFFERS"] == "1 VERIFY 1" and lines["TAMPERED_VERIFY"] == "0"
    assert lines["THROW"] == "attempt to aggregate proof with invalid number of inputs"     # aggregator_circuit.tcc:138-141
    assert lines["OFFCURVE"].startswith("nested proof or verification key has a point that is not on its curve")
    assert lines["GPUWITNESS"] == "VERIFY 1 INPUTS_EQUAL 1"
    # the reference's domain is the default, and the key decides: generate_trusted_setup gives the forced power of
    # two; a 65,536-point key handed over as raw arrays proves; a step-domain key (49,152) proves on the same
    # circuit object, and the first key again after it; a key whose domain cannot hold the 44,188 points is refused
    assert lines["DOMAIN"] == "65536"
    assert lines["IMPORTED_KEY"] == "domain=65536 key_domain=65536 VERIFY 1"
    assert lines["SMALL_DOMAIN"] == "refused"
    assert lines["STEP_KEY"] == "domain=49152 VERIFY 1 CROSS 0 BACK 1"
    assert "DONE" in out.stdout
    # the printed JSON is the reference's encoding: decode it here and verify with the host pairing check
    vk = E.verification_key_from_json(json.loads(lines["VK"]))
    proof, inputs = E.extended_proof_from_json(json.loads(lines["PROOF"]))
    assert zk.groth16_verify(vk, inputs, proof)
    want = [R.nested_vk_hash(nvk), 3, x1, x2]
    assert [fr_int(x) for x in inputs] == want
    streams = [ln.split(" ", 4) for ln in out.stdout.splitlines() if ln.startswith("STREAM")]
    assert len(streams) == 3 and all(s[3] == "1" for s in streams)
    for s in streams:
        proof, inputs = E.extended_proof_from_json(json.loads(s[4]))
        assert zk.groth16_verify(vk, inputs, proof)
        idx = int(s[1])
        assert [fr_int(x) for x in inputs] == [want[0], 3] + ([x2, x1] if idx == 1 else [x1, x2])


def test_two_prover_contexts_in_one_process(tmp_path):
    """tests/cpp/multi_device.cpp through the C++ adapter: hip_proving_key over a device list (two contexts on GPU 0:
    the wrapping key partitioned, zkhip_multi_prover) returns the whole-key proof limb for limb, and aggregator_circuit::open_node_stream
    over the same list (replicas behind zkhip_dispatcher) proves and verifies eight batches on both entries."""
    exe = _build(tmp_path, "multi_device")
    path, _, _ = _fixture_file(tmp_path)
    out = subprocess.run([str(exe), str(path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert any(ln.startswith("PARTITIONED devices=2") and ln.endswith("same_as_whole_key=1") for ln in lines), out.stdout
    assert "PARTITIONED verifies=1" in lines and "DONE" in lines, out.stdout
    rep = [ln for ln in lines if ln.startswith("REPLICAS")]
    assert rep and "entries=2 verified=8 of 8" in rep[0], out.stdout
    per = [int(x) for x in rep[0].rsplit("=", 1)[1].split(",")]
    assert sum(per) == 8 and min(per) >= 1, out.stdout
