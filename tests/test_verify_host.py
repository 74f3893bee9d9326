"""The product's host-side Groth16/BW6-761 verifier (zkhip_groth16_verify: host field, group law and Tate
pairing of zecale_amd/csrc/host_field.hpp + pairing_host.hpp) against the reference's own known-answer
fixtures: batch1.json verifies, batch1-invalid.json does not, under aggregator_vk.json
(client/test_commands/test_bw6_761_groth16_contract.py:66-79).  CPU only: the verifier needs no device."""
import time

import numpy as np

from tests.helpers import aff_limbs, fr_limbs, golden, h2i, pt_from_json


def _vk(j):
    return dict(alpha=aff_limbs(pt_from_json(j["alpha"])), beta=aff_limbs(pt_from_json(j["beta"])),
                delta=aff_limbs(pt_from_json(j["delta"])), ABC=np.array([aff_limbs(pt_from_json(p)) for p in j["ABC"]]))


def _proof(j):
    return np.concatenate([aff_limbs(pt_from_json(j["a"])), aff_limbs(pt_from_json(j["b"])), aff_limbs(pt_from_json(j["c"]))])


def test_reference_kat_valid_and_invalid():
    from zecale_amd import zkhip
    vk = _vk(golden("dummy_app/aggregator_vk.json"))
    for name, expect in (("batch1.json", True), ("batch1-invalid.json", False)):
        ep = golden("dummy_app/" + name)["ext_proof"]
        inputs = np.array([fr_limbs(h2i(x)) for x in ep["inputs"]])
        t = time.time()
        assert zkhip.groth16_verify(vk, inputs, _proof(ep["proof"])) is expect, name
        print(name, "verify took %.3f s" % (time.time() - t))


def test_golden_small_proof_verifies_and_tampering_fails():
    from zecale_amd import zkhip
    g = golden("groth16_small.json")
    vk = _vk(g["vk"])
    z = [h2i(x) for x in g["z"]]
    inputs = np.array([fr_limbs(x) for x in z[1:1 + g["n_primary"]]])
    proof = _proof(g["proof"])
    assert zkhip.groth16_verify(vk, inputs, proof)
    bad = inputs.copy(); bad[0] = fr_limbs(z[1] + 1)
    assert not zkhip.groth16_verify(vk, bad, proof)
    badp = proof.copy(); badp[48:] = proof[:24]           # C := A
    assert not zkhip.groth16_verify(vk, inputs, badp)
