"""End to end on the GPU, as libzecale/tests/aggregator/aggregator_dummy_test.cpp does for BLS12-377 -> BW6-761
Groth16 (:201-211): aggregator circuit -> trusted setup -> witness from two nested dummy-app proofs (the
reference's own fixtures, a = 7 and 8) -> wrapping proof on the MI355X -> wsnark::verify == true (:61-62),
primary input[0] == compute_hash(vk) (:70-73), input[1] == packed result bits (:77-84; {1,0} when the second
nested input is bumped, :162-186), inputs[2..] == nested inputs (:87-96)."""
import numpy as np
import pytest

from oracle import pyref as R
from tests.helpers import fr_int, fr_limbs
from tests.test_aggregator_host import nested_proof_limbs, nested_vk_limbs
from tests.test_oracle_pins import load_nested_fixtures

pytestmark = pytest.mark.gpu


def test_aggregate_two_dummy_app_proofs(zk, oracle_lib):
    agg = zk.AggregatorCircuit(2, 1)
    desc = zk.r1cs_desc_from_aggregator(agg)
    kp = zk.Keypair(desc, fr_limbs(0x1234567), fr_limbs(0x2345678), fr_limbs(0x3456789), fr_limbs(0x456789a))
    vk = kp.vk()
    assert vk["ABC"].shape[0] == agg.num_primary_inputs() + 1       # aggregator_server.cpp:490 sanity check
    crs = kp.upload_crs()
    r1 = zk.r1cs_from_desc(desc)
    nvk, proofs = load_nested_fixtures()
    nvk_l = nested_vk_limbs(nvk)
    (p1, in1), (p2, in2) = proofs[0], proofs[1]
    for bump, expected_bits in ((0, 3), (1, 1)):
        x1, x2 = in1[0], in2[0] + bump
        z = agg.witness(nvk_l, np.concatenate([nested_proof_limbs(p1), nested_proof_limbs(p2)]), np.array([fr_limbs(x1), fr_limbs(x2)]))
        assert r1.is_satisfied(z)
        proof = zk.groth16_prove(crs, r1, z, fr_limbs(0xabcdef), fr_limbs(0xfedcba))
        primary = z[1:1 + agg.num_primary_inputs()]
        assert zk.groth16_verify(vk, primary, proof)                            # wsnark::verify(...) == true
        assert (primary[0] == zk.aggregator_vk_hash(nvk_l, 1)).all()
        assert fr_int(primary[1]) == expected_bits
        assert [fr_int(primary[2]), fr_int(primary[3])] == [x1, x2]
        tampered = primary.copy(); tampered[1] = fr_limbs(expected_bits ^ 2)      # claiming the other result must fail
        assert not zk.groth16_verify(vk, tampered, proof)
        print("wrapping proof ok, result bits", expected_bits, zk.last_prove_timings())
    crs.free(); r1.free(); kp.free(); agg.free()
