"""Witness generation on the GPU (zecale_amd/csrc/witness.hip, witness_tape.cpp) against the host generator: the same assignment,
limb for limb, for the cases of libzecale/tests/aggregator/aggregator_dummy_test.cpp (valid batch, invalid nested proof) and the
other circuit shapes (batches of 1 and 3, nine inputs per nested proof); the assignment satisfies every constraint (GPU SpMV
check), a proof from the device-resident assignment equals the proof from the host one."""
import time

import numpy as np
import pytest

from tests.helpers import fr_int, fr_limbs
from tests.test_aggregator_host import nested_proof_limbs, nested_vk_limbs
from tests.test_oracle_pins import load_nested_fixtures

pytestmark = pytest.mark.gpu


def _batch(num_proofs, k, bump_last=False, first=0):
    nvk, proofs = load_nested_fixtures()
    if k > 1:
        nvk = dict(nvk)
        nvk["ABC"] = list(nvk["ABC"]) + [proofs[i][0]["a"] for i in range(6)] + [proofs[0][0]["c"], proofs[1][0]["c"]]
        nvk["ABC"] = nvk["ABC"][:k + 1]
    chosen = [proofs[(first + i) % 6] for i in range(num_proofs)]
    xs = [[inp[0] + j * 1000 for j in range(k)] for _, inp in chosen]
    if bump_last:
        xs[-1][0] += 1
    return (nested_vk_limbs(nvk), np.concatenate([nested_proof_limbs(p) for p, _ in chosen]),
            np.array([fr_limbs(x) for row in xs for x in row]))


@pytest.mark.parametrize("num_proofs,k,bump", [(2, 1, False), (2, 1, True), (1, 1, False), (3, 1, True), (2, 9, False)])
def test_gpu_witness_equals_host_witness(zk, num_proofs, k, bump):
    agg = zk.AggregatorCircuit(num_proofs, k)
    vk, pr, inp = _batch(num_proofs, k, bump)
    z_host = agg.witness(vk, pr, inp)
    t = time.time()
    z_gpu = agg.witness_gpu(vk, pr, inp)
    dt = time.time() - t
    st = agg.gpu_witness_stats()
    print("GPU witness:", st, "first call %.1f ms" % (dt * 1e3))
    assert st["multiplications"] > 10000 and st["inversions"] > 20      # (54 since the G2 line denominators come from one Jacobian run per point)
    assert (z_gpu == z_host).all(), "first difference at variable %d" % int(np.nonzero((z_gpu != z_host).any(axis=1))[0][0])
    r1 = zk.r1cs_from_desc(zk.r1cs_desc_from_aggregator(agg))
    assert r1.is_satisfied(z_gpu)
    t = time.time()
    z2 = agg.witness_gpu(vk, pr, inp)
    print("second call %.1f ms" % ((time.time() - t) * 1e3))
    assert (z2 == z_host).all()
    r1.free(); agg.free()


def test_gpu_witness_other_batches_and_degenerate_input(zk):
    agg = zk.AggregatorCircuit(2, 1)
    for first in (1, 2, 4):
        vk, pr, inp = _batch(2, 1, first=first)
        assert (agg.witness_gpu(vk, pr, inp) == agg.witness(vk, pr, inp)).all()
    # the same nested proof twice (equal G2 points walk the same lines: still generic for the program)
    vk, pr, inp = _batch(2, 1)
    pr2 = np.concatenate([pr[:48], pr[:48]]); inp2 = np.concatenate([inp[:1], inp[:1]])
    assert (agg.witness_gpu(vk, pr2, inp2) == agg.witness(vk, pr2, inp2)).all()
    # a degenerate input (ABC_1 = ABC_0: the input accumulator adds a point to itself, the slope's denominator is zero): the host
    # generator branches there, the recorded program cannot - the device notices the inversion of zero and refuses
    vk_bad = vk.copy(); vk_bad[72:84] = vk_bad[60:72]
    with pytest.raises(zk.ZkhipError):
        agg.witness_gpu(vk_bad, pr, inp)
    agg.free()


def test_pipeline_with_gpu_witness_matches_serial_path(zk):
    """The streaming prover with ZKHIP_PIPELINE_GPU_WITNESS: the assignment is generated on the device and never leaves it; every
    extended proof equals host witness + zkhip_groth16_prove; a degenerate batch falls back to the host generator."""
    from tests.test_aggregator_gpu import _setup
    agg, desc, kp, nvk_l, proofs = _setup(zk)
    vk = kp.vk()
    crs, r1 = kp.upload_crs(), zk.r1cs_from_desc(desc)
    pipe = zk.AggregatorPipeline(agg, crs, gpu_slots=2, witness_workers=6, gpu_witness=True)
    jobs = []
    for i, (a, b, bump) in enumerate(((0, 1, 0), (2, 3, 0), (4, 5, 1), (1, 2, 0), (3, 3, 0), (5, 0, 0), (0, 1, 0), (2, 4, 1))):
        (pa, ia), (pb, ib) = proofs[a], proofs[b]
        npr = np.concatenate([nested_proof_limbs(pa), nested_proof_limbs(pb)])
        nin = np.array([fr_limbs(ia[0]), fr_limbs(ib[0] + bump)])
        r, s = fr_limbs(0xaaaa + i), fr_limbs(0xbbbb + 7 * i)
        jobs.append((nvk_l, npr, nin, r, s, 1 if bump else 3, pipe.submit(nvk_l, npr, nin, r, s)))
    # a degenerate key (ABC_1 = ABC_0): the device refuses, the worker falls back to the host generator, the batch is proved all the same
    vk_deg = nvk_l.copy(); vk_deg[72:84] = vk_deg[60:72]
    (pa, ia), (pb, ib) = proofs[0], proofs[1]
    npr = np.concatenate([nested_proof_limbs(pa), nested_proof_limbs(pb)])
    nin = np.array([fr_limbs(ia[0]), fr_limbs(ib[0])])
    jobs.append((vk_deg, npr, nin, fr_limbs(5), fr_limbs(6), 0, pipe.submit(vk_deg, npr, nin, fr_limbs(5), fr_limbs(6))))
    for nvk_j, npr, nin, r, s, bits, ticket in reversed(jobs):
        prim, proof = pipe.wait(ticket)
        z = agg.witness(nvk_j, npr, nin)
        assert (prim == z[1:1 + agg.num_primary_inputs()]).all()
        assert (proof == zk.groth16_prove(crs, r1, z, r, s)).all()
        assert fr_int(prim[1]) == bits
        assert zk.groth16_verify(vk, prim, proof)
    pipe.free()
    crs.free(); r1.free(); kp.free(); agg.free()
