"""Round 6: the ACCEPT branch of the in-circuit verifier for nested proofs with MORE THAN ONE input, on every
generator and prover path of the product.  The reference's slow test aggregates two valid nine-input Zeth proofs and asserts
verify == true with result bits {1,1} (libzecale/tests/aggregator/aggregator_test.cpp:222-254,293-314); no Zeth proof is in the
tree, so the nested statements come from a known trapdoor (tests/golden/nested_k{3,9}.json, written by tests/golden/gen_golden.py,
pinned by the oracle's verifier in tests/test_oracle_pins.py::test_nested_statements_from_a_trapdoor).  Until this round every
nine-input run in the tree used a key padded with unrelated points and had only ever produced result bits 0: a defect in the input
accumulator acc = ABC_0 + sum x_j ABC_j for j >= 2 (253-bit scalars; the doubling chains the application handle folds into its
constants) that turned a valid proof's bit into 0 would have passed everything."""
import numpy as np
import pytest

from oracle import pyref as R
from tests.helpers import fr_int, fr_limbs, random_fr_uniform
from tests.test_aggregator_host import nested_proof_limbs, nested_vk_limbs
from tests.test_oracle_pins import load_nested_statement

pytestmark = pytest.mark.gpu

TRAPDOOR = (0x1234567, 0x2345678, 0x3456789, 0x456789a)


def _cases(k):
    """(bumped (proof, input) pairs, result bits): accept / accept, and each reject pattern at the first, a middle, the last input."""
    out = [((), 3)]
    for j in sorted({0, k // 2, k - 1}):
        out += [(((0, j),), 2), (((1, j),), 1), (((0, j), (1, j)), 0)]
    return out


def _inputs(valid, bumps, which=(0, 1)):
    xs = [list(valid[w][1]) for w in which]
    for p, j in bumps:
        xs[p][j] = (xs[p][j] + 1) % R.BLS_R
    return xs, np.array([fr_limbs(x) for row in xs for x in row])


@pytest.mark.parametrize("k", [3, 9])
def test_every_generator_accepts_valid_proofs_and_rejects_bumped_inputs(zk, k):
    """Host generator, GPU generator, the application's masked host generator and its own GPU program: the same assignment limb for
    limb, satisfied on the GPU (SpMV check), result bits 3 / 2 / 1 / 0 as the bumps say."""
    agg = zk.AggregatorCircuit(2, k)
    desc = zk.r1cs_desc_from_aggregator(agg)
    kp = zk.Keypair(desc, *(fr_limbs(t) for t in TRAPDOOR))
    crs, r1 = kp.upload_crs(zk.key_opts(table_naf=True)), zk.r1cs_from_desc(desc)
    nvk, valid = load_nested_statement(k)
    nvk_l = nested_vk_limbs(nvk)
    app = zk.AggregatorApp(agg, crs, nvk_l)
    npr = np.concatenate([nested_proof_limbs(valid[0][0]), nested_proof_limbs(valid[1][0])])
    l = agg.num_primary_inputs()
    cases = _cases(k)
    batches = []
    for bumps, bits in cases:
        xs, nin = _inputs(valid, bumps)
        z = agg.witness(nvk_l, npr, nin)
        assert fr_int(z[2]) == bits, (bumps, fr_int(z[2]))
        assert r1.is_satisfied(z)
        assert fr_int(z[1]) == R.nested_vk_hash(nvk) and [fr_int(x) for x in z[3:1 + l]] == xs[0] + xs[1]
        zg = agg.witness_gpu(nvk_l, npr, nin)
        assert (zg == z).all(), "GPU generator differs at variable %d" % int(np.nonzero((zg != z).any(axis=1))[0][0])
        zm = app.witness(npr, nin)
        assert (zm == app.mask(z)).all()
        batches.append((npr, nin, zm, bits))
    zs, prims = app.witness_gpu([(b[0], b[1]) for b in batches])          # all the cases in ONE launch of the application's program
    for (npr_, nin, zm, bits), zd, pi in zip(batches, zs, prims):
        assert zd is not None and (zd == zm).all()
        assert fr_int(pi[1]) == bits and (pi == zm[1:1 + l]).all()
    # the third proof, and a proof paired with another proof's inputs
    xs, nin = _inputs(valid, (), which=(2, 0))
    npr2 = np.concatenate([nested_proof_limbs(valid[2][0]), nested_proof_limbs(valid[0][0])])
    assert fr_int(agg.witness_gpu(nvk_l, npr2, nin)[2]) == 3
    xs, nin = _inputs(valid, (), which=(1, 0))
    assert fr_int(agg.witness_gpu(nvk_l, npr2, nin)[2]) == 2                # proof 2 with proof 1's inputs rejected, proof 0 accepted
    app.free(); crs.free(); r1.free(); kp.free(); agg.free()


@pytest.mark.parametrize("gpu_witness,hybrid", [(False, False), (True, False), (True, True)], ids=["host-witness", "gpu-witness", "hybrid"])
def test_streaming_prover_on_valid_nine_input_proofs(zk, gpu_witness, hybrid):
    """zkhip_aggregator_pipeline with the application registered (RegisterApplication, aggregator_server.cpp:170-235) and with the
    cache off: every wrapping proof equals the plain serial proof of the full host assignment, verifies, and carries the expected
    result bits - 3 for the valid batch."""
    k = 9
    agg = zk.AggregatorCircuit(2, k)
    desc = zk.r1cs_desc_from_aggregator(agg)
    kp = zk.Keypair(desc, *(fr_limbs(t) for t in TRAPDOOR))
    vk = kp.vk()
    crs, r1 = kp.upload_crs(zk.key_opts(table_naf=True)), zk.r1cs_from_desc(desc)
    nvk, valid = load_nested_statement(k)
    nvk_l = nested_vk_limbs(nvk)
    npr = np.concatenate([nested_proof_limbs(valid[0][0]), nested_proof_limbs(valid[1][0])])
    jobs = []
    for i, (bumps, bits) in enumerate(_cases(k) + [((), 3)] * 3):
        xs, nin = _inputs(valid, bumps)
        jobs.append((nin, fr_limbs(0x5000 + i), fr_limbs(0x6000 + 5 * i), bits))
    results = {}
    for cache in (True, False):
        pipe = zk.AggregatorPipeline(agg, crs, gpu_slots=3, witness_workers=(6 if gpu_witness else 3), gpu_witness=gpu_witness, hybrid=hybrid,
                                     app_cache=cache)
        if cache:
            pipe.register_app(nvk_l)
        tickets = [pipe.submit(nvk_l, npr, nin, r, s) for nin, r, s, _ in jobs]
        results[cache] = [pipe.wait(t) for t in tickets]
        assert (pipe.app_hits() == len(jobs)) if cache else (pipe.app_hits() == 0)
        pipe.free()
    for i, (nin, r, s, bits) in enumerate(jobs):
        z = agg.witness(nvk_l, npr, nin)
        plain = zk.groth16_prove(crs, r1, z, r, s) if i < 4 or i % 3 == 0 else None
        for prim, proof in (results[True][i], results[False][i]):
            assert (prim == z[1:1 + agg.num_primary_inputs()]).all() and fr_int(prim[1]) == bits
            assert plain is None or (proof == plain).all()
        assert (results[True][i][1] == results[False][i][1]).all()
        assert zk.groth16_verify(vk, results[True][i][0], results[True][i][1])
    crs.free(); r1.free(); kp.free(); agg.free()


@pytest.mark.parametrize("k", [3, 9])
def test_wrapping_proof_of_valid_multi_input_proofs_equals_oracle(zk, oracle_lib, k):
    """tests/test_aggregator_gpu.py::test_wrapping_proof_equals_oracle for k > 1: the C restatement of r1cs_to_qap_witness_map +
    r1cs_gg_ppzksnark_prover proves the assignment of the VALID batch (result bits 3) once; the product's proof from the host
    assignment, from the GPU generator's, from the application's masked assignment (host and device program) and from an instance
    must equal it limb for limb, and verify."""
    O = oracle_lib
    agg = zk.AggregatorCircuit(2, k)
    desc = zk.r1cs_desc_from_aggregator(agg)
    kp = zk.Keypair(desc, *(fr_limbs(t) for t in TRAPDOOR))
    crs, r1 = kp.upload_crs(zk.key_opts(table_naf=True)), zk.r1cs_from_desc(desc)
    nvk, valid = load_nested_statement(k)
    nvk_l = nested_vk_limbs(nvk)
    npr = np.concatenate([nested_proof_limbs(valid[0][0]), nested_proof_limbs(valid[1][0])])
    l = agg.num_primary_inputs()
    pk, m, l_pk, dom = kp.pk_arrays()
    assert l_pk == l and dom == O.qap_domain_size(agg.num_constraints, l, None)
    A, B, C = agg.get_constraint_system()
    rs = random_fr_uniform(4242 + k, 2)
    _, nin = _inputs(valid, ())
    z = agg.witness(nvk_l, npr, nin)
    assert fr_int(z[2]) == 3 and O.r1cs_first_unsatisfied(A, B, C, z) == -1
    h = O.qap_h(A, B, C, z, agg.num_constraints, l, dom)
    assert (r1.qap_h(z) == h).all()
    expect = O.groth16_prove(pk, z, l, h, rs[0], rs[1])
    assert zk.groth16_verify(kp.vk(), z[1:1 + l], expect)
    assert (zk.groth16_prove(crs, r1, z, rs[0], rs[1]) == expect).all()
    assert (zk.groth16_prove(crs, r1, agg.witness_gpu(nvk_l, npr, nin), rs[0], rs[1]) == expect).all()
    app = zk.AggregatorApp(agg, crs, nvk_l)
    assert (app.prove(r1, app.witness(npr, nin), rs[0], rs[1]) == expect).all()
    (zd,), _ = app.witness_gpu([(npr, nin)])
    pr = zk.Prover(crs, desc)
    assert (pr.prove_app(app, zd, rs[0], rs[1]) == expect).all()
    assert (pr.prove(z, rs[0], rs[1]) == expect).all()
    pr.free(); app.free(); crs.free(); r1.free(); kp.free(); agg.free()
