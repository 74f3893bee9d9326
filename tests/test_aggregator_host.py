"""The wrapping (aggregator) circuit on the host: structure, witness, public inputs.  Mirrors the checks of the
reference's libzecale/tests/aggregator/aggregator_dummy_test.cpp:64-96 that do not need a proof:
  input[0] == compute_hash(vk), input[1] == packed result bits ({1,1} valid / {1,0} when the 2nd nested input is
  bumped, :162-186), inputs[2..] == nested inputs; plus: the assignment satisfies every constraint (CPU oracle), and
  verification_key_hash_gadget_test.cpp:20-47 (hash non-zero, differs between keys).
Nested proofs and key are the reference's own fixtures (testdata/dummy_app/vk.json, extproof1..6.json).  CPU only."""
import numpy as np
import pytest

from oracle import pyref as R
from tests.helpers import fr_int, fr_limbs
from tests.test_oracle_pins import load_nested_fixtures, load_nested_statement


def _fl(x):
    return R.int_to_limbs(R.to_mont(x % R.BLS_Q, R.BLS_Q, 6), 6)


def g1_limbs(p):
    return _fl(p[0]) + _fl(p[1])


def g2_limbs(p):
    return _fl(p[0][0]) + _fl(p[0][1]) + _fl(p[1][0]) + _fl(p[1][1])


def nested_vk_limbs(nvk):
    return np.array(g1_limbs(nvk["alpha"]) + g2_limbs(nvk["beta"]) + g2_limbs(nvk["delta"]) + sum((g1_limbs(p) for p in nvk["ABC"]), []),
                    dtype=np.uint64)


def nested_proof_limbs(pr):
    return np.array(g1_limbs(pr["a"]) + g2_limbs(pr["b"]) + g1_limbs(pr["c"]), dtype=np.uint64)


@pytest.fixture(scope="module")
def circuit():
    from zecale_amd import zkhip
    return zkhip.AggregatorCircuit(2, 1)


def test_native_nested_verifier_on_reference_fixtures():
    from zecale_amd import zkhip
    nvk, proofs = load_nested_fixtures()
    vk = nested_vk_limbs(nvk)
    for pr, inputs in proofs:
        assert zkhip.bls12_377_groth16_verify(vk, np.array([fr_limbs(inputs[0])]), nested_proof_limbs(pr))
    pr, inputs = proofs[2]
    assert not zkhip.bls12_377_groth16_verify(vk, np.array([fr_limbs(inputs[0] + 1)]), nested_proof_limbs(pr))


@pytest.mark.parametrize("k", [3, 9])
def test_native_nested_verifier_with_several_inputs(k):
    """zkhip_bls12_377_groth16_verify on VALID statements with more than one input (tests/golden/nested_k{3,9}.json, pinned by
    tests/test_oracle_pins.py::test_nested_statements_from_a_trapdoor): all three proofs accepted, every single bumped input and
    a foreign proof's inputs rejected - the same decisions as the oracle's verifier, position by position."""
    from zecale_amd import zkhip
    nvk, proofs = load_nested_statement(k)
    vk = nested_vk_limbs(nvk)
    ins = lambda xs: np.array([fr_limbs(x) for x in xs])
    for pr, xs in proofs:
        assert zkhip.bls12_377_groth16_verify(vk, ins(xs), nested_proof_limbs(pr))
        for j in range(k):
            bad = list(xs); bad[j] = (bad[j] + 1) % R.BLS_R
            assert not zkhip.bls12_377_groth16_verify(vk, ins(bad), nested_proof_limbs(pr)), j
    assert not zkhip.bls12_377_groth16_verify(vk, ins(proofs[1][1]), nested_proof_limbs(proofs[0][0]))


def test_circuit_shape(circuit):
    assert circuit.num_primary_inputs() == 4          # 1 + 1 + 2 * 1 (aggregator_circuit.tcc:172-180)
    assert 20000 < circuit.num_constraints < 200000
    A, B, C = circuit.get_constraint_system()
    assert len(A[0]) == circuit.num_constraints + 1
    assert int(max(A[1].max(), B[1].max(), C[1].max())) < circuit.num_variables


@pytest.mark.parametrize("bump_second", [False, True])
def test_witness_and_public_inputs(circuit, oracle_lib, bump_second):
    from zecale_amd import zkhip
    nvk, proofs = load_nested_fixtures()
    vk = nested_vk_limbs(nvk)
    (p1, in1), (p2, in2) = proofs[0], proofs[1]       # a = 7, 8 (fees 12, 11: the first batch, scripts/test-client:67-70)
    x1, x2 = in1[0], in2[0] + (1 if bump_second else 0)
    z = circuit.witness(vk, np.concatenate([nested_proof_limbs(p1), nested_proof_limbs(p2)]), np.array([fr_limbs(x1), fr_limbs(x2)]))
    A, B, C = circuit.get_constraint_system()
    assert oracle_lib.r1cs_first_unsatisfied(A, B, C, z) == -1
    assert fr_int(z[0]) == 1
    assert (z[1] == zkhip.aggregator_vk_hash(vk, 1)).all() and fr_int(z[1]) != 0
    assert fr_int(z[2]) == (1 if bump_second else 3)     # results packed LSB-first: {1,1} -> 3, {1,0} -> 1
    assert fr_int(z[3]) == x1 and fr_int(z[4]) == x2
    # a corrupted assignment is rejected
    zb = z.copy(); zb[2] = fr_limbs(fr_int(z[2]) ^ 2)
    assert oracle_lib.r1cs_first_unsatisfied(A, B, C, zb) >= 0


def test_vk_hash_differs_between_keys():
    from zecale_amd import zkhip
    nvk, _ = load_nested_fixtures()
    h1 = zkhip.aggregator_vk_hash(nested_vk_limbs(nvk), 1)
    nvk2 = dict(nvk); nvk2["alpha"] = nvk["ABC"][0]
    h2 = zkhip.aggregator_vk_hash(nested_vk_limbs(nvk2), 1)
    assert fr_int(h1) != 0 and fr_int(h2) != 0 and fr_int(h1) != fr_int(h2)


@pytest.mark.parametrize("num_proofs", [1, 3])
def test_other_batch_sizes(oracle_lib, num_proofs):
    """NumProofs is a template parameter of the reference (aggregator_circuit.hpp:32-37; the server fixes it at 2): batches of 1 and 3
    nested proofs give 1 + 1 + NumProofs primary inputs, a satisfying witness, and result bits packed LSB-first (the last
    proof of the batch of 3 gets a bumped input: bits {1,1,0} = 3)."""
    from zecale_amd import zkhip
    c = zkhip.AggregatorCircuit(num_proofs, 1)
    assert c.num_primary_inputs() == 2 + num_proofs
    nvk, proofs = load_nested_fixtures()
    vk = nested_vk_limbs(nvk)
    chosen = [proofs[k] for k in range(num_proofs)]
    xs = [inp[0] for _, inp in chosen]
    if num_proofs == 3:
        xs[2] += 1
    z = c.witness(vk, np.concatenate([nested_proof_limbs(p) for p, _ in chosen]), np.array([fr_limbs(x) for x in xs]))
    A, B, C = c.get_constraint_system()
    assert oracle_lib.r1cs_first_unsatisfied(A, B, C, z) == -1
    assert (z[1] == zkhip.aggregator_vk_hash(vk, 1)).all()
    assert fr_int(z[2]) == (1 if num_proofs == 1 else 3)
    assert [fr_int(z[3 + k]) for k in range(num_proofs)] == xs
    c.free()


def _bumps(k):
    """(proof, input position) pairs to bump, and the result bits each leaves: the accept branch and all three reject patterns,
    at the first, a middle and the last input of the accumulator acc = ABC_0 + sum x_j ABC_j."""
    out = [((), 3)]
    for j in sorted({0, k // 2, k - 1}):
        out += [(((0, j),), 2), (((1, j),), 1), (((0, j), (1, j)), 0)]
    return out


@pytest.mark.parametrize("k", [3, 9])
def test_valid_nested_proofs_with_several_inputs(oracle_lib, k):
    """The reference's slow test (libzecale/tests/aggregator/aggregator_test.cpp:222-254,293-314): a batch of two VALID nine-input
    nested proofs gives result bits {1,1}; here with the trapdoor-built statements of tests/golden/nested_k{3,9}.json (no Zeth
    proof is in the tree).  The in-circuit verifier's ACCEPT branch for inputs 2..k: bits 3 for the valid batch, 2 / 1 / 0 when
    input j of proof 0 / 1 / both is bumped (j = first, middle, last) - and every assignment satisfies every constraint under
    the oracle, so the bit is what the constraints force, not what the generator chose."""
    from zecale_amd import zkhip
    c = zkhip.AggregatorCircuit(2, k)
    nvk, proofs = load_nested_statement(k)
    vk = nested_vk_limbs(nvk)
    npr = np.concatenate([nested_proof_limbs(proofs[0][0]), nested_proof_limbs(proofs[1][0])])
    A, B, C = c.get_constraint_system()
    for bumps, bits in _bumps(k):
        xs = [list(proofs[0][1]), list(proofs[1][1])]
        for p, j in bumps:
            xs[p][j] = (xs[p][j] + 1) % R.BLS_R
        z = c.witness(vk, npr, np.array([fr_limbs(x) for row in xs for x in row]))
        assert fr_int(z[2]) == bits, (bumps, fr_int(z[2]))
        assert (z[1] == zkhip.aggregator_vk_hash(vk, k)).all() and fr_int(z[1]) == R.nested_vk_hash(nvk)
        assert [fr_int(z[3 + i]) for i in range(2 * k)] == xs[0] + xs[1]
        if len(bumps) != 1 or bumps[0][1] == k // 2:                 # (the oracle's pass over 92 k constraints: ~1 s each)
            assert oracle_lib.r1cs_first_unsatisfied(A, B, C, z) == -1
            zb = z.copy(); zb[2] = fr_limbs(bits ^ 1)               # ... and the OTHER value of a result bit violates one
            assert oracle_lib.r1cs_first_unsatisfied(A, B, C, zb) >= 0
    # the second and third proof in the other order: bits follow the proofs, not the positions
    z = c.witness(vk, np.concatenate([nested_proof_limbs(proofs[2][0]), nested_proof_limbs(proofs[1][0])]),
                  np.array([fr_limbs(x) for x in proofs[2][1] + proofs[0][1]]))
    assert fr_int(z[2]) == 1                                         # proof 1 with proof 0's inputs: rejected
    c.free()


def test_nine_inputs_under_an_unrelated_key(oracle_lib):
    """(Rounds 3-5's nine-input case, kept as the all-reject case.)  The nested
    key is padded with further G1 points from the fixtures and the inputs are arbitrary: the nested proofs are then INVALID, which
    the circuit must accept with result bits 0 (aggregator_circuit.hpp:51-54) - the witness satisfies every constraint."""
    from zecale_amd import zkhip
    k = 9
    c = zkhip.AggregatorCircuit(2, k)
    assert c.num_primary_inputs() == 2 + 2 * k
    nvk, proofs = load_nested_fixtures()
    nvk9 = dict(nvk)
    nvk9["ABC"] = list(nvk["ABC"]) + [proofs[i][0]["a"] for i in range(6)] + [proofs[0][0]["c"], proofs[1][0]["c"]]
    assert len(nvk9["ABC"]) == k + 1
    vk = nested_vk_limbs(nvk9)
    xs = [[1000 * p + j for j in range(k)] for p in range(2)]
    z = c.witness(vk, np.concatenate([nested_proof_limbs(proofs[0][0]), nested_proof_limbs(proofs[1][0])]),
                  np.array([fr_limbs(x) for row in xs for x in row]))
    A, B, C = c.get_constraint_system()
    assert oracle_lib.r1cs_first_unsatisfied(A, B, C, z) == -1
    assert (z[1] == zkhip.aggregator_vk_hash(vk, k)).all()
    assert fr_int(z[2]) == 0                                  # both nested proofs rejected
    assert [fr_int(z[3 + i]) for i in range(2 * k)] == xs[0] + xs[1]
    c.free()


@pytest.mark.parametrize("inputs_per_proof", [1, 9])
def test_application_host_generator_equals_the_full_generator(inputs_per_proof):
    """The host generator of a REGISTERED application (aggregator.cpp: zk_app_host_* - what zkhip_aggregator_witness_app runs on the
    host; the reference registers a key once, aggregator_server.cpp:170-235) computes the proof sections only: the key's hash and lines
    are not recomputed (their slices stay zero) and the doubling chains 2^j ABC_k of the input accumulators are allocated from the
    values the handle keeps.  Everything it does write equals the full generator's assignment, limb for limb - one and nine inputs per
    nested proof, a valid batch and one with a bumped input.  CPU only: the library's internal entry points, no device."""
    import ctypes
    import bench
    from zecale_amd import zkhip
    lib = zkhip.load()
    lib.zk_app_host_new.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    lib.zk_app_host_witness.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    lib.zk_app_host_free.argtypes = [ctypes.c_void_p]
    agg = zkhip.AggregatorCircuit(2, inputs_per_proof)
    nvk, proofs, inputs, _ = bench.aggregator_inputs(inputs_per_proof)
    nvk, proofs = np.ascontiguousarray(nvk), np.ascontiguousarray(proofs)
    st = ctypes.c_void_p()
    assert lib.zk_app_host_new(agg.handle, nvk.ctypes.data, ctypes.byref(st)) == 0
    for bump in (0, 1):
        nin = np.ascontiguousarray(inputs).copy()
        if bump:
            nin.reshape(-1, 6)[-1] = fr_limbs(fr_int(nin.reshape(-1, 6)[-1]) + 1)
        z = agg.witness(nvk, proofs, nin)
        out = np.full_like(z, 0xFFFFFFFFFFFFFFFF)                     # (every entry must be written)
        h = np.ascontiguousarray(z[1])
        assert lib.zk_app_host_witness(agg.handle, st, nvk.ctypes.data, proofs.ctypes.data, nin.ctypes.data, None, 0, h.ctypes.data, out.ctypes.data) == 0
        differ = np.nonzero((out != z).any(axis=1))[0]
        assert len(differ) > 8000 and not out[differ].any()          # the key's own sections, left at zero
        assert differ.max() < len(z) // 2                             # ... which precede the proof sections
        assert (out[:3] == z[:3]).all()                               # ONE, the key's hash, the packed result bits
    lib.zk_app_host_free(st)
    agg.free()
