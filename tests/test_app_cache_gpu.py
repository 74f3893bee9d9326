"""Per-application constants (zkhip_aggregator_app).  The reference registers a nested verification key once
(RegisterApplication, aggregator_server/aggregator_server.cpp:170-235) and aggregates batch after batch under it
(GenerateAggregatedTransaction, :279-348 -> aggregator_circuit::prove, libzecale/circuits/aggregator_circuit.tcc:120-170): the part
of the assignment that depends on the key alone, and its share of four of the five MSMs, is computed once per application here.
Everything below is a statement of EQUALITY with the path that recomputes it per proof - which tests/test_aggregator_gpu.py pins
against the oracle - so a proof with the handle is the oracle's proof too."""
import numpy as np
import pytest

from tests.helpers import fr_int, fr_limbs, random_fr_uniform
from tests.test_aggregator_gpu import _setup
from tests.test_aggregator_host import nested_proof_limbs

pytestmark = pytest.mark.gpu


def _batch(proofs, a, b, bump=0):
    (pa, ia), (pb, ib) = proofs[a], proofs[b]
    return np.concatenate([nested_proof_limbs(pa), nested_proof_limbs(pb)]), np.array([fr_limbs(ia[0]), fr_limbs(ib[0] + bump)])


def _second_key(nvk_l):
    """Another application on the same circuit: the fixture's key with ABC_0 and ABC_1 exchanged (points on the curve; the fixture's
    proofs do not verify under it: result bits 0)."""
    v = nvk_l.copy()
    v[60:72], v[72:84] = nvk_l[72:84], nvk_l[60:72]
    return v


def test_constants_are_the_full_witness_at_their_positions(zk, oracle_lib):
    """The handle's positions hold, in EVERY batch of the application, exactly the handle's values (three batches, one with an invalid
    nested proof); the masked generator = the full generator with those positions zeroed; the hash is primary input 0; the four
    cached points are the oracle's MSM of (values, bases at the positions) for the A, B-G2, B-G1 and L queries."""
    O = oracle_lib
    agg, desc, kp, nvk_l, proofs = _setup(zk)
    crs = kp.upload_crs()
    app = zk.AggregatorApp(agg, crs, nvk_l)
    pos, val, h, pts = app.constants()
    l, m = agg.num_primary_inputs(), agg.num_variables
    # batch 2, one input per nested proof: the key's 14 variables, its hash chain and the lines of -beta / -delta (8,745) and what the
    # recorder folds of the doubling chains of ABC_i inside the two proof sections
    assert app.num_constants == len(pos) and 8759 <= len(pos) <= 12000 and (np.diff(pos.astype(np.int64)) > 0).all() and pos[0] > l and pos[-1] < m
    print("constants per application: %d of %d variables" % (len(pos), m))
    assert (h == zk.aggregator_vk_hash(nvk_l, 1)).all()
    for a, b, bump in ((0, 1, 0), (2, 3, 0), (4, 5, 1)):
        npr, nin = _batch(proofs, a, b, bump)
        z = agg.witness(nvk_l, npr, nin)
        assert (z[pos] == val).all() and (z[1] == h).all()
        zm = app.witness(npr, nin)
        assert (zm == app.mask(z)).all()
        assert not zm[pos].any() and (zm[:l + 1] == z[:l + 1]).all()
        back = zm.copy(); back[pos] = val
        assert (back == z).all()
    pk, m_pk, l_pk, dom = kp.pk_arrays()
    for k, (name, off) in enumerate((("A", 0), ("B2", 0), ("B1", 0), ("L", l + 1))):
        expect = O.jac_to_affine(O.msm(pk[name][pos - off], val))
        assert (zk.jac_to_affine(pts[k]) == expect).all(), name
    # a full assignment of ANOTHER key is not maskable; an off-curve key gets no handle
    other = agg.witness(_second_key(nvk_l), *_batch(proofs, 0, 1))
    with pytest.raises(zk.ZkhipError):
        app.mask(other)
    bad = nvk_l.copy(); bad[6] ^= np.uint64(1)
    with pytest.raises(zk.ZkhipError):
        zk.AggregatorApp(agg, crs, bad)
    app.free(); crs.free(); kp.free(); agg.free()


@pytest.mark.parametrize("naf", [False, True], ids=["window-tables", "naf-tables"])
@pytest.mark.parametrize("domain", [None, "step"], ids=["forced-pow2-domain", "step-domain"])
def test_proof_from_the_masked_assignment_is_the_plain_proof(zk, naf, domain):
    """zkhip_groth16_prove_app / zkhip_prover_prove_app on the masked assignment = zkhip_groth16_prove on the full one, limb for limb
    (the A, B and L sums are the same group elements: masked MSM + cached point), for both kinds of table, both evaluation domains,
    (r, s) uniform in Fr, a valid and an invalid batch; an assignment that is not masked is refused, so is a handle of another key."""
    agg, desc, kp, nvk_l, proofs = _setup(zk, domain)
    crs = kp.upload_crs(zk.key_opts(table_naf=naf))
    r1 = zk.r1cs_from_desc(desc)
    app = zk.AggregatorApp(agg, crs, nvk_l)
    pr = zk.Prover(crs, desc)
    rs = random_fr_uniform(4321, 2)
    for a, b, bump in ((0, 1, 0), (3, 2, 1)):
        npr, nin = _batch(proofs, a, b, bump)
        z = agg.witness(nvk_l, npr, nin)
        plain = zk.groth16_prove(crs, r1, z, rs[0], rs[1])
        zm = app.witness(npr, nin)
        assert (app.prove(r1, zm, rs[0], rs[1]) == plain).all()
        assert (pr.prove_app(app, zm, rs[0], rs[1]) == plain).all()
        assert (pr.prove(z, rs[0], rs[1]) == plain).all()                  # (the instance still proves full assignments)
        assert zk.groth16_verify(kp.vk(), z[1:1 + agg.num_primary_inputs()], plain)
    with pytest.raises(zk.ZkhipError):
        app.prove(r1, z, rs[0], rs[1])                                     # not masked
    crs2 = kp.upload_crs()
    with pytest.raises(zk.ZkhipError):
        zk.Prover(crs2, desc).prove_app(app, zm, rs[0], rs[1])             # the handle belongs to crs
    crs2.free()
    pr.free(); app.free(); crs.free(); r1.free(); kp.free(); agg.free()


@pytest.mark.parametrize("num_proofs", [1, 3])
def test_other_batch_sizes(zk, num_proofs):
    """The handle for circuits of one and of three nested proofs (the reference's batch size is a template parameter,
    aggregator_circuit.tcc:33-60): the constants scale with the proof sections (the key's share once, the folded doubling chains per
    section), the masked generators - host and GPU - agree, and the proof from the masked assignment is the plain proof."""
    from tests.test_aggregator_host import load_nested_fixtures, nested_vk_limbs
    agg = zk.AggregatorCircuit(num_proofs, 1)
    desc = zk.r1cs_desc_from_aggregator(agg)
    kp = zk.Keypair(desc, fr_limbs(0x1234567), fr_limbs(0x2345678), fr_limbs(0x3456789), fr_limbs(0x456789a))
    nvk, proofs = load_nested_fixtures()
    nvk_l = nested_vk_limbs(nvk)
    crs, r1 = kp.upload_crs(), zk.r1cs_from_desc(desc)
    app = zk.AggregatorApp(agg, crs, nvk_l)
    print("batch %d: %d constants of %d variables" % (num_proofs, app.num_constants, agg.num_variables))
    assert 8759 <= app.num_constants < agg.num_variables // 2
    rs = random_fr_uniform(99 + num_proofs, 2)
    bumps = [0, 1, 0][:num_proofs]
    npr = np.concatenate([nested_proof_limbs(proofs[k][0]) for k in range(num_proofs)])
    nin = np.array([fr_limbs(proofs[k][1][0] + bumps[k]) for k in range(num_proofs)])
    z = agg.witness(nvk_l, npr, nin)
    zm = app.witness(npr, nin)
    assert (zm == app.mask(z)).all()
    (zg,), (pi,) = app.witness_gpu([(npr, nin)])
    assert zg is not None and (zg == zm).all()
    expect_bits = sum(0 if bumps[k] else (1 << k) for k in range(num_proofs))
    assert fr_int(pi[1]) == expect_bits
    plain = zk.groth16_prove(crs, r1, z, rs[0], rs[1])
    assert (app.prove(r1, zm, rs[0], rs[1]) == plain).all()
    assert zk.groth16_verify(kp.vk(), z[1:1 + agg.num_primary_inputs()], plain)
    app.free(); crs.free(); r1.free(); kp.free(); agg.free()


def test_gpu_generator_of_an_application(zk):
    """zkhip_gpu_witness_run_batched_app: the application's own device program (its key folded in as constants: no key-hash launch)
    writes the MASKED assignment - three batches in one launch, equal to the masked host generator limb for limb; the primary inputs
    come back complete (hash, result bits, nested inputs)."""
    agg, desc, kp, nvk_l, proofs = _setup(zk)
    crs = kp.upload_crs()
    app = zk.AggregatorApp(agg, crs, nvk_l)
    batches = [_batch(proofs, 0, 1), _batch(proofs, 2, 3), _batch(proofs, 4, 5, 1)]
    zs, prim = app.witness_gpu(batches)
    l = agg.num_primary_inputs()
    for (npr, nin), z, pi, bits in zip(batches, zs, prim, (3, 3, 1)):
        assert z is not None
        zm = app.witness(npr, nin)
        assert (z == zm).all()
        assert (pi == zm[1:1 + l]).all() and fr_int(pi[1]) == bits and (pi[0] == zk.aggregator_vk_hash(nvk_l, 1)).all()
    app.free(); crs.free(); kp.free(); agg.free()


@pytest.mark.parametrize("gpu_witness", [False, True], ids=["host-witness", "gpu-witness"])
def test_pipeline_with_and_without_the_cache_gives_the_same_proofs(zk, gpu_witness):
    """The streaming prover keeps a handle per nested key it meets (two applications here, interleaved, one registered ahead, one met
    by its first batch) - every extended proof equals the one of a pipeline with the cache OFF and the plain serial proof, and the
    cache was actually used."""
    agg, desc, kp, nvk_l, proofs = _setup(zk)
    vk = kp.vk()
    crs, r1 = kp.upload_crs(zk.key_opts(table_naf=True)), zk.r1cs_from_desc(desc)
    nvk2 = _second_key(nvk_l)
    jobs = []
    for i, (a, b, bump) in enumerate(((0, 1, 0), (2, 3, 0), (4, 5, 1), (1, 2, 0), (3, 3, 0), (5, 0, 0), (0, 1, 0), (2, 4, 1), (1, 0, 0), (5, 4, 0))):
        npr, nin = _batch(proofs, a, b, bump)
        key = nvk2 if i % 3 == 2 else nvk_l
        jobs.append((key, npr, nin, fr_limbs(0xaaaa + i), fr_limbs(0xbbbb + 7 * i), (0 if key is nvk2 else (1 if bump else 3))))
    results = {}
    for cache in (True, False):
        pipe = zk.AggregatorPipeline(agg, crs, gpu_slots=3, witness_workers=2, gpu_witness=gpu_witness, app_cache=cache)
        if cache:
            pipe.register_app(nvk_l)                                        # RegisterApplication; the second key is met by its first batch
        tickets = [pipe.submit(k, npr, nin, r, s) for k, npr, nin, r, s, _ in jobs]
        if cache:                                                           # a second round: by now both handles exist
            tickets += [pipe.submit(k, npr, nin, r, s) for k, npr, nin, r, s, _ in jobs]
        results[cache] = [pipe.wait(t) for t in tickets]
        hits = pipe.app_hits()
        assert (hits >= len(jobs)) if cache else (hits == 0), hits
        pipe.free()
    for i, (k, npr, nin, r, s, bits) in enumerate(jobs):
        z = agg.witness(k, npr, nin)
        plain = zk.groth16_prove(crs, r1, z, r, s)
        for prim, proof in (results[True][i], results[True][i + len(jobs)], results[False][i]):
            assert (prim == z[1:1 + agg.num_primary_inputs()]).all() and fr_int(prim[1]) == bits
            assert (proof == plain).all()
        assert zk.groth16_verify(vk, results[True][i][0], results[True][i][1])
    crs.free(); r1.free(); kp.free(); agg.free()


def test_degenerate_key_gets_no_handle_and_is_still_proved(zk):
    """ABC_1 = ABC_0: the input accumulator adds a point to itself - the host generator branches there, the recorded program cannot.
    The recording with the key folded in meets an inversion of a constant zero: no handle (zkhip_aggregator_app_new refuses), the
    pipeline proves the key's batches by the plain path."""
    agg, desc, kp, nvk_l, proofs = _setup(zk)
    crs, r1 = kp.upload_crs(), zk.r1cs_from_desc(desc)
    vk_deg = nvk_l.copy(); vk_deg[72:84] = vk_deg[60:72]
    with pytest.raises(zk.ZkhipError):
        zk.AggregatorApp(agg, crs, vk_deg)
    pipe = zk.AggregatorPipeline(agg, crs, gpu_slots=2, witness_workers=2)
    with pytest.raises(zk.ZkhipError):
        pipe.register_app(vk_deg)
    npr, nin = _batch(proofs, 0, 1)
    prim, proof = pipe.wait(pipe.submit(vk_deg, npr, nin, fr_limbs(5), fr_limbs(6)))
    z = agg.witness(vk_deg, npr, nin)
    assert (proof == zk.groth16_prove(crs, r1, z, fr_limbs(5), fr_limbs(6))).all() and (prim == z[1:5]).all()
    assert zk.groth16_verify(kp.vk(), prim, proof) and pipe.app_hits() == 0
    pipe.free(); crs.free(); r1.free(); kp.free(); agg.free()


def test_nine_inputs_application(zk):
    """The Zeth-shaped circuit (nine inputs per nested proof, aggregator_test.cpp:222-254): the handle's constants, the masked proof =
    the plain proof, and the count the bench reports (the key's 30 variables, its hash and lines, eighteen doubling chains)."""
    import bench
    nvk_l, npr, nin, trapdoor = bench.aggregator_inputs(9)
    agg = zk.AggregatorCircuit(2, 9)
    desc = zk.r1cs_desc_from_aggregator(agg)
    kp = zk.Keypair(desc, *trapdoor)
    crs, r1 = kp.upload_crs(zk.key_opts(table_naf=True)), zk.r1cs_from_desc(desc)
    app = zk.AggregatorApp(agg, crs, nvk_l)
    pos, val, h, pts = app.constants()
    print("nine inputs: constants per application: %d of %d variables" % (len(pos), agg.num_variables))
    z = agg.witness(nvk_l, npr, nin)
    assert (z[pos] == val).all() and len(pos) > 20000
    zm = app.witness(npr, nin)
    assert (zm == app.mask(z)).all()
    rs = random_fr_uniform(99, 2)
    plain = zk.groth16_prove(crs, r1, z, rs[0], rs[1])
    assert (app.prove(r1, zm, rs[0], rs[1]) == plain).all()
    assert zk.groth16_verify(kp.vk(), z[1:1 + agg.num_primary_inputs()], plain)
    app.free(); crs.free(); r1.free(); kp.free(); agg.free()


def test_hybrid_witness_and_dispatcher_registration(zk):
    """ZKHIP_PIPELINE_HYBRID_WITNESS: host generators and GPU batchers on one queue - every proof equals the plain serial proof whichever
    generator produced its assignment; zkhip_dispatcher_register_app registers the application on every entry of a device list."""
    agg, desc, kp, nvk_l, proofs = _setup(zk)
    crs, r1 = kp.upload_crs(), zk.r1cs_from_desc(desc)
    pipe = zk.AggregatorPipeline(agg, crs, gpu_slots=3, witness_workers=2, gpu_witness=True, hybrid=True)
    pipe.register_app(nvk_l)
    jobs = []
    for i in range(40):
        npr, nin = _batch(proofs, i % 6, (i * 5 + 1) % 6, int(i % 7 == 3))
        r, s = fr_limbs(0x1000 + i), fr_limbs(0x2000 + 3 * i)
        jobs.append((npr, nin, r, s, pipe.submit(nvk_l, npr, nin, r, s)))
    for npr, nin, r, s, t in jobs:
        prim, proof = pipe.wait(t)
        z = agg.witness(nvk_l, npr, nin)
        assert (prim == z[1:1 + agg.num_primary_inputs()]).all()
        assert (proof == zk.groth16_prove(crs, r1, z, r, s)).all()
    assert pipe.app_hits() == 40
    pipe.free()
    disp = zk.AggregatorDispatcher(agg, kp, [0, 0], zk.key_opts(table_naf=False), gpu_slots=2, witness_workers=2)
    disp.register_app(nvk_l)
    npr, nin = _batch(proofs, 0, 1)
    ts = [disp.submit(nvk_l, npr, nin, fr_limbs(3), fr_limbs(4)) for _ in range(4)]
    z = agg.witness(nvk_l, npr, nin)
    want = zk.groth16_prove(crs, r1, z, fr_limbs(3), fr_limbs(4))
    for t in ts:
        prim, proof = disp.wait(t)
        assert (proof == want).all()
    bad = nvk_l.copy(); bad[6] ^= np.uint64(1)
    with pytest.raises(zk.ZkhipError):
        disp.register_app(bad)                     # off-curve key: refused on the first entry
    disp.free()
    crs.free(); r1.free(); kp.free(); agg.free()


def test_application_table_evicts_and_failed_keys_hold_no_place(zk):
    """a pipeline's table of application handles holds 32.  Until round 6 the 33rd key of a pipeline's life - and every
    key after it, valid or not - took the plain path for good, and a key that could not be given a handle held a place for ever.  Now
    the least recently used ready handle is evicted and failed keys go to a negative cache: 36 DISTINCT on-curve keys (the fixture's
    key with ABC_0 / ABC_1 replaced by pairs of other G1 points of the fixtures), one batch each, are ALL proved from a handle -
    after a degenerate key (ABC_1 = ABC_0: no handle) has been met several times - and the first key, evicted meanwhile, gets its
    handle again.  Every proof equals the plain serial proof of the full assignment."""
    agg, desc, kp, nvk_l, proofs = _setup(zk)
    crs, r1 = kp.upload_crs(), zk.r1cs_from_desc(desc)
    pts = [nvk_l[60:72].copy(), nvk_l[72:84].copy()]
    for pr, _ in proofs:
        l = nested_proof_limbs(pr)
        pts += [l[:12].copy(), l[36:48].copy()]                              # A and C of the six fixture proofs: points of G1
    keys = []
    for i in range(len(pts)):
        for j in range(len(pts)):
            if i != j and len(keys) < 36 and not (pts[i] == pts[j]).all():
                k = nvk_l.copy(); k[60:72] = pts[i]; k[72:84] = pts[j]
                keys.append(k)
    assert len(keys) == 36 and len({bytes(k) for k in keys}) == 36
    pipe = zk.AggregatorPipeline(agg, crs, gpu_slots=2, witness_workers=2)
    npr, nin = _batch(proofs, 0, 1)
    deg = nvk_l.copy(); deg[72:84] = deg[60:72]
    for _ in range(3):                                                        # no handle, no place taken, proved all the same
        prim, proof = pipe.wait(pipe.submit(deg, npr, nin, fr_limbs(5), fr_limbs(6)))
        assert zk.groth16_verify(kp.vk(), prim, proof)
    assert pipe.app_hits() == 0
    with pytest.raises(zk.ZkhipError):
        pipe.register_app(deg)
    for n, k in enumerate(keys):
        prim, proof = pipe.wait(pipe.submit(k, npr, nin, fr_limbs(0x100 + n), fr_limbs(0x200 + n)))
        assert pipe.app_hits() == n + 1, (n, pipe.app_hits())                # (before round 6: stuck at 32)
        if n in (0, 31, 32, 35):
            z = agg.witness(k, npr, nin)
            assert (prim == z[1:1 + agg.num_primary_inputs()]).all()
            assert (proof == zk.groth16_prove(crs, r1, z, fr_limbs(0x100 + n), fr_limbs(0x200 + n))).all()
    prim, proof = pipe.wait(pipe.submit(keys[0], npr, nin, fr_limbs(7), fr_limbs(8)))      # evicted by now: built again, used at once
    assert pipe.app_hits() == 37 and zk.groth16_verify(kp.vk(), prim, proof)
    pipe.free(); crs.free(); r1.free(); kp.free(); agg.free()
