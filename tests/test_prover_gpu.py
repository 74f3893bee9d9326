"""Parity of the HIP Groth16 prover (zkhip_groth16_prove: SpMV + 7 NTT + 5 MSM + tail) with the CPU
oracle and with the trapdoor closed form; mirrors the checks of the reference's
libzecale/tests/aggregator/aggregator_dummy_test.cpp:61-62 (proof verifies) at the level this
path is pinned: limb-exact proof elements for injected (r, s)."""
import random

import numpy as np
import pytest

from oracle import pyref as R
from tests.helpers import (aff_limbs, aff_point, crs_from_trapdoor, csr_from_rows, fr_array, fr_ints, fr_limbs, golden, h2i,
                           make_r1cs, pt_from_json)

pytestmark = pytest.mark.gpu


def _golden_case():
    g = golden("groth16_small.json")
    pts = lambda L: np.array([aff_limbs(pt_from_json(p)) for p in L]).reshape(-1, 24)
    pk = {k: (aff_limbs(pt_from_json(v)) if k in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2") else pts(v))
          for k, v in g["pk"].items()}
    return g, pk


@pytest.fixture(params=["tables+batched", "tables", "plain"])
def key_mode(request):
    """The three ways a proving key is held and its five MSMs are launched (all must give the same proof): window tables with
    the five MSMs in one launch sequence (default), window tables with one launch sequence per MSM, plain base sets.  The mode is an
    option of the KEY (zkhip_key_opts, resolved at upload; round 4: these tests used to steer through the deprecated process-wide
    zkhip_set_crs_precompute / zkhip_set_batch_msms)."""
    return request.param


def _opts(zk, mode):
    return zk.key_opts(precompute=mode != "plain", batch_msms=mode == "tables+batched")


def test_golden_small_circuit(zk, key_mode):
    g, pk = _golden_case()
    A, B, C = (csr_from_rows(g[k]) for k in "ABC")
    z = fr_array([h2i(x) for x in g["z"]])
    r1 = zk.R1cs(A, B, C, len(g["z"]), g["n_primary"])
    assert r1.log_d == g["log_d"]
    assert r1.is_satisfied(z)
    zbad = z.copy(); zbad[3] = fr_limbs(12345)
    assert not r1.is_satisfied(zbad)
    assert fr_ints(r1.qap_h(z)) == [h2i(x) for x in g["h"]]
    crs = zk.Crs(pk, len(g["z"]), g["n_primary"], 1 << g["log_d"], opts=_opts(zk, key_mode))
    assert (crs.table_window > 0) == (key_mode != "plain")
    proof = zk.groth16_prove(crs, r1, z, fr_limbs(h2i(g["r"])), fr_limbs(h2i(g["s"])))
    assert aff_point(proof[:24]) == pt_from_json(g["proof"]["a"])
    assert aff_point(proof[24:48]) == pt_from_json(g["proof"]["b"])
    assert aff_point(proof[48:]) == pt_from_json(g["proof"]["c"])
    crs.free(); r1.free()


@pytest.mark.parametrize("n,bool_frac,domain", [(3000, 0.0, None), (3000, 0.0, "step"), (4000, 0.6, None)])
def test_synthetic_circuit_vs_oracle_and_trapdoor(zk, oracle_lib, key_mode, n, bool_frac, domain):
    """domain None: the reference's forced power of two (libzeth passes force_pow_2_domain) - the default of every
    entry point; "step": libfqfft's unforced step_radix2_domain, the library's explicit option.  The constraint-system handle is
    created on the DEFAULT domain in both cases: zkhip_groth16_prove moves it to the key's."""
    O = oracle_lib
    n_primary, n_aux = 4, n                # the wrapping circuit has 4 primary inputs (aggregator_circuit.tcc:172-180)
    A, B, C, z = make_r1cs(11 + n, n, n_primary, n_aux, bool_frac)
    m = len(z)
    rng = random.Random(5)
    tau, alpha, beta, delta, r, s = (rng.randrange(1, R.R_MOD) for _ in range(6))
    dom = R.STEP if domain == "step" else None
    pk, d = crs_from_trapdoor(zk, A, B, C, m, n_primary, tau, alpha, beta, delta, dom)
    Ac, Bc, Cc = csr_from_rows(A), csr_from_rows(B), csr_from_rows(C)
    zl = fr_array(z)
    r1 = zk.R1cs(Ac, Bc, Cc, m, n_primary)
    # n = 3000: 3,005 points -> 4,096 forced, 2,048 + 1,024 unforced (step_radix2_domain); n = 4000: 4,005 -> 4,096 either way
    assert r1.domain_size == 4096 == O.qap_domain_size(n, n_primary) == zk.domain_size(n + n_primary + 1)
    assert d == O.qap_domain_size(n, n_primary, O.STEP if domain == "step" else None) == {(3000, None): 4096, (3000, "step"): 3072, (4000, None): 4096}[(n, domain)]
    assert zk.step_domain_size(n + n_primary + 1) == {3000: 3072, 4000: 4096}[n]
    assert r1.is_satisfied(zl)
    if d != r1.domain_size:
        r1.set_domain(d)
        assert r1.domain_size == d and r1.is_satisfied(zl)
    h = r1.qap_h(zl)
    h_or = O.qap_h(Ac, Bc, Cc, zl, n, n_primary, d)
    r1.set_domain(None)                    # back on the default: the proof below must follow the KEY
    assert (h == h_or).all()
    assert (h[-1] == 0).all()
    crs = zk.Crs(pk, m, n_primary, d, opts=_opts(zk, key_mode))
    proof = zk.groth16_prove(crs, r1, zl, fr_limbs(r), fr_limbs(s))
    assert r1.domain_size == d
    proof_or = O.groth16_prove(pk, zl, n_primary, h_or, fr_limbs(r), fr_limbs(s))
    assert (proof == proof_or).all()
    # trapdoor closed form: A = (alpha + a(tau) + r delta) G1 etc. with h from the oracle
    st = R.groth16_setup_scalars(A, B, C, m, n_primary, tau, alpha, beta, delta, dom)
    hi = fr_ints(h_or)
    a_t = sum(zi * x for zi, x in zip(z, st["At"])) % R.R_MOD
    b_t = sum(zi * x for zi, x in zip(z, st["Bt"])) % R.R_MOD
    dinv = pow(delta, -1, R.R_MOD)
    h_t = R.poly_eval(hi[: d - 1], tau) * st["Zt"] % R.R_MOD * dinv % R.R_MOD
    l_t = sum(z[i] * ((beta * st["At"][i] + alpha * st["Bt"][i] + st["Ct"][i]) % R.R_MOD) for i in range(n_primary + 1, m)) % R.R_MOD * dinv % R.R_MOD
    sa = (alpha + a_t + r * delta) % R.R_MOD
    sb = (beta + b_t + s * delta) % R.R_MOD
    sc = (h_t + l_t + s * sa + r * sb - r * s % R.R_MOD * delta) % R.R_MOD
    g1, g2 = aff_limbs(R.G1_GEN), aff_limbs(R.G2_GEN)
    assert (proof[:24] == O.jac_to_affine(O.scalar_mul(g1, fr_limbs(sa)))).all()
    assert (proof[24:48] == O.jac_to_affine(O.scalar_mul(g2, fr_limbs(sb)))).all()
    assert (proof[48:] == O.jac_to_affine(O.scalar_mul(g1, fr_limbs(sc)))).all()
    # and the proof verifies under the key's verification half (wsnark::verify, aggregator_dummy_test.cpp:61-62);
    # a bumped public input does not (aggregator_dummy_test.cpp:162-186 does the same to a nested input)
    assert zk.groth16_verify(pk["vk"], zl[1:1 + n_primary], proof)
    bad = zl[1:1 + n_primary].copy(); bad[1] = fr_limbs(z[2] + 1)
    assert not zk.groth16_verify(pk["vk"], bad, proof)
    print(zk.last_prove_timings())
    crs.free(); r1.free()


def test_bad_arguments(zk):
    g, pk = _golden_case()
    A, B, C = (csr_from_rows(g[k]) for k in "ABC")
    r1 = zk.R1cs(A, B, C, len(g["z"]), g["n_primary"])
    with pytest.raises(zk.ZkhipError):           # proving key for another size
        crs = zk.Crs(pk, len(g["z"]), g["n_primary"], 1 << g["log_d"])
        r2 = zk.R1cs(A, B, C, len(g["z"]), g["n_primary"] + 1)
        zk.groth16_prove(crs, r2, fr_array([h2i(x) for x in g["z"]]), fr_limbs(1), fr_limbs(1))
    r1.free()


def test_key_names_the_domain(zk):
    """a prover works on the domain its KEY was generated on.  One 2,505-point system, three keys from the same
    toxic waste: the reference's forced 4,096 points, libfqfft's unforced 2,048 + 512, and 8,192 (a larger valid domain).  Every key
    proves through the shared handle and through a prover instance; the proofs differ (other Lagrange bases, other H) and each verifies
    under ITS key only; a key whose domain is too small (2,048), or not a domain (3,000), is refused."""
    n, n_primary = 2500, 4
    A, B, C, z = make_r1cs(78, n, n_primary, n, 0.3)
    m = len(z)
    rng = random.Random(10)
    tau, alpha, beta, delta, r, s = (rng.randrange(1, R.R_MOD) for _ in range(6))
    csr = (csr_from_rows(A), csr_from_rows(B), csr_from_rows(C))
    desc, keep = zk.make_r1cs_desc(*csr, m, n_primary)
    zl = fr_array(z)
    r1 = zk.R1cs(*csr, m, n_primary)
    assert r1.domain_size == 4096
    proofs = {}
    for dom, want in ((None, 4096), (R.STEP, 2560), (8192, 8192)):
        pk, d = crs_from_trapdoor(zk, A, B, C, m, n_primary, tau, alpha, beta, delta, dom)
        assert d == want
        crs = zk.Crs(pk, m, n_primary, d)
        proof = zk.groth16_prove(crs, r1, zl, fr_limbs(r), fr_limbs(s))
        assert r1.domain_size == d
        pr = zk.Prover(crs, desc)
        assert (pr.prove(zl, fr_limbs(r), fr_limbs(s)) == proof).all()
        pr.free()
        assert zk.groth16_verify(pk["vk"], zl[1:1 + n_primary], proof)
        proofs[want] = (proof, pk["vk"])
        crs.free()
    assert not zk.groth16_verify(proofs[4096][1], zl[1:1 + n_primary], proofs[2560][0])
    assert not (proofs[4096][0][48:] == proofs[8192][0][48:]).all()
    pk, d = crs_from_trapdoor(zk, A, B, C, m, n_primary, tau, alpha, beta, delta, None)
    for bad_d in (2048, 3000):
        with pytest.raises(zk.ZkhipError):
            bad = dict(pk); bad["H"] = pk["H"][:bad_d - 1]
            crs = zk.Crs(bad, m, n_primary, bad_d)
            try:
                zk.groth16_prove(crs, r1, zl, fr_limbs(r), fr_limbs(s))
            finally:
                crs.free()
    with pytest.raises(zk.ZkhipError):
        zk.R1cs(*csr, m, n_primary, domain=2048)
    r1.free()


def test_key_partitioned_prover_matches_whole_key(zk):
    """SURVEY 8e on one device: the proving key cut into 3 uneven slices (as 3 ranks would hold it), partial sums
    added, tail run once == the single-GPU proof, limb for limb."""
    from zecale_amd import dist as zdist
    n, n_primary = 2500, 4
    A, B, C, z = make_r1cs(77, n, n_primary, n, 0.5)
    m = len(z)
    rng = random.Random(9)
    tau, alpha, beta, delta, r, s = (rng.randrange(1, R.R_MOD) for _ in range(6))
    pk, d = crs_from_trapdoor(zk, A, B, C, m, n_primary, tau, alpha, beta, delta, R.STEP)
    assert d == 2048 + 512                               # 2,505 points: the OPTIONAL step domain with four rows per column (slices of a non-power-of-two H query)
    r1 = zk.R1cs(csr_from_rows(A), csr_from_rows(B), csr_from_rows(C), m, n_primary, domain="step")
    zl = fr_array(z)
    whole = zk.Crs(pk, m, n_primary, d)
    expect = zk.groth16_prove(whole, r1, zl, fr_limbs(r), fr_limbs(s))
    world = 3
    total = None
    for rank in range(world):
        a_rng, h_rng, l_rng = zdist.key_slices(m, n_primary, d, world, rank)
        sl = zk.Crs.upload_slice(pk, m, n_primary, d, a_rng, h_rng, l_rng)
        part = zk.groth16_prove_partial(sl, r1, zl)
        total = part if total is None else np.array([zk.jac_add(total[k], part[k]) for k in range(5)])
        sl.free()
    got = zk.groth16_finish(pk, total, fr_limbs(r), fr_limbs(s))
    assert (got == expect).all()
    assert zk.groth16_verify(pk["vk"], zl[1:1 + n_primary], got)
    whole.free(); r1.free()


@pytest.mark.parametrize("domain", [None, "step"], ids=["forced-pow2-domain", "step-domain"])
def test_setup_slice_is_the_slice_of_the_whole_setup(zk, domain):
    """zkhip_groth16_setup_slice (round 6): every rank of a partitioned key evaluates the exponents, cuts by FINITE terms from the
    exponents (zero exponent = base at infinity: the cuts zkhip_key_partition makes on the finished key) and multiplies only its own
    slice on the device.  Against the whole setup (zkhip_groth16_setup_ex) with the same toxic waste: the ranges equal
    zkhip_key_partition's on the whole key, the partial sums equal those of the same slice uploaded from the whole key, the
    verification half and the tail's constants are the whole key's, and the three ranks' sums finish to the whole-key proof.
    A boolean-heavy system with unused variables: a third of its B query is the point at infinity, so the cuts are uneven."""
    n, n_primary = 3000, 3
    A, B, C, z = make_r1cs(4242, n, n_primary, n + 700, 0.4)          # (700 more variables than constraints: some occur in no B row)
    m = len(z)
    csr = (csr_from_rows(A), csr_from_rows(B), csr_from_rows(C))
    desc, keep = zk.make_r1cs_desc(*csr, m, n_primary)
    td = [fr_limbs(x) for x in (0x1111111, 0x2222223, 0x3333335, 0x4444447)]
    kp = zk.Keypair(desc, *td, domain=domain)
    pk, m_pk, l_pk, dom = kp.pk_arrays()
    finite = lambda a: int(np.count_nonzero(a.reshape(-1, 24).any(axis=1)))
    assert finite(pk["B2"]) < m and finite(pk["B2"]) == finite(pk["B1"])          # (there ARE bases at infinity to cut around)
    r1 = zk.R1cs(*csr, m, n_primary, domain=domain)
    zl = fr_array(z)
    rr, ss = fr_limbs(0xabcdef1), fr_limbs(0x1fedcba)
    whole = kp.upload_crs()
    expect = zk.groth16_prove(whole, r1, zl, rr, ss)
    world = 3
    cuts = zk.key_partition(pk, m, n_primary, dom, world)
    total = None
    for rank in range(world):
        vk_only, sl = zk.Keypair.setup_slice(desc, *td, world, rank, domain=domain)
        want = tuple((int(c[rank]), int(c[rank + 1])) for c in cuts)
        assert sl.ranges == want, (sl.ranges, want)
        ref = zk.Crs.upload_slice(pk, m, n_primary, dom, *want)
        assert sl.finite_terms() == ref.finite_terms() and sl.table_window == ref.table_window
        part, part_ref = zk.groth16_prove_partial(sl, r1, zl), zk.groth16_prove_partial(ref, r1, zl)
        for k in range(5):          # (the SAME group elements: a sum's Jacobian representation depends on the order its pieces were stitched in)
            assert (zk.jac_to_affine(part[k]) == zk.jac_to_affine(part_ref[k])).all(), (rank, k)
        total = part if total is None else np.array([zk.jac_add(total[k], part[k]) for k in range(5)])
        v1, v0 = vk_only.vk(), kp.vk()
        assert all((v1[k] == v0[k]).all() for k in ("alpha", "beta", "delta", "ABC")) and vk_only.domain_size == dom
        c1, c0 = vk_only.consts(), kp.consts()
        assert all((c1[k] == c0[k]).all() for k in c0)
        if rank == 0:                       # the keypair of a slice setup holds no queries: asking for them is an error, not a crash
            for call in (vk_only.pk_arrays, vk_only.upload_crs, lambda: vk_only.write("/tmp/zkhip_no_queries.bin")):
                with pytest.raises(zk.ZkhipError):
                    call()
        sl.free(); ref.free(); vk_only.free()
    got = zk.groth16_finish(kp.consts(), total, rr, ss)
    assert (got == expect).all() and zk.groth16_verify(kp.vk(), zl[1:1 + n_primary], got)
    with pytest.raises(zk.ZkhipError):
        zk.Keypair.setup_slice(desc, *td, 3, 3)
    whole.free(); r1.free(); kp.free()


def test_split_proof_option_gives_the_same_proof(zk):
    """zkhip_set_prove_split (round 6; measured slower and OFF by default, profiles/r06_split_proof_ab.txt): the four MSMs that need
    only the assignment on one plan beside the QAP map, the H MSM on a second plan behind it - the proof and the partial sums of a
    key slice are the same group elements as with one launch sequence, for a whole key, a slice, and a prover instance."""
    n, n_primary = 6000, 2
    A, B, C, z = make_r1cs(99, n, n_primary, n, 0.3)
    m = len(z)
    csr = (csr_from_rows(A), csr_from_rows(B), csr_from_rows(C))
    desc, keep = zk.make_r1cs_desc(*csr, m, n_primary)
    kp = zk.Keypair(desc, *(fr_limbs(x) for x in (0x1111111, 0x2222223, 0x3333335, 0x4444447)))
    r1, zl = zk.R1cs(*csr, m, n_primary), fr_array(z)
    crs = kp.upload_crs()
    rr, ss = fr_limbs(0x5151), fr_limbs(0x7171)
    pk, _, _, dom = kp.pk_arrays()
    cuts = zk.key_partition(pk, m, n_primary, dom, 2)
    sl = zk.Crs.upload_slice(pk, m, n_primary, dom, *((int(c[1]), int(c[2])) for c in cuts))
    try:
        zk.set_prove_split(0)
        expect = zk.groth16_prove(crs, r1, zl, rr, ss)
        assert not zk.last_prove_split()
        part0 = zk.groth16_prove_partial(sl, r1, zl)
        for mode in (1, 2):
            zk.set_prove_split(mode)
            for _ in range(2):
                assert (zk.groth16_prove(crs, r1, zl, rr, ss) == expect).all() and zk.last_prove_split()
            part = zk.groth16_prove_partial(sl, r1, zl)
            assert all((zk.jac_to_affine(part[k]) == zk.jac_to_affine(part0[k])).all() for k in range(5))
            pr = zk.Prover(crs, desc)
            assert (pr.prove(zl, rr, ss) == expect).all()
            pr.free()
        with pytest.raises(zk.ZkhipError):
            zk.set_prove_split(3)
    finally:
        zk.set_prove_split(0)
    assert zk.groth16_verify(kp.vk(), zl[1:1 + n_primary], expect)
    sl.free(); crs.free(); r1.free(); kp.free()


def test_full_size_2_20_key_modes_agree(zk):
    """BASELINE configs[2] size.  No oracle finishes a 2^20-constraint proof in seconds, so the full-size check is a consistency
    property: the three key modes (window tables + one launch sequence for the five MSMs, window tables + one launch sequence per
    MSM, plain base sets) are three different schedules of the same sums and must return the SAME proof, limb for limb; the small
    circuits above pin each mode against the oracle and the trapdoor closed form."""
    import bench
    n = (1 << 20) - 8
    m, l = n + 5, 4
    rng = np.random.default_rng(7)

    def rand_csr(terms):
        cols = rng.integers(0, m, size=(n, terms), dtype=np.uint32).reshape(-1)
        rp = (np.arange(n + 1, dtype=np.uint32) * terms)
        return rp, cols, bench.random_fr_canonical(int(rng.integers(1 << 30)), n * terms)
    csr = (rand_csr(2), rand_csr(2), rand_csr(2))
    r1 = zk.R1cs(*csr, m, l)
    d = r1.domain_size
    assert d == 1 << 20
    g1 = bench.g1_generator_limbs()
    consts = dict(alpha_g1=g1, beta_g1=g1, beta_g2=g1, delta_g1=g1, delta_g2=g1)
    pk = dict(consts)
    for key, cnt, seed in (("A", m, 1), ("B2", m, 2), ("B1", m, 3), ("H", d - 1, 4), ("L", m - l - 1, 5)):
        pk[key] = zk.fixed_base_mul(g1, bench.random_fr_canonical(seed * 7919, cnt), montgomery=False)
    pk["B2"][::3] = 0                                   # a sparse B query, as a real key has
    pk["B1"][::3] = 0
    z = bench.random_fr_canonical(99, m)
    z[::50] = 0
    z[0] = np.array(bench.zkhip_fr_one(), dtype=np.uint64)
    r, s = bench.random_fr_canonical(5, 1)[0], bench.random_fr_canonical(6, 1)[0]
    proofs = {}
    for mode in ("tables+batched", "tables", "plain"):
        crs = zk.Crs(pk, m, l, d, opts=_opts(zk, mode))
        proofs[mode] = zk.groth16_prove(crs, r1, z, r, s)
        crs.free()
    assert (proofs["tables+batched"] == proofs["tables"]).all() and (proofs["tables"] == proofs["plain"]).all()
    assert proofs["plain"][:24].any()
    r1.free()


def _full_size_proof_verifies(zk, log_n):
    from zecale_amd.encoding import R_MOD
    n, l = (1 << log_n) - 8, 4
    m = n + l + 1
    rng = np.random.default_rng(11)
    vals = [1, 3, 5, 7, 11] + [0] * n
    lo = rng.integers(0, 1 << 62, size=n)
    grow = np.arange(l + 1, l + 1 + n, dtype=np.uint64)      # constraint i may use the l + 1 + i earlier variables
    a_idx = (lo.astype(np.uint64) % grow).astype(np.uint32)
    b_idx = ((lo.astype(np.uint64) >> np.uint64(31)) % grow).astype(np.uint32)
    al, bl = a_idx.tolist(), b_idx.tolist()
    for i in range(n):                                   # earlier variables only: satisfiable by construction
        vals[l + 1 + i] = vals[al[i]] * vals[bl[i]] % R_MOD
    shift = 1 << 384
    z = np.frombuffer(b"".join((v * shift % R_MOD).to_bytes(48, "little") for v in vals), dtype=np.uint64).reshape(m, 6).copy()
    del vals
    one = z[0].copy()
    rp = np.arange(n + 1, dtype=np.uint32)
    ones = np.tile(one, (n, 1))
    A = (rp, a_idx, ones); B = (rp, b_idx, ones); C = (rp, np.arange(l + 1, m, dtype=np.uint32), ones)
    desc, keep = zk.make_r1cs_desc(A, B, C, m, l)
    r1 = zk.R1cs(A, B, C, m, l)
    assert r1.log_d == log_n and r1.domain_size == 1 << log_n and r1.is_satisfied(z)
    kp = zk.Keypair(desc, fr_limbs(0x1234567), fr_limbs(0x2345678), fr_limbs(0x3456789), fr_limbs(0x456789a))
    crs = kp.upload_crs()
    proof = zk.groth16_prove(crs, r1, z, fr_limbs(0xabcdef), fr_limbs(0xfedcba))
    print("2^%d constraints:" % log_n, zk.last_prove_timings())
    vk = kp.vk()
    assert zk.groth16_verify(vk, z[1:1 + l], proof)
    bad = z[1:1 + l].copy(); bad[2] = fr_limbs(6)
    assert not zk.groth16_verify(vk, bad, proof)
    zb = z.copy(); zb[m // 2] = fr_limbs(12345)          # a wrong witness is noticed by the satisfiability check (a6)
    assert not r1.is_satisfied(zb)
    crs.free(); kp.free(); r1.free()


def test_full_size_2_20_proof_verifies(zk):
    """BASELINE configs[2] at full size, end to end: a satisfiable 2^20-constraint system (each constraint multiplies two earlier
    variables into a new one), trusted setup on the GPU from a known trapdoor, witness, proof, and wsnark::verify == true with the
    host pairing verifier (aggregator_dummy_test.cpp:61-62); a wrong public input must not verify."""
    _full_size_proof_verifies(zk, 20)


@pytest.mark.parametrize("log_n", [13, 17, 19, 21])
def test_odd_domain_sizes_proof_verifies(zk, log_n):
    """Domains 2^odd split into UNEQUAL factors (K = 2 N2): every shape of the two-pass transforms - one sub-transform per
    workgroup (2^13, 2^17), four / two per workgroup (2^19: 2^9 x 4, 2^10 x 2; 2^21: 2^11, 2^10 x 2), both the natural -> transposed
    and the transposed -> natural order - sits between the SpMV and a proof that verifies."""
    _full_size_proof_verifies(zk, log_n)


def test_full_size_2_22_proof_verifies(zk):
    """BASELINE configs[3]'s size on ONE GPU (the 8-GPU form partitions this key): 2^22 constraints, 4.2 M-point query vectors with
    their window tables (69 GB of HBM), the same end-to-end check as at 2^20."""
    _full_size_proof_verifies(zk, 22)


def test_both_domains_golden_instance(zk):
    """tests/golden/step_domain.json: ONE system (7 constraints + 2 inputs + 1 = 10 points) and trapdoor on both evaluation domains -
    16 points (the reference's forced power of two: the default) and 10 = 8 + 2 points (libfqfft's unforced step_radix2_domain: the
    option) - the coefficients of H and the proof that oracle/pyref.py computed for each (its transforms checked against naive
    evaluation, the proofs against the pinned pairing check), limb for limb.  One handle serves both keys."""
    g = golden("step_domain.json")["groth16"]
    pts = lambda L: np.array([aff_limbs(pt_from_json(p)) for p in L]).reshape(-1, 24)
    A, B, C = (csr_from_rows(g[k]) for k in "ABC")
    z = fr_array([h2i(x) for x in g["z"]])
    r1 = zk.R1cs(A, B, C, len(g["z"]), g["n_primary"])
    assert r1.domain_size == 16 == zk.domain_size(10) and zk.step_domain_size(10) == 10 and r1.is_satisfied(z)
    for case in (g["other_domains"][0], g, g["other_domains"][0]):
        pk = {k: (aff_limbs(pt_from_json(v)) if k in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2") else pts(v)) for k, v in case["pk"].items()}
        crs = zk.Crs(pk, len(g["z"]), g["n_primary"], case["d"])
        proof = zk.groth16_prove(crs, r1, z, fr_limbs(h2i(g["r"])), fr_limbs(h2i(g["s"])))
        assert r1.domain_size == case["d"] and r1.log_d == 4 and r1.is_satisfied(z)
        assert fr_ints(r1.qap_h(z)) == [h2i(x) for x in case["h"]]
        assert aff_point(proof[:24]) == pt_from_json(case["proof"]["a"])
        assert aff_point(proof[24:48]) == pt_from_json(case["proof"]["b"])
        assert aff_point(proof[48:]) == pt_from_json(case["proof"]["c"])
        crs.free()
    r1.free()


@pytest.mark.parametrize("points,domain", [(3, 3), (6, 6), (11, 12), (1025, 1025), (1030, 1032), (2500, 2560), (4100, 4100), (6000, 6144), (20000, 20480),
                                           (33000, 33024)])
def test_step_domains_qap_against_the_oracle(zk, oracle_lib, points, domain):
    """The library's OPTION (ZKHIP_DOMAIN_STEP; the default is the reference's forced power of two, checked beside it here).
    The QAP map over the domains libfqfft picks, unforced, for sizes that are not powers of two (2^k + 2^r points), against the C oracle:
    one row per column up to a thousand (1,025 = 1,024 + 1: the whole big part folds into ONE point), parts below and above the
    2^12 points from which a transform's input order is transposed, the two parts in different orders (4,100 = 4,096 + 4)."""
    O = oracle_lib
    l = 1
    n = points - l - 1
    rng = np.random.default_rng(points)
    m = n + l + 1 + 5
    import bench
    def rand_csr(terms):
        cols = rng.integers(0, m, size=(n, terms), dtype=np.uint32).reshape(-1)
        rp = (np.arange(n + 1, dtype=np.uint32) * terms)
        return rp, cols, bench.random_fr_canonical(int(rng.integers(1 << 30)), n * terms)
    csr = (rand_csr(2), rand_csr(1), rand_csr(2))
    z = bench.random_fr_canonical(7 + points, m)
    r1 = zk.R1cs(*csr, m, l, domain="step")
    assert r1.domain_size == domain == O.qap_domain_size(n, l, O.STEP) == zk.step_domain_size(points)
    assert (r1.qap_h(z) == O.qap_h(*csr, z, n, l, O.STEP)).all()          # (an unsatisfied system: H is then a quotient with a remainder - the same one)
    # the same handle on the reference's domain: the forced power of two
    r1.set_domain(None)
    forced = 1 << (points - 1).bit_length()
    assert r1.domain_size == forced == O.qap_domain_size(n, l) == zk.domain_size(points)
    assert (r1.qap_h(z) == O.qap_h(*csr, z, n, l)).all()
    r1.free()
