#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from oracle/pyref.py (pure-Python big
integers; pyref itself is pinned on the reference's BW6-761 Groth16 KAT fixtures, see
tests/test_oracle_pins.py).  No reference code is involved: the reference tree holds no
implementation of this path (its arithmetic lives in an absent submodule).

The JSON files tests/golden/dummy_app/*.json are DATA FILES copied verbatim from the
reference's testdata/dummy_app/ (fixtures its own tests consume).

Run from the repo root:  python tests/golden/gen_golden.py
All values are canonical (non-Montgomery) integers as hex strings.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyref as R  # noqa: E402

hx = lambda v: hex(v)
pt = lambda P: None if P is None else [hex(P[0]), hex(P[1])]


def gen_fields(rng):
    out = {}
    for name, p in (("fq", R.Q_MOD), ("fr", R.R_MOD)):
        vec = []
        specials = [(0, 0), (1, 1), (p - 1, p - 1), (p - 1, 1), (2, (p + 1) // 2)]
        for i in range(24):
            a, b = specials[i] if i < len(specials) else (rng.randrange(p), rng.randrange(p))
            vec.append(dict(a=hx(a), b=hx(b), mul=hx(a * b % p), add=hx((a + b) % p), sub=hx((a - b) % p),
                            inv_a=hx(pow(a, -1, p)) if a else None))
        out[name] = vec
    return out


def gen_curve(rng):
    out = {}
    for name, G, b in (("g1", R.G1_GEN, R.G1_B), ("g2", R.G2_GEN, R.G2_B)):
        ks = [1, 2, 3, R.R_MOD - 1, R.R_MOD - 2] + [rng.randrange(R.R_MOD) for _ in range(5)]
        muls = [dict(k=hx(k), P=pt(R.ec_mul(k, G))) for k in ks]
        P, Q = R.ec_mul(ks[5], G), R.ec_mul(ks[6], G)
        assert R.on_curve(P, b) and R.on_curve(Q, b)
        out[name] = dict(gen=pt(G), muls=muls, add=dict(P=pt(P), Q=pt(Q), sum=pt(R.ec_add(P, Q)), dblP=pt(R.ec_add(P, P)),
                                                         PminusP=pt(R.ec_add(P, R.ec_neg(P)))))
    return out


def gen_msm(rng):
    cases = []
    for name, G, n in (("g1_random_64", R.G1_GEN, 64), ("g2_random_48", R.G2_GEN, 48), ("g1_edge_40", R.G1_GEN, 40)):
        P = R.ec_mul(rng.randrange(R.R_MOD), G)
        D = R.ec_mul(rng.randrange(R.R_MOD), G)
        pts = []
        for _ in range(n):
            pts.append(P)
            P = R.ec_add(P, D)
        sc = [rng.randrange(R.R_MOD) for _ in range(n)]
        if "edge" in name:
            sc[0] = 0; sc[1] = 1; sc[2] = R.R_MOD - 1; sc[3] = 2; sc[4] = 1
            pts[6] = pts[5]                      # duplicate point
            pts[8] = R.ec_neg(pts[7]); sc[8] = sc[7]   # P + (-P)
            pts[10] = None                       # infinity base
            sc[11] = (1 << 376)                  # top bit region
            sc[12] = (1 << 16) - 1; sc[13] = 1 << 15; sc[14] = (1 << 15) + 1   # signed-digit boundaries
        res = R.msm_naive(sc, pts)
        assert res == R.msm_pippenger(sc, pts, c=7)
        cases.append(dict(name=name, bases=[pt(p) for p in pts], scalars=[hx(s) for s in sc], result=pt(res)))
    # all-zero scalars, all-one scalars, single term
    P = R.ec_mul(99, R.G1_GEN)
    pts = [R.ec_mul(i + 5, R.G1_GEN) for i in range(8)]
    cases.append(dict(name="g1_all_zero", bases=[pt(p) for p in pts], scalars=[hx(0)] * 8, result=None))
    cases.append(dict(name="g1_all_one", bases=[pt(p) for p in pts], scalars=[hx(1)] * 8, result=pt(R.msm_naive([1] * 8, pts))))
    cases.append(dict(name="g1_single", bases=[pt(P)], scalars=[hx(R.R_MOD - 5)], result=pt(R.ec_mul(R.R_MOD - 5, P))))
    return cases


def gen_ntt(rng):
    out = []
    for log_d in (1, 3, 5, 7):
        d = 1 << log_d
        a = [rng.randrange(R.R_MOD) for _ in range(d)]
        if log_d == 3:
            assert R.fft_domain(a, log_d) == R.dft_naive(a, R.fr_root_of_unity(log_d))
        out.append(dict(log_d=log_d, input=[hx(x) for x in a],
                        fft=[hx(x) for x in R.fft_domain(a, log_d)],
                        ifft=[hx(x) for x in R.ifft_domain(a, log_d)],
                        coset_fft=[hx(x) for x in R.coset_fft_domain(a, log_d)],
                        icoset_fft=[hx(x) for x in R.icoset_fft_domain(a, log_d)]))
    return out


def small_r1cs(rng, n_constraints, n_primary, n_aux):
    """Satisfiable by construction: constraint j: (sum a_i z_i) * (sum b_i z_i) = z_out_j (+ slack lc)."""
    m = 1 + n_primary + n_aux
    z = [1] + [rng.randrange(R.R_MOD) for _ in range(m - 1)]
    A, B, C = [], [], []
    for j in range(n_constraints):
        ra = [(rng.randrange(m), rng.randrange(1, 1 << 20)) for _ in range(rng.randrange(1, 4))]
        rb = [(rng.randrange(m), rng.randrange(1, R.R_MOD)) for _ in range(rng.randrange(1, 3))]
        va, vb = R.r1cs_eval_row(ra, z), R.r1cs_eval_row(rb, z)
        # C row: c1*z_k + c0*1 with c0 chosen to satisfy
        k = rng.randrange(1, m)
        c1 = rng.randrange(1, R.R_MOD)
        c0 = (va * vb - c1 * z[k]) % R.R_MOD
        rc = [(k, c1), (0, c0)]
        A.append(ra); B.append(rb); C.append(rc)
    assert R.r1cs_is_satisfied(A, B, C, z)
    return A, B, C, z


def gen_groth16(rng, n_constraints=5, domain=None, also=()):
    """domain: the QAP's evaluation domain (oracle/pyref.py qap_domain_size): None = the reference's forced power of two, R.STEP =
    libfqfft's unforced choice.  also: further domains for the SAME system, trapdoor and (r, s), returned under "other_domains"."""
    n_primary = 2
    A, B, C, z = small_r1cs(rng, n_constraints, n_primary, 4)
    tau, alpha, beta, delta = (rng.randrange(1, R.R_MOD) for _ in range(4))
    r, s = rng.randrange(R.R_MOD), rng.randrange(R.R_MOD)
    out = _groth16_on(A, B, C, z, n_primary, tau, alpha, beta, delta, r, s, domain)
    if also:
        out["other_domains"] = [_groth16_on(A, B, C, z, n_primary, tau, alpha, beta, delta, r, s, dm, key_only=True) for dm in also]
    return out


def _groth16_on(A, B, C, z, n_primary, tau, alpha, beta, delta, r, s, domain, key_only=False):
    pk, vk = R.groth16_generate_keypair(A, B, C, len(z), n_primary, tau, alpha, beta, delta, domain)
    proof = R.groth16_prove(pk, A, B, C, z, r, s)
    expect = R.groth16_expected_proof_from_trapdoor(A, B, C, z, n_primary, tau, alpha, beta, delta, r, s, domain)
    assert proof == expect
    h, d = R.qap_witness_map(A, B, C, z, n_primary, domain)
    log_d = R.qap_domain_log(len(A), n_primary, domain)
    # pairing check of the proof under the vk with the reference's verification equation
    assert R.bw6_groth16_verify(dict(alpha=vk["alpha_g1"], beta=vk["beta_g2"], delta=vk["delta_g2"], ABC=vk["ABC_g1"]),
                                dict(a=proof[0], b=proof[1], c=proof[2]), z[1:1 + n_primary])
    rows = lambda M: [[[i, hx(c)] for i, c in row] for row in M]
    system = dict() if key_only else dict(n_primary=n_primary, A=rows(A), B=rows(B), C=rows(C), z=[hx(x) for x in z],
                                          trapdoor=dict(tau=hx(tau), alpha=hx(alpha), beta=hx(beta), delta=hx(delta)), r=hx(r), s=hx(s))
    return dict(**system,
                log_d=log_d, d=d, h=[hx(x) for x in h],
                pk=dict(alpha_g1=pt(pk["alpha_g1"]), beta_g1=pt(pk["beta_g1"]), beta_g2=pt(pk["beta_g2"]),
                        delta_g1=pt(pk["delta_g1"]), delta_g2=pt(pk["delta_g2"]),
                        A=[pt(p) for p in pk["A_query"]], B2=[pt(p) for p in pk["B_query_g2"]],
                        B1=[pt(p) for p in pk["B_query_g1"]], H=[pt(p) for p in pk["H_query"]],
                        L=[pt(p) for p in pk["L_query"]]),
                vk=dict(alpha=pt(vk["alpha_g1"]), beta=pt(vk["beta_g2"]), delta=pt(vk["delta_g2"]),
                        ABC=[pt(p) for p in vk["ABC_g1"]]),
                proof=dict(a=pt(proof[0]), b=pt(proof[1]), c=pt(proof[2])))


def gen_step_domain(rng):
    """Round 4: the evaluation domains libfqfft picks for sizes that are not powers of two (step_radix2_domain: 2^k + 2^r points).
    FFT / iFFT / cosetFFT / icosetFFT vectors, each checked against naive evaluation at the domain's points, and a Groth16 instance
    of 7 constraints + 2 inputs + 1 = 10 = 8 + 2 points whose proof passes the pinned pairing check."""
    vecs = []
    for d in (3, 6, 10, 12, 24, 40):
        dom = R.EvalDomain(d)
        a = [rng.randrange(R.R_MOD) for _ in range(d)]
        pts = dom.points()
        assert dom.fft(a) == [R.poly_eval(a, x) for x in pts]
        assert dom.coset_fft(a) == [R.poly_eval(a, R.FR_GENERATOR * x % R.R_MOD) for x in pts]
        assert dom.ifft(dom.fft(a)) == a and dom.icoset_fft(dom.coset_fft(a)) == a
        vecs.append(dict(d=d, input=[hx(x) for x in a], fft=[hx(x) for x in dom.fft(a)], ifft=[hx(x) for x in dom.ifft(a)],
                         coset_fft=[hx(x) for x in dom.coset_fft(a)], icoset_fft=[hx(x) for x in dom.icoset_fft(a)]))
    sizes = {str(m): R.evaluation_domain_size(m) for m in (1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 13, 17, 33, 100, 1025, 44188, 92060, 1048573, 4194301)}
    # the SAME system and toxic waste on both domains: "groth16" = libfqfft's unforced step domain (10 = 8 + 2 points; the library's
    # option), "other_domains"[0] = the reference's forced power of two (16 points; the default) - key, h and proof of each
    g = gen_groth16(rng, n_constraints=7, domain=R.STEP, also=(None,))
    assert g["d"] == 10 and g["other_domains"][0]["d"] == 16
    return dict(domain_sizes=sizes, forced_domain_sizes={m: R.forced_domain_size(int(m)) for m in sizes}, fft_vectors=vecs, groth16=g)


def gen_nested(seed, n_inputs, n_proofs=3):
    """Round 6: VALID nested BLS12-377 Groth16 statements with more than one public input, from a known trapdoor
    (pyref.bls12_377_groth16_statement_from_trapdoor) - the reference's slow test aggregates valid nine-input Zeth proofs
    (libzecale/tests/aggregator/aggregator_test.cpp:222-254,293-314) and no such proof is in the tree.  Same JSON shapes as
    the reference's testdata/dummy_app/vk.json and extproof*.json (G2 coordinates as [c1, c0]).  Checked here: every point on
    its curve and of order r, every proof accepted by the pinned verifier, and REJECTED with any one input bumped."""
    vk, proofs = R.bls12_377_groth16_statement_from_trapdoor(random.Random(seed), n_inputs, n_proofs)
    g1ok = lambda P: R.on_curve(P, R.BLS_G1_B, R.BLS_Q) and R.ec_mul(R.BLS_R, P, R.BLS_Q) is None
    g2ok = lambda Q: R.bls_g2_on_curve(Q) and R.bls_g2_mul(R.BLS_R - 1, Q) == R.bls_g2_neg(Q)
    assert g1ok(vk["alpha"]) and g2ok(vk["beta"]) and g2ok(vk["delta"]) and all(g1ok(P) for P in vk["ABC"])
    g2 = lambda Q: [[hx(Q[0][1]), hx(Q[0][0])], [hx(Q[1][1]), hx(Q[1][0])]]
    out = dict(vk=dict(alpha=pt(vk["alpha"]), beta=g2(vk["beta"]), delta=g2(vk["delta"]), ABC=[pt(P) for P in vk["ABC"]]), proofs=[])
    for pr, xs in proofs:
        assert g1ok(pr["a"]) and g2ok(pr["b"]) and g1ok(pr["c"])
        assert R.bls12_377_groth16_verify(vk, pr, xs)
        for j in range(n_inputs):
            bad = list(xs); bad[j] = (bad[j] + 1) % R.BLS_R
            assert not R.bls12_377_groth16_verify(vk, pr, bad), j
        out["proofs"].append(dict(proof=dict(a=pt(pr["a"]), b=g2(pr["b"]), c=pt(pr["c"])), inputs=[hx(x) for x in xs]))
    return out


def main():
    if "--nested-only" in sys.argv:
        for k, seed in ((9, 0x9E57ED), (3, 0x3E57ED)):
            with open(os.path.join(HERE, f"nested_k{k}.json"), "w") as f:
                json.dump(gen_nested(seed, k), f, indent=0)
            print("wrote nested_k%d" % k)
        return
    if "--step-only" in sys.argv:          # (the other files are unchanged by rounds 4-5: their domains are powers of two under both rules)
        with open(os.path.join(HERE, "step_domain.json"), "w") as f:
            json.dump(gen_step_domain(random.Random(0x57E9)), f, indent=0)
        print("wrote step_domain")
        return
    rng = random.Random(0x5EED)
    files = dict(field_vectors=gen_fields(rng), curve_vectors=gen_curve(rng), msm_vectors=gen_msm(rng),
                 ntt_vectors=gen_ntt(rng), groth16_small=gen_groth16(rng))
    files["step_domain"] = gen_step_domain(random.Random(0x57E9))
    files["nested_k9"] = gen_nested(0x9E57ED, 9)
    files["nested_k3"] = gen_nested(0x3E57ED, 3)
    for name, data in files.items():
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(data, f, indent=0)
        print("wrote", name)


if __name__ == "__main__":
    main()
