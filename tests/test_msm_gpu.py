"""Parity of the HIP MSM (through the C ABI, include/zkhip.h) with the CPU oracle.  Bit-exact:
results are compared as affine-normalised Montgomery limbs.  Cases follow what the reference's
path sees: random, all-zero, all-one, single term, duplicate points, P + (-P),
maximal scalar r - 1, infinity bases, boolean-heavy witness-like scalars; G1 and G2."""
import numpy as np
import pytest

from oracle import pyref as R
from tests.helpers import aff_limbs, aff_point, fr_array, golden, h2i, pt_from_json, random_fr_canonical, random_fr_uniform

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[-1, 1, 3], ids=["auto", "affine1", "affine3"])
def affine_levels(request, zk):
    """Every case runs with the default accumulation (XYZZ mixed additions only) and with one and three batched-affine levels
    forced in front of it (pairwise affine sums inside the buckets, one shared inversion per lane: the same group element)."""
    name = request.node.name
    if request.param != -1:
        wanted = ("golden", "random_vs_oracle", "witness_like", "heavy_buckets", "order_two", "table_random", "submit_collect")
        big3 = request.param == 3 and ("full_size_2_20_closed_form" in name or "table_full_size" in name)
        if not (any(w in name for w in wanted) or big3):
            pytest.skip("affine-level variants run on the cases that reach the level kernel's branches")
    zk.set_affine_levels(request.param)
    yield request.param
    zk.set_affine_levels(-1)


@pytest.fixture(params=[False, True], ids=["windows", "naf"])
def table_kind(request):
    """The table-backed cases run with both kinds of table: one level per window, and every bit position with the scalars recoded in
    non-adjacent form (odd signed digits at arbitrary positions, the same group element).  The kind is an option of the base set's
    HANDLE (zkhip_bases_precompute_ex; round 4: these tests used to steer through the deprecated process-wide zkhip_set_table_naf)."""
    return request.param


def _msm_aff(zk, bases, scal, montgomery=True, window=0):
    """One MSM over a plain base set; the window is an option of the handle (zkhip_bases_set_window)."""
    if len(bases) == 0:
        return zk.jac_to_affine(zk.msm_raw(bases, scal, montgomery=montgomery))
    b = zk.Bases.upload(bases).set_window(window)
    out = zk.jac_to_affine(b.msm(scal, montgomery=montgomery))
    b.free()
    return out


@pytest.mark.parametrize("window", [0, 5, 9])
def test_golden_vectors(zk, window):
    for case in golden("msm_vectors.json"):
        bases = np.array([aff_limbs(pt_from_json(p)) for p in case["bases"]])
        scal = fr_array([h2i(s) for s in case["scalars"]])
        got = aff_point(_msm_aff(zk, bases, scal, window=window))
        assert got == pt_from_json(case["result"]), case["name"]


def test_empty(zk):
    b = zk.Bases.upload(np.zeros((4, 24), dtype=np.uint64))
    out = b.msm(np.zeros((0, 6), dtype=np.uint64))
    assert (zk.jac_to_affine(out) == 0).all()
    b.free()


def test_fixed_base_mul_matches_oracle(zk, oracle_lib):
    O = oracle_lib
    for G in (R.G1_GEN, R.G2_GEN):
        g = aff_limbs(G)
        ks = random_fr_canonical(7, 64)
        ks[0] = 0
        ks[1, :] = 0; ks[1, 0] = 1
        pts = zk.fixed_base_mul(g, ks, montgomery=True)
        for i in range(64):
            assert (pts[i] == O.jac_to_affine(O.scalar_mul(g, ks[i]))).all(), i


@pytest.mark.parametrize("n,window", [(1 << 10, 0), (5000, 11), (1 << 14, 0), (1 << 14, 13)])
def test_random_vs_oracle_g1(zk, oracle_lib, n, window):
    O = oracle_lib
    bases = zk.fixed_base_mul(aff_limbs(R.G1_GEN), random_fr_uniform(100 + n, n), montgomery=False)
    scal = random_fr_uniform(200 + n, n)         # uniform in [0, r): every residue the ABI allows, top bit included (bench.py draws the same way)
    assert (_msm_aff(zk, bases, scal, window=window) == O.jac_to_affine(O.msm(bases, scal))).all()
    # the same limbs as CANONICAL integers (scalars_montgomery = 0): digits up to the top bit of r
    scal_m = np.array([O.f_op("from_canonical", 1, s_) for s_ in scal[:2000]])
    got = zk.jac_to_affine(zk.msm_raw(bases[:2000], scal[:2000], montgomery=False))
    assert (got == O.jac_to_affine(O.msm(bases[:2000], scal_m))).all()


def test_random_vs_oracle_g2(zk, oracle_lib):
    O = oracle_lib
    n = 3000
    bases = zk.fixed_base_mul(aff_limbs(R.G2_GEN), random_fr_canonical(31, n), montgomery=False)
    assert O.on_curve(bases[5], g2=True)
    scal = random_fr_uniform(32, n)
    assert (_msm_aff(zk, bases, scal) == O.jac_to_affine(O.msm(bases, scal))).all()


def test_witness_like_scalars(zk, oracle_lib):
    """70 % of the scalars in {0, 1} (boolean-heavy R1CS witness), Montgomery form."""
    O = oracle_lib
    n = 1 << 13
    bases = zk.fixed_base_mul(aff_limbs(R.G1_GEN), random_fr_canonical(41, n), montgomery=False)
    rng = np.random.default_rng(5)
    sel = rng.random(n)
    vals = [0 if s < 0.35 else 1 if s < 0.7 else int(x) for s, x in zip(sel, rng.integers(2, 1 << 62, n))]
    scal = fr_array(vals)
    big = random_fr_canonical(42, n)
    scal[sel > 0.85] = big[sel > 0.85]       # some full-size ones (treated as Montgomery residues)
    assert (_msm_aff(zk, bases, scal) == O.jac_to_affine(O.msm(bases, scal))).all()


def test_the_accumulation_reports_the_entries_it_sorted(zk):
    """zkhip_last_accumulate_entries (bench.py's roofline numerator for witness-like scalars): zero scalars produce no entry, a scalar
    of one digit produces one, a full-size scalar one per non-zero window."""
    n = 2048
    bases = zk.fixed_base_mul(aff_limbs(R.G1_GEN), random_fr_canonical(61, n), montgomery=False)
    b = zk.Bases.upload(bases)
    scal = fr_array([0] * 1000 + [1] * 1048)             # (Montgomery residues)
    b.msm(scal)
    assert zk.last_accumulate_entries() == 1048
    b.msm(random_fr_canonical(62, n), montgomery=False)
    full = zk.last_accumulate_entries()
    assert 20 * n < full <= 48 * n              # (c-bit windows over 378 bits, nearly all of them non-zero)
    b.free()


def test_resident_bases_offset_and_reuse(zk, oracle_lib):
    O = oracle_lib
    n = 2048
    bases = zk.fixed_base_mul(aff_limbs(R.G1_GEN), random_fr_canonical(51, n), montgomery=False)
    b = zk.Bases.upload(bases)
    assert len(b) == n
    for off, ln in ((0, n), (100, 1000), (2047, 1)):
        scal = random_fr_canonical(52 + off, ln)
        got = zk.jac_to_affine(b.msm(scal, offset=off))
        assert (got == O.jac_to_affine(O.msm(bases[off:off + ln], scal))).all()
    with pytest.raises(zk.ZkhipError):
        b.msm(random_fr_canonical(1, 10), offset=n - 5)
    b.free()


def test_full_size_2_20_closed_form(zk, oracle_lib):
    """BASELINE config 2 size.  Bases k_i*G (k_i known) => sum s_i (k_i G) = (sum s_i k_i mod r) G:
    a size-independent check of the 2^20 MSM against ONE oracle scalar multiplication; plus
    linearity MSM(s) + MSM(t) = MSM(s + t) on the same bases."""
    O = oracle_lib
    n = 1 << 20
    g = aff_limbs(R.G1_GEN)
    ks = random_fr_canonical(61, n)
    bases = zk.fixed_base_mul(g, ks, montgomery=False)
    # spot-check the generator itself against the oracle
    for i in (0, 12345, n - 1):
        k_m = np.array(R.int_to_limbs(R.to_mont(R.limbs_to_int(ks[i]), R.R_MOD, 6), 6), dtype=np.uint64)
        assert (bases[i] == O.jac_to_affine(O.scalar_mul(g, k_m))).all()
    b = zk.Bases.upload(bases)
    s = random_fr_canonical(62, n)
    t = random_fr_canonical(63, n)
    to_int = lambda a: [int(x[0]) | int(x[1]) << 64 | int(x[2]) << 128 | int(x[3]) << 192 | int(x[4]) << 256 | int(x[5]) << 320
                        for x in a.tolist()]
    ki, si, ti = to_int(ks), to_int(s), to_int(t)
    ms = b.msm(s, montgomery=False)
    mt = b.msm(t, montgomery=False)
    dot = sum(a * k for a, k in zip(si, ki)) % R.R_MOD
    exp = O.jac_to_affine(O.scalar_mul(g, np.array(R.int_to_limbs(R.to_mont(dot, R.R_MOD, 6), 6), dtype=np.uint64)))
    assert (zk.jac_to_affine(ms) == exp).all()
    st = np.array([R.int_to_limbs((a + c) % R.R_MOD, 6) for a, c in zip(si, ti)], dtype=np.uint64)
    mst = b.msm(st, montgomery=False)
    assert (zk.jac_to_affine(zk.jac_add(ms, mt)) == zk.jac_to_affine(mst)).all()
    b.free()


def test_heavy_buckets_are_stitched(zk, oracle_lib):
    """Buckets far larger than a slice (half the scalars equal to 1, a fifth equal to one constant):
    exercises the logarithmic stitching of cut buckets (k_fixup_round)."""
    O = oracle_lib
    n = 1 << 14
    bases = zk.fixed_base_mul(aff_limbs(R.G1_GEN), random_fr_canonical(71, n), montgomery=False)
    scal = random_fr_canonical(72, n)
    rng = np.random.default_rng(9)
    sel = rng.random(n)
    one = np.zeros(6, dtype=np.uint64); one[0] = 1
    scal[sel < 0.5] = one
    scal[(sel >= 0.5) & (sel < 0.7)] = scal[0]
    for window in (0, 8):
        got = _msm_aff(zk, bases, scal, montgomery=False, window=window)
        # oracle takes Montgomery scalars: convert canonical -> Montgomery on the oracle side
        scal_m = np.array([O.f_op("from_canonical", 1, s) for s in scal])
        assert (got == O.jac_to_affine(O.msm(bases, scal_m))).all()


# ---- window tables (zkhip_bases_precompute): same group element as the plain path and as the oracle ----

@pytest.mark.parametrize("window", [0, 5, 9, 13])
def test_table_golden_vectors(zk, window, table_kind):
    """Every golden MSM vector through a table-backed base set (includes infinity bases, P + (-P), r - 1)."""
    for case in golden("msm_vectors.json"):
        bases = np.array([aff_limbs(pt_from_json(p)) for p in case["bases"]])
        scal = fr_array([h2i(s) for s in case["scalars"]])
        b = zk.Bases.upload(bases).precompute(window, table_naf=table_kind)
        assert b.table_window == (window or b.table_window) and b.table_window > 0
        got = aff_point(zk.jac_to_affine(b.msm(scal)))
        b.free()
        assert got == pt_from_json(case["result"]), case["name"]


@pytest.mark.parametrize("n,window", [(1 << 10, 0), (5000, 11), (1 << 14, 0), (1 << 14, 17), (3001, 20)])
def test_table_random_vs_oracle(zk, oracle_lib, n, window, table_kind):
    O = oracle_lib
    bases = zk.fixed_base_mul(aff_limbs(R.G1_GEN), random_fr_canonical(300 + n, n), montgomery=False)
    bases[7] = 0                                     # a base at infinity
    scal = random_fr_canonical(400 + n, n)
    scal[3] = 0
    scal[4, :] = 0; scal[4, 0] = 1
    b = zk.Bases.upload(bases).precompute(window, table_naf=table_kind)
    exp = O.jac_to_affine(O.msm(bases, scal))
    assert (zk.jac_to_affine(b.msm(scal)) == exp).all()
    # sub-range of a table-backed set
    off, ln = 17, n - 100
    assert (zk.jac_to_affine(b.msm(scal[:ln], offset=off)) == O.jac_to_affine(O.msm(bases[off:off + ln], scal[:ln]))).all()
    with pytest.raises(zk.ZkhipError):
        b.precompute(window, table_naf=table_kind)   # only once
    b.free()


def test_table_g2_and_witness_like(zk, oracle_lib, table_kind):
    O = oracle_lib
    n = 4096
    bases = zk.fixed_base_mul(aff_limbs(R.G2_GEN), random_fr_canonical(501, n), montgomery=False)
    rng = np.random.default_rng(6)
    sel = rng.random(n)
    scal = fr_array([0 if s < 0.35 else 1 if s < 0.7 else int(x) for s, x in zip(sel, rng.integers(2, 1 << 62, n))])
    big = random_fr_canonical(502, n)
    scal[sel > 0.85] = big[sel > 0.85]
    b = zk.Bases.upload(bases).precompute(table_naf=table_kind)
    assert (zk.jac_to_affine(b.msm(scal)) == O.jac_to_affine(O.msm(bases, scal))).all()
    b.free()


def test_table_point_of_order_two(zk, oracle_lib, table_kind):
    """(1, 0) lies on G1's curve y^2 = x^3 - 1 and has order 2: every table level above 0 is the point at infinity.
    Not a proving-key element, but zkhip_msm is a general group operation."""
    O = oracle_lib
    n = 64
    bases = zk.fixed_base_mul(aff_limbs(R.G1_GEN), random_fr_canonical(601, n), montgomery=False)
    bases[5] = aff_limbs((1, 0))
    assert O.on_curve(bases[5])
    scal = random_fr_canonical(602, n)
    b = zk.Bases.upload(bases).precompute(6, table_naf=table_kind)
    assert (zk.jac_to_affine(b.msm(scal)) == O.jac_to_affine(O.msm(bases, scal))).all()
    b.free()


def test_table_full_size_2_20_matches_plain_path(zk):
    """BASELINE config 2 size: the table-backed MSM returns the same point as the plain path (itself pinned against the
    oracle by test_full_size_2_20_closed_form)."""
    n = 1 << 20
    bases = zk.fixed_base_mul(aff_limbs(R.G1_GEN), random_fr_canonical(61, n), montgomery=False)
    s = random_fr_canonical(62, n)
    b = zk.Bases.upload(bases)
    plain = zk.jac_to_affine(b.msm(s, montgomery=False))
    b.precompute()
    assert (zk.jac_to_affine(b.msm(s, montgomery=False)) == plain).all()
    b.free()


def test_submit_collect_slots_in_flight(zk, oracle_lib, table_kind):
    """zkhip_msm_submit / zkhip_msm_collect: two MSMs in flight on two slots return what the blocking call returns;
    a busy slot refuses a second submit, an idle slot has nothing to collect."""
    O = oracle_lib
    n = 6000
    bases = zk.fixed_base_mul(aff_limbs(R.G1_GEN), random_fr_canonical(701, n), montgomery=False)
    b = zk.Bases.upload(bases).precompute(table_naf=table_kind)
    scal = [random_fr_canonical(710 + i, n) for i in range(3)]
    dev = [zk.DeviceBuffer(s) for s in scal]
    exp = [O.jac_to_affine(O.msm(bases, s)) for s in scal]
    b.msm_submit(dev[0].ptr, n, slot=0)
    b.msm_submit(dev[1].ptr, n, slot=1)
    with pytest.raises(zk.ZkhipError):
        b.msm_submit(dev[2].ptr, n, slot=1)
    assert (zk.jac_to_affine(zk.msm_collect(0)) == exp[0]).all()
    b.msm_submit(dev[2].ptr, n, slot=0)
    assert (zk.jac_to_affine(zk.msm_collect(1)) == exp[1]).all()
    assert (zk.jac_to_affine(zk.msm_collect(0)) == exp[2]).all()
    with pytest.raises(zk.ZkhipError):
        zk.msm_collect(0)
    # all eight slots in flight at once (the depth bench.py streams with), collected in submission order; the accumulation
    # launches' intervals on the device's time base are well-formed and as long as the reported duration
    for k in range(8):
        b.msm_submit(dev[k % 3].ptr, n, slot=k)
    with pytest.raises(zk.ZkhipError):
        b.msm_submit(dev[0].ptr, n, slot=8)
    for k in range(8):
        assert (zk.jac_to_affine(zk.msm_collect(k)) == exp[k % 3]).all()
        t0, t1 = zk.last_accumulate_interval()
        assert t1 > t0 > 0 and abs((t1 - t0) - zk.last_accumulate_ms()) < 0.05
    for d in dev:
        d.free()
    b.free()


def test_table_2_22_closed_form(zk, oracle_lib):
    """BASELINE config 4 size on one GPU (2^22 terms, window table with c = 21): sum s_i (k_i G) = (sum s_i k_i mod r) G
    against ONE oracle scalar multiplication - the size-independent check at the largest configured size."""
    O = oracle_lib
    n = 1 << 22
    g = aff_limbs(R.G1_GEN)
    ks = random_fr_canonical(81, n)
    bases = zk.fixed_base_mul(g, ks, montgomery=False)
    b = zk.Bases.upload(bases).precompute()
    del bases
    assert b.table_window == 21
    s = random_fr_canonical(82, n)
    to_int = lambda a: [int(x[0]) | int(x[1]) << 64 | int(x[2]) << 128 | int(x[3]) << 192 | int(x[4]) << 256 | int(x[5]) << 320
                        for x in a.tolist()]
    dot = sum(a * k for a, k in zip(to_int(s), to_int(ks))) % R.R_MOD
    got = zk.jac_to_affine(b.msm(s, montgomery=False))
    exp = O.jac_to_affine(O.scalar_mul(g, np.array(R.int_to_limbs(R.to_mont(dot, R.R_MOD, 6), 6), dtype=np.uint64)))
    assert (got == exp).all()
    b.free()


def test_headline_2_20_uniform_scalars_eight_in_flight_against_the_oracle(zk, oracle_lib):
    """BASELINE configs[1] as bench.py's headline runs it (this comparison lived in bench.py only): 2^20 bases with
    their window table resident, scalars UNIFORM in [0, r) as Montgomery residues, EIGHT MSMs in flight on a handle-owned stream
    (zkhip_msm_stream_*), every result compared with ONE libff-shaped multi_exp of the C oracle at the full size."""
    O = oracle_lib
    n = 1 << 20
    bases = zk.fixed_base_mul(aff_limbs(R.G1_GEN), random_fr_canonical(0x5EED, n), montgomery=False)
    scal = random_fr_uniform(0xABC0, n)
    exp = O.jac_to_affine(O.msm(bases, scal))
    b = zk.Bases.upload(bases).precompute()
    assert b.table_window == 20
    dev = zk.DeviceBuffer(scal)
    stream = zk.MsmStream(b, depth=8)
    tickets = [stream.submit(dev.ptr, n) for _ in range(8)]
    with pytest.raises(zk.ZkhipError):
        stream.submit(dev.ptr, n)
    for t in tickets:
        assert (zk.jac_to_affine(stream.collect(t)) == exp).all()
    stream.free(); dev.free(); b.free()


def test_the_library_measures_its_multiplier_peak_and_rebases_its_clock(zk):
    """zkhip_measure_fq_mul_rate: Fq products per second of this device, now (what bench.py's fq_mul_frac divides by: the rate differs
    between boxes of the same model) - a plausible figure for an MI355X; zkhip_reset_time_base: the accumulate intervals restart near zero."""
    rate = zk.measure_fq_mul_rate()
    assert 8e9 < rate < 40e9, rate
    bases = zk.fixed_base_mul(aff_limbs(R.G1_GEN), random_fr_canonical(801, 3000), montgomery=False)
    b = zk.Bases.upload(bases)
    stream = zk.MsmStream(b, depth=2)
    dev = zk.DeviceBuffer(random_fr_canonical(802, 3000))
    zk.reset_time_base()
    stream.collect(stream.submit(dev.ptr, 3000))
    t0, t1 = stream.last_accumulate_interval()
    assert 0 <= t0 < t1 < 5000.0                     # milliseconds since the reset, not since the library first planned an MSM
    assert abs((t1 - t0) - stream.last_accumulate_ms()) < 0.05
    stream.free(); dev.free(); b.free()
