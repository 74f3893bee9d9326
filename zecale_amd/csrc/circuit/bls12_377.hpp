// BLS12-377 (the nested curve, npp = other_curve<bw6_761_pp>: aggregator_server.cpp:48) generic over the DSL
// field type: group law in affine coordinates with witnessed slopes, optimal ate Miller loop, final
// exponentiation, and the Groth16 verification equation of the Clearmatics variant without gamma.
// Instantiated with NF this is a native verifier (pinned on testdata/dummy_app/{vk,extproof1..6}.json);
// instantiated with CV it is the in-circuit verifier the reference takes from libsnark
// (r1cs_gg_ppzksnark_*_gadget<bw6_761_pp>, groth16_verifier_parameters.hpp:16-27).
//
// E : y^2 = x^3 + 1 over Fq;  E' : y^2 = x^3 + 1/u over Fq2 (D-twist);  psi(x', y') = (x' w^2, y' w^3).
// Loop parameter u = 0x8508c00000000001 (r = u^4 - u^2 + 1).
#pragma once
#include <array>
#include "tower.hpp"

namespace zkhip {
namespace ZK_CIRCUIT_NS {

static const uint64_t BLS_U = 0x8508c00000000001ull;

template <class F> struct G1 { F x, y; };
template <class F> struct G2 { Fq2<F> x, y; };

// dinv (optional): the inverse of the slope's denominator, when the caller has batched it with others
template <class F> inline G1<F> g1_add(const G1<F>& p, const G1<F>& q, const HFr* dinv = nullptr) {       // p != +-q
  F lam = f_div(q.y - p.y, q.x - p.x, dinv);
  F x3 = f_mul_minus(lam, lam, p.x + q.x);
  F y3 = f_mul_minus(lam, p.x - x3, p.y);
  return G1<F>{x3, y3};
}
template <class F> inline G1<F> g1_dbl(const G1<F>& p, const HFr* dinv = nullptr) {
  F xx = p.x * p.x;
  F lam = f_div(xx + xx + xx, p.y + p.y, dinv);
  F x3 = f_mul_minus(lam, lam, p.x + p.x);
  F y3 = f_mul_minus(lam, p.x - x3, p.y);
  return G1<F>{x3, y3};
}
// Curve membership of a witnessed point (the reference's proof variables carry libsnark's G1 / G2 checker gadgets,
// groth16_verifier_parameters.hpp:16-27 -> r1cs_gg_ppzksnark_proof_variable): without it the chord-and-tangent arithmetic below
// would run on points of ANOTHER curve y^2 = x^3 + b' (invalid-curve setting) and the result bit would not be bound to a valid
// nested proof.   G1: y^2 = x^3 + 1  (3 constraints);   G2: y^2 = x^3 + 1/u over Fq2, 1/u = -u/5  (7 constraints).
template <class F> inline void g1_assert_on_curve(const G1<F>& p) {
  F xx = p.x * p.x, yy = p.y * p.y;
  F::assert_product(xx, p.x, yy - f_one<F>());
}
template <class F> inline void g2_assert_on_curve(const G2<F>& p) {
  Fq2<F> xx = p.x.sqr(), yy = p.y.sqr();
  Fq2<F> b(f_zero<F>(), F::constant(small_consts().fifth_neg));
  fq2_assert_mul(xx, p.x, yy - b);
}
template <class F> inline G1<F> g1_select(const F& bit, const G1<F>& a, const G1<F>& b) {
  return G1<F>{f_select(bit, a.x, b.x), f_select(bit, a.y, b.y)};
}

// ---- Miller loop over precomputed lines ---------------------------------------------------------------------------------
// The line through psi(T) with slope lambda' w, evaluated at P in G1, is  yP - lambda' xP w + (lambda' xT - yT) w^3.
// Everything that depends on the G2 argument only - the chain of points T, the slopes, b = lambda' xT - yT - is computed ONCE per
// G2 point (g2_precompute): 13 of the 15 constraints of a step.  The nested key's beta and delta are shared by all proofs of a
// batch (the reference precomputes the key the same way: aggregator_gadget.tcc:93), the generator's lines are constants.
template <class F> struct LineCoeffs { Fq2<F> lam, b; };
template <class F> using G2Lines = std::vector<LineCoeffs<F>>;          // Miller-schedule order: per bit of u a doubling, then an addition if set

// The inverse NORMS of the slope denominators of the schedule below for one G2 point Q, in schedule order, WITHOUT walking the affine
// chain (round 5): a Jacobian run of the same doublings and additions over Fq2 inverts nothing, and
//   2 y(T) = 2 Y / Z^3            so  1 / N(2 y(T))       = N(Z)^3 / N(2 Y)
//   x(Q) - x(T) = (x(Q) Z^2 - X) / Z^2   so  1 / N(x(Q) - x(T)) = N(Z)^2 / N(x(Q) Z^2 - X)         (N = the norm to Fq)
// - the 69 denominators N(2 Y), N(x(Q) Z^2 - X) of the schedule are then inverted TOGETHER (one inversion, a product tree).  The same
// field elements as the step-by-step inversions give (69 per proof section: a tenth of the host generator's time, and 69 dependent
// inversion levels of the GPU generator's program).  false: a denominator is zero (host build; the caller keeps the step-by-step form).
inline void batch_inv_tree(std::vector<HFr>& v);
struct JacV2 { V2 X, Y, Z; };
inline JacV2 jacv2_dbl(const JacV2& p) {                     // y^2 = x^3 + b over Fq2 (dbl-2009-l)
  V2 A = v2_mul(p.X, p.X), B = v2_mul(p.Y, p.Y), C = v2_mul(B, B);
  V2 t = v2_add(p.X, B);
  V2 D = v2_dbl(v2_sub(v2_sub(v2_mul(t, t), A), C));
  V2 E = v2_add(v2_add(A, A), A), Fv = v2_mul(E, E);
  JacV2 r;
  r.X = v2_sub(Fv, v2_dbl(D));
  r.Y = v2_sub(v2_mul(E, v2_sub(D, r.X)), v2_dbl(v2_dbl(v2_dbl(C))));
  r.Z = v2_dbl(v2_mul(p.Y, p.Z));
  return r;
}
inline bool g2_schedule_norm_invs(const V2& qx, const V2& qy, std::vector<HFr>& out) {
  JacV2 T{qx, qy, V2{HFr::one(), HFr::zero()}};
  std::vector<HFr> den, num;
  for (int i = 62; i >= 0; i--) {
    {
      const HFr nz = v2_norm(T.Z);
      den.push_back(v2_norm(v2_dbl(T.Y))); num.push_back(nz * nz * nz);
      T = jacv2_dbl(T);
    }
    if ((BLS_U >> i) & 1) {                                   // T + Q, Q affine (madd-2007-bl)
      const V2 ZZ = v2_mul(T.Z, T.Z);
      const V2 H = v2_sub(v2_mul(qx, ZZ), T.X);
      const HFr nz = v2_norm(T.Z);
      den.push_back(v2_norm(H)); num.push_back(nz * nz);
      const V2 S2 = v2_mul(qy, v2_mul(T.Z, ZZ));
      const V2 HH = v2_mul(H, H), I = v2_dbl(v2_dbl(HH)), Jv = v2_mul(H, I), r = v2_dbl(v2_sub(S2, T.Y)), V = v2_mul(T.X, I);
      JacV2 n;
      n.X = v2_sub(v2_sub(v2_mul(r, r), Jv), v2_dbl(V));
      n.Y = v2_sub(v2_mul(r, v2_sub(V, n.X)), v2_dbl(v2_mul(T.Y, Jv)));
      const V2 zh = v2_add(T.Z, H);
      n.Z = v2_sub(v2_sub(v2_mul(zh, zh), ZZ), HH);
      T = n;
    }
  }
#ifdef ZK_CIRCUIT_FR
  {
    std::vector<HFr> var;
    std::vector<size_t> where;
    for (size_t i = 0; i < den.size(); i++) {
      if (den[i].is_const()) den[i] = den[i].inv();          // (the generator's lines, an application's key: folded by the recorder)
      else { var.push_back(den[i]); where.push_back(i); }
    }
    if (!var.empty()) batch_inv_tree(var);
    for (size_t i = 0; i < where.size(); i++) den[where[i]] = var[i];
  }
#else
  for (const HFr& d : den) if (d.is_zero()) return false;
  batch_inv_tree(den);
#endif
  out.resize(den.size());
  for (size_t i = 0; i < den.size(); i++) out[i] = num[i] * den[i];
  return true;
}

// Walks the schedule for several G2 points in lock-step.  The slopes of one step need one Fq2 inversion each: their inverse norms
// come from g2_schedule_norm_invs (one inversion per POINT), or - a degenerate point, host build - from one inversion per step
// shared by the points (batch_inv).
template <class F> inline std::vector<G2Lines<F>> g2_precompute(const std::vector<G2<F>>& Qs) {
  const int n = (int)Qs.size();
  std::vector<G2<F>> T = Qs;
  std::vector<G2Lines<F>> out(n);
  HFr ninv[8];
  std::vector<std::vector<HFr>> pre(n);
  bool have_pre = true;
  for (int k = 0; k < n && have_pre; k++) have_pre = g2_schedule_norm_invs(v2_of(Qs[k].x), v2_of(Qs[k].y), pre[k]);
  size_t at = 0;
  for (int i = 62; i >= 0; i--) {
    if (have_pre) { for (int k = 0; k < n; k++) ninv[k] = pre[k][at]; at++; }
    else {
      for (int k = 0; k < n; k++) ninv[k] = v2_norm(v2_of(T[k].y + T[k].y));
      batch_inv(ninv, n);
    }
    for (int k = 0; k < n; k++) {
      Fq2<F> xx = T[k].x.sqr();
      Fq2<F> lam = fq2_div(xx + xx + xx, T[k].y + T[k].y, &ninv[k]);
      out[k].push_back(LineCoeffs<F>{lam, lam * T[k].x - T[k].y});
      Fq2<F> x3 = fq2_sqr_minus(lam, T[k].x + T[k].x);
      Fq2<F> y3 = fq2_mul_minus(lam, T[k].x - x3, T[k].y);
      T[k].x = x3; T[k].y = y3;
    }
    if ((BLS_U >> i) & 1) {
      if (have_pre) { for (int k = 0; k < n; k++) ninv[k] = pre[k][at]; at++; }
      else {
        for (int k = 0; k < n; k++) ninv[k] = v2_norm(v2_of(Qs[k].x - T[k].x));
        batch_inv(ninv, n);
      }
      for (int k = 0; k < n; k++) {
        Fq2<F> lam = fq2_div(Qs[k].y - T[k].y, Qs[k].x - T[k].x, &ninv[k]);
        out[k].push_back(LineCoeffs<F>{lam, lam * T[k].x - T[k].y});
        Fq2<F> x3 = fq2_sqr_minus(lam, T[k].x + Qs[k].x);
        Fq2<F> y3 = fq2_mul_minus(lam, T[k].x - x3, T[k].y);
        T[k].x = x3; T[k].y = y3;
      }
    }
  }
  return out;
}

template <class F> inline Fq12<F> line_at(const LineCoeffs<F>& lc, const G1<F>& P) {
  Fq12<F> l;
  Fq2<F> a = lc.lam.mul_base(P.x).neg();
  l.c[0] = P.y;
  l.c[1] = a.c0; l.c[7] = a.c1;
  l.c[3] = lc.b.c0; l.c[9] = lc.b.c1;
  return l;
}

template <class F> struct MillerPair { const G2Lines<F>* lines; G1<F> P; };

// prod_i f_{u,Q_i}(P_i): one shared accumulator (one squaring per bit for the whole product)
template <class F> inline Fq12<F> multi_miller_loop(const std::vector<MillerPair<F>>& ps) {
  Fq12<F> f = Fq12<F>::one();
  bool first = true;
  size_t idx = 0;
  for (int i = 62; i >= 0; i--) {
    if (!first) f = f.sqr();
    for (const auto& s : ps) { Fq12<F> l = line_at((*s.lines)[idx], s.P); f = first ? l : fq12_mul_line(f, l); first = false; }
    idx++;
    if ((BLS_U >> i) & 1) {
      for (const auto& s : ps) f = fq12_mul_line(f, line_at((*s.lines)[idx], s.P));
      idx++;
    }
  }
  return f;
}

// x^u for x in the cyclotomic subgroup (every use is after the easy part of the final exponentiation)
template <class F> inline Fq12<F> exp_by_u(const Fq12<F>& x) {
  Fq12<F> acc = x;
  for (int i = 62; i >= 0; i--) {
    acc = fq12_cyclotomic_sqr(acc);
    if ((BLS_U >> i) & 1) acc = acc * x;
  }
  return acc;
}

// f^(3 (q^12 - 1) / r): easy part (q^6 - 1)(q^2 + 1), then 3 (q^4 - q^2 + 1)/r = l0 + l1 q + l2 q^2 + l3 q^3 with
// l3 = (u-1)^2, l2 = l3 u, l1 = l2 u - l3, l0 = l1 u + 3  (identity checked in tests against big integers).
// The factor 3 is coprime to r: the cube of the reduced ate pairing is as good a pairing for an "== 1" check.
template <class F> inline Fq12<F> final_exponentiation(const Fq12<F>& f) {
  Fq12<F> f1 = f.conjugate() * fq12_inverse(f);
  Fq12<F> f2 = f1.frobenius(2) * f1;
  Fq12<F> t0 = exp_by_u(f2) * f2.conjugate();          // f2^(u-1)      (inverse = conjugate in the cyclotomic subgroup)
  Fq12<F> t1 = exp_by_u(t0) * t0.conjugate();          // f2^l3
  Fq12<F> t2 = exp_by_u(t1);                           // f2^l2
  Fq12<F> t3 = exp_by_u(t2) * t1.conjugate();          // f2^l1
  Fq12<F> t4 = exp_by_u(t3) * (fq12_cyclotomic_sqr(f2) * f2);   // f2^l0
  return t4 * t3.frobenius(1) * t2.frobenius(2) * t1.frobenius(3);
}

// [x == 1] in Fq12 as a field element (1 / 0)
template <class F> inline F fq12_is_one(const Fq12<F>& x) {
  F acc = f_is_zero(x.c[0] - f_one<F>());
  for (int i = 1; i < 12; i++) acc = acc * f_is_zero(x.c[i]);
  return acc;
}

// libff's bls12_377_G2::G2_one (not in the reference tree; confirmed by the nested fixtures: see oracle/pyref.py BLS_G2_GEN)
struct BlsG2Gen {
  HFr x0, x1, y0, y1;
  static HFr dec(const char* s) {
    HFr acc = HFr::zero(), ten = HFr::from_u64(10);
    for (; *s; s++) acc = acc * ten + HFr::from_u64((uint64_t)(*s - '0'));
    return acc;
  }
  BlsG2Gen() {
    x0 = dec("111583945774695116443911226257823823434468740249883042837745151039122196680777376765707574547389190084887628324746");
    x1 = dec("129066980656703085518157301154335215886082112524378686555873161080604845924984124025594590925548060469686767592854");
    y0 = dec("168863299724668977183029941347596462608978380503965103341003918678547611204475537878680436662916294540335494194722");
    y1 = dec("233892497287475762251335351893618429603672921469864392767514552093535653615809913098097380147379993375817193725968");
  }
};
inline const BlsG2Gen& bls_g2_gen() { static BlsG2Gen g; return g; }

// dbl_chain (host generator of a REGISTERED application, aggregator.cpp: AppHost): the doubling chains 2^j ABC_k of the input
// accumulators depend on the key alone - [k][j] holds the values of the four variables doubling j of input k allocates
// (x^2, slope, x', y'), computed once per application; the generator then allocates them without computing them.
using DblChain = std::vector<std::vector<std::array<HFr, 4>>>;
template <class F> struct NestedVk {
  G1<F> alpha; G2<F> beta, delta; std::vector<G1<F>> abc;
  G2Lines<F> neg_beta_lines, neg_delta_lines;       // vk_precompute: shared by every proof verified under this key
  const DblChain* dbl_chain = nullptr;
};
template <class F> struct NestedProof { G1<F> a; G2<F> b; G1<F> c; };

template <class F> inline G2<F> g2_neg(const G2<F>& p) { return G2<F>{p.x, p.y.neg()}; }

// The recorded witness program (witness_tape.cpp) must not contain the accumulator's chain of 2 x 253 dependent inversions - an
// inversion costs a GPU lane what thirty multiplications do and a level of the program waits for it.  The denominators of the
// slopes are therefore derived from a Jacobian run of the same two chains (the doublings pw_j = 2^j ABC_k, the conditional
// additions acc_j), where nothing is inverted:
//   x(pw_j) - x(acc_j) = H_j / (Za Zp)^2      so  1 / (...) = (Za Zp)^3 / Zs_j       (Zs_j = Za Zp H_j, the Z of the Jacobian sum)
//   2 y(pw_j)          = 2 Yp / Zp^3          so  1 / (...) = Zp^4 / Zp_{j+1}        (Zp_{j+1} = 2 Yp Zp, the Z of the double)
// and every Zs_j, Zp_{j+1} of the accumulator is inverted in ONE inversion (a product tree).  The same field elements as the
// affine chain computes, by another route.  Round 5: the HOST generator takes this route too (253 inversions of 4.7 us per input and
// proof section were a fifth of its time: one inversion and ~7 k products instead) - unless a Z of the run is zero (degenerate
// points: pw = +-acc), where it keeps the affine chain with its zero-stays-zero inversions.
struct JacH { HFr X, Y, Z; };
inline JacH jac_dbl(const JacH& p) {                       // y^2 = x^3 + b
  HFr A = p.X * p.X, B = p.Y * p.Y, C = B * B;
  HFr t = p.X + B;
  HFr D = t * t - A - C; D = D + D;
  HFr E = A + A + A, F = E * E;
  JacH r;
  r.X = F - (D + D);
  HFr C8 = C + C; C8 = C8 + C8; C8 = C8 + C8;
  r.Y = E * (D - r.X) - C8;
  HFr yz = p.Y * p.Z;
  r.Z = yz + yz;
  return r;
}
inline JacH jac_add(const JacH& p, const JacH& q) {        // p != +-q (else Z = 0: the inversion below meets zero and says so)
  HFr z1z1 = p.Z * p.Z, z2z2 = q.Z * q.Z;
  HFr U1 = p.X * z2z2, U2 = q.X * z1z1;
  HFr S1 = p.Y * (q.Z * z2z2), S2 = q.Y * (p.Z * z1z1);
  HFr H = U2 - U1, R = S2 - S1;
  HFr HH = H * H, HHH = H * HH, V = U1 * HH;
  JacH r;
  r.X = R * R - HHH - (V + V);
  r.Y = R * (V - r.X) - S1 * HHH;
  r.Z = (p.Z * q.Z) * H;
  return r;
}
inline void batch_inv_tree(std::vector<HFr>& v) {          // none of them zero
  std::vector<std::vector<HFr>> lv;
  lv.push_back(v);
  while (lv.back().size() > 1) {
    const std::vector<HFr>& c = lv.back();
    std::vector<HFr> n;
    for (size_t i = 0; i + 1 < c.size(); i += 2) n.push_back(c[i] * c[i + 1]);
    if (c.size() & 1) n.push_back(c.back());
    lv.push_back(std::move(n));
  }
  std::vector<HFr> inv = {lv.back()[0].inv()};
  for (size_t L = lv.size() - 1; L-- > 0;) {
    const std::vector<HFr>& c = lv[L];
    std::vector<HFr> ni(c.size());
    for (size_t i = 0; i + 1 < c.size(); i += 2) { ni[i] = inv[i / 2] * c[i + 1]; ni[i + 1] = inv[i / 2] * c[i]; }
    if (c.size() & 1) ni[c.size() - 1] = inv[c.size() / 2];
    inv.swap(ni);
  }
  v.swap(inv);
}
// p + (x2, y2), the second point affine (madd-2007-bl); p != +-q as in jac_add
inline JacH jac_madd(const JacH& p, const HFr& x2, const HFr& y2) {
  HFr z1z1 = p.Z * p.Z;
  HFr U2 = x2 * z1z1, S2 = y2 * (p.Z * z1z1);
  HFr H = U2 - p.X, R = S2 - p.Y;
  HFr HH = H * H, HHH = H * HH, V = p.X * HH;
  JacH r;
  r.X = R * R - HHH - (V + V);
  r.Y = R * (V - r.X) - p.Y * HHH;
  r.Z = p.Z * H;
  return r;
}
#ifndef ZK_CIRCUIT_FR
// The addition denominators alone, for a generator that has the doubling chain from its application (NestedVk::dbl_chain): the powers
// pw_j are affine points it already knows, so the Jacobian run is the accumulator's only (mixed additions) and half as long a product
// tree.  1 / (x(pw_j) - x(acc_j)) = Za^3 / Zs_j with Zs_j = Za H_j.  Empty: a degenerate step (the caller keeps the affine chain).
template <class F> inline std::vector<std::vector<HFr>> accumulator_add_denominators(const NestedVk<F>& vk, const std::vector<std::vector<F>>& input_bits) {
  std::vector<std::vector<HFr>> out(input_bits.size());
  std::vector<HFr> zs, num;
  const DblChain& ch = *vk.dbl_chain;
  JacH acc{vk.abc[0].x.value(), vk.abc[0].y.value(), HFr::one()};
  for (size_t k = 0; k < input_bits.size(); k++) {
    HFr px = vk.abc[k + 1].x.value(), py = vk.abc[k + 1].y.value();
    for (size_t j = 0; j < input_bits[k].size(); j++) {
      JacH s = jac_madd(acc, px, py);
      zs.push_back(s.Z); num.push_back(acc.Z * acc.Z * acc.Z);
      const HFr b = input_bits[k][j].value();
      acc = JacH{acc.X + b * (s.X - acc.X), acc.Y + b * (s.Y - acc.Y), acc.Z + b * (s.Z - acc.Z)};
      if (j + 1 < input_bits[k].size()) { px = ch[k][j][2]; py = ch[k][j][3]; }
    }
  }
  for (const HFr& zv : zs) if (zv.is_zero()) return {};
  batch_inv_tree(zs);
  size_t at = 0;
  for (size_t k = 0; k < input_bits.size(); k++) {
    out[k].resize(input_bits[k].size());
    for (size_t j = 0; j < input_bits[k].size(); j++) { out[k][j] = num[at] * zs[at]; at++; }
  }
  return out;
}
// the chain itself, once per application: plain values (the inversions of 252 doublings per input are paid once); empty: degenerate
inline DblChain doubling_chain_values(const std::vector<std::array<HFr, 2>>& abc_from_1, size_t bits) {
  DblChain ch(abc_from_1.size());
  for (size_t k = 0; k < abc_from_1.size(); k++) {
    HFr x = abc_from_1[k][0], y = abc_from_1[k][1];
    for (size_t j = 0; j + 1 < bits; j++) {
      if ((y + y).is_zero()) return {};                      // a point of order two in the key: no chain, the generator computes as ever
      const HFr xx = x * x, lam = (xx + xx + xx) * (y + y).inv();
      const HFr x3 = lam * lam - (x + x), y3 = lam * (x - x3) - y;
      ch[k].push_back({xx, lam, x3, y3});
      x = x3; y = y3;
    }
  }
  return ch;
}
#endif
// dinv[k][j] = {1 / (x(pw) - x(acc)), 1 / (2 y(pw))} at step j of input k
template <class F> inline std::vector<std::vector<std::array<HFr, 2>>> accumulator_denominators(const NestedVk<F>& vk, const std::vector<std::vector<F>>& input_bits) {
  std::vector<std::vector<std::array<HFr, 2>>> out(input_bits.size());
  std::vector<HFr> zs, num;                                 // the Z to invert and what its inverse is multiplied by, in step order
  JacH acc{vk.abc[0].x.value(), vk.abc[0].y.value(), HFr::one()};
  for (size_t k = 0; k < input_bits.size(); k++) {
    JacH pw{vk.abc[k + 1].x.value(), vk.abc[k + 1].y.value(), HFr::one()};
    for (size_t j = 0; j < input_bits[k].size(); j++) {
      const bool more = j + 1 < input_bits[k].size();
      JacH s = jac_add(acc, pw);
      HFr zz = acc.Z * pw.Z;
      zs.push_back(s.Z); num.push_back(zz * zz * zz);
      const HFr b = input_bits[k][j].value();
      acc = JacH{acc.X + b * (s.X - acc.X), acc.Y + b * (s.Y - acc.Y), acc.Z + b * (s.Z - acc.Z)};
      if (more) {
        JacH d = jac_dbl(pw);
        HFr z2 = pw.Z * pw.Z;
        zs.push_back(d.Z); num.push_back(z2 * z2);
        pw = d;
      }
    }
  }
  // With an application's key folded into the recording (witness_tape_build(fixed_vk)) the doubling chain 2^j ABC_k is made of
  // CONSTANTS: their Z are inverted where they stand (folded by the recorder: no instruction) and only the accumulator's own Z - which
  // depend on the proofs' inputs - share the one inversion.  Mixed into the same product tree the constants would come back as
  // computed values, and the doubling chain's variables (2,016 of the batch-2 circuit's) would not fold.  Generic recording: no constants.
#ifdef ZK_CIRCUIT_FR
  {
    std::vector<HFr> var;
    std::vector<size_t> where;
    for (size_t i = 0; i < zs.size(); i++) {
      if (zs[i].is_const()) zs[i] = zs[i].inv();
      else { var.push_back(zs[i]); where.push_back(i); }
    }
    if (!var.empty()) batch_inv_tree(var);
    for (size_t i = 0; i < where.size(); i++) zs[where[i]] = var[i];
  }
#else
  for (const HFr& zv : zs) if (zv.is_zero()) return {};          // a degenerate step: the caller keeps the affine chain
  batch_inv_tree(zs);
#endif
  size_t at = 0;
  for (size_t k = 0; k < input_bits.size(); k++) {
    out[k].resize(input_bits[k].size());
    for (size_t j = 0; j < input_bits[k].size(); j++) {
      out[k][j][0] = num[at] * zs[at]; at++;
      if (j + 1 < input_bits[k].size()) { out[k][j][1] = num[at] * zs[at]; at++; }
    }
  }
  return out;
}

// acc = ABC_0 + sum_j bits_j (2^j ABC_1) ...: one input, bits little-endian
template <class F> inline G1<F> input_accumulator(const NestedVk<F>& vk, const std::vector<std::vector<F>>& input_bits) {
#ifndef ZK_CIRCUIT_FR
  if constexpr (witness_only<F>::value) {
    // a registered application's generator: the doubling chain comes from the handle (its variables are allocated with the values
    // kept there - four per doubling, in g1_dbl's order), the accumulator's additions are all that is computed
    if (vk.dbl_chain && vk.dbl_chain->size() == input_bits.size()) {
      const auto den = accumulator_add_denominators(vk, input_bits);
      if (!den.empty() || input_bits.empty()) {
        const DblChain& ch = *vk.dbl_chain;
        G1<F> acc = vk.abc[0];
        for (size_t k = 0; k < input_bits.size(); k++) {
          G1<F> pw = vk.abc[k + 1];
          for (size_t j = 0; j < input_bits[k].size(); j++) {
            G1<F> s = g1_add(acc, pw, &den[k][j]);
            acc = g1_select(input_bits[k][j], s, acc);
            if (j + 1 < input_bits[k].size()) {
              const std::array<HFr, 4>& c = ch[k][j];
              (void)F::witness(c[0]); (void)F::witness(c[1]);
              pw = G1<F>{F::witness(c[2]), F::witness(c[3])};
            }
          }
        }
        return acc;
      }
    }
  }
#endif
  const auto den = accumulator_denominators(vk, input_bits);       // (host build: empty when a step is degenerate)
#ifdef ZK_CIRCUIT_FR
  const bool have_den = true;
#else
  const bool have_den = !den.empty() || input_bits.empty();
#endif
  G1<F> acc = vk.abc[0];
  for (size_t k = 0; k < input_bits.size(); k++) {
    G1<F> pw = vk.abc[k + 1];
    for (size_t j = 0; j < input_bits[k].size(); j++) {
      const bool more = j + 1 < input_bits[k].size();
      HFr dinv[2];
      if (have_den) { dinv[0] = den[k][j][0]; dinv[1] = den[k][j][1]; }
      else {
        dinv[0] = (pw.x - acc.x).value(); dinv[1] = (pw.y + pw.y).value();      // denominators of the addition and of the doubling
        batch_inv(dinv, more ? 2 : 1);
      }
      G1<F> s = g1_add(acc, pw, &dinv[0]);
      acc = g1_select(input_bits[k][j], s, acc);
      if (more) pw = g1_dbl(pw, &dinv[1]);
    }
  }
  return acc;
}

// 1 / 0 : e(A,B) e(acc,-G2_one) e(alpha,-beta) e(C,-delta) == 1
// (reference: the check the online verifier gadget performs; result is NOT enforced to be 1:
//  aggregator_circuit.hpp:51-54)
// the part of a verification that depends on the key only (libsnark's "processed verification key")
template <class F> inline void vk_precompute(NestedVk<F>& vk) {
  std::vector<G2Lines<F>> l = g2_precompute<F>({g2_neg(vk.beta), g2_neg(vk.delta)});
  vk.neg_beta_lines = std::move(l[0]);
  vk.neg_delta_lines = std::move(l[1]);
}

template <class F> inline void proof_assert_well_formed(const NestedProof<F>& pr) {
  g1_assert_on_curve(pr.a);
  g2_assert_on_curve(pr.b);
  g1_assert_on_curve(pr.c);
}

template <class F> inline F groth16_verify_bit(const NestedVk<F>& vk, const NestedProof<F>& pr, const G1<F>& acc) {
  if (vk.neg_beta_lines.empty()) throw std::runtime_error("vk_precompute has not run");
  const BlsG2Gen& g = bls_g2_gen();
  G2<F> gen{Fq2<F>::constant(g.x0, g.x1), Fq2<F>::constant(g.y0, g.y1)};
  std::vector<G2Lines<F>> own = g2_precompute<F>({pr.b, g2_neg(gen)});       // the proof's B; the generator's lines are constants
  std::vector<MillerPair<F>> ps = {{&own[0], pr.a}, {&own[1], acc}, {&vk.neg_beta_lines, vk.alpha}, {&vk.neg_delta_lines, pr.c}};
  Fq12<F> f = multi_miller_loop(ps);
  return fq12_is_one(final_exponentiation(f));
}

}  // namespace ZK_CIRCUIT_NS
}  // namespace zkhip
