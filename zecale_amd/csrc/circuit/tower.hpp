// Extension towers of the BLS12-377 base field, generic over the DSL's field type F (dsl.hpp):
//   Fq2  = Fq[u]  / (u^2 + 5)
//   Fq12 = Fq[w]  / (w^12 + 5),  u = w^6       (direct degree-12 extension: -5 is neither a square nor a cube in
//                                                Fq, q = 1 mod 12, and 5/4 is not a fourth power - checked in
//                                                tests/test_circuit_algebra.py against big integers)
// The direct representation makes the Frobenius maps coefficient-wise scalings by constants (free in a circuit)
// and lets a full Fq12 multiplication cost 23 constraints: c(X) = a(X) b(X) has degree 22, so it is pinned by
// a(x_k) b(x_k) = c(x_k) at 23 points, and the reduction modulo w^12 + 5 is linear.
#pragma once
#include "dsl.hpp"

namespace zkhip {
namespace ZK_CIRCUIT_NS {

template <class F>
struct Fq2 {
  F c0, c1;
  Fq2() : c0(f_zero<F>()), c1(f_zero<F>()) {}
  Fq2(const F& a, const F& b) : c0(a), c1(b) {}
  static Fq2 zero() { return Fq2(); }
  static Fq2 one() { return Fq2(f_one<F>(), f_zero<F>()); }
  static Fq2 constant(const HFr& a, const HFr& b) { return Fq2(F::constant(a), F::constant(b)); }
  static Fq2 witness(const HFr& a, const HFr& b) { return Fq2(F::witness(a), F::witness(b)); }
  Fq2 operator+(const Fq2& o) const { return Fq2(c0 + o.c0, c1 + o.c1); }
  Fq2 operator-(const Fq2& o) const { return Fq2(c0 - o.c0, c1 - o.c1); }
  Fq2 neg() const { return Fq2(c0.neg(), c1.neg()); }
  Fq2 dbl() const { return *this + *this; }
  Fq2 mulc(const HFr& k) const { return Fq2(c0.mulc(k), c1.mulc(k)); }
  Fq2 mul_base(const F& k) const { return Fq2(c0 * k, c1 * k); }     // by an element of Fq: 2 products
  Fq2 operator*(const Fq2& o) const {                                // Karatsuba: 3 products
    F v0 = c0 * o.c0, v1 = c1 * o.c1;
    F m = (c0 + c1) * (o.c0 + o.c1);
    return Fq2(v0 - v1.mulc(HFr::from_u64(5)), m - v0 - v1);
  }
  Fq2 sqr() const {                                                  // 2 products
    F t = c0 * c1;
    F s = (c0 + c1) * (c0 - c1.mulc(HFr::from_u64(5)));              // a0^2 - 5 a1^2 - 4 t
    return Fq2(s + t.mulc(HFr::from_u64(4)), t + t);
  }
  static void assert_eq(const Fq2& a, const Fq2& b) { F::assert_eq(a.c0, b.c0); F::assert_eq(a.c1, b.c1); }
};

// native value helpers on Fq2 (for witnesses)
struct V2 { HFr a, b; };
inline HFr fr_times5(const HFr& v) { HFr t = v + v; t = t + t; return t + v; }              // three additions, not a product
inline V2 v2_mul(const V2& x, const V2& y) {                                               // u^2 = -5: three products (Karatsuba)
  HFr v0 = x.a * y.a, v1 = x.b * y.b;
  return V2{v0 - fr_times5(v1), (x.a + x.b) * (y.a + y.b) - v0 - v1};
}
inline HFr v2_norm(const V2& x) { return x.a * x.a + fr_times5(x.b * x.b); }               // a^2 + 5 b^2
// ninv: the inverse of the norm if the caller already has it (batch_inv), else null
inline V2 v2_inv(const V2& x, const HFr* ninv = nullptr) {
  HFr ni;
  if (ninv) ni = *ninv;
  else {
    ni = fr_inv0(v2_norm(x));                 // (0 -> 0: the structure pass runs on zeros)
  }
  return V2{x.a * ni, (x.b * ni).neg()};
}
template <class F> inline V2 v2_of(const Fq2<F>& x) { return V2{x.c0.value(), x.c1.value()}; }

template <class F> inline bool fq2_is_const(const Fq2<F>& x) { return f_is_const(x.c0) && f_is_const(x.c1); }

// enforce x y = t for a given (linear) t: one auxiliary variable v1 = x1 y1 and
//   x0 y0 = t0 + 5 v1,   (x0 + x1)(y0 + y1) = t1 + t0 + 6 v1          (3 constraints, like a Karatsuba product)
template <class F> inline void fq2_assert_mul(const Fq2<F>& x, const Fq2<F>& y, const Fq2<F>& t) {
  F v1 = x.c1 * y.c1;
  if (witness_only<F>::value) return;                      // (the variable above is allocated either way)
  F::assert_product(x.c0, y.c0, t.c0 + v1.mulc(HFr::from_u64(5)));
  F::assert_product(x.c0 + x.c1, y.c0 + y.c1, t.c1 + t.c0 + v1.mulc(HFr::from_u64(6)));
}
// r = a b - off with r FRESH variables (see f_mul_minus): 3 constraints, 3 variables
template <class F> inline Fq2<F> fq2_mul_minus(const Fq2<F>& a, const Fq2<F>& b, const Fq2<F>& off) {
  if (fq2_is_const(a) || fq2_is_const(b)) return a * b - off;
  V2 p = v2_mul(v2_of(a), v2_of(b));
  Fq2<F> r = Fq2<F>::witness(p.a - off.c0.value(), p.b - off.c1.value());
  fq2_assert_mul(a, b, r + off);
  return r;
}
// r = a^2 - off with r FRESH: a0 a1 = t1 / 2,  (a0 + a1)(a0 - 5 a1) = t0 - 2 t1  with t = r + off     (2 constraints, 2 variables)
template <class F> inline Fq2<F> fq2_sqr_minus(const Fq2<F>& a, const Fq2<F>& off) {
  if (fq2_is_const(a)) return a.sqr() - off;
  V2 p = v2_mul(v2_of(a), v2_of(a));
  Fq2<F> r = Fq2<F>::witness(p.a - off.c0.value(), p.b - off.c1.value());
  Fq2<F> t = r + off;
  static const HFr half = HFr::from_u64(2).inv();
  F::assert_product(a.c0, a.c1, t.c1.mulc(half));
  F::assert_product(a.c0 + a.c1, a.c0 - a.c1.mulc(HFr::from_u64(5)), t.c0 - t.c1.mulc(HFr::from_u64(2)));
  return r;
}

// a / b in Fq2: witness q, enforce q b = a   (3 constraints, 3 variables)
template <class F> inline Fq2<F> fq2_div(const Fq2<F>& a, const Fq2<F>& b, const HFr* b_norm_inv = nullptr) {
  V2 q = v2_mul(v2_of(a), v2_inv(v2_of(b), b_norm_inv));
  if (fq2_is_const(a) && fq2_is_const(b)) return Fq2<F>::constant(q.a, q.b);
  Fq2<F> w = Fq2<F>::witness(q.a, q.b);
  fq2_assert_mul(w, b, a);
  return w;
}

// ---- Fq12 -----------------------------------------------------------------------------------------
struct Fq12Consts {
  HFr pw[23][23];      // pw[k][m] = k^m : evaluation points 0..22
  HFr frob[12];        // c1^i, c1 = (-5)^((q-1)/12): Frobenius scales coefficient i by frob[(k i) mod 12] for x -> x^(q^k)
  Fq12Consts() {
    for (int k = 0; k < 23; k++) {
      HFr x = HFr::from_u64((uint64_t)k), p = HFr::one();
      for (int m = 0; m < 23; m++) { pw[k][m] = p; p = p * x; }
    }
    // (q - 1) / 12 as limbs
    uint64_t e[6];
    memcpy(e, FrParams::P64, sizeof e);
    e[0] -= 1;
    uint64_t rem = 0;
    for (int i = 5; i >= 0; i--) {
      unsigned __int128 cur = ((unsigned __int128)rem << 64) | e[i];
      e[i] = (uint64_t)(cur / 12); rem = (uint64_t)(cur % 12);
    }
    HFr c1 = HFr::from_u64(5).neg().pow_limbs(e, 6);
    frob[0] = HFr::one();
    for (int i = 1; i < 12; i++) frob[i] = frob[i - 1] * c1;
  }
};
inline const Fq12Consts& fq12_consts() { static Fq12Consts c; return c; }

template <class F> struct Fq12;
template <class F> Fq12<F> fq12_mul_impl(const Fq12<F>& a, const Fq12<F>& b);

template <class F>
struct Fq12 {
  F c[12];
  Fq12() { for (auto& x : c) x = f_zero<F>(); }
  static Fq12 one() { Fq12 r; r.c[0] = f_one<F>(); return r; }
  Fq12 operator+(const Fq12& o) const { Fq12 r; for (int i = 0; i < 12; i++) r.c[i] = c[i] + o.c[i]; return r; }
  Fq12 operator-(const Fq12& o) const { Fq12 r; for (int i = 0; i < 12; i++) r.c[i] = c[i] - o.c[i]; return r; }
  Fq12 operator*(const Fq12& o) const { return fq12_mul_impl(*this, o); }
  Fq12 sqr() const { return fq12_mul_impl(*this, *this); }
  // x -> x^(q^k): coefficient-wise scaling (w^(q^k) = c1^k w)
  Fq12 frobenius(int k) const {
    Fq12 r;
    for (int i = 0; i < 12; i++) r.c[i] = c[i].mulc(fq12_consts().frob[(k * i) % 12]);
    return r;
  }
  Fq12 conjugate() const { return frobenius(6); }       // x^(q^6): odd coefficients negated
  static void assert_eq(const Fq12& a, const Fq12& b) { for (int i = 0; i < 12; i++) F::assert_eq(a.c[i], b.c[i]); }
};

// native product of coefficient vectors: full 23-coefficient product
#ifdef ZK_CIRCUIT_FR
// (the recording build: every product and sum is an instruction of the GPU generator's program; products by the structural zeros of a
//  line are folded by the recorder)
inline void v12_full_product(const HFr* a, const HFr* b, HFr* c /*23*/) {
  for (int i = 0; i < 23; i++) c[i] = HFr::zero();
  for (int i = 0; i < 12; i++)
    for (int j = 0; j < 12; j++) c[i + j] = c[i + j] + a[i] * b[j];
}
inline void v12_line_product(const HFr* a, const HFr* l, HFr* c /*23*/) { v12_full_product(a, l, c); }
inline void v12_square(const HFr* a, HFr* c /*23*/) { v12_full_product(a, a, c); }
#else
// Host generator (round 5): these products were 40 % of a witness - 378 per proof section, 144 Montgomery products each.  A
// coefficient of the product is a SUM of products: its terms are accumulated as plain 768-bit integers and reduced ONCE (lazy
// reduction: 36 limb products per term + 42 per coefficient instead of 78 per term); a LINE has five non-zero coefficients (w^0, w^1,
// w^3, w^7, w^9: 280 of the 378 products are by a line: 60 terms instead of 144); a square needs each cross term once.  The same
// field elements, fully reduced: the assignment is unchanged limb for limb (tests/test_aggregator_host.py, tests/test_witness_gpu.py).
inline void v12_full_product(const HFr* a, const HFr* b, HFr* c /*23*/) {
  for (int k = 0; k < 23; k++) {
    uint64_t acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int lo = k > 11 ? k - 11 : 0, hi = k < 11 ? k : 11;
    for (int i = lo; i <= hi; i++) HFr::mul_wide_add(acc, a[i], b[k - i]);
    c[k] = HFr::redc_wide(acc);
  }
}
// l: a line - non-zero at the indices below ONLY (line_at, bls12_377.hpp); c[21], c[22] are zero
inline void v12_line_product(const HFr* a, const HFr* l, HFr* c /*23*/) {
  static const int LI[5] = {0, 1, 3, 7, 9};
  for (int k = 0; k < 21; k++) {
    uint64_t acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j : LI) { const int i = k - j; if (i >= 0 && i < 12) HFr::mul_wide_add(acc, a[i], l[j]); }
    c[k] = HFr::redc_wide(acc);
  }
  c[21] = HFr::zero(); c[22] = HFr::zero();
}
inline void v12_square(const HFr* a, HFr* c /*23*/) {
  for (int k = 0; k < 23; k++) {
    uint64_t acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int lo = k > 11 ? k - 11 : 0, hi = k < 11 ? k : 11;
    for (int i = lo; 2 * i < k && i <= hi; i++) HFr::mul_wide_add(acc, a[i], a[k - i]);      // cross terms, once
    uint64_t carry = 0;                                                                      // ... doubled (at most 6 terms: no overflow)
    for (int t = 0; t < 12; t++) { const uint64_t w = acc[t]; acc[t] = (w << 1) | carry; carry = w >> 63; }
    if (!(k & 1) && k / 2 >= lo && k / 2 <= hi) HFr::mul_wide_add(acc, a[k / 2], a[k / 2]);
    c[k] = HFr::redc_wide(acc);
  }
}
#endif

template <> inline Fq12<NF> fq12_mul_impl<NF>(const Fq12<NF>& a, const Fq12<NF>& b) {
  HFr av[12], bv[12], cv[23];
  for (int i = 0; i < 12; i++) { av[i] = a.c[i].v; bv[i] = b.c[i].v; }
  v12_full_product(av, bv, cv);
  Fq12<NF> r;
  HFr five = HFr::from_u64(5);
  for (int i = 0; i < 12; i++) r.c[i] = NF(i + 12 < 23 ? cv[i] - cv[i + 12] * five : cv[i]);
  return r;
}

template <> inline Fq12<CV> fq12_mul_impl<CV>(const Fq12<CV>& a, const Fq12<CV>& b) {
  const Fq12Consts& K = fq12_consts();
  HFr av[12], bv[12], cv[23];
  for (int i = 0; i < 12; i++) { av[i] = a.c[i].val; bv[i] = b.c[i].val; }
  v12_full_product(av, bv, cv);
  CV cc[23];
  for (int m = 0; m < 23; m++) cc[m] = CV::witness(cv[m]);
  for (int k = 0; k < 23; k++) {
    CV ea, eb, ec;
    for (int i = 0; i < 12; i++) { ea = ea + a.c[i].mulc(K.pw[k][i]); eb = eb + b.c[i].mulc(K.pw[k][i]); }
    for (int m = 0; m < 23; m++) ec = ec + cc[m].mulc(K.pw[k][m]);
    CV::assert_product(ea, eb, ec);
  }
  Fq12<CV> r;
  HFr five = HFr::from_u64(5);
  for (int i = 0; i < 12; i++) r.c[i] = (i + 12 < 23) ? cc[i] - cc[i + 12].mulc(five) : cc[i];
  return r;
}

template <> inline Fq12<WV> fq12_mul_impl<WV>(const Fq12<WV>& a, const Fq12<WV>& b) {
  HFr av[12], bv[12], cv[23];
  for (int i = 0; i < 12; i++) { av[i] = a.c[i].val; bv[i] = b.c[i].val; }
  if (&a == &b) v12_square(av, cv);                      // Fq12::sqr()
  else v12_full_product(av, bv, cv);
  WV cc[23];
  for (int m = 0; m < 23; m++) cc[m] = WV::witness(cv[m]);       // same 23 allocations as the CV version
  Fq12<WV> r;
  HFr five = HFr::from_u64(5);
  for (int i = 0; i < 12; i++) r.c[i] = (i + 12 < 23) ? cc[i] - cc[i + 12].mulc(five) : cc[i];
  return r;
}

// a * l for a LINE l (non-zero coefficients at w^0, w^1, w^3, w^7, w^9 only: degree 9): the unreduced product has degree 20, so 21
// evaluation points pin it - two constraints and two variables fewer than a general multiplication (280 line multiplications per
// in-circuit verification).  The caller guarantees l.c[i] = 0 for i > 9 (structural zeros: never read here).
constexpr int FQ12_LINE_PTS = 21;
template <class F> Fq12<F> fq12_mul_line(const Fq12<F>& a, const Fq12<F>& l);
template <> inline Fq12<NF> fq12_mul_line<NF>(const Fq12<NF>& a, const Fq12<NF>& l) { return fq12_mul_impl<NF>(a, l); }
template <> inline Fq12<CV> fq12_mul_line<CV>(const Fq12<CV>& a, const Fq12<CV>& l) {
  const Fq12Consts& K = fq12_consts();
  HFr av[12], bv[12], cv[23];
  for (int i = 0; i < 12; i++) { av[i] = a.c[i].val; bv[i] = l.c[i].val; }
  v12_full_product(av, bv, cv);
  CV cc[FQ12_LINE_PTS];
  for (int m = 0; m < FQ12_LINE_PTS; m++) cc[m] = CV::witness(cv[m]);
  for (int k = 0; k < FQ12_LINE_PTS; k++) {
    CV ea, eb, ec;
    for (int i = 0; i < 12; i++) ea = ea + a.c[i].mulc(K.pw[k][i]);
    for (int i = 0; i <= 9; i++) eb = eb + l.c[i].mulc(K.pw[k][i]);
    for (int m = 0; m < FQ12_LINE_PTS; m++) ec = ec + cc[m].mulc(K.pw[k][m]);
    CV::assert_product(ea, eb, ec);
  }
  Fq12<CV> r;
  HFr five = HFr::from_u64(5);
  for (int i = 0; i < 12; i++) r.c[i] = (i + 12 < FQ12_LINE_PTS) ? cc[i] - cc[i + 12].mulc(five) : cc[i];
  return r;
}
template <> inline Fq12<WV> fq12_mul_line<WV>(const Fq12<WV>& a, const Fq12<WV>& l) {
  HFr av[12], bv[12], cv[23];
  for (int i = 0; i < 12; i++) { av[i] = a.c[i].val; bv[i] = l.c[i].val; }
  v12_line_product(av, bv, cv);
  WV cc[FQ12_LINE_PTS];
  for (int m = 0; m < FQ12_LINE_PTS; m++) cc[m] = WV::witness(cv[m]);       // same 21 allocations as the CV version
  Fq12<WV> r;
  HFr five = HFr::from_u64(5);
  for (int i = 0; i < 12; i++) r.c[i] = (i + 12 < FQ12_LINE_PTS) ? cc[i] - cc[i + 12].mulc(five) : cc[i];
  return r;
}

// ---- squaring in the cyclotomic subgroup (Granger-Scott) --------------------------------------------------------------
// After the easy part of the final exponentiation every element satisfies x^(q^4 - q^2 + 1) = 1, and its square needs only three
// squarings in Fq4.  Tower inside the direct representation: u = w^6 (Fq2 = Fq[u]), s = w^3 (Fq4 = Fq2[s], s^2 = u),
// Fq12 = Fq4[w]/(w^3 - s):  x = g0 + g1 w + g2 w^2,  g_k = a_k + a_(k+3) s,  a_j = c[j] + c[j+6] u.  Then
//   x^2 = (3 g0^2 - 2 conj g0) + (3 s g2^2 + 2 conj g1) w + (3 g1^2 - 2 conj g2) w^2        (conj: s -> -s)
// 9 squarings in Fq2 = 18 constraints instead of 23, and - as everywhere in a chain - the six Fq2 coefficients of the result are
// FRESH variables: the constraints read  a^2 = P, b^2 = Q, (a + b)^2 = P + Q + S  with P and S expressed through the outputs.
struct SmallConsts {       // 2, 3, 5 and their inverses in Montgomery form, computed once
  HFr two, three, five, half, third, fifth_neg, five_neg;
  SmallConsts() {
    two = HFr::from_u64(2); three = HFr::from_u64(3); five = HFr::from_u64(5);
    half = two.inv(); third = three.inv(); fifth_neg = five.inv().neg(); five_neg = five.neg();
  }
};
inline const SmallConsts& small_consts() { static SmallConsts c; return c; }
template <class F> inline Fq2<F> fq2_mul_u(const Fq2<F>& a) { return Fq2<F>(a.c1.mulc(small_consts().five_neg), a.c0); }       // u (c0 + c1 u)
template <class F> inline Fq2<F> fq2_div_u(const Fq2<F>& a) { return Fq2<F>(a.c1, a.c0.mulc(small_consts().fifth_neg)); }      // (c0 + c1 u) / u
// enforce a^2 = t for a linear t (2 constraints)
template <class F> inline void fq2_assert_sqr(const Fq2<F>& a, const Fq2<F>& t) {
  if (witness_only<F>::value) return;
  const SmallConsts& K = small_consts();
  F::assert_product(a.c0, a.c1, t.c1.mulc(K.half));
  F::assert_product(a.c0 + a.c1, a.c0 - a.c1.mulc(K.five), t.c0 - t.c1.mulc(K.two));
}
inline V2 v2_add(const V2& x, const V2& y) { return V2{x.a + y.a, x.b + y.b}; }
inline V2 v2_sub(const V2& x, const V2& y) { return V2{x.a - y.a, x.b - y.b}; }
inline V2 v2_dbl(const V2& x) { return V2{x.a + x.a, x.b + x.b}; }
inline V2 v2_tpl(const V2& x) { return V2{x.a + x.a + x.a, x.b + x.b + x.b}; }
inline V2 v2_mul_u(const V2& x) { return V2{fr_times5(x.b).neg(), x.a}; }

// One Fq4 squaring g = a + b s:  g^2 = (a^2 + u b^2) + (2 a b) s.  The caller supplies P (= a^2) and S (= 2 a b) as LINEAR
// expressions in its own fresh outputs; Q = b^2 is a fresh variable here.  6 constraints.
template <class F> inline Fq2<F> fq4_sqr_witness_q(const Fq2<F>& b) {
  V2 q = v2_mul(v2_of(b), v2_of(b));
  return Fq2<F>::witness(q.a, q.b);
}
template <class F> inline void fq4_sqr_assert(const Fq2<F>& a, const Fq2<F>& b, const Fq2<F>& P, const Fq2<F>& Q, const Fq2<F>& S) {
  fq2_assert_sqr(a, P);
  fq2_assert_sqr(b, Q);
  fq2_assert_sqr(a + b, P + Q + S);
}

template <class F> inline Fq12<F> fq12_cyclotomic_sqr(const Fq12<F>& x) {
  const HFr& third = small_consts().third;
  Fq2<F> a[6];
  for (int j = 0; j < 6; j++) a[j] = Fq2<F>(x.c[j], x.c[j + 6]);
  auto val = [](const Fq2<F>& t) { return v2_of(t); };
  Fq2<F> h[6];
  // native values of the three Fq4 squares
  auto sq = [&](const Fq2<F>& p, const Fq2<F>& q, V2& re, V2& im, V2& qq) {      // (p + q s)^2 = re + im s;  qq = q^2
    V2 pp = v2_mul(val(p), val(p)), pq = v2_mul(val(p), val(q));
    qq = v2_mul(val(q), val(q));
    re = v2_add(pp, v2_mul_u(qq)); im = v2_add(pq, pq);
  };
  V2 r0, i0, r1, i1, r2, i2, q0, q1, q2;
  sq(a[0], a[3], r0, i0, q0); sq(a[1], a[4], r1, i1, q1); sq(a[2], a[5], r2, i2, q2);
  constexpr bool W = witness_only<F>::value;            // values only: P and S below exist for the assertions alone
  // h0 = 3 g0^2 - 2 conj g0:  (3 r0 - 2 a0) + (3 i0 + 2 a3) s
  {
    V2 hx = v2_sub(v2_tpl(r0), v2_dbl(val(a[0]))), hy = v2_add(v2_tpl(i0), v2_dbl(val(a[3])));
    h[0] = Fq2<F>::witness(hx.a, hx.b); h[3] = Fq2<F>::witness(hy.a, hy.b);
    Fq2<F> Q = Fq2<F>::witness(q0.a, q0.b);                              // a3^2
    if (!W) {
      Fq2<F> P = (h[0] + a[0].dbl()).mulc(third) - fq2_mul_u(Q);         // a0^2 = (h0x + 2 a0)/3 - u Q
      Fq2<F> S = (h[3] - a[3].dbl()).mulc(third);                        // 2 a0 a3 = (h0y - 2 a3)/3
      fq4_sqr_assert(a[0], a[3], P, Q, S);
    }
  }
  // h2 = 3 g1^2 - 2 conj g2:  (3 r1 - 2 a2) + (3 i1 + 2 a5) s
  {
    V2 hx = v2_sub(v2_tpl(r1), v2_dbl(val(a[2]))), hy = v2_add(v2_tpl(i1), v2_dbl(val(a[5])));
    h[2] = Fq2<F>::witness(hx.a, hx.b); h[5] = Fq2<F>::witness(hy.a, hy.b);
    Fq2<F> Q = Fq2<F>::witness(q1.a, q1.b);                              // a4^2
    if (!W) {
      Fq2<F> P = (h[2] + a[2].dbl()).mulc(third) - fq2_mul_u(Q);
      Fq2<F> S = (h[5] - a[5].dbl()).mulc(third);
      fq4_sqr_assert(a[1], a[4], P, Q, S);
    }
  }
  // h1 = 3 s g2^2 + 2 conj g1:  s (r2 + i2 s) = u i2 + r2 s  ->  (3 u i2 + 2 a1) + (3 r2 - 2 a4) s
  {
    V2 hx = v2_add(v2_tpl(v2_mul_u(i2)), v2_dbl(val(a[1]))), hy = v2_sub(v2_tpl(r2), v2_dbl(val(a[4])));
    h[1] = Fq2<F>::witness(hx.a, hx.b); h[4] = Fq2<F>::witness(hy.a, hy.b);
    Fq2<F> Q = Fq2<F>::witness(q2.a, q2.b);                              // a5^2
    if (!W) {
      Fq2<F> P = (h[4] + a[4].dbl()).mulc(third) - fq2_mul_u(Q);         // a2^2 = (h1y + 2 a4)/3 - u Q
      Fq2<F> S = fq2_div_u(h[1] - a[1].dbl()).mulc(third);                // 2 a2 a5 = (h1x - 2 a1) / (3 u)
      fq4_sqr_assert(a[2], a[5], P, Q, S);
    }
  }
  Fq12<F> r;
  for (int j = 0; j < 6; j++) { r.c[j] = h[j].c0; r.c[j + 6] = h[j].c1; }
  return r;
}

// 1 / a: witness + a * inv = 1
template <class F> inline void v12_of(const Fq12<F>& a, HFr* out) { for (int i = 0; i < 12; i++) out[i] = a.c[i].value(); }

// native inverse in Fq12 by solving with the norm chain is overkill here: use a^(q^12 - 2) ... too slow;
// instead: inverse = conj-product trick over the quadratic tower Fq12 = Fq6[w] (w^2 = v, v = w^2 in our basis):
// a = e + o (even part e, odd part o);  a^-1 = (e - o) / (e^2 - o^2),  and e^2 - o^2 lies in the even subalgebra
// Fq6 = Fq[w^2]; repeat with Fq6 = Fq2[...]: simpler to do Gaussian elimination on the 12x12 multiplication matrix.
inline bool v12_inverse(const HFr* a, HFr* out) {
  // solve M x = e0 where column j of M is a * w^j reduced
  HFr M[12][13];
  HFr five = HFr::from_u64(5);
  for (int j = 0; j < 12; j++) {
    // a * w^j : shift coefficients by j with wrap  w^12 = -5
    for (int i = 0; i < 12; i++) {
      int d = i + j;
      HFr v = a[i];
      if (d >= 12) { d -= 12; v = (v * five).neg(); }
      M[d][j] = v;
    }
  }
  for (int i = 0; i < 12; i++) M[i][12] = (i == 0) ? HFr::one() : HFr::zero();
  for (int col = 0; col < 12; col++) {
    int piv = -1;
    for (int r = col; r < 12; r++) if (!M[r][col].is_zero()) { piv = r; break; }
    if (piv < 0) return false;
    if (piv != col) for (int k = 0; k < 13; k++) std::swap(M[piv][k], M[col][k]);
    HFr inv = M[col][col].inv();
    for (int k = 0; k < 13; k++) M[col][k] = M[col][k] * inv;
    for (int r = 0; r < 12; r++) {
      if (r == col || M[r][col].is_zero()) continue;
      HFr f = M[r][col];
      for (int k = 0; k < 13; k++) M[r][k] = M[r][k] - f * M[col][k];
    }
  }
  for (int i = 0; i < 12; i++) out[i] = M[i][12];
  return true;
}

template <class F> inline Fq12<F> fq12_inverse(const Fq12<F>& a) {
  HFr av[12], iv[12];
  v12_of(a, av);
  if (!v12_inverse(av, iv)) for (auto& x : iv) x = HFr::zero();
  Fq12<F> w;
  for (int i = 0; i < 12; i++) w.c[i] = F::witness(iv[i]);
  Fq12<F>::assert_eq(w * a, Fq12<F>::one());
  return w;
}

}  // namespace ZK_CIRCUIT_NS
}  // namespace zkhip
