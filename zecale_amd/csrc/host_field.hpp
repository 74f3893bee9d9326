// Host-side (CPU) field and curve arithmetic of the PRODUCT library: 64-bit limbs, CIOS Montgomery
// with unsigned __int128, same in-memory form as the C ABI (N64 limbs, radix 2^(64*N64)).
// Used for the short serial tails that do not belong on a throughput machine (the ~380 doublings
// of the final window combination of a one-shot MSM, affine normalisation of 1..5 result points,
// the 3 scalar multiplications by r, s of the Groth16 tail - reference: SURVEY 8(a) row a9) and,
// later, for setup / verification.  It is NOT a fallback for the HIP kernels: every function
// that needs the GPU fails with ZKHIP_ERR_NO_DEVICE when no gfx950 device is present.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "bw6_params.h"
#include "fp_inv.cuh"

namespace zkhip {
namespace host {

typedef unsigned __int128 u128;

template <class PR>
struct HF {
  static constexpr int N = PR::N64;
  uint64_t v[N];

  static HF zero() { HF r; memset(r.v, 0, sizeof r.v); return r; }
  static HF one() { HF r; for (int i = 0; i < N; i++) r.v[i] = PR::ONE64[i]; return r; }
  static HF from_limbs(const uint64_t* p) { HF r; memcpy(r.v, p, sizeof r.v); return r; }
  void to_limbs(uint64_t* p) const { memcpy(p, v, sizeof v); }
  bool is_zero() const { uint64_t o = 0; for (int i = 0; i < N; i++) o |= v[i]; return o == 0; }
  bool operator==(const HF& b) const { return memcmp(v, b.v, sizeof v) == 0; }
  bool operator!=(const HF& b) const { return !(*this == b); }

  static bool geq_p(const uint64_t* a) {
    for (int i = N - 1; i >= 0; i--) {
      if (a[i] > PR::P64[i]) return true;
      if (a[i] < PR::P64[i]) return false;
    }
    return true;
  }
  static void sub_p(uint64_t* a) {
    uint64_t br = 0;
    for (int i = 0; i < N; i++) {
      u128 t = (u128)a[i] - PR::P64[i] - br;
      a[i] = (uint64_t)t;
      br = (uint64_t)(t >> 64) & 1;
    }
  }
  HF operator+(const HF& b) const {
    HF r; uint64_t c = 0;
    for (int i = 0; i < N; i++) { u128 t = (u128)v[i] + b.v[i] + c; r.v[i] = (uint64_t)t; c = (uint64_t)(t >> 64); }
    if (c || geq_p(r.v)) sub_p(r.v);   // moduli leave >= 7 spare bits, c is always 0
    return r;
  }
  HF operator-(const HF& b) const {
    HF r; uint64_t br = 0;
    for (int i = 0; i < N; i++) { u128 t = (u128)v[i] - b.v[i] - br; r.v[i] = (uint64_t)t; br = (uint64_t)(t >> 64) & 1; }
    if (br) { uint64_t c = 0; for (int i = 0; i < N; i++) { u128 t = (u128)r.v[i] + PR::P64[i] + c; r.v[i] = (uint64_t)t; c = (uint64_t)(t >> 64); } }
    return r;
  }
  HF neg() const { return zero() - *this; }
  HF dbl() const { return *this + *this; }
  // CIOS Montgomery product, the form for a modulus with spare top bits (both moduli leave seven: the running value never needs an
  // extra carry word): per row one chain for a b_i and one for m p, fused, fully unrolled.  15-20 % faster than the textbook loop it
  // replaces (round 5) - the host witness generator spends two thirds of its time here (105 k products per proof section).
  HF operator*(const HF& b) const {
    uint64_t t[N];
    memset(t, 0, sizeof t);
#pragma GCC unroll 12
    for (int i = 0; i < N; i++) {
      const uint64_t bi = b.v[i];
      u128 x = (u128)v[0] * bi + t[0];
      uint64_t ca = (uint64_t)(x >> 64);                 // carry of the a b_i chain
      const uint64_t m = (uint64_t)x * PR::PINV64;
      u128 y = (u128)m * PR::P64[0] + (uint64_t)x;
      uint64_t cm = (uint64_t)(y >> 64);                 // carry of the m p chain
#pragma GCC unroll 12
      for (int j = 1; j < N; j++) {
        x = (u128)v[j] * bi + t[j] + ca;
        ca = (uint64_t)(x >> 64);
        y = (u128)m * PR::P64[j] + (uint64_t)x + cm;
        cm = (uint64_t)(y >> 64);
        t[j - 1] = (uint64_t)y;
      }
      t[N - 1] = ca + cm;                                // (no overflow: p < 2^(64 N - 7))
    }
    HF r; memcpy(r.v, t, sizeof r.v);
    if (geq_p(r.v)) sub_p(r.v);
    return r;
  }
  HF sqr() const { return (*this) * (*this); }
  // Lazy reduction for sums of products (the host witness generator's Fq12 products: 144 products feed 23 coefficients):
  //   mul_wide_add   acc (2N limbs) += a b, the plain 2N-limb product - half the work of a Montgomery product;
  //   redc_wide      T -> T / R mod p, fully reduced, for T < p R (R = 2^(64 N)): sums of up to 2^7 products of reduced elements fit
  //                  (the moduli leave seven spare bits); T is consumed.
  // sum_i a_i b_i / R through these = the sum of the Montgomery products a_i * b_i, the same fully reduced limbs.
  static inline void mul_wide_add(uint64_t* acc, const HF& a, const HF& b) {
#pragma GCC unroll 12
    for (int i = 0; i < N; i++) {
      const uint64_t bi = b.v[i];
      uint64_t c = 0;
#pragma GCC unroll 12
      for (int j = 0; j < N; j++) {
        u128 x = (u128)a.v[j] * bi + acc[i + j] + c;
        acc[i + j] = (uint64_t)x; c = (uint64_t)(x >> 64);
      }
      for (int k = i + N; c && k < 2 * N; k++) { u128 x = (u128)acc[k] + c; acc[k] = (uint64_t)x; c = (uint64_t)(x >> 64); }
    }
  }
  static inline HF redc_wide(uint64_t* T) {
#pragma GCC unroll 12
    for (int i = 0; i < N; i++) {
      const uint64_t m = T[i] * PR::PINV64;
      uint64_t c = 0;
#pragma GCC unroll 12
      for (int j = 0; j < N; j++) {
        u128 x = (u128)m * PR::P64[j] + T[i + j] + c;
        T[i + j] = (uint64_t)x; c = (uint64_t)(x >> 64);
      }
      for (int k = i + N; c && k < 2 * N; k++) { u128 x = (u128)T[k] + c; T[k] = (uint64_t)x; c = (uint64_t)(x >> 64); }
    }
    HF r; memcpy(r.v, T + N, sizeof r.v);
    if (geq_p(r.v)) sub_p(r.v);
    return r;
  }
  // canonical integer (little-endian limbs) of the represented value
  void to_canonical(uint64_t* out) const {
    HF o = zero(); o.v[0] = 1;
    HF r = (*this) * o;
    memcpy(out, r.v, sizeof r.v);
  }
  static HF from_canonical(const uint64_t* in) {
    HF x = from_limbs(in), r2;
    for (int i = 0; i < N; i++) r2.v[i] = PR::R2_64[i];
    return x * r2;
  }
  static HF from_u64(uint64_t x) { uint64_t l[N] = {0}; l[0] = x; return from_canonical(l); }
  HF pow_limbs(const uint64_t* e, int nlimbs) const {
    HF acc = one();
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
      acc = acc.sqr();
      if ((e[i / 64] >> (i % 64)) & 1) acc = acc * (*this);
    }
    return acc;
  }
  HF inv_fermat() const {  // a^(p-2)
    uint64_t e[N];
    memcpy(e, PR::P64, sizeof e);
    e[0] -= 2;  // p odd and p0 >= 2
    return pow_limbs(e, N);
  }
  // Inverse by Bernstein-Yang division steps (fp_inv.cuh, the routine the device uses, compiled for the host): 4.7 us for Fr
  // against 18 us for a binary extended Euclid on the limbs and 60 us for Fermat - the wrapping circuit's witness needs
  // hundreds of inversions per proof.  Input aR -> output a^-1 R.  0 -> 0.
  HF inv() const {
    if (is_zero()) return zero();
    uint64_t l[N], o[N];
    to_limbs(l);
    fp_to_abi<PR>(fp_inv<PR>(fp_from_abi<PR>(l)), o);
    return from_limbs(o);
  }
};

typedef HF<FqParams> HFq;
typedef HF<FrParams> HFr;

// Jacobian point on y^2 = x^3 + b over Fq (a = 0).  Infinity: Z = 0.
struct HJac {
  HFq X, Y, Z;
  static HJac infinity() { HJac p; p.X = HFq::zero(); p.Y = HFq::one(); p.Z = HFq::zero(); return p; }
  bool is_inf() const { return Z.is_zero(); }
  static HJac from_affine(const HFq& x, const HFq& y) {
    if (x.is_zero() && y.is_zero()) return infinity();   // ABI encoding of the point at infinity
    HJac p; p.X = x; p.Y = y; p.Z = HFq::one(); return p;
  }
  HJac dbl() const {
    if (is_inf()) return *this;
    // dbl-2009-l (a = 0)
    HFq A = X.sqr(), B = Y.sqr(), C = B.sqr();
    HFq D = ((X + B).sqr() - A - C).dbl();
    HFq E = A.dbl() + A, F = E.sqr();
    HJac r;
    r.X = F - D.dbl();
    r.Y = E * (D - r.X) - C.dbl().dbl().dbl();
    r.Z = (Y * Z).dbl();
    return r;
  }
  HJac add(const HJac& o) const {
    if (is_inf()) return o;
    if (o.is_inf()) return *this;
    HFq Z1Z1 = Z.sqr(), Z2Z2 = o.Z.sqr();
    HFq U1 = X * Z2Z2, U2 = o.X * Z1Z1;
    HFq S1 = Y * o.Z * Z2Z2, S2 = o.Y * Z * Z1Z1;
    if (U1 == U2) {
      if (S1 == S2) return dbl();
      return infinity();
    }
    HFq H = U2 - U1, I = H.dbl().sqr(), J = H * I, r = (S2 - S1).dbl(), V = U1 * I;
    HJac q;
    q.X = r.sqr() - J - V.dbl();
    q.Y = r * (V - q.X) - (S1 * J).dbl();
    q.Z = ((Z + o.Z).sqr() - Z1Z1 - Z2Z2) * H;
    return q;
  }
  HJac neg() const { HJac r = *this; r.Y = Y.neg(); return r; }
  // this + (x2, y2), the second point affine and finite (madd-2007-bl: 7 products and 4 squarings against add's 11 and 5)
  HJac add_affine(const HFq& x2, const HFq& y2) const {
    if (is_inf()) return from_affine(x2, y2);
    HFq Z1Z1 = Z.sqr();
    HFq U2 = x2 * Z1Z1, S2 = y2 * Z * Z1Z1;
    if (U2 == X) {
      if (S2 == Y) return dbl();
      return infinity();
    }
    HFq H = U2 - X, HH = H.sqr(), I = HH.dbl().dbl(), J = H * I, r = (S2 - Y).dbl(), V = X * I;
    HJac q;
    q.X = r.sqr() - J - V.dbl();
    q.Y = r * (V - q.X) - (Y * J).dbl();
    q.Z = (Z + H).sqr() - Z1Z1 - HH;
    return q;
  }
  // k * P for a canonical little-endian scalar: 4-bit fixed windows from the top (14 additions for the table, then four doublings
  // and at most one addition per window: ~100 additions instead of the ~190 of double-and-add on a 377-bit scalar)
  HJac mul_canonical(const uint64_t* k, int nlimbs) const {
    HJac tab[16];
    tab[0] = infinity(); tab[1] = *this;
    for (int d = 2; d < 16; d++) tab[d] = (d & 1) ? tab[d - 1].add(*this) : tab[d / 2].dbl();
    HJac acc = infinity();
    bool any = false;
    for (int w = nlimbs * 16 - 1; w >= 0; w--) {
      if (any) { acc = acc.dbl(); acc = acc.dbl(); acc = acc.dbl(); acc = acc.dbl(); }
      const unsigned d = (unsigned)(k[w / 16] >> ((w % 16) * 4)) & 15u;
      if (d) { acc = any ? acc.add(tab[d]) : tab[d]; any = true; }
    }
    return acc;
  }
  // affine (x, y); infinity -> (0, 0)
  void to_affine(HFq& x, HFq& y) const {
    if (is_inf()) { x = HFq::zero(); y = HFq::zero(); return; }
    HFq zi = Z.inv(), zi2 = zi.sqr();
    x = X * zi2; y = Y * zi2 * zi;
  }
};

// k P for a FIXED point P and many 377-bit scalars k (the prover's tail multiplies delta_1 and delta_2 of its key by r, s, r s in
// every proof: r1cs_gg_ppzksnark_prover, reached from aggregator_circuit.tcc:168): 48 windows of 8 bits, the 255 non-zero multiples
// d 2^(8 w) P of every window kept AFFINE (one batch inversion when the table is built: ~0.1 s, 4.7 MB per point) - a product is at
// most 48 mixed additions and no doubling, against 377 doublings and ~95 additions for the variable-base routine above.
struct FixedBase8 {
  static constexpr int W = 48;                 // ceil(377 / 8)
  std::vector<HFq> xy;                         // [w][d - 1] -> x, y
  bool base_inf = true;
  bool ok = true;                              // false: a multiple was the point at infinity (see build) - the table must not be used
  void build(const HJac& P) {
    base_inf = P.is_inf();
    ok = true;
    xy.assign((size_t)W * 255 * 2, HFq::zero());
    if (base_inf) return;
    std::vector<HJac> pts((size_t)W * 255);
    HJac step = P;                             // 2^(8 w) P
    for (int w = 0; w < W; w++) {
      HJac acc = step;
      for (int d = 1; d <= 255; d++) { pts[(size_t)w * 255 + d - 1] = acc; acc = acc.add(step); }
      step = acc;                              // 256 step
    }
    // batch normalisation: one inversion for all 48 x 255 multiples.  A multiple d 2^(8w) P (< r) of a point of prime order r is
    // never the point at infinity; a key read from a FILE is only checked to lie on the curve, and a delta with a small-order
    // component would put a zero Z into the running product and turn every entry into garbage: such a table is
    // marked unusable and the tail falls back to variable-base multiplications, which are correct for any point.
    for (const HJac& q : pts)
      if (q.is_inf()) { ok = false; xy.clear(); return; }
    std::vector<HFq> pre(pts.size());
    HFq run = HFq::one();
    for (size_t i = 0; i < pts.size(); i++) { pre[i] = run; run = run * pts[i].Z; }
    HFq inv = run.inv();
    for (size_t i = pts.size(); i-- > 0;) {
      const HFq zi = inv * pre[i];
      inv = inv * pts[i].Z;
      const HFq zi2 = zi.sqr();
      xy[i * 2] = pts[i].X * zi2; xy[i * 2 + 1] = pts[i].Y * zi2 * zi;
    }
  }
  // k: canonical little-endian limbs, below 2^384
  HJac mul(const uint64_t* k) const {
    HJac acc = HJac::infinity();
    if (base_inf) return acc;
    for (int w = 0; w < W; w++) {
      const unsigned d = (unsigned)(k[w / 8] >> ((w % 8) * 8)) & 255u;
      if (d) { const size_t i = ((size_t)w * 255 + d - 1) * 2; acc = acc.add_affine(xy[i], xy[i + 1]); }
    }
    return acc;
  }
};

}  // namespace host
}  // namespace zkhip
