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
  HF operator*(const HF& b) const {
    uint64_t t[N + 2];
    memset(t, 0, sizeof t);
    for (int i = 0; i < N; i++) {
      uint64_t c = 0;
      for (int j = 0; j < N; j++) {
        u128 x = (u128)v[j] * b.v[i] + t[j] + c;
        t[j] = (uint64_t)x; c = (uint64_t)(x >> 64);
      }
      u128 x = (u128)t[N] + c; t[N] = (uint64_t)x; t[N + 1] = (uint64_t)(x >> 64);
      uint64_t m = t[0] * PR::PINV64;
      x = (u128)m * PR::P64[0] + t[0]; c = (uint64_t)(x >> 64);
      for (int j = 1; j < N; j++) {
        x = (u128)m * PR::P64[j] + t[j] + c;
        t[j - 1] = (uint64_t)x; c = (uint64_t)(x >> 64);
      }
      x = (u128)t[N] + c; t[N - 1] = (uint64_t)x; t[N] = t[N + 1] + (uint64_t)(x >> 64);
    }
    HF r; memcpy(r.v, t, sizeof r.v);
    if (t[N] || geq_p(r.v)) sub_p(r.v);
    return r;
  }
  HF sqr() const { return (*this) * (*this); }
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

}  // namespace host
}  // namespace zkhip
