// Modular inversion for the reduced-radix fields of fp29.cuh, written for a GPU lane.
//
// Why not Fermat: a^(p-2) is ~1,140 sequential 761-bit multiplications - 6.6 ms of latency for ONE lane on gfx950 and, worse,
// ~1,140 multiplier-bodies of issue slots for the whole wave.  The bucket accumulation (msm.hip, batched affine additions) shares
// one inversion among the m additions a lane performs, so the inversion must cost a few tens of multiplications, not a thousand.
//
// Algorithm: Bernstein-Yang division steps ("safegcd", delta = 1/2 variant) in batches of 29 steps: the 29 steps are decided on
// the low limbs only and summarised in a 2 x 2 integer matrix, which is then applied to the full-width f, g (exact division by
// 2^29) and, modulo p, to the Bezout coefficients d, e.  Everything is branch-free per lane (the 64 lanes of a wave run in
// lockstep: all invert at once for the price of one), ~55 k VALU instructions for Fq = ~30 multiplications' worth.
// Bound on the number of steps for a 761-bit modulus: (45907 * 761 + 26313) / 19929 = 1755 (Pornin, "Bounds on divsteps
// iterations"); 62 batches = 1798 steps.  The loop leaves early once g = 0 on every lane of the wave.
//
// This is the inversion the reference reaches through libff::Fp_model::inverse() inside mixed additions / to_affine
// (reached from libzecale/circuits/aggregator_circuit.tcc:168); libff is not in the reference tree, nothing here derives from it.
#pragma once
#include "fp29.cuh"

namespace zkhip {

// Signed little-endian numbers in N limbs of 29 bits: limbs 0 .. N-2 in [0, 2^29), the top limb carries the sign.
template <int N>
struct S29 {
  int32_t v[N];
};

// 29 division steps on the low limbs; returns the new zeta and the transition matrix t = [u v; q r]:
//   t * [f, g] = 2^29 * [f', g']
ZK_HD ZK_INL int32_t divsteps_29(int32_t zeta, uint32_t f0, uint32_t g0, int32_t& U, int32_t& V, int32_t& Q, int32_t& R) {
  uint32_t u = 1, v = 0, q = 0, r = 1;
  uint32_t f = f0, g = g0;
#pragma unroll
  for (int i = 0; i < 29; i++) {
    uint32_t c1 = (uint32_t)(zeta >> 31);            // all ones if zeta < 0
    uint32_t c2 = (uint32_t)0 - (g & 1u);            // all ones if g odd
    uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // conditionally negated f, u, v
    g += x & c2; q += y & c2; r += z & c2;
    c1 &= c2;
    zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;
    f += g & c1; u += q & c1; v += r & c1;
    g >>= 1;
    u <<= 1; v <<= 1;
  }
  U = (int32_t)u; V = (int32_t)v; Q = (int32_t)q; R = (int32_t)r;
  return zeta;
}

// (f, g) <- t * (f, g) / 2^29   (exact)
template <int N>
ZK_HD ZK_INL void update_fg_29(S29<N>& f, S29<N>& g, int32_t u, int32_t v, int32_t q, int32_t r) {
  int64_t cf = (int64_t)u * f.v[0] + (int64_t)v * g.v[0];
  int64_t cg = (int64_t)q * f.v[0] + (int64_t)r * g.v[0];
  cf >>= 29; cg >>= 29;                              // low 29 bits are zero by construction
#pragma unroll
  for (int i = 1; i < N; i++) {
    cf += (int64_t)u * f.v[i] + (int64_t)v * g.v[i];
    cg += (int64_t)q * f.v[i] + (int64_t)r * g.v[i];
    f.v[i - 1] = (int32_t)((uint32_t)cf & M29); cf >>= 29;
    g.v[i - 1] = (int32_t)((uint32_t)cg & M29); cg >>= 29;
  }
  f.v[N - 1] = (int32_t)cf;
  g.v[N - 1] = (int32_t)cg;
}

// (d, e) <- t * (d, e) / 2^29 mod p,  d, e kept in (-2p, p)
template <class PR>
ZK_HD ZK_INL void update_de_29(S29<PR::NL>& d, S29<PR::NL>& e, int32_t u, int32_t v, int32_t q, int32_t r) {
  constexpr int N = PR::NL;
  const int32_t sd = d.v[N - 1] >> 31, se = e.v[N - 1] >> 31;
  int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
  int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0];
  int64_t ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
  // multiples of p that make the low limb vanish (PINV = -p^-1 mod 2^29): md = PINV cd (mod 2^29), chosen in (md - 2^29, md]
  // so that the results stay in (-2p, p): u d' + v e' with d', e' in (-p, p) lies in (-2^29 p, 2^29 p), minus [0, 2^29) p
  md -= (int32_t)(((uint32_t)md - PR::PINV * (uint32_t)cd) & M29);
  me -= (int32_t)(((uint32_t)me - PR::PINV * (uint32_t)ce) & M29);
  cd += (int64_t)PR::P[0] * md;
  ce += (int64_t)PR::P[0] * me;
  cd >>= 29; ce >>= 29;
#pragma unroll
  for (int i = 1; i < N; i++) {
    cd += (int64_t)u * d.v[i] + (int64_t)v * e.v[i] + (int64_t)PR::P[i] * md;
    ce += (int64_t)q * d.v[i] + (int64_t)r * e.v[i] + (int64_t)PR::P[i] * me;
    d.v[i - 1] = (int32_t)((uint32_t)cd & M29); cd >>= 29;
    e.v[i - 1] = (int32_t)((uint32_t)ce & M29); ce >>= 29;
  }
  d.v[N - 1] = (int32_t)cd;
  e.v[N - 1] = (int32_t)ce;
}

constexpr int safegcd_batches(int nbits) { return ((45907 * nbits + 26313) / 19929 + 29 + 28) / 29; }   // bound + one spare batch

// a^-1 in Montgomery form for a in Montgomery form (lazily bounded: a < 2p, limbs normalised).  Result < 2p.  0 -> 0.
template <class PR>
ZK_HD ZK_INL Fp<PR> fp_inv(const Fp<PR>& a_in) {
  constexpr int N = PR::NL;
  const Fp<PR> a = fp_cond_sub_p(a_in);
  S29<N> f, g, d, e;
#pragma unroll
  for (int i = 0; i < N; i++) { f.v[i] = (int32_t)PR::P[i]; g.v[i] = (int32_t)a.l[i]; d.v[i] = 0; e.v[i] = 0; }
  e.v[0] = 1;
  int32_t zeta = -1;
#pragma unroll 1
  for (int it = 0; it < safegcd_batches(PR::NBITS); it++) {
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < N; i++) nz |= (uint32_t)g.v[i];
#if defined(__HIP_DEVICE_COMPILE__)
    if (__all(nz == 0)) break;                       // wave-uniform exit: every lane has reached g = 0
#else
    if (nz == 0) break;
#endif
    int32_t u, v, q, r;
    zeta = divsteps_29(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], u, v, q, r);
    update_de_29<PR>(d, e, u, v, q, r);
    update_fg_29<N>(f, g, u, v, q, r);
  }
  // g = 0, f = +-gcd = +-1 (or +-p for a = 0, where d = 0): inverse = sign(f) * d mod p
  const int32_t sf = f.v[N - 1] >> 31;               // all ones if f < 0
  // bring d from (-2p, p) to [0, p): add p if negative, negate if f < 0, add p if negative
  S29<N> t = d;
  auto cond_add_p = [&](S29<N>& x) {
    const int32_t neg = x.v[N - 1] >> 31;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) { int32_t s = x.v[i] + (int32_t)(PR::P[i] & (uint32_t)neg) + c; x.v[i] = s & (int32_t)M29; c = s >> 29; }
    x.v[N - 1] += (int32_t)(PR::P[N - 1] & (uint32_t)neg) + c;
  };
  cond_add_p(t);
  {
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) { int32_t s = ((t.v[i] ^ sf) - sf) + c; t.v[i] = s & (int32_t)M29; c = s >> 29; }
    t.v[N - 1] = ((t.v[N - 1] ^ sf) - sf) + c;
  }
  cond_add_p(t);
  Fp<PR> x;
#pragma unroll
  for (int i = 0; i < N; i++) x.l[i] = (uint32_t)t.v[i];
  // x = (a R)^-1 as an integer; a^-1 R = x * R^3 / R
  const Fp<PR> r2 = fp_const<PR>(PR::R2);
  return fp_mul(fp_mul(x, r2), r2);
}

}  // namespace zkhip
