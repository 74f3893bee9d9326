// Group law for G1 / G2 of BW6-761 on the device.  Both groups live on curves y^2 = x^3 + b over
// the SAME base field Fq (G1: b = -1, G2: b = 4, reference testdata/dummy_app/aggregator_vk.json
// has single-Fq coordinates for both); the formulas below never use b (a = 0), so one code path
// serves both groups.
//
// Coordinates: extended Jacobian "XYZZ" (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2), infinity <=> ZZ = 0.
//   mixed add  (XYZZ + affine): 8M + 2S      full add (XYZZ + XYZZ): 12M + 2S     double: 6M + 4S
// Lazy bounds (multiples of p) are tracked in the comments: [k] means value < k*p.  Every fp_mul
// output is [2]; stored points keep X [10], Y [4], ZZ [2], ZZZ [2].
#pragma once
#include "fp29.cuh"

namespace zkhip {

struct XYZZ {
  Fq X, Y, ZZ, ZZZ;
};

// affine point in device Montgomery form, 24 packed 32-bit words per coordinate (canonical < p);
// the all-zero pattern is the point at infinity ((0,0) is on neither curve).
struct AffPacked {
  uint32_t x[24];
  uint32_t y[24];
};
static_assert(sizeof(AffPacked) == 192, "packed affine point is 192 bytes");

struct AffineDev {
  Fq x, y;
  bool inf;
};

ZK_HD ZK_INL XYZZ xyzz_infinity() {
  XYZZ r;
  r.X = fp_zero<FqParams>();
  r.Y = fp_zero<FqParams>();
  r.ZZ = fp_zero<FqParams>();
  r.ZZZ = fp_zero<FqParams>();
  return r;
}

// ZZ is kept as a multiplication output ([2]) or exact 0, so the test is exact.
ZK_HD ZK_INL bool xyzz_is_inf(const XYZZ& p) { return fp_is_zero_2p(p.ZZ); }

ZK_HD ZK_INL XYZZ xyzz_from_affine(const Fq& x, const Fq& y) {
  XYZZ r;
  r.X = x;
  r.Y = y;
  r.ZZ = fp_one<FqParams>();
  r.ZZZ = fp_one<FqParams>();
  return r;
}

// 2*(x, y) for an affine point (mdbl-2008-s-1): U = 2y, V = U^2, W = U V, S = x V, M = 3x^2,
// X3 = M^2 - 2S, Y3 = M (S - X3) - W y, ZZ3 = V, ZZZ3 = W.       x, y: [2]
ZK_HD ZK_INL XYZZ xyzz_dbl_affine(const Fq& x, const Fq& y) {
  XYZZ r;
  Fq U = fp_dbl(y);                                  // [4]
  Fq V = fp_sqr(U);                                  // [2]
  Fq W = fp_mul(U, V);                               // [2]
  Fq S = fp_mul(x, V);                               // [2]
  Fq xx = fp_sqr(x);                                 // [2]
  Fq M = fp_add(fp_dbl(xx), xx);                     // [6]
  Fq MM = fp_sqr(M);                                 // [2]
  r.X = fp_sub<FqParams, 4>(MM, fp_dbl(S));          // [6]
  Fq t = fp_sub<FqParams, 8>(S, r.X);                // [10]
  Fq Wy = fp_mul(W, y);                              // [2]
  r.Y = fp_sub<FqParams, 2>(fp_mul(M, t), Wy);       // [4]
  r.ZZ = V;
  r.ZZZ = W;
  return r;
}

// 2*P for XYZZ P (dbl-2008-s-1).  P.X [10], P.Y [4].
ZK_HD ZK_INL XYZZ xyzz_dbl(const XYZZ& p) {
  XYZZ r;
  Fq U = fp_dbl(p.Y);                                // [8]
  Fq V = fp_sqr(U);                                  // [2]
  Fq W = fp_mul(U, V);                               // [2]
  Fq S = fp_mul(p.X, V);                             // [2]
  Fq xx = fp_sqr(p.X);                               // [2]
  Fq M = fp_add(fp_dbl(xx), xx);                     // [6]
  Fq MM = fp_sqr(M);                                 // [2]
  r.X = fp_sub<FqParams, 4>(MM, fp_dbl(S));          // [6]
  Fq t = fp_sub<FqParams, 8>(S, r.X);                // [10]
  Fq Wy = fp_mul(W, p.Y);                            // [2]
  r.Y = fp_sub<FqParams, 2>(fp_mul(M, t), Wy);       // [4]
  r.ZZ = fp_mul(V, p.ZZ);                            // [2]  (U = 0, i.e. a 2-torsion point, gives ZZ = 0 = infinity)
  r.ZZZ = fp_mul(W, p.ZZZ);                          // [2]
  return r;
}

// acc += (x2, y2), (x2, y2) a finite affine point with coordinates [2].   madd-2008-s
ZK_HD ZK_INL void xyzz_madd(XYZZ& acc, const Fq& x2, const Fq& y2) {
  if (xyzz_is_inf(acc)) {
    acc = xyzz_from_affine(x2, y2);
    return;
  }
  Fq U2 = fp_mul(x2, acc.ZZ);                        // [2]
  Fq S2 = fp_mul(y2, acc.ZZZ);                       // [2]
  Fq P = fp_sub<FqParams, 16>(U2, acc.X);            // [18]   acc.X [10]
  Fq R = fp_sub<FqParams, 4>(S2, acc.Y);             // [6]    acc.Y [4]
  Fq PP = fp_sqr(P);                                 // [2]
  Fq RR = fp_sqr(R);                                 // [2]
  if (fp_is_zero_2p(PP)) {                           // same x: P = +-Q
    if (fp_is_zero_2p(RR)) acc = xyzz_dbl_affine(x2, y2);
    else acc = xyzz_infinity();
    return;
  }
  Fq PPP = fp_mul(P, PP);                            // [2]
  Fq Q = fp_mul(acc.X, PP);                          // [2]
  Fq X3 = fp_sub<FqParams, 4>(fp_sub<FqParams, 4>(RR, PPP), fp_dbl(Q));   // [2+4+4 = 10]
  Fq t = fp_sub<FqParams, 16>(Q, X3);                // [18]
  Fq Y3a = fp_mul(R, t);                             // [2]
  Fq Y3b = fp_mul(acc.Y, PPP);                       // [2]
  acc.X = X3;
  acc.Y = fp_sub<FqParams, 2>(Y3a, Y3b);             // [4]
  acc.ZZ = fp_mul(acc.ZZ, PP);                       // [2]
  acc.ZZZ = fp_mul(acc.ZZZ, PPP);                    // [2]
}

// a += b (both XYZZ).  add-2008-s
ZK_HD ZK_INL void xyzz_add(XYZZ& a, const XYZZ& b) {
  if (xyzz_is_inf(b)) return;
  if (xyzz_is_inf(a)) { a = b; return; }
  Fq U1 = fp_mul(a.X, b.ZZ);                         // [2]
  Fq U2 = fp_mul(b.X, a.ZZ);                         // [2]
  Fq S1 = fp_mul(a.Y, b.ZZZ);                        // [2]
  Fq S2 = fp_mul(b.Y, a.ZZZ);                        // [2]
  Fq P = fp_sub<FqParams, 2>(U2, U1);                // [4]
  Fq R = fp_sub<FqParams, 2>(S2, S1);                // [4]
  Fq PP = fp_sqr(P);                                 // [2]
  Fq RR = fp_sqr(R);                                 // [2]
  if (fp_is_zero_2p(PP)) {
    if (fp_is_zero_2p(RR)) a = xyzz_dbl(a);
    else a = xyzz_infinity();
    return;
  }
  Fq PPP = fp_mul(P, PP);                            // [2]
  Fq Q = fp_mul(U1, PP);                             // [2]
  Fq X3 = fp_sub<FqParams, 4>(fp_sub<FqParams, 4>(RR, PPP), fp_dbl(Q));   // [10]
  Fq t = fp_sub<FqParams, 16>(Q, X3);                // [18]
  Fq Y3a = fp_mul(R, t);                             // [2]
  Fq Y3b = fp_mul(S1, PPP);                          // [2]
  Fq zz = fp_mul(a.ZZ, b.ZZ);                        // [2]
  Fq zzz = fp_mul(a.ZZZ, b.ZZZ);                     // [2]
  a.X = X3;
  a.Y = fp_sub<FqParams, 2>(Y3a, Y3b);               // [4]
  a.ZZ = fp_mul(zz, PP);                             // [2]
  a.ZZZ = fp_mul(zzz, PPP);                          // [2]
}

ZK_HD ZK_INL void xyzz_neg(XYZZ& a) {
  a.Y = fp_sub<FqParams, 4>(fp_zero<FqParams>(), a.Y);   // [4]
}

// ---- memory forms -------------------------------------------------------------------------
ZK_HD ZK_INL AffineDev aff_load(const AffPacked* p) {
  AffineDev r;
  uint32_t nz = 0;
  uint32_t w[24];
#pragma unroll
  for (int i = 0; i < 24; i++) { w[i] = p->x[i]; nz |= w[i]; }
  r.x = fp_unpack32<FqParams>(w);
#pragma unroll
  for (int i = 0; i < 24; i++) { w[i] = p->y[i]; nz |= w[i]; }
  r.y = fp_unpack32<FqParams>(w);
  r.inf = (nz == 0);
  return r;
}

// XYZZ in memory: 4 x 27 limbs, stored limb-major with a caller-chosen stride so that
// consecutive threads touch consecutive words (coalesced): word (k, idx) at base[k*stride + idx].
ZK_HD ZK_INL void xyzz_store(uint32_t* base, size_t stride, size_t idx, const XYZZ& p) {
  const Fq* f[4] = {&p.X, &p.Y, &p.ZZ, &p.ZZZ};
#pragma unroll
  for (int c = 0; c < 4; c++)
#pragma unroll
    for (int i = 0; i < 27; i++) base[(size_t)(c * 27 + i) * stride + idx] = f[c]->l[i];
}

ZK_HD ZK_INL XYZZ xyzz_load(const uint32_t* base, size_t stride, size_t idx) {
  XYZZ p;
  Fq* f[4] = {&p.X, &p.Y, &p.ZZ, &p.ZZZ};
#pragma unroll
  for (int c = 0; c < 4; c++)
#pragma unroll
    for (int i = 0; i < 27; i++) f[c]->l[i] = base[(size_t)(c * 27 + i) * stride + idx];
  return p;
}

}  // namespace zkhip
