// Host-side Groth16 verification over BW6-761 (product code, CPU): the `wsnarkT::verify` member of
// the snark policy class (reference libzecale/tests/aggregator/aggregator_dummy_test.cpp:61-62) and
// the equation the on-chain verifier checks (contracts/Groth16BW6_761.sol:166-176):
//     e(A, B) * e(acc, -g2) * e(alpha, -beta) * e(C, -delta) = 1,   acc = ABC_0 + sum x_i ABC_i.
// Verification is not on the prover's hot path; this is a plain, serial implementation:
// reduced Tate pairing t(P, Q) = f_{r,P}(psi(Q))^((q^6-1)/r), P in G1(Fq), Q on the sextic twist
// y^2 = x^3 + 4 over Fq, untwisted into Fq6 = Fq[w]/(w^6 + 4) by psi(x', y') = (x'/w^2, y'/w^3).
// Miller lines are scaled by Fq factors and vertical lines dropped (both die in the final exponentiation);
// the four Miller loops share one accumulator (one Fq6 squaring per bit for the whole product).
#pragma once
#include <vector>

#include "host_field.hpp"

namespace zkhip {
namespace host {

struct Fq6 {
  HFq c[6];   // sum c_i w^i,  w^6 = -4
  static Fq6 one() { Fq6 r; for (int i = 0; i < 6; i++) r.c[i] = HFq::zero(); r.c[0] = HFq::one(); return r; }
  bool is_one() const {
    if (c[0] != HFq::one()) return false;
    for (int i = 1; i < 6; i++) if (!c[i].is_zero()) return false;
    return true;
  }
  static HFq times_m4(const HFq& x) { HFq d = x.dbl().dbl(); return d.neg(); }
  Fq6 operator*(const Fq6& o) const {
    HFq t[11];
    for (int i = 0; i < 11; i++) t[i] = HFq::zero();
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) t[i + j] = t[i + j] + c[i] * o.c[j];
    Fq6 r;
    for (int i = 0; i < 6; i++) r.c[i] = (i + 6 < 11) ? t[i] + times_m4(t[i + 6]) : t[i];
    return r;
  }
  Fq6 sqr() const { return (*this) * (*this); }
  // multiply by the sparse line value  l0 + l3 w^3 + l4 w^4
  Fq6 mul_line(const HFq& l0, const HFq& l3, const HFq& l4) const {
    HFq t[11];
    for (int i = 0; i < 11; i++) t[i] = HFq::zero();
    for (int i = 0; i < 6; i++) {
      t[i] = t[i] + c[i] * l0;
      t[i + 3] = t[i + 3] + c[i] * l3;
      t[i + 4] = t[i + 4] + c[i] * l4;
    }
    Fq6 r;
    for (int i = 0; i < 6; i++) r.c[i] = (i + 6 < 11) ? t[i] + times_m4(t[i + 6]) : t[i];
    return r;
  }
  Fq6 pow_limbs(const uint64_t* e, int nlimbs) const {
    Fq6 acc = one();
    bool started = false;
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
      if (started) acc = acc.sqr();
      if ((e[i / 64] >> (i % 64)) & 1) { acc = started ? acc * (*this) : *this; started = true; }
    }
    return acc;
  }
};

struct MillerPair {
  HFq px, py;        // P in G1, affine
  HFq qx4, qy4;      // -xQ/4 and -yQ/4: psi(Q) = qx4 w^4 , qy4 w^3   (coefficients of w^4, w^3)
  HFq X, Y, Z;       // running T = [k] P, Jacobian
  bool done;         // T reached infinity (last addition is a vertical line)
};

// f <- f * line(T, T)(psi(Q)); T <- 2T.   Line scaled by 2 Y Z^3:
//   l = (2 Y Z^3) yq + (-3 X^2 Z^2) xq + (3 X^3 - 2 Y^2)
inline void miller_double(MillerPair& m, Fq6& f) {
  HFq XX = m.X.sqr(), YY = m.Y.sqr(), ZZ = m.Z.sqr();
  HFq threeXX = XX.dbl() + XX;
  HFq A = (m.Y * m.Z * ZZ).dbl();
  HFq B = (threeXX * ZZ).neg();
  HFq C = threeXX * m.X - YY.dbl();
  f = f.mul_line(C, A * m.qy4, B * m.qx4);
  // dbl-2009-l
  HFq YYYY = YY.sqr();
  HFq D = ((m.X + YY).sqr() - XX - YYYY).dbl();
  HFq F = threeXX.sqr();
  HFq X3 = F - D.dbl();
  HFq Y3 = threeXX * (D - X3) - YYYY.dbl().dbl().dbl();
  HFq Z3 = (m.Y * m.Z).dbl();
  m.X = X3; m.Y = Y3; m.Z = Z3;
}

// f <- f * line(T, P)(psi(Q)); T <- T + P.   Line scaled by D = Z (x2 Z^2 - X):
//   l = D yq - N xq + (N x2 - D y2),  N = y2 Z^3 - Y
inline void miller_add(MillerPair& m, Fq6& f) {
  HFq ZZ = m.Z.sqr();
  HFq H = m.px * ZZ - m.X;               // x2 Z^2 - X
  HFq N = m.py * ZZ * m.Z - m.Y;         // y2 Z^3 - Y
  if (H.is_zero()) {                     // T = +-P: for prime-order points only T = -P occurs (last step): vertical line
    m.done = true;
    return;
  }
  HFq D = m.Z * H;
  f = f.mul_line(N * m.px - D * m.py, D * m.qy4, N.neg() * m.qx4);
  // madd: X3 = N^2 - H^3 - 2 X H^2, Y3 = N (X H^2 - X3) - Y H^3, Z3 = Z H
  HFq HH = H.sqr(), HHH = HH * H, V = m.X * HH;
  HFq X3 = N.sqr() - HHH - V.dbl();
  HFq Y3 = N * (V - X3) - m.Y * HHH;
  m.X = X3; m.Y = Y3; m.Z = D;
}

// prod_i t(P_i, Q_i) == 1 ?   points affine (x, y) in ABI limbs; infinity (all zero) contributes 1.
inline bool pairing_product_is_one(const std::vector<const uint64_t*>& g1, const std::vector<const uint64_t*>& g2) {
  std::vector<MillerPair> ms;
  HFq quarter_neg = HFq::from_u64(4).inv().neg();    // -1/4
  for (size_t i = 0; i < g1.size(); i++) {
    HFq px = HFq::from_limbs(g1[i]), py = HFq::from_limbs(g1[i] + 12);
    HFq qx = HFq::from_limbs(g2[i]), qy = HFq::from_limbs(g2[i] + 12);
    if ((px.is_zero() && py.is_zero()) || (qx.is_zero() && qy.is_zero())) continue;
    MillerPair m;
    m.px = px; m.py = py; m.qx4 = qx * quarter_neg; m.qy4 = qy * quarter_neg;
    m.X = px; m.Y = py; m.Z = HFq::one(); m.done = false;
    ms.push_back(m);
  }
  Fq6 f = Fq6::one();
  const uint64_t* r = FqParams::R_ORDER64;
  int top = 6 * 64 - 1;
  while (!((r[top / 64] >> (top % 64)) & 1)) top--;
  for (int i = top - 1; i >= 0; i--) {
    f = f.sqr();
    for (auto& m : ms) if (!m.done) miller_double(m, f);
    if ((r[i / 64] >> (i % 64)) & 1)
      for (auto& m : ms) if (!m.done) miller_add(m, f);
  }
  return f.pow_limbs(FqParams::FINAL_EXP, FqParams::FINAL_EXP_LIMBS).is_one();
}

}  // namespace host
}  // namespace zkhip
