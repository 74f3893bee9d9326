// Reduced-radix Montgomery arithmetic for the BW6-761 fields, written for gfx950.
//
// Representation ("device form"): NL limbs of 29 bits held in u32 (Fq: 27 limbs = 783 bits,
// Fr: 14 limbs = 406 bits), Montgomery radix R = 2^(29*NL).  Why 29 bits and not 32:
//   * gfx950 has v_mad_u64_u32 (32x32+64 -> 64) but no multiply with carry-in, and its 64-bit
//     addend must be an aligned VGPR pair.  With 29-bit limbs a whole product-scanning column
//     (<= 2*NL products of < 2^58) fits one 64-bit accumulator, so a field multiplication is
//     a pure chain of v_mad_u64_u32 on one register pair: no carry flags, no moves.
//   * >= 22 spare bits above the modulus let every multiplication skip the final conditional
//     subtraction: values are kept lazily in [0, k*p), k small (bounds are stated per function).
// This is the arithmetic the reference reaches through libff::Fp_model (bw6_761_Fq / bw6_761_Fr;
// call site libzecale/circuits/aggregator_circuit.tcc:168); libff itself is not in the reference
// tree (empty submodule depends/zeth), so nothing here is derived from its source.
//
// The same header compiles for the host (g++) so host-side unit tests can run without a GPU.
#pragma once
#include <stdint.h>
#include "bw6_params.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__
#define ZK_INL __forceinline__
#define ZK_NOINL __noinline__
#else
#define ZK_HD
#define ZK_INL inline __attribute__((always_inline))
#define ZK_NOINL __attribute__((noinline))
#endif

#ifndef ZK_MUL_INLINE
#define ZK_MUL_INLINE 0
#endif
#if ZK_MUL_INLINE
#define ZK_MULATTR ZK_INL
#else
#define ZK_MULATTR ZK_NOINL
#endif

namespace zkhip {

constexpr uint32_t M29 = (1u << 29) - 1;

template <class PR>
struct Fp {
  static constexpr int NL = PR::NL;
  uint32_t l[NL];
};

// Device code: the three multiplier bodies below run every column's v_mad_u64_u32 as ONE dependent chain on one register pair
// (fp29_chain.cuh: asm blocks, because hipcc re-associates the C++ column sums - carry last - and then needs a 64-bit addition per
// column and eight accumulator pairs).  ZK_NO_ASM_CHAIN keeps the C++ bodies (host builds always use them).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM_CHAIN)
#define ZK_ASM_CHAIN 1
#include "fp29_chain.cuh"
#else
#define ZK_ASM_CHAIN 0
#endif

// r = a*b/R mod p.  Requires a*b < R*p*2^k' such that a*b/R + p fits (a, b < 2^10 p is ample);
// limbs of a and b normalised (< 2^29).  Result < a*b/R + p  (< 2p when a*b < R*p).
template <class PR>
ZK_HD ZK_MULATTR Fp<PR> fp_mul(Fp<PR> a, Fp<PR> b) {
#if ZK_ASM_CHAIN
  return fp_mul_chain2<PR>(a, b);
#endif
  constexpr int N = PR::NL;
  Fp<PR> r;
  uint32_t m[N];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * PR::P[k - i];
    m[k] = ((uint32_t)acc * PR::PINV) & M29;
    acc += (uint64_t)m[k] * PR::P[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
    for (int i = k - N + 1; i < N; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - N + 1; i < N; i++) acc += (uint64_t)m[i] * PR::P[k - i];
    r.l[k - N] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  r.l[N - 1] = (uint32_t)acc;
  return r;
}

// r = a*a/R mod p; same bounds as fp_mul.  Off-diagonal products are computed once and doubled.
template <class PR>
ZK_HD ZK_MULATTR Fp<PR> fp_sqr(Fp<PR> a) {
#if ZK_ASM_CHAIN
  return fp_sqr_chain<PR>(a);
#endif
  constexpr int N = PR::NL;
  Fp<PR> r;
  uint32_t m[N];
  uint32_t a2[N];  // 2*a_i (< 2^30)
#pragma unroll
  for (int i = 0; i < N; i++) a2[i] = a.l[i] << 1;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) acc += (uint64_t)a2[i] * a.l[k - i];
    if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * PR::P[k - i];
    m[k] = ((uint32_t)acc * PR::PINV) & M29;
    acc += (uint64_t)m[k] * PR::P[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
    for (int i = k - N + 1; 2 * i < k; i++) acc += (uint64_t)a2[i] * a.l[k - i];
    if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
    for (int i = k - N + 1; i < N; i++) acc += (uint64_t)m[i] * PR::P[k - i];
    r.l[k - N] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  r.l[N - 1] = (uint32_t)acc;
  return r;
}

// r = (a*b + c*d)/R mod p with ONE Montgomery reduction: 2 * NL^2 product terms and NL^2 reduction terms instead of 2 * (NL^2 + NL^2).
// (The y coordinate of every addition formula has this shape: Y3 = R (Q - X3) - Y1 PPP.)  A column now holds up to 2 * NL products
// of < 2^58 plus NL reduction terms: 3 * 27 * 2^58 > 2^64.  So the products of a column are summed on their own (< 2^64: at most 26 pairs
// of full 29-bit limbs per product - a pair that involves a top limb is small), their upper part goes straight to the next
// column's carry, and only their low 29 bits meet the reduction terms: three more shift / add operations per column - in the 11 long
// columns of the middle only; the 42 columns of at most 21 terms per operand pair fit one accumulator as in fp_mul.
// Requires normalised limbs and a*b + c*d < 2^10 R p; result < (a*b + c*d)/R + p.
template <class PR>
ZK_HD ZK_MULATTR Fp<PR> fp_mul2(Fp<PR> a, Fp<PR> b, Fp<PR> c, Fp<PR> d) {
#if ZK_ASM_CHAIN
  return fp_mul2_chain<PR>(a, b, c, d);
#endif
  constexpr int N = PR::NL;
  Fp<PR> r;
  uint32_t m[N];
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
    if (3 * (k + 1) <= 63) {
      // a short column: 3 (k + 1) terms of < 2^58 and the carry fit one accumulator, as in fp_mul
      uint64_t acc = carry;
#pragma unroll
      for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = 0; i <= k; i++) acc += (uint64_t)c.l[i] * d.l[k - i];
#pragma unroll
      for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * PR::P[k - i];
      m[k] = ((uint32_t)acc * PR::PINV) & M29;
      acc += (uint64_t)m[k] * PR::P[0];
      carry = acc >> 29;
    } else {
      uint64_t t = 0;
#pragma unroll
      for (int i = 0; i <= k; i++) t += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = 0; i <= k; i++) t += (uint64_t)c.l[i] * d.l[k - i];
      // (the reduction terms do not depend on this column's products: their chain starts from the carry and runs beside the chain
      //  of the products - two independent accumulators for the scheduler to interleave)
      uint64_t acc = carry;
#pragma unroll
      for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * PR::P[k - i];
      acc += (uint32_t)t & M29;
      m[k] = ((uint32_t)acc * PR::PINV) & M29;
      acc += (uint64_t)m[k] * PR::P[0];
      carry = (acc >> 29) + (t >> 29);
    }
  }
#pragma unroll
  for (int k = N; k < 2 * N - 1; k++) {
    if (3 * (2 * N - 1 - k) <= 63) {
      uint64_t acc = carry;
#pragma unroll
      for (int i = k - N + 1; i < N; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = k - N + 1; i < N; i++) acc += (uint64_t)c.l[i] * d.l[k - i];
#pragma unroll
      for (int i = k - N + 1; i < N; i++) acc += (uint64_t)m[i] * PR::P[k - i];
      r.l[k - N] = (uint32_t)acc & M29;
      carry = acc >> 29;
    } else {
      uint64_t t = 0;
#pragma unroll
      for (int i = k - N + 1; i < N; i++) t += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = k - N + 1; i < N; i++) t += (uint64_t)c.l[i] * d.l[k - i];
      uint64_t acc = carry;
#pragma unroll
      for (int i = k - N + 1; i < N; i++) acc += (uint64_t)m[i] * PR::P[k - i];
      acc += (uint32_t)t & M29;
      r.l[k - N] = (uint32_t)acc & M29;
      carry = (acc >> 29) + (t >> 29);
    }
  }
  r.l[N - 1] = (uint32_t)carry;
  return r;
}

// carry-normalise limbs that may have grown up to 2^32-1 (top limb keeps the overflow)
template <class PR>
ZK_HD ZK_INL void fp_normalise(Fp<PR>& a) {
  constexpr int N = PR::NL;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < N - 1; i++) {
    uint32_t t = a.l[i] + c;
    a.l[i] = t & M29;
    c = t >> 29;
  }
  a.l[N - 1] += c;
}

// r = a + b (no reduction): bound(r) = bound(a) + bound(b)
template <class PR>
ZK_HD ZK_INL Fp<PR> fp_add(const Fp<PR>& a, const Fp<PR>& b) {
  Fp<PR> r;
#pragma unroll
  for (int i = 0; i < PR::NL; i++) r.l[i] = a.l[i] + b.l[i];
  fp_normalise(r);
  return r;
}

// r = 2a
template <class PR>
ZK_HD ZK_INL Fp<PR> fp_dbl(const Fp<PR>& a) {
  Fp<PR> r;
#pragma unroll
  for (int i = 0; i < PR::NL; i++) r.l[i] = a.l[i] << 1;
  fp_normalise(r);
  return r;
}

// r = a - b + K*p, requires b <= K*p (K in {2,4,8,16}); bound(r) = bound(a) + K
template <class PR, int K>
ZK_HD ZK_INL Fp<PR> fp_sub(const Fp<PR>& a, const Fp<PR>& b) {
  constexpr int N = PR::NL;
  Fp<PR> r;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint32_t kp = (K == 2) ? PR::SUBK2[i] : (K == 4) ? PR::SUBK4[i] : (K == 8) ? PR::SUBK8[i] : PR::SUBK16[i];
    r.l[i] = a.l[i] + kp - b.l[i];
  }
  fp_normalise(r);
  return r;
}

// K * p with every limb but the top raised by 2^29 at its upper neighbour's expense (same number): a_i + d_i - b_i never goes
// below zero for normalised a_i, b_i.  Any K with K p < 2^(29 NL) (the generated SUBK2..16 of the parameter file are this for K <= 16).
template <class PR, int K>
struct SubSafeKP {
  uint32_t l[PR::NL];
  constexpr SubSafeKP() : l{} {
    uint64_t c = 0;
    for (int i = 0; i < PR::NL; i++) {
      c += (uint64_t)PR::P[i] * K;
      l[i] = (i + 1 < PR::NL) ? (uint32_t)(c & M29) : (uint32_t)c;
      c >>= 29;
    }
    for (int i = 0; i + 1 < PR::NL; i++) { l[i] += 1u << 29; l[i + 1] -= 1u; }
  }
};
// r = a - b + K*p, requires b <= K*p; bound(r) = bound(a) + K
template <class PR, int K>
ZK_HD ZK_INL Fp<PR> fp_sub_k(const Fp<PR>& a, const Fp<PR>& b) {
  constexpr SubSafeKP<PR, K> kp{};
  Fp<PR> r;
#pragma unroll
  for (int i = 0; i < PR::NL; i++) r.l[i] = a.l[i] + kp.l[i] - b.l[i];
  fp_normalise(r);
  return r;
}

// K * p with every limb but the top raised by 4 * 2^29 at its upper neighbour's expense (same number): a_i + d_i - b_i - 2 c_i never
// goes below zero for normalised a_i, b_i, c_i.
template <class PR, int K>
struct SubSafe3KP {
  uint32_t l[PR::NL];
  constexpr SubSafe3KP() : l{} {
    uint64_t c = 0;
    for (int i = 0; i < PR::NL; i++) {
      c += (uint64_t)PR::P[i] * K;
      l[i] = (i + 1 < PR::NL) ? (uint32_t)(c & M29) : (uint32_t)c;
      c >>= 29;
    }
    for (int i = 0; i + 1 < PR::NL; i++) { l[i] += 4u << 29; l[i + 1] -= 4u; }
  }
};
// r = a - b - 2c + K*p with ONE carry pass (X3 = RR - PPP - 2Q of every addition formula: two subtractions and a doubling, each
// with its own carry normalisation, cost three passes).  Requires b + 2c <= K*p and K*p's top limb >= 4; limbs of a, b, c normalised.
// bound(r) = bound(a) + K.
template <class PR, int K>
ZK_HD ZK_INL Fp<PR> fp_sub_sub2(const Fp<PR>& a, const Fp<PR>& b, const Fp<PR>& c) {
  constexpr SubSafe3KP<PR, K> kp{};
  static_assert(PR::NL >= 2, "");
  Fp<PR> r;
#pragma unroll
  for (int i = 0; i < PR::NL; i++) r.l[i] = a.l[i] + kp.l[i] - b.l[i] - (c.l[i] << 1);
  fp_normalise(r);
  return r;
}

template <class PR>
ZK_HD ZK_INL Fp<PR> fp_const(const uint32_t (&c)[PR::NL]) {
  Fp<PR> r;
#pragma unroll
  for (int i = 0; i < PR::NL; i++) r.l[i] = c[i];
  return r;
}

template <class PR>
ZK_HD ZK_INL Fp<PR> fp_zero() {
  Fp<PR> r;
#pragma unroll
  for (int i = 0; i < PR::NL; i++) r.l[i] = 0;
  return r;
}

template <class PR>
ZK_HD ZK_INL Fp<PR> fp_one() {
  Fp<PR> r;
#pragma unroll
  for (int i = 0; i < PR::NL; i++) r.l[i] = PR::ONE[i];
  return r;
}

// a in [0, 2p) -> [0, p)
template <class PR>
ZK_HD ZK_INL Fp<PR> fp_cond_sub_p(const Fp<PR>& a) {
  constexpr int N = PR::NL;
  Fp<PR> d;
  int32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    int32_t t = (int32_t)a.l[i] - (int32_t)PR::P[i] + borrow;
    d.l[i] = (uint32_t)t & M29;
    borrow = t >> 29;  // arithmetic shift: 0 or -1
  }
  // top limb of d has the borrow folded in through the mask; borrow != 0 <=> a < p
  Fp<PR> r;
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
  return r;
}

// canonical limbs of K * p
template <class PR, int K>
struct KTimesP {
  uint32_t l[PR::NL];
  constexpr KTimesP() : l{} {
    uint64_t c = 0;
    for (int i = 0; i < PR::NL; i++) {
      c += (uint64_t)PR::P[i] * K;
      l[i] = (i + 1 < PR::NL) ? (uint32_t)(c & M29) : (uint32_t)c;
      c >>= 29;
    }
  }
};

// a in [0, 2 K p) -> [0, K p)
template <class PR, int K>
ZK_HD ZK_INL Fp<PR> fp_cond_sub_kp(const Fp<PR>& a) {
  constexpr int N = PR::NL;
  constexpr KTimesP<PR, K> kp{};
  Fp<PR> d;
  int32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    int32_t t = (int32_t)a.l[i] - (int32_t)kp.l[i] + borrow;
    d.l[i] = (i + 1 < N) ? ((uint32_t)t & M29) : (uint32_t)t;
    borrow = (i + 1 < N) ? (t >> 29) : (t >> 31);
  }
  Fp<PR> r;
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
  return r;
}
// full reduction of a lazily bounded value (< 2^10 p) to the canonical representative
template <class PR>
ZK_HD ZK_INL Fp<PR> fp_canon(const Fp<PR>& a) {
  Fp<PR> one = fp_const<PR>(PR::ONE);  // a * R / R = a, result < a/ R*R ... < 2p
  // mont_mul(a, R mod p) = a mod p (in [0, 2p))
  return fp_cond_sub_p(fp_mul(a, one));
}

// a == 0 (mod p) for a in [0, 2p)
template <class PR>
ZK_HD ZK_INL bool fp_is_zero_2p(const Fp<PR>& a) {
  uint32_t z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < PR::NL; i++) {
    z |= a.l[i];
    e |= a.l[i] ^ PR::P[i];
  }
  return z == 0 || e == 0;
}

// ---- packing: canonical-size values <-> packed little-endian 32-bit words (N32 words) ----
template <class PR>
ZK_HD ZK_INL void fp_pack32(const Fp<PR>& a /* canonical, < p */, uint32_t* w) {
  constexpr int N = PR::NL, W = PR::N32;
#pragma unroll
  for (int j = 0; j < W; j++) {
    const int bit = 32 * j, i = bit / 29, sh = bit % 29;
    uint64_t v = (uint64_t)a.l[i] >> sh;
    if (i + 1 < N) v |= (uint64_t)a.l[i + 1] << (29 - sh);
    if (i + 2 < N && (58 - sh) < 32) v |= (uint64_t)a.l[i + 2] << (58 - sh);
    w[j] = (uint32_t)v;
  }
}

template <class PR>
ZK_HD ZK_INL Fp<PR> fp_unpack32(const uint32_t* w) {
  constexpr int N = PR::NL, W = PR::N32;
  Fp<PR> r;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const int bit = 29 * i, j = bit / 32, sh = bit % 32;
    uint64_t v = 0;
    if (j < W) v = (uint64_t)w[j] >> sh;
    if (j + 1 < W) v |= (uint64_t)w[j + 1] << (32 - sh);
    r.l[i] = (uint32_t)v & M29;
  }
  return r;
}

// ABI form (N64 x u64, Montgomery radix 2^(64*N64), canonical < p)  ->  device form (< 2p)
template <class PR>
ZK_HD ZK_INL Fp<PR> fp_from_abi(const uint64_t* x) {
  uint32_t w[2 * PR::N64];
#pragma unroll
  for (int i = 0; i < PR::N64; i++) {
    w[2 * i] = (uint32_t)x[i];
    w[2 * i + 1] = (uint32_t)(x[i] >> 32);
  }
  Fp<PR> v = fp_unpack32<PR>(w);
  return fp_mul(v, fp_const<PR>(PR::ABI2DEV));
}

// device form (lazy, < 2^10 p) -> ABI form, canonical
template <class PR>
ZK_HD ZK_INL void fp_to_abi(const Fp<PR>& a, uint64_t* x) {
  Fp<PR> c = fp_cond_sub_p(fp_mul(a, fp_const<PR>(PR::DEV2ABI)));
  uint32_t w[2 * PR::N64];
#pragma unroll
  for (int i = 0; i < 2 * PR::N64; i++) w[i] = 0;
  fp_pack32<PR>(c, w);
#pragma unroll
  for (int i = 0; i < PR::N64; i++) x[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}

// ABI Montgomery form -> canonical integer limbs (for scalars): x * 2^-(64*N64) mod p
template <class PR>
ZK_HD ZK_INL void fp_abi_to_canonical_words(const uint64_t* x, uint32_t* w /* N32 words */) {
  // device form of value v is v*Rdev; from_abi gives (x_int)*Rdev with x_int the represented value
  Fp<PR> d = fp_from_abi<PR>(x);
  Fp<PR> one_raw = fp_zero<PR>();
  one_raw.l[0] = 1;                     // mont_mul(d, 1) = x_int mod p, in [0, 2p)
  Fp<PR> c = fp_cond_sub_p(fp_mul(d, one_raw));
  fp_pack32<PR>(c, w);
}

using Fq = Fp<FqParams>;
using Fr = Fp<FrParams>;

}  // namespace zkhip
