// The three multiplier bodies of fp29.cuh (fp_mul, fp_sqr, fp_mul2) for the DEVICE, every column's v_mad_u64_u32 as ONE dependent chain
// on one register pair.  Included by fp29.cuh inside namespace zkhip (device pass only); same arithmetic, same bounds, same results.
//
// Why asm.  fp29.cuh's C++ bodies have one accumulator in the source, but hipcc's reassociation pass orders every column sum so that the
// loop-carried value - the carry of the previous column - is added LAST.  The columns then look independent: the scheduler runs eight
// of them side by side on eight register pairs and joins each with the carry by a 64-bit addition (v_lshl_add_u64, as dear as a mad):
// 52 per product.  With two waves per SIMD that parallelism buys nothing - a wave's dependent mads issue back to back - and there is
// no builtin for a mad with its addend, no switch for the pass, and an empty asm barrier moves with the sum it is attached to.
//
// How.  zk_chain.inc (tools/gen_chain_inc.py) holds asm BLOCKS of 1 .. 13 mads on one accumulator: an asm statement takes at most 30
// operands - the accumulator, the carry-out pair, 13 pairs of factors.  The carry-out pair (unused; the instruction must name one) is
// an EARLY-CLOBBER output: without "&" the allocator may give it the registers of a modulus limb that a later mad of the block
// still reads.  Between two blocks the compiler puts one s_nop: a VGPR written by inline asm and read by the next instruction is
// treated as gfx950's dst_sel forwarding hazard.  That is why the blocks are as long as the operand limit allows (~150 nops per
// product, free beside a second wave; one statement per mad - 1,458 nops - was measurably slower and 20 KB larger).
// Shifts, quotient digits (m = acc * PINV mod 2^29) and the dual product's column split stay C++ between the blocks.
// Checked against big integers on the device: tests/test_field_gpu.py.  -DZK_NO_ASM_CHAIN keeps the C++ bodies.
template <int CNT> struct ZkMadV;
template <class PC, int J0, int CNT> struct ZkMadS;
#include "zk_chain.inc"
// acc += sum_{i = I0 .. I1} x[i] * y[K - i], in blocks of at most 13 terms
template <int K, int I0, int I1>
__device__ __forceinline__ void zk_prod_terms(uint64_t& acc, const uint32_t* a, const uint32_t* b) {
  if constexpr (I0 <= I1) {
    constexpr int CNT = (I1 - I0 + 1) < 13 ? (I1 - I0 + 1) : 13;
    ZkMadV<CNT>::run(acc, a + I0, b + (K - I0));
    zk_prod_terms<K, I0 + CNT, I1>(acc, a, b);
  }
}
// acc += sum_{i = I0 .. I1} m[i] * P[K - i]  (the modulus limbs are SGPR operands)
template <class PC, int K, int I0, int I1>
__device__ __forceinline__ void zk_red_terms(uint64_t& acc, const uint32_t* m) {
  if constexpr (I0 <= I1) {
    constexpr int CNT = (I1 - I0 + 1) < 13 ? (I1 - I0 + 1) : 13;
    ZkMadS<PC, K - I0, CNT>::run(acc, m + I0);
    zk_red_terms<PC, K, I0 + CNT, I1>(acc, m);
  }
}
// column K of the interleaved Montgomery product, then column K + 1 (compile-time recursion: the block sizes are template arguments)
template <class PR, int K>
__device__ __forceinline__ void zk_mul_col(uint64_t& acc, const Fp<PR>& a, const Fp<PR>& b, uint32_t* m, Fp<PR>& r) {
  constexpr int N = PR::NL;
  if constexpr (K < N) {
    zk_prod_terms<K, 0, K>(acc, a.l, b.l);
    zk_red_terms<PR, K, 0, K - 1>(acc, m);
    m[K] = ((uint32_t)acc * PR::PINV) & M29;
    ZkMadS<PR, 0, 1>::run(acc, m + K);
    acc >>= 29;
  } else {
    zk_prod_terms<K, K - N + 1, N - 1>(acc, a.l, b.l);
    zk_red_terms<PR, K, K - N + 1, N - 1>(acc, m);
    r.l[K - N] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  if constexpr (K + 1 < 2 * N - 1) zk_mul_col<PR, K + 1>(acc, a, b, m, r);
}
template <class PR>
__device__ __forceinline__ Fp<PR> fp_mul_chain2(Fp<PR> a, Fp<PR> b) {
  Fp<PR> r;
  uint32_t m[PR::NL];
  uint64_t acc = 0;
  zk_mul_col<PR, 0>(acc, a, b, m, r);
  r.l[PR::NL - 1] = (uint32_t)acc;
  return r;
}

// ---- squaring: off-diagonal products once, against the doubled operand
template <class PR, int K>
__device__ __forceinline__ void zk_sqr_col(uint64_t& acc, const uint32_t* a, const uint32_t* a2, uint32_t* m, Fp<PR>& r) {
  constexpr int N = PR::NL;
  constexpr int I0 = K < N ? 0 : K - N + 1, I1 = (K + 1) / 2 - 1;        // 2 i < K
  zk_prod_terms<K, I0, I1>(acc, a2, a);
  if constexpr ((K & 1) == 0) ZkMadV<1>::run(acc, a + K / 2, a + K / 2);
  if constexpr (K < N) {
    zk_red_terms<PR, K, 0, K - 1>(acc, m);
    m[K] = ((uint32_t)acc * PR::PINV) & M29;
    ZkMadS<PR, 0, 1>::run(acc, m + K);
    acc >>= 29;
  } else {
    zk_red_terms<PR, K, K - N + 1, N - 1>(acc, m);
    r.l[K - N] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  if constexpr (K + 1 < 2 * N - 1) zk_sqr_col<PR, K + 1>(acc, a, a2, m, r);
}
template <class PR>
__device__ __forceinline__ Fp<PR> fp_sqr_chain(Fp<PR> a) {
  Fp<PR> r;
  uint32_t m[PR::NL], a2[PR::NL];
#pragma unroll
  for (int i = 0; i < PR::NL; i++) a2[i] = a.l[i] << 1;
  uint64_t acc = 0;
  zk_sqr_col<PR, 0>(acc, a.l, a2, m, r);
  r.l[PR::NL - 1] = (uint32_t)acc;
  return r;
}

// ---- dual product (a b + c d) / R with one reduction: fp_mul2's columns, each chain on its own register pair
template <class PR, int K>
__device__ __forceinline__ void zk_mul2_col(uint64_t& carry, const Fp<PR>& a, const Fp<PR>& b, const Fp<PR>& c, const Fp<PR>& d, uint32_t* m, Fp<PR>& r) {
  constexpr int N = PR::NL;
  constexpr int I0 = K < N ? 0 : K - N + 1, I1 = K < N ? K : N - 1;
  constexpr bool LONG = 3 * (I1 - I0 + 1) > 63;
  if constexpr (!LONG) {
    uint64_t acc = carry;
    zk_prod_terms<K, I0, I1>(acc, a.l, b.l);
    zk_prod_terms<K, I0, I1>(acc, c.l, d.l);
    if constexpr (K < N) {
      zk_red_terms<PR, K, 0, K - 1>(acc, m);
      m[K] = ((uint32_t)acc * PR::PINV) & M29;
      ZkMadS<PR, 0, 1>::run(acc, m + K);
    } else {
      zk_red_terms<PR, K, K - N + 1, N - 1>(acc, m);
      r.l[K - N] = (uint32_t)acc & M29;
    }
    carry = acc >> 29;
  } else {
    uint64_t t = 0;
    zk_prod_terms<K, I0, I1>(t, a.l, b.l);
    zk_prod_terms<K, I0, I1>(t, c.l, d.l);
    uint64_t acc = carry;
    if constexpr (K < N) {
      zk_red_terms<PR, K, 0, K - 1>(acc, m);
      acc += (uint32_t)t & M29;
      m[K] = ((uint32_t)acc * PR::PINV) & M29;
      ZkMadS<PR, 0, 1>::run(acc, m + K);
    } else {
      zk_red_terms<PR, K, K - N + 1, N - 1>(acc, m);
      acc += (uint32_t)t & M29;
      r.l[K - N] = (uint32_t)acc & M29;
    }
    carry = (acc >> 29) + (t >> 29);
  }
  if constexpr (K + 1 < 2 * N - 1) zk_mul2_col<PR, K + 1>(carry, a, b, c, d, m, r);
}
template <class PR>
__device__ __forceinline__ Fp<PR> fp_mul2_chain(Fp<PR> a, Fp<PR> b, Fp<PR> c, Fp<PR> d) {
  Fp<PR> r;
  uint32_t m[PR::NL];
  uint64_t carry = 0;
  zk_mul2_col<PR, 0>(carry, a, b, c, d, m, r);
  r.l[PR::NL - 1] = (uint32_t)carry;
  return r;
}
