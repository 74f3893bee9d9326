// single-accumulator Montgomery product: every column's mads in asm blocks on one register pair
template <int CNT> struct ZkMadV;
template <class PC, int J0, int CNT> struct ZkMadS;
#include "zk_chain.inc"
template <int K, int I0, int I1>
__device__ __forceinline__ void zk_prod_terms(uint64_t& acc, const uint32_t* a, const uint32_t* b) {
  if constexpr (I0 <= I1) {
    constexpr int CNT = (I1 - I0 + 1) < 13 ? (I1 - I0 + 1) : 13;
    ZkMadV<CNT>::run(acc, a + I0, b + (K - I0));
    zk_prod_terms<K, I0 + CNT, I1>(acc, a, b);
  }
}
template <class PC, int K, int I0, int I1>
__device__ __forceinline__ void zk_red_terms(uint64_t& acc, const uint32_t* m) {
  if constexpr (I0 <= I1) {
    constexpr int CNT = (I1 - I0 + 1) < 13 ? (I1 - I0 + 1) : 13;
    ZkMadS<PC, K - I0, CNT>::run(acc, m + I0);
    zk_red_terms<PC, K, I0 + CNT, I1>(acc, m);
  }
}
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
