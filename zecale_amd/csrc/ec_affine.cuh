// Batched AFFINE additions for the bucket accumulation (msm.hip: k_affine_level).
//
// A mixed addition into an XYZZ accumulator costs 8 M + 2 S.  The sum of two AFFINE points costs one inversion, 2 M and 1 S -
// and a lane that performs m independent additions shares ONE inversion among them (Montgomery's trick: 3 M per addition), so
// an addition costs 5 M + 1 S plus 1/m of an inversion (fp_inv.cuh: ~30 M).  The additions inside one bucket are made
// independent by summing the bucket's sorted entries PAIRWISE, level by level (msm.hip).
//
//   lambda = (y2 - y1) / (x2 - x1)            (P1 = P2:  lambda = 3 x1^2 / (2 y1))
//   x3 = lambda^2 - x1 - x2,   y3 = lambda (x1 - x3) - y1
//
// This replaces the same libff::multi_exp inner loop as ec_mem.cuh (reached from libzecale/circuits/aggregator_circuit.tcc:168).
#pragma once
#include "ec_mem.cuh"
#include "fp_inv.cuh"

namespace zkhip {

// a in [0, 4p) -> [0, p)
template <class PR>
ZK_HD ZK_INL Fp<PR> fp_canon_4p(const Fp<PR>& a) { return fp_cond_sub_kp<PR, 1>(fp_cond_sub_kp<PR, 2>(a)); }

// Dense intermediate points (the outputs of a level) use AffPacked too; the point at infinity - which only arises when a
// bucket holds P and -P, or twice a point of order 2 - is encoded as x = all ones (canonical x < p < 2^761 never has word 23
// set to all ones; the all-zero encoding of the base sets is not available here: (0, +-2) lies on G2's curve, so x = 0 alone
// proves nothing and the levels only look at x first).
#define ZK_AFF_INF_WORD 0xFFFFFFFFu

enum PairKind : uint32_t { PK_ADD = 0, PK_DBL = 1, PK_FIRST = 2, PK_SECOND = 3, PK_INF = 4 };   // result: sum, double, P1, P2, infinity

struct PairRef {
  const AffPacked* p1;
  const AffPacked* p2;     // null: a bucket's odd entry passes through
  bool neg1, neg2;
};

__device__ __forceinline__ void aff_ld_words(const uint32_t* src, uint32_t* w) {
  const uint4* q = reinterpret_cast<const uint4*>(src);
#pragma unroll
  for (int i = 0; i < 6; i++) { uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
}
__device__ __forceinline__ void aff_st_words(uint32_t* dst, const uint32_t* w) {
  uint4* q = reinterpret_cast<uint4*>(dst);
#pragma unroll
  for (int i = 0; i < 6; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
__device__ __forceinline__ Fq aff_y_eff(const uint32_t* wy, bool neg) {      // [2]
  Fq y = fp_unpack32<FqParams>(wy);
  if (neg) y = fp_sub<FqParams, 2>(fp_zero<FqParams>(), y);
  return y;
}

// same x coordinate (rare): equal points double, opposite points (and points of order 2) cancel - the decision madd_same_x
// takes from R = 0 for the XYZZ accumulators
__device__ __noinline__ uint32_t pair_kind_same_x(const AffPacked* p1, bool neg1, const AffPacked* p2, bool neg2) {
  uint32_t w1[24], w2[24];
  aff_ld_words(p1->y, w1); aff_ld_words(p2->y, w2);
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < 24; i++) nz |= w1[i];
  if (nz == 0) return PK_INF;                          // y = 0: a point of order 2, P + P = O (and -P = P)
  Fq t = fp_sub<FqParams, 2>(aff_y_eff(w1, neg1), aff_y_eff(w2, neg2));       // [4]
  return fp_is_zero_2p(fp_mul(t, fp_one<FqParams>())) ? PK_DBL : PK_INF;
}

// classify a pair from the x coordinates (already loaded as packed words)
__device__ __forceinline__ uint32_t pair_kind(const PairRef& pr, const uint32_t* wx1, const uint32_t* wx2) {
  if (!pr.p2) return PK_FIRST;
  const bool inf1 = wx1[23] == ZK_AFF_INF_WORD, inf2 = wx2[23] == ZK_AFF_INF_WORD;
  if (inf1) return inf2 ? PK_INF : PK_SECOND;
  if (inf2) return PK_FIRST;
  uint32_t diff = 0;
#pragma unroll
  for (int i = 0; i < 24; i++) diff |= wx1[i] ^ wx2[i];
  if (diff == 0) return pair_kind_same_x(pr.p1, pr.neg1, pr.p2, pr.neg2);
  return PK_ADD;
}

// the denominator of a pair's slope ([4]); 1 for the pairs without an addition
__device__ __forceinline__ Fq pair_denominator(const PairRef& pr, uint32_t kind, const uint32_t* wx1, const uint32_t* wx2) {
  if (kind == PK_ADD) return fp_sub<FqParams, 2>(fp_unpack32<FqParams>(wx2), fp_unpack32<FqParams>(wx1));     // [3]
  if (kind == PK_DBL) {
    uint32_t wy[24];
    aff_ld_words(pr.p1->y, wy);
    return fp_dbl(aff_y_eff(wy, pr.neg1));                                                                      // [4]
  }
  return fp_one<FqParams>();
}

}  // namespace zkhip
