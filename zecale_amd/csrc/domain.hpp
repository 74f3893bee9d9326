// Evaluation domains of the QAP (host side): the domains libfqfft offers over Fr of BW6-761 (2-adicity 46), their points, their
// vanishing polynomial and the Lagrange basis at a point.  [UPSTREAM-RECALL: libfqfft is an absent sub-submodule of the reference
// (SURVEY 0.1); libfqfft/evaluation_domain/get_evaluation_domain.tcc and domains/{basic,step}_radix2_domain.tcc.]  Reached in the
// reference from r1cs_gg_ppzksnark_generator / _prover through aggregator_circuit.tcc:108 and :168.
//
// WHICH domain a system of n constraints and l inputs gets (n + l + 1 points):
//   * the reference: libzeth's groth16_snark passes force_pow_2_domain = true to the generator and to the prover (SURVEY App. B.1,
//     B.2, row a7): basic_radix2_domain of 2^ceil(log2(n + l + 1)) points.  This is the DEFAULT here (forced_domain_size): the
//     wrapping circuit (44,183 constraints + 5) lives on 65,536 points and a reference key's H query has 65,535 entries.
//   * libfqfft's unforced get_evaluation_domain (eval_domain_size), an explicit OPTION (ZKHIP_DOMAIN_STEP): a power of two gets the
//     basic domain; anything else step_radix2_domain of m = big + small points, big = the largest power of two below min_size, small =
//     min_size - big rounded up to a power of two (if that makes m = 2 big, the basic domain of that size): points big_w^i (i < big),
//     then w small_w^i (i < small) with w of order 2 big, big_w = w^2, small_w of order small; Z(x) = (x^big - 1)(x^small - w^small).
//     The wrapping circuit would get 32,768 + 16,384 = 49,152 points: a quarter fewer H-query terms - but NOT the domain of a key
//     the reference generates, and nothing in the reference's tree can pin it.
// A proving key is authoritative: a prover works on the domain its key was generated for (zkhip_crs_desc.domain_size), whichever
// of the two it is (is_valid_domain).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>

#include "host_field.hpp"

namespace zkhip {
namespace host {

inline int ceil_log2(size_t n) { int k = 0; while (((size_t)1 << k) < n) k++; return k; }

// the reference's choice (force_pow_2_domain): the power of two at or above min_size
inline size_t forced_domain_size(size_t min_size) { return (size_t)1 << ceil_log2(min_size < 1 ? 1 : min_size); }
// libfqfft's unforced choice
inline size_t eval_domain_size(size_t min_size) {
  if (min_size <= 1) return 1;
  if ((min_size & (min_size - 1)) == 0) return min_size;
  const size_t big = (size_t)1 << (ceil_log2(min_size) - 1), small = min_size - big;
  return big + ((size_t)1 << ceil_log2(small));
}

// a size get_evaluation_domain can return: a power of two, or 2^k + 2^r with r < k (a fixed point of eval_domain_size)
inline bool is_valid_domain(size_t d) { return d >= 1 && eval_domain_size(d) == d; }
// the domain of a system with `points` = n + l + 1 interpolation points: requested == 0: the forced power of two (default);
// requested == (size_t)-1: libfqfft's unforced choice; else `requested` itself if it is valid and large enough; 0 = refused
inline size_t resolve_domain(size_t points, size_t requested) {
  if (requested == 0) return forced_domain_size(points);
  if (requested == (size_t)-1) return eval_domain_size(points);
  return (is_valid_domain(requested) && requested >= points) ? requested : 0;
}

inline HFr fr_pow_u64(const HFr& b, uint64_t e) { uint64_t ee[1] = {e}; return b.pow_limbs(ee, 1); }
inline HFr fr_root_of_unity(int log_n) {            // of order 2^log_n
  HFr w = HFr::from_limbs(FrParams::ROOT_2_46_64);
  for (int i = 0; i < FrParams::TWO_ADICITY - log_n; i++) w = w.sqr();
  return w;
}

struct EvalDomain {
  size_t m = 1, big = 1, small = 0;       // small == 0: a radix-2 domain of `big` points
  int log_big = 0, log_small = 0;
  HFr omega, big_omega, small_omega;      // step domain: see above; radix-2 domain: big_omega = the root of order m
  explicit EvalDomain(size_t size) : m(size) {
    if ((m & (m - 1)) == 0) {
      big = m; small = 0; log_big = ceil_log2(m);
      big_omega = fr_root_of_unity(log_big); omega = big_omega; small_omega = HFr::one();
    } else {
      log_big = ceil_log2(m) - 1; big = (size_t)1 << log_big; small = m - big; log_small = ceil_log2(small);
      omega = fr_root_of_unity(log_big + 1); big_omega = omega.sqr(); small_omega = fr_root_of_unity(log_small);
    }
  }
  bool is_step() const { return small != 0; }
  size_t compr() const { return small ? big / small : 1; }
  HFr vanishing(const HFr& x) const {
    if (!small) return fr_pow_u64(x, m) - HFr::one();
    return (fr_pow_u64(x, big) - HFr::one()) * (fr_pow_u64(x, small) - fr_pow_u64(omega, small));
  }
  // L_j(t) = Z(t) / ((t - x_j) Z'(x_j)) for every point x_j; false if t lies in the domain
  bool lagrange_at(const HFr& t, std::vector<HFr>& out) const {
    std::vector<HFr> x(m), den(m), pref(m);
    HFr xi = HFr::one();
    for (size_t j = 0; j < big; j++) { x[j] = xi; xi = xi * big_omega; }
    xi = omega;
    for (size_t j = 0; j < small; j++) { x[big + j] = xi; xi = xi * small_omega; }
    // denominators (t - x_j) Z'(x_j) x_j^-1 ... kept as: den_j = (t - x_j) * dz_j with L_j = Z(t) x_j / den_j
    //   radix-2: Z'(x) = m / x;  step, big part (x^big = 1): Z'(x) = big (x^small - w^small) / x;
    //   step, small part (x^small = w^small, x^big = -1): Z'(x) = -2 small w^small / x
    const HFr ws = small ? fr_pow_u64(omega, small) : HFr::one();
    const HFr bs = small ? fr_pow_u64(big_omega, small) : HFr::one();     // x_j^small for the big part: bs^j, period big / small
    const HFr big_f = HFr::from_u64((uint64_t)big), dz_small = small ? (HFr::from_u64(2 * (uint64_t)small) * ws).neg() : HFr::one();
    HFr acc = HFr::one(), xs = HFr::one();
    for (size_t j = 0; j < m; j++) {
      HFr dz;
      if (!small) dz = big_f;
      else if (j < big) { dz = big_f * (xs - ws); xs = xs * bs; }
      else dz = dz_small;
      den[j] = (t - x[j]) * dz;
      if (den[j].is_zero()) return false;
      pref[j] = acc; acc = acc * den[j];
    }
    const HFr zt = vanishing(t);
    HFr inv_all = acc.inv();
    out.resize(m);
    for (size_t j = m; j-- > 0;) {
      const HFr dj_inv = inv_all * pref[j];
      inv_all = inv_all * den[j];
      out[j] = zt * x[j] * dj_inv;
    }
    return true;
  }
};

}  // namespace host
}  // namespace zkhip
