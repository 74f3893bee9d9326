// Host side of the wrapping circuit: native BLS12-377 Groth16 verification and (below) the aggregator circuit.
#include <string.h>
#include <mutex>

#include "../../include/zkhip.h"
#include "circuit/bls12_377.hpp"

using namespace zkhip;
using namespace zkhip::circuit;

namespace {
template <class F> G1<F> g1_from(const uint64_t* p, bool witness) {
  HFr x = HFr::from_limbs(p), y = HFr::from_limbs(p + 6);
  return witness ? G1<F>{F::witness(x), F::witness(y)} : G1<F>{F::constant(x), F::constant(y)};
}
template <class F> G2<F> g2_from(const uint64_t* p, bool witness) {
  HFr a = HFr::from_limbs(p), b = HFr::from_limbs(p + 6), c = HFr::from_limbs(p + 12), d = HFr::from_limbs(p + 18);
  return witness ? G2<F>{Fq2<F>::witness(a, b), Fq2<F>::witness(c, d)} : G2<F>{Fq2<F>::constant(a, b), Fq2<F>::constant(c, d)};
}
}  // namespace

extern "C" int zkhip_bls12_377_groth16_verify(const uint64_t vk_alpha_g1[12], const uint64_t vk_beta_g2[24], const uint64_t vk_delta_g2[24],
                                               const uint64_t* vk_abc, const uint64_t* inputs, size_t n_inputs,
                                               const uint64_t proof_a[12], const uint64_t proof_b[24], const uint64_t proof_c[12], int* ok) {
  if (!vk_alpha_g1 || !vk_beta_g2 || !vk_delta_g2 || !vk_abc || !proof_a || !proof_b || !proof_c || !ok || (n_inputs && !inputs))
    return ZKHIP_ERR_ARG;
  try {
    NestedVk<NF> vk;
    vk.alpha = g1_from<NF>(vk_alpha_g1, false);
    vk.beta = g2_from<NF>(vk_beta_g2, false);
    vk.delta = g2_from<NF>(vk_delta_g2, false);
    for (size_t i = 0; i <= n_inputs; i++) vk.abc.push_back(g1_from<NF>(vk_abc + i * 12, false));
    NestedProof<NF> pr{g1_from<NF>(proof_a, false), g2_from<NF>(proof_b, false), g1_from<NF>(proof_c, false)};
    std::vector<std::vector<NF>> bits(n_inputs);
    for (size_t k = 0; k < n_inputs; k++) {
      uint64_t c[6];
      HFr::from_limbs(inputs + k * 6).to_canonical(c);
      for (int j = 0; j < 253; j++) bits[k].push_back(NF::witness_bit((c[j / 64] >> (j % 64)) & 1));
    }
    G1<NF> acc = input_accumulator(vk, bits);
    *ok = groth16_verify_bit(vk, pr, acc).value().is_zero() ? 0 : 1;
  } catch (const std::exception&) {
    *ok = 0;
  }
  return ZKHIP_OK;
}
