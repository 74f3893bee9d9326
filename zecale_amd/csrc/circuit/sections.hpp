// The wrapping circuit section by section: shared by the host build (aggregator.cpp: constraints, host witness) and by the
// recording build (witness_tape.cpp: the straight-line program of the GPU witness generator).  See dsl.hpp for the mechanism.
// Mirrors libzecale::aggregator_circuit / aggregator_gadget (libzecale/circuits/aggregator_circuit.tcc:17-98, 120-180;
// aggregator_gadget.tcc:13-112).
#pragma once
#include "bls12_377.hpp"
#include "mimc.hpp"

namespace zkhip {
namespace ZK_CIRCUIT_NS {

template <class F> G1<F> g1_from(const uint64_t* p, bool witness) {
  HFr x = HFr::from_limbs(p), y = HFr::from_limbs(p + 6);
  return witness ? G1<F>{F::witness(x), F::witness(y)} : G1<F>{F::constant(x), F::constant(y)};
}
template <class F> G2<F> g2_from(const uint64_t* p, bool witness) {
  HFr a = HFr::from_limbs(p), b = HFr::from_limbs(p + 6), c = HFr::from_limbs(p + 12), d = HFr::from_limbs(p + 18);
  return witness ? G2<F>{Fq2<F>::witness(a, b), Fq2<F>::witness(c, d)} : G2<F>{Fq2<F>::constant(a, b), Fq2<F>::constant(c, d)};
}

constexpr int NESTED_INPUT_BITS = 253;   // Fr of BLS12-377 (aggregator_gadget.tcc:42)

template <class F> std::vector<F> vk_all_vars(const NestedVk<F>& vk) {
  std::vector<F> v = {vk.alpha.x, vk.alpha.y, vk.beta.x.c0, vk.beta.x.c1, vk.beta.y.c0, vk.beta.y.c1,
                      vk.delta.x.c0, vk.delta.x.c1, vk.delta.y.c0, vk.delta.y.c1};
  for (const auto& p : vk.abc) { v.push_back(p.x); v.push_back(p.y); }
  return v;
}

struct NestedData {
  const uint64_t* vk;        // alpha (12) | beta (24) | delta (24) | abc ((k+1) x 12)
  const uint64_t* proofs;    // num_proofs x [a (12) | b (24) | c (12)]
  const uint64_t* inputs;    // num_proofs x k x 6
};

// The circuit in four kinds of sections, allocated in this order:
//   inputs:   primary inputs (vk hash, packed results, nested inputs), then the nested key and proofs
//   hash:     MiMC of the key's variables                          -> value of primary input 0
//   key:      the lines of -beta and -delta (vk_precompute), shared by every proof of the batch (aggregator_gadget.tcc:93)
//   proof p:  253 bits per nested input, accumulator, verification -> result bit p
// With V = CV one pass emits constraints and the assignment.  With V = WV (assignment only) the sections only READ values
// of earlier sections, so they run on separate host threads (the proof sections start when the key section is done), each
// filling its own slice of the assignment; concatenated in section order they reproduce the circuit's variable numbering.
template <class V>
struct Inputs {
  V vk_hash, packed;
  std::vector<std::vector<V>> nin;
  std::vector<std::vector<HFr>> nin_val;
  NestedVk<V> vk;
  std::vector<NestedProof<V>> proofs;
};

template <class V>
void alloc_inputs(Inputs<V>& in, size_t num_proofs, size_t k, const NestedData* data) {
  static const uint64_t zeros[48 * 8] = {0};
  auto limbs = [&](const uint64_t* p, size_t off) { return data ? p + off : zeros; };
  in.vk_hash = V::witness(HFr::zero());                  // variable 1, value patched by the caller
  in.packed = V::witness(HFr::zero());                   // variable 2
  in.nin.resize(num_proofs); in.nin_val.resize(num_proofs);
  for (size_t p = 0; p < num_proofs; p++)
    for (size_t j = 0; j < k; j++) {
      HFr v = HFr::from_limbs(limbs(data ? data->inputs : nullptr, (p * k + j) * 6));
      in.nin_val[p].push_back(v);
      in.nin[p].push_back(V::witness(v));
    }
  const uint64_t* vkp = data ? data->vk : nullptr;
  in.vk.alpha = g1_from<V>(limbs(vkp, 0), true);
  in.vk.beta = g2_from<V>(limbs(vkp, 12), true);
  in.vk.delta = g2_from<V>(limbs(vkp, 36), true);
  for (size_t i = 0; i <= k; i++) in.vk.abc.push_back(g1_from<V>(limbs(vkp, 60 + i * 12), true));
  for (size_t p = 0; p < num_proofs; p++) {
    const uint64_t* pp = data ? data->proofs : nullptr;
    in.proofs.push_back(NestedProof<V>{g1_from<V>(limbs(pp, p * 48), true), g2_from<V>(limbs(pp, p * 48 + 12), true),
                                       g1_from<V>(limbs(pp, p * 48 + 36), true)});
  }
}

template <class V> V section_hash(const Inputs<V>& in) {
  V h = mimc_hash(vk_all_vars(in.vk));
  V::assert_eq(in.vk_hash, h);
  return h;
}

template <class V> V section_proof(const Inputs<V>& in, size_t p, size_t k) {
  std::vector<std::vector<V>> bits(k);
  for (size_t j = 0; j < k; j++) {
    V sum;
    HFr w = HFr::one();
    for (int t = 0; t < NESTED_INPUT_BITS; t++) {
      V bit = V::witness_bitv(fr_bit(in.nin_val[p][j], t));
      bits[j].push_back(bit);
      sum = sum + bit.mulc(w);
      w = w + w;
    }
    V::assert_eq(sum, in.nin[p][j]);                      // packing (multipacking_gadget in the reference)
  }
  proof_assert_well_formed(in.proofs[p]);                 // proof_variable_gadget's curve checks (13 constraints per proof)
  G1<V> acc = input_accumulator(in.vk, bits);
  return groth16_verify_bit(in.vk, in.proofs[p], acc);
}

// structure pass (and single-threaded assignment)
template <class V>
void synthesize(Builder& b, size_t num_proofs, size_t k, const NestedData* data) {
  current_builder() = &b;
  Inputs<V> in;
  alloc_inputs(in, num_proofs, k, data);
  if (b.on_section) b.on_section(0);
  V h = section_hash(in);
  if (b.on_section) b.on_section(1);
  b.z[1] = h.value();
  vk_precompute(in.vk);
  if (b.on_section) b.on_section(2);
  V packed_lc;
  HFr pow2 = HFr::one();
  for (size_t p = 0; p < num_proofs; p++) {
    V res = section_proof(in, p, k);
    packed_lc = packed_lc + res.mulc(pow2);
    pow2 = pow2 + pow2;
  }
  V::assert_eq(in.packed, packed_lc);                     // packing_gadget::generate_r1cs_witness_from_bits (.tcc:157)
  b.z[2] = packed_lc.value();
  current_builder() = nullptr;
}

}  // namespace ZK_CIRCUIT_NS
}  // namespace zkhip
