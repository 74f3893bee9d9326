// Host side of the wrapping circuit: native BLS12-377 Groth16 verification and (below) the aggregator circuit.
#include <string.h>
#include <mutex>
#include <thread>
#include <exception>

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
    proof_assert_well_formed(pr);                     // off-curve points: rejected (libsnark proof.is_well_formed())
    g1_assert_on_curve(vk.alpha); g2_assert_on_curve(vk.beta); g2_assert_on_curve(vk.delta);
    for (const auto& q : vk.abc) g1_assert_on_curve(q);
    std::vector<std::vector<NF>> bits(n_inputs);
    for (size_t k = 0; k < n_inputs; k++) {
      uint64_t c[6];
      HFr::from_limbs(inputs + k * 6).to_canonical(c);
      for (int j = 0; j < 253; j++) bits[k].push_back(NF::witness_bit((c[j / 64] >> (j % 64)) & 1));
    }
    G1<NF> acc = input_accumulator(vk, bits);
    vk_precompute(vk);
    *ok = groth16_verify_bit(vk, pr, acc).value().is_zero() ? 0 : 1;
  } catch (const std::exception&) {
    *ok = 0;
  }
  return ZKHIP_OK;
}

// ================================================================================================
// The wrapping ("aggregator") circuit.  Mirrors libzecale::aggregator_circuit<wppT, wsnarkT, nverifierT, NumProofs>
// (libzecale/circuits/aggregator_circuit.hpp:32-114, .tcc:17-180) and aggregator_gadget (.tcc:13-112):
//   primary inputs, allocated first (aggregator_circuit.hpp:19-31, .tcc:172-180):
//     [ hash of the nested verification key, packed verification results (LSB = proof 0),
//       nested primary inputs of proof 0, ..., of proof NumProofs-1 ]
//   auxiliary: the nested verification key, the nested proofs, 253 bits per nested input
//     (aggregator_gadget.tcc:42), and every intermediate of the in-circuit Groth16 verifications.
//   The result bits are NOT enforced to be 1: an invalid nested proof yields a valid wrapping proof whose
//   result bit is 0 (aggregator_circuit.hpp:51-54; aggregator_dummy_test.cpp:162-186).
// ================================================================================================
#include "circuit/mimc.hpp"

struct zkhip_aggregator {
  size_t num_proofs, inputs_per_proof;
  size_t n_vars = 0, n_primary = 0, n_constraints = 0;
  std::vector<uint32_t> rp[3], col[3];
  std::vector<uint64_t> val[3];
};

namespace {

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
    uint64_t c[6];
    in.nin_val[p][j].to_canonical(c);
    V sum;
    HFr w = HFr::one();
    for (int t = 0; t < NESTED_INPUT_BITS; t++) {
      V bit = V::witness_bit((c[t / 64] >> (t % 64)) & 1);
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
  V h = section_hash(in);
  b.z[1] = h.value();
  vk_precompute(in.vk);
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

// assignment only, sections on separate threads
void witness_parallel(std::vector<HFr>& z, size_t num_proofs, size_t k, const NestedData* data) {
  Builder b0;
  current_builder() = &b0;
  Inputs<WV> in;
  alloc_inputs(in, num_proofs, k, data);
  current_builder() = nullptr;
  // sections: 0 = hash, 1 = key lines, 2 + p = proof p
  const size_t ns = num_proofs + 2;
  std::vector<std::vector<HFr>> parts(ns);
  std::vector<HFr> results(ns);
  std::vector<std::exception_ptr> errs(ns);
  auto run = [&](size_t s) {
    try {
      Builder bs;
      bs.z.clear();                                     // a section holds no constant ONE of its own
      current_builder() = &bs;
      if (s == 0) results[s] = section_hash(in).value();
      else if (s == 1) vk_precompute(in.vk);
      else results[s] = section_proof(in, s - 2, k).value();
      current_builder() = nullptr;
      parts[s] = std::move(bs.z);
    } catch (...) { errs[s] = std::current_exception(); current_builder() = nullptr; }
  };
  std::thread t_hash(run, 0);
  run(1);                                               // the proof sections read the key's lines
  std::vector<std::thread> th;
  if (!errs[1]) for (size_t s = 2; s < ns; s++) th.emplace_back(run, s);
  t_hash.join();
  for (auto& t : th) t.join();
  for (auto& e : errs) if (e) std::rethrow_exception(e);
  z = std::move(b0.z);
  for (auto& p : parts) z.insert(z.end(), p.begin(), p.end());
  z[1] = results[0];
  HFr packed = HFr::zero(), pow2 = HFr::one();
  for (size_t p = 0; p < num_proofs; p++) { packed = packed + results[p + 2] * pow2; pow2 = pow2 + pow2; }
  z[2] = packed;
}

void to_csr(const std::vector<LC>& M, std::vector<uint32_t>& rp, std::vector<uint32_t>& col, std::vector<uint64_t>& val) {
  rp.assign(1, 0);
  for (const LC& row : M) {
    for (const Term& t : row) {
      col.push_back(t.var);
      uint64_t l[6];
      t.coeff.to_limbs(l);
      val.insert(val.end(), l, l + 6);
    }
    rp.push_back((uint32_t)col.size());
  }
}

}  // namespace

extern "C" {

int zkhip_aggregator_new(size_t num_proofs, size_t inputs_per_proof, zkhip_aggregator** out) {
  if (!out || num_proofs == 0 || num_proofs > 16 || inputs_per_proof == 0 || inputs_per_proof > 16) return ZKHIP_ERR_ARG;
  zkhip_aggregator* a = new zkhip_aggregator();
  a->num_proofs = num_proofs; a->inputs_per_proof = inputs_per_proof;
  Builder b;
  b.record = true;
  synthesize<CV>(b, num_proofs, inputs_per_proof, nullptr);
  a->n_vars = b.z.size();
  a->n_primary = 2 + num_proofs * inputs_per_proof;       // aggregator_circuit.tcc:172-180
  a->n_constraints = b.num_constraints();
  to_csr(b.A, a->rp[0], a->col[0], a->val[0]);
  to_csr(b.B, a->rp[1], a->col[1], a->val[1]);
  to_csr(b.C, a->rp[2], a->col[2], a->val[2]);
  *out = a;
  return ZKHIP_OK;
}

void zkhip_aggregator_free(zkhip_aggregator* a) { delete a; }
size_t zkhip_aggregator_num_proofs(const zkhip_aggregator* a) { return a ? a->num_proofs : 0; }
size_t zkhip_aggregator_inputs_per_proof(const zkhip_aggregator* a) { return a ? a->inputs_per_proof : 0; }
size_t zkhip_aggregator_num_constraints(const zkhip_aggregator* a) { return a ? a->n_constraints : 0; }
size_t zkhip_aggregator_num_variables(const zkhip_aggregator* a) { return a ? a->n_vars : 0; }
size_t zkhip_aggregator_num_primary_inputs(const zkhip_aggregator* a) { return a ? a->n_primary : 0; }

int zkhip_aggregator_get_r1cs(const zkhip_aggregator* a, zkhip_r1cs_desc* d) {
  if (!a || !d) return ZKHIP_ERR_ARG;
  d->n_constraints = a->n_constraints; d->n_vars = a->n_vars; d->n_primary = a->n_primary;
  d->a_row_ptr = a->rp[0].data(); d->a_col = a->col[0].data(); d->a_val = a->val[0].data();
  d->b_row_ptr = a->rp[1].data(); d->b_col = a->col[1].data(); d->b_val = a->val[1].data();
  d->c_row_ptr = a->rp[2].data(); d->c_col = a->col[2].data(); d->c_val = a->val[2].data();
  return ZKHIP_OK;
}

int zkhip_aggregator_witness(zkhip_aggregator* a, const uint64_t* nested_vk, const uint64_t* nested_proofs,
                             const uint64_t* nested_inputs, uint64_t* z_out) {
  if (!a || !nested_vk || !nested_proofs || !nested_inputs || !z_out) return ZKHIP_ERR_ARG;
  std::vector<HFr> z;                  // re-entrant: the circuit description is read-only after zkhip_aggregator_new
  NestedData d{nested_vk, nested_proofs, nested_inputs};
  try {
    witness_parallel(z, a->num_proofs, a->inputs_per_proof, &d);
  } catch (const std::exception&) {
    current_builder() = nullptr;
    return ZKHIP_ERR_ARG;
  }
  if (z.size() != a->n_vars) return ZKHIP_ERR_STATE;
  for (size_t i = 0; i < z.size(); i++) z[i].to_limbs(z_out + i * 6);
  return ZKHIP_OK;
}

int zkhip_aggregator_check_inputs(const zkhip_aggregator* a, const uint64_t* nested_vk, const uint64_t* nested_proofs, int* ok) {
  if (!a || !nested_vk || !nested_proofs || !ok) return ZKHIP_ERR_ARG;
  *ok = 1;
  try {
    g1_assert_on_curve(g1_from<NF>(nested_vk, false));
    g2_assert_on_curve(g2_from<NF>(nested_vk + 12, false));
    g2_assert_on_curve(g2_from<NF>(nested_vk + 36, false));
    for (size_t i = 0; i <= a->inputs_per_proof; i++) g1_assert_on_curve(g1_from<NF>(nested_vk + 60 + i * 12, false));
    for (size_t p = 0; p < a->num_proofs; p++)
      proof_assert_well_formed(NestedProof<NF>{g1_from<NF>(nested_proofs + p * 48, false), g2_from<NF>(nested_proofs + p * 48 + 12, false),
                                               g1_from<NF>(nested_proofs + p * 48 + 36, false)});
  } catch (const std::exception&) {
    *ok = 0;
  }
  return ZKHIP_OK;
}

int zkhip_aggregator_vk_hash(const uint64_t* nested_vk, size_t inputs_per_proof, uint64_t out[6]) {
  if (!nested_vk || !out) return ZKHIP_ERR_ARG;
  NestedVk<NF> vk;
  vk.alpha = g1_from<NF>(nested_vk, false);
  vk.beta = g2_from<NF>(nested_vk + 12, false);
  vk.delta = g2_from<NF>(nested_vk + 36, false);
  for (size_t i = 0; i <= inputs_per_proof; i++) vk.abc.push_back(g1_from<NF>(nested_vk + 60 + i * 12, false));
  mimc_hash(vk_all_vars(vk)).value().to_limbs(out);
  return ZKHIP_OK;
}

}  // extern "C"
