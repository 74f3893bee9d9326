// Host side of the wrapping circuit: native BLS12-377 Groth16 verification and (below) the aggregator circuit.
#include <string.h>
#include <mutex>
#include <thread>
#include <exception>

#include "../../include/zkhip.h"
#include "circuit/sections.hpp"

using namespace zkhip;
using namespace zkhip::circuit;


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

#include "aggregator_internal.h"

namespace {

// A section's values live for one witness (a few ms) in a vector of 1-2 MB: glibc would serve each from mmap and return it with munmap -
// a dozen address-space changes and a thousand page faults per witness under the process's memory-map lock, which throttled ten
// generator threads (round 5: nine inputs per nested proof, MORE workers gave FEWER proofs).  The vectors are kept and reused instead
// (at most 48 of them: ten generators x four sections in flight, and what a burst leaves behind is freed as it comes back).
struct SectionPool {
  std::mutex mu;
  std::vector<std::vector<HFr>> free_;
  std::vector<HFr> get(size_t cap) {
    std::vector<HFr> v;
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!free_.empty()) { v.swap(free_.back()); free_.pop_back(); }
    }
    v.clear();
    v.reserve(cap);                                       // (one allocation per section, not one per doubling of the vector)
    return v;
  }
  void put(std::vector<HFr>& v) {
    std::lock_guard<std::mutex> lk(mu);
    if (free_.size() < 48) { free_.emplace_back(); free_.back().swap(v); }
  }
};
SectionPool& section_pool() { static SectionPool* p = new SectionPool(); return *p; }     // (never destroyed: generator threads may outlive main)

// the assembled assignment, straight into the caller's buffer: [ONE, primary inputs, inputs' variables | section 0 | section 1 | ...]
// (`gap`: entries left ZERO between the head and the first section - the sections an application's handle holds)
void write_assignment(uint64_t* z_out, size_t n_vars, const std::vector<HFr>& head, size_t gap, std::vector<std::vector<HFr>>& parts) {
  size_t total = head.size() + gap;
  for (auto& p : parts) total += p.size();
  if (total != n_vars) throw std::runtime_error("assignment layout changed");
  static_assert(sizeof(HFr) == 48, "an assignment entry is six u64 limbs");
  memcpy(z_out, head.data(), head.size() * 48);
  size_t at = head.size();
  if (gap) { memset(z_out + at * 6, 0, gap * 48); at += gap; }
  for (auto& p : parts) {
    if (!p.empty()) memcpy(z_out + at * 6, p.data(), p.size() * 48);
    at += p.size();
    section_pool().put(p);
  }
}

// assignment only, sections on separate threads
void witness_parallel(uint64_t* z_out, size_t n_vars, size_t num_proofs, size_t k, const NestedData* data) {
  const size_t section_cap = n_vars / (num_proofs ? num_proofs : 1) + 64;
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
      bs.z = section_pool().get(section_cap);           // a section holds no constant ONE of its own
      current_builder() = &bs;
      if (s == 0) results[s] = section_hash(in).value();
      else if (s == 1) vk_precompute(in.vk);
      else results[s] = section_proof(in, s - 2, k).value();
      current_builder() = nullptr;
      parts[s].swap(bs.z);
    } catch (...) { errs[s] = std::current_exception(); current_builder() = nullptr; }
  };
  std::thread t_hash(run, 0);
  run(1);                                               // the proof sections read the key's lines
  std::vector<std::thread> th;
  if (!errs[1]) for (size_t s = 2; s < ns; s++) th.emplace_back(run, s);
  t_hash.join();
  for (auto& t : th) t.join();
  for (auto& e : errs) if (e) { for (auto& p : parts) section_pool().put(p); std::rethrow_exception(e); }
  b0.z[1] = results[0];
  HFr packed = HFr::zero(), pow2 = HFr::one();
  for (size_t p = 0; p < num_proofs; p++) { packed = packed + results[p + 2] * pow2; pow2 = pow2 + pow2; }
  b0.z[2] = packed;
  write_assignment(z_out, n_vars, b0.z, 0, parts);
}

// Proof sections only, for a batch under a REGISTERED key (zk_app_host_witness): the key's variables are allocated as always (the
// numbering must not move), its hash and its lines are not recomputed - `vk` carries the lines - and their slices stay zero.
void witness_proofs_only(uint64_t* z_out, const zkhip_aggregator* a, const NestedVk<WV>& vk, const NestedData* data, const DblChain* chain) {
  const size_t num_proofs = a->num_proofs, k = a->inputs_per_proof;
  Builder b0;
  current_builder() = &b0;
  Inputs<WV> in;
  alloc_inputs(in, num_proofs, k, data);
  current_builder() = nullptr;
  in.vk.neg_beta_lines = vk.neg_beta_lines;
  in.vk.neg_delta_lines = vk.neg_delta_lines;
  in.vk.dbl_chain = chain;                              // (read-only, shared by the section threads)
  std::vector<std::vector<HFr>> parts(num_proofs);
  std::vector<HFr> results(num_proofs);
  std::vector<std::exception_ptr> errs(num_proofs);
  auto run = [&](size_t p) {
    try {
      Builder bs;
      bs.z = section_pool().get((a->n_vars - a->sec_proofs) / num_proofs + 64);
      current_builder() = &bs;
      results[p] = section_proof(in, p, k).value();
      current_builder() = nullptr;
      parts[p].swap(bs.z);
    } catch (...) { errs[p] = std::current_exception(); current_builder() = nullptr; }
  };
  std::vector<std::thread> th;
  for (size_t p = 1; p < num_proofs; p++) th.emplace_back(run, p);
  run(0);
  for (auto& t : th) t.join();
  for (auto& e : errs) if (e) { for (auto& p : parts) section_pool().put(p); std::rethrow_exception(e); }
  if (b0.z.size() != a->sec_hash) throw std::runtime_error("assignment layout changed");
  HFr packed = HFr::zero(), pow2 = HFr::one();
  for (size_t p = 0; p < num_proofs; p++) { packed = packed + results[p] * pow2; pow2 = pow2 + pow2; }
  b0.z[2] = packed;
  write_assignment(z_out, a->n_vars, b0.z, a->sec_proofs - a->sec_hash, parts);     // the hash and key sections: the application's constants, left at zero
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
  static thread_local size_t marks[3];
  static thread_local Builder* marking;
  marking = &b;
  b.on_section = [](int which) { if (which >= 0 && which < 3) marks[which] = marking->z.size(); };
  synthesize<CV>(b, num_proofs, inputs_per_proof, nullptr);
  a->sec_hash = marks[0]; a->sec_key = marks[1]; a->sec_proofs = marks[2];
  a->n_vars = b.z.size();
  a->n_primary = 2 + num_proofs * inputs_per_proof;       // aggregator_circuit.tcc:172-180
  a->n_constraints = b.num_constraints();
  to_csr(b.A, a->rp[0], a->col[0], a->val[0]);
  to_csr(b.B, a->rp[1], a->col[1], a->val[1]);
  to_csr(b.C, a->rp[2], a->col[2], a->val[2]);
  *out = a;
  return ZKHIP_OK;
}

void zkhip_aggregator_free(zkhip_aggregator* a) {
  if (!a) return;
  if (a->gpu_release) a->gpu_release(a);            // device copies of the witness program (witness.hip)
  delete a;
}
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
  NestedData d{nested_vk, nested_proofs, nested_inputs};     // re-entrant: the circuit description is read-only after zkhip_aggregator_new
  try {
    witness_parallel(z_out, a->n_vars, a->num_proofs, a->inputs_per_proof, &d);
  } catch (const std::exception&) {
    current_builder() = nullptr;
    return ZKHIP_ERR_ARG;
  }
  return ZKHIP_OK;
}

struct AppHost { NestedVk<WV> vk; DblChain chain; };     // chain: the doubling chains 2^j ABC_k of the input accumulators (bls12_377.hpp)

int zk_app_host_new(const zkhip_aggregator* a, const uint64_t* nested_vk, void** state) {
  if (!a || !nested_vk || !state) return ZKHIP_ERR_ARG;
  AppHost* st = new AppHost();
  try {
    Builder b;                                     // scratch: only the VALUES of the key's lines are kept
    b.record = false;
    current_builder() = &b;
    NestedData d{nested_vk, nullptr, nullptr};
    st->vk.alpha = g1_from<WV>(d.vk, true);
    st->vk.beta = g2_from<WV>(d.vk + 12, true);
    st->vk.delta = g2_from<WV>(d.vk + 36, true);
    for (size_t i = 0; i <= a->inputs_per_proof; i++) st->vk.abc.push_back(g1_from<WV>(d.vk + 60 + i * 12, true));
    vk_precompute(st->vk);
    current_builder() = nullptr;
    std::vector<std::array<HFr, 2>> abc;
    for (size_t i = 1; i <= a->inputs_per_proof; i++) abc.push_back({st->vk.abc[i].x.value(), st->vk.abc[i].y.value()});
    st->chain = doubling_chain_values(abc, NESTED_INPUT_BITS);
  } catch (const std::exception&) {
    current_builder() = nullptr;
    delete st;
    return ZKHIP_ERR_ARG;
  }
  *state = st;
  return ZKHIP_OK;
}

void zk_app_host_free(void* state) { delete (AppHost*)state; }

int zk_app_host_witness(const zkhip_aggregator* a, const void* state, const uint64_t* nested_vk, const uint64_t* nested_proofs,
                        const uint64_t* nested_inputs, const uint32_t* s_idx, size_t n_s, const uint64_t vk_hash[6], uint64_t* z_out) {
  if (!a || !state || !nested_vk || !nested_proofs || !nested_inputs || !vk_hash || !z_out || (n_s && !s_idx)) return ZKHIP_ERR_ARG;
  NestedData d{nested_vk, nested_proofs, nested_inputs};
  try {
    witness_proofs_only(z_out, a, ((const AppHost*)state)->vk, &d, ((const AppHost*)state)->chain.empty() ? nullptr : &((const AppHost*)state)->chain);
  } catch (const std::exception&) {
    current_builder() = nullptr;
    return ZKHIP_ERR_ARG;
  }
  memcpy(z_out + 6, vk_hash, 48);                              // primary input 0 (not a masked position: the caller gets it back)
  for (size_t j = 0; j < n_s; j++) {
    if (s_idx[j] <= a->n_primary || s_idx[j] >= a->n_vars) return ZKHIP_ERR_ARG;
    memset(z_out + (size_t)s_idx[j] * 6, 0, 48);
  }
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
