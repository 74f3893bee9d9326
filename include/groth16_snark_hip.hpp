// groth16_snark_hip.hpp - C++ host layer above the zkhip C ABI, mirroring the reference's snark policy
// class for the ONE path this library replaces.
//
// In the reference the wrapping prover is selected by a template parameter:
//     using wsnark = libzeth::groth16_snark<wpp>;                 aggregator_server/aggregator_server.cpp:61
//     libzecale::aggregator_circuit<wpp, wsnark, nverifier, batch_size>            ... :65-66
// and aggregator_circuit only ever touches these static members of wsnarkT (seam 3):
//     generate_setup(pb)            libzecale/circuits/aggregator_circuit.tcc:108
//     generate_proof(pk, pb)        libzecale/circuits/aggregator_circuit.tcc:168      <-- the hot path
//     verify(inputs, proof, vk)     libzecale/tests/aggregator/aggregator_dummy_test.cpp:61-62
// This header provides (a) a self-contained RAII prover over raw limb arrays (usable and tested without
// libsnark, which is not in this image) and (b) the policy-class adapter `groth16_snark_hip<ppT, baseT>`
// that a maintainer drops in as `wsnark`: it inherits everything from the CPU policy
// class and overrides generate_proof only.  Same names, same argument meaning, same error behaviour
// (std::runtime_error, as aggregator_circuit::prove throws at aggregator_circuit.tcc:138-141).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "zkhip.h"

namespace zecale_amd {

inline void zk_check(int rc, const char* what) {
  if (rc != ZKHIP_OK)
    throw std::runtime_error(std::string(what) + ": " + zkhip_strerror(rc) + " (" + zkhip_last_error() + ")");
}

// A Groth16 proof over BW6-761 in the ABI's affine Montgomery limbs: A in G1, B in G2, C in G1.
struct groth16_proof {
  std::array<uint64_t, 24> a, b, c;
};

// Constraint system in CSR form (values in Montgomery limbs); see zkhip_r1cs_desc.
struct csr_matrix {
  std::vector<uint32_t> row_ptr, col;
  std::vector<uint64_t> val;   // 6 limbs per entry
};

// The proving key + constraint system resident in HBM.  Non-copyable, like aggregator_circuit
// (libzecale/circuits/aggregator_circuit.hpp:95-97).
class hip_proving_key {
 public:
  // opts: this key's own table / launch options (zkhip_key_opts travel with the handle; nullptr = the defaults)
  // The KEY names the QAP's evaluation domain (crs.domain_size): a key from the reference's generate_setup - libzeth forces a power
  // of two, 65,536 points for the wrapping circuit (aggregator_circuit.tcc:108) - proves as it is; so does a key generated here on
  // libfqfft's unforced step domain.  The constraint system is uploaded on the key's domain.
  hip_proving_key(const zkhip_crs_desc& crs, const zkhip_r1cs_desc& cs, const zkhip_key_opts* opts = nullptr) {
    zk_check(zkhip_r1cs_upload_ex(&cs, crs.domain_size, &r1cs_), "zkhip_r1cs_upload_ex");
    int rc = zkhip_crs_upload_ex(&crs, opts, &crs_);
    if (rc != ZKHIP_OK) { zkhip_r1cs_free(r1cs_); zk_check(rc, "zkhip_crs_upload_ex"); }
    n_vars_ = cs.n_vars;
  }
  // The key PARTITIONED over the GPUs of a node (zkhip_multi_prover: a contiguous slice of every query vector and a prover instance
  // per entry of `devices`, partial sums added on the host, one tail; BASELINE configs[3]).  An index may repeat: {0, 0} is two
  // contexts on GPU 0.  The reference is one process that owns its prover (aggregator_server.cpp:106-118, 390-416): so is this.
  hip_proving_key(const zkhip_crs_desc& crs, const zkhip_r1cs_desc& cs, const std::vector<int>& devices, const zkhip_key_opts* opts = nullptr) {
    if (devices.empty()) throw std::runtime_error("hip_proving_key: empty device list");
    zk_check(zkhip_multi_prover_new(&crs, &cs, opts, devices.data(), (int)devices.size(), &multi_), "zkhip_multi_prover_new");
    int rc = zkhip_set_device(devices[0]);                       // (is_satisfied runs on the first GPU of the list)
    if (rc == ZKHIP_OK) rc = zkhip_r1cs_upload_ex(&cs, crs.domain_size, &r1cs_);
    if (rc != ZKHIP_OK) { zkhip_multi_prover_free(multi_); zk_check(rc, "zkhip_r1cs_upload_ex"); }
    n_vars_ = cs.n_vars;
  }
  hip_proving_key(const hip_proving_key&) = delete;
  hip_proving_key& operator=(const hip_proving_key&) = delete;
  ~hip_proving_key() { zkhip_multi_prover_free(multi_); zkhip_crs_free(crs_); zkhip_r1cs_free(r1cs_); }
  size_t num_devices() const { return multi_ ? (size_t)zkhip_multi_prover_size(multi_) : 1; }

  size_t num_variables() const { return n_vars_; }
  size_t domain_size() const { return zkhip_r1cs_domain_size(r1cs_); }          // the key's: 2^k (a reference key), or 2^k + 2^r
  unsigned log_domain_size() const { return zkhip_r1cs_log_domain(r1cs_); }      // ceil(log2 domain_size())

  // full assignment z = (1, primary, auxiliary), n_vars x 6 limbs
  bool is_satisfied(const uint64_t* z) const {
    int ok = 0;
    zk_check(zkhip_r1cs_is_satisfied(r1cs_, z, &ok), "zkhip_r1cs_is_satisfied");
    return ok != 0;
  }
  // r, s: the prover's randomisers (6 limbs each); libsnark draws them with Fr::random_element().
  groth16_proof generate_proof(const uint64_t* z, const uint64_t r[6], const uint64_t s[6]) const {
    uint64_t out[72];
    if (multi_) zk_check(zkhip_multi_prover_prove(multi_, z, r, s, out), "zkhip_multi_prover_prove");
    else zk_check(zkhip_groth16_prove(crs_, r1cs_, z, r, s, out), "zkhip_groth16_prove");
    groth16_proof p;
    std::memcpy(p.a.data(), out, 192);
    std::memcpy(p.b.data(), out + 24, 192);
    std::memcpy(p.c.data(), out + 48, 192);
    return p;
  }

 private:
  zkhip_crs* crs_ = nullptr;
  zkhip_r1cs* r1cs_ = nullptr;
  zkhip_multi_prover* multi_ = nullptr;     // set: the key lives in slices on several GPUs / contexts, crs_ stays null
  size_t n_vars_ = 0;
};

// ---------------------------------------------------------------------------------------------
// Policy-class adapter.  `baseT` is the reference's CPU policy class (libzeth::groth16_snark<ppT>);
// `bridgeT` converts its types to limb arrays (three small functions, shown in INTEGRATION.md):
//     static hip_proving_key*  bridgeT::upload(const typename baseT::proving_key&);      // once, cached
//     static void              bridgeT::assignment(const protoboard&, std::vector<uint64_t>& z);
//     static typename baseT::proof bridgeT::proof_from_limbs(const groth16_proof&);
//     static void              bridgeT::random_scalars(uint64_t r[6], uint64_t s[6]);
// Everything else (keypair I/O, verify, JSON, name) is inherited unchanged, so the server, the tests and
// the on-disk key format of the reference keep working.
template <class baseT, class bridgeT>
class groth16_snark_hip : public baseT {
 public:
  template <class protoboardT>
  static typename baseT::proof generate_proof(const typename baseT::proving_key& pk, const protoboardT& pb) {
    hip_proving_key* dev = bridgeT::upload(pk);          // HBM-resident after the first call
    std::vector<uint64_t> z;
    bridgeT::assignment(pb, z);                          // (1, primary_input, auxiliary_input) as limbs
    if (z.size() != dev->num_variables() * 6) throw std::runtime_error("assignment size does not match the proving key");
    uint64_t r[6], s[6];
    bridgeT::random_scalars(r, s);
    return bridgeT::proof_from_limbs(dev->generate_proof(z.data(), r, s));
  }
};

}  // namespace zecale_amd
