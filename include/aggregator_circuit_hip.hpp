// aggregator_circuit_hip.hpp - C++ mirror of libzecale::aggregator_circuit<wppT, wsnarkT, nverifierT, NumProofs>
// (reference libzecale/circuits/aggregator_circuit.hpp:32-114) over the zkhip C ABI: same member names, argument
// meaning and error behaviour, so that aggregator_server.cpp (:480-514, :318-319) and the reference's tests read the same.
//   explicit aggregator_circuit(size_t inputs_per_nested_proof);          hpp:85      (copy deleted, hpp:95-97)
//   keypair generate_trusted_setup() const;                               tcc:100-109
//   size_t num_primary_inputs() const;                                    tcc:172-180
//   const r1cs_constraint_system& get_constraint_system() const;          hpp:99-101
//   extended_proof prove(nested_vk, nested_proofs, proving_key);          tcc:120-170  (throws std::runtime_error on a
//                                                                         wrong nested input count, tcc:138-141)
// Types are plain limb containers (the libff/libsnark types are not in this image); INTEGRATION.md shows the bridge.
#pragma once
#include <array>
#include <cstdio>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "groth16_snark_hip.hpp"

namespace zecale_amd {

// Nested (BLS12-377) objects: coordinates are 6-limb Montgomery elements of Fr(BW6-761) = Fq(BLS12-377).
struct nested_verification_key {           // libzeth groth16 verification_key<npp>: alpha, beta, delta, ABC (testdata/dummy_app/vk.json)
  std::array<uint64_t, 12> alpha_g1;
  std::array<uint64_t, 24> beta_g2, delta_g2;
  std::vector<std::array<uint64_t, 12>> abc_g1;
  std::vector<uint64_t> flat() const {
    std::vector<uint64_t> v(alpha_g1.begin(), alpha_g1.end());
    v.insert(v.end(), beta_g2.begin(), beta_g2.end());
    v.insert(v.end(), delta_g2.begin(), delta_g2.end());
    for (const auto& p : abc_g1) v.insert(v.end(), p.begin(), p.end());
    return v;
  }
};
struct nested_proof {
  std::array<uint64_t, 12> a, c;
  std::array<uint64_t, 24> b;
};
struct nested_extended_proof {             // libzeth::extended_proof<npp, nsnark>: get_proof(), get_primary_inputs()
  nested_proof proof;
  std::vector<std::array<uint64_t, 6>> primary_inputs;
  const nested_proof& get_proof() const { return proof; }
  const std::vector<std::array<uint64_t, 6>>& get_primary_inputs() const { return primary_inputs; }
};

// libzeth::extended_proof<wpp, wsnark> (used at aggregator_circuit.tcc:167-169; write_json at aggregator_server.cpp:322)
struct extended_proof {
  groth16_proof proof;
  std::vector<std::array<uint64_t, 6>> primary_inputs;
  const groth16_proof& get_proof() const { return proof; }
  const std::vector<std::array<uint64_t, 6>>& get_primary_inputs() const { return primary_inputs; }

  static std::string hex_be(int which, const uint64_t* mont) {     // "0x" + fixed-width big-endian hex of the canonical value
    const int n = which == 0 ? 12 : 6;
    uint64_t c[12];
    zk_check(zkhip_to_canonical(which, mont, c), "zkhip_to_canonical");
    std::string s = "0x";
    char buf[17];
    for (int i = n - 1; i >= 0; i--) { std::snprintf(buf, sizeof buf, "%016llx", (unsigned long long)c[i]); s += buf; }
    return s;
  }
  // same shape as testdata/dummy_app/batch1.json "ext_proof" 
  std::string to_json() const {
    auto pt = [](const uint64_t* p) { return "[\"" + hex_be(0, p) + "\", \"" + hex_be(0, p + 12) + "\"]"; };
    std::ostringstream o;
    o << "{\"proof\": {\"a\": " << pt(proof.a.data()) << ", \"b\": " << pt(proof.b.data()) << ", \"c\": " << pt(proof.c.data())
      << "}, \"inputs\": [";
    for (size_t i = 0; i < primary_inputs.size(); i++) o << (i ? ", " : "") << "\"" << hex_be(1, primary_inputs[i].data()) << "\"";
    o << "]}";
    return o.str();
  }
};

class keypair {                            // wsnarkT::keypair: pk (HBM-resident) + vk
 public:
  // opts: the key's own table / launch options (a server that proves a stream asks for table_naf = 1); nullptr = the defaults
  explicit keypair(zkhip_keypair* kp, const zkhip_key_opts* opts = nullptr) : kp_(kp) {
    zkhip_crs_desc d;
    zk_check(zkhip_keypair_crs_desc(kp_, &d), "zkhip_keypair_crs_desc");
    zk_check(zkhip_crs_upload_ex(&d, opts, &crs_), "zkhip_crs_upload_ex");
  }
  keypair(const keypair&) = delete;
  keypair& operator=(const keypair&) = delete;
  ~keypair() { zkhip_crs_free(crs_); zkhip_keypair_free(kp_); }
  const zkhip_crs* pk() const { return crs_; }
  // points of the evaluation domain the key was generated on (its H query has one fewer): a prover follows the key
  size_t domain_size() const { zkhip_crs_desc d; zk_check(zkhip_keypair_crs_desc(kp_, &d), "zkhip_keypair_crs_desc"); return d.domain_size; }
  const zkhip_keypair* host() const { return kp_; }      // the key in host memory (what a multi-GPU stream uploads to every GPU)
  // vk: alpha (G1), beta, delta (G2), ABC; `vk_abc_size() == num_primary_inputs() + 1` is the server's sanity check (aggregator_server.cpp:490)
  size_t vk_abc_size() const { uint64_t a[24], b[24], d[24]; const uint64_t* abc; return zkhip_keypair_vk(kp_, a, b, d, &abc); }
  // wsnarkT::verification_key_write_json (aggregator_server.cpp:185,223): the shape of testdata/dummy_app/aggregator_vk.json
  std::string verification_key_to_json() const {
    uint64_t a[24], b[24], d[24];
    const uint64_t* abc;
    const size_t n = zkhip_keypair_vk(kp_, a, b, d, &abc);
    auto pt = [](const uint64_t* p) { return "[\"" + extended_proof::hex_be(0, p) + "\", \"" + extended_proof::hex_be(0, p + 12) + "\"]"; };
    std::ostringstream o;
    o << "{\"alpha\": " << pt(a) << ", \"beta\": " << pt(b) << ", \"delta\": " << pt(d) << ", \"ABC\": [";
    for (size_t i = 0; i < n; i++) o << (i ? ", " : "") << pt(abc + i * 24);
    o << "]}";
    return o.str();
  }
  bool verify(const extended_proof& ep) const {      // wsnarkT::verify(inputs, proof, vk)
    uint64_t a[24], b[24], d[24];
    const uint64_t* abc;
    size_t n = zkhip_keypair_vk(kp_, a, b, d, &abc);
    if (ep.primary_inputs.size() + 1 != n) return false;
    std::vector<uint64_t> in;
    for (const auto& x : ep.primary_inputs) in.insert(in.end(), x.begin(), x.end());
    uint64_t pr[72];
    std::memcpy(pr, ep.proof.a.data(), 192); std::memcpy(pr + 24, ep.proof.b.data(), 192); std::memcpy(pr + 48, ep.proof.c.data(), 192);
    int ok = 0;
    zk_check(zkhip_groth16_verify(a, b, d, abc, in.data(), ep.primary_inputs.size(), pr, &ok), "zkhip_groth16_verify");
    return ok != 0;
  }
 private:
  zkhip_keypair* kp_;
  zkhip_crs* crs_ = nullptr;
};

template <size_t NumProofs>
class aggregator_circuit {
 public:
  explicit aggregator_circuit(size_t inputs_per_nested_proof) : inputs_per_nested_proof_(inputs_per_nested_proof) {
    zk_check(zkhip_aggregator_new(NumProofs, inputs_per_nested_proof, &agg_), "zkhip_aggregator_new");
    zk_check(zkhip_aggregator_get_r1cs(agg_, &cs_), "zkhip_aggregator_get_r1cs");
  }
  aggregator_circuit(const aggregator_circuit&) = delete;
  aggregator_circuit& operator=(const aggregator_circuit&) = delete;
  ~aggregator_circuit() { if (r1cs_) zkhip_r1cs_free(r1cs_); zkhip_aggregator_free(agg_); }

  size_t num_primary_inputs() const { return zkhip_aggregator_num_primary_inputs(agg_); }
  const zkhip_r1cs_desc& get_constraint_system() const { return cs_; }

  // needs a device (the batch exponentiations run on the GPU); fresh toxic waste, uniform in Fr, from the OS, discarded on return.
  // domain_size: ZKHIP_DOMAIN_DEFAULT = the power of two libzeth's generate_setup forces (the reference: tcc:108; 65,536 points for
  // batch 2); ZKHIP_DOMAIN_STEP = libfqfft's unforced step domain (49,152: an option of this library, not a reference key).
  std::unique_ptr<keypair> generate_trusted_setup(size_t domain_size = ZKHIP_DOMAIN_DEFAULT) const {
    uint64_t t[4][6];
    for (auto& s : t) zk_check(zkhip_fr_random(s), "zkhip_fr_random");
    zkhip_keypair* kp = nullptr;
    zk_check(zkhip_groth16_setup_ex(&cs_, t[0], t[1], t[2], t[3], domain_size, &kp), "zkhip_groth16_setup_ex");
    return std::unique_ptr<keypair>(new keypair(kp));
  }

  // Streaming form for a server that wraps batch after batch (zkhip_aggregator_pipeline_*): submit() returns a ticket at
  // once, wait(ticket) the extended proof.  Witness generation, the GPU prover (gpu_slots proofs in flight) and the host
  // tail of successive batches overlap.  The stream must not outlive the circuit or the keypair.
  class stream {
   public:
    stream(aggregator_circuit& c, const keypair& kp, int gpu_slots, int witness_workers, bool witness_on_gpu = false) : c_(c) {
      zk_check(zkhip_aggregator_pipeline_new_ex(c.agg_, kp.pk(), gpu_slots, witness_workers, witness_on_gpu ? ZKHIP_PIPELINE_GPU_WITNESS : 0u, &p_),
               "zkhip_aggregator_pipeline_new_ex");
    }
    stream(const stream&) = delete;
    stream& operator=(const stream&) = delete;
    ~stream() { zkhip_aggregator_pipeline_free(p_); }
    uint64_t submit(const nested_verification_key& nested_vk, const std::array<const nested_extended_proof*, NumProofs>& nested_proofs) {
      std::vector<uint64_t> vk, proofs, inputs;
      c_.flatten(nested_vk, nested_proofs, vk, proofs, inputs);
      uint64_t r[6], s[6], ticket = 0;
      random_scalars(r, s);
      zk_check(zkhip_aggregator_pipeline_submit(p_, vk.data(), proofs.data(), inputs.data(), r, s, &ticket), "zkhip_aggregator_pipeline_submit");
      return ticket;
    }
    extended_proof wait(uint64_t ticket) {
      const size_t np = c_.num_primary_inputs();
      std::vector<uint64_t> prim(np * 6);
      uint64_t out[72];
      zk_check(zkhip_aggregator_pipeline_wait(p_, ticket, prim.data(), out), "zkhip_aggregator_pipeline_wait");
      extended_proof ep;
      std::memcpy(ep.proof.a.data(), out, 192); std::memcpy(ep.proof.b.data(), out + 24, 192); std::memcpy(ep.proof.c.data(), out + 48, 192);
      for (size_t i = 0; i < np; i++) {
        std::array<uint64_t, 6> x;
        std::memcpy(x.data(), &prim[i * 6], 48);
        ep.primary_inputs.push_back(x);
      }
      return ep;
    }
   private:
    aggregator_circuit& c_;
    zkhip_pipeline* p_ = nullptr;
  };
  // witness_on_gpu: the assignments are generated by a device kernel (zkhip_gpu_witness_*): fewer host cores, a deeper stream
  std::unique_ptr<stream> open_stream(const keypair& kp, int gpu_slots = 32, int witness_workers = 10, bool witness_on_gpu = false) {
    return std::unique_ptr<stream>(new stream(*this, kp, gpu_slots, witness_workers, witness_on_gpu));
  }

  // The same stream over EVERY GPU of a node (zkhip_dispatcher_*): a resident copy of the key and a pipeline per entry of `devices`
  // (an index may repeat), each batch goes to the entry with the fewest batches outstanding.  One process, as the reference server
  // is (aggregator_server.cpp:106-118, 390-416); whole proofs are independent, so there is no exchange between the GPUs.
  class node_stream {
   public:
    node_stream(aggregator_circuit& c, const keypair& kp, const std::vector<int>& devices, int gpu_slots, int witness_workers,
                bool witness_on_gpu = false, const zkhip_key_opts* opts = nullptr) : c_(c) {
      zkhip_crs_desc d;
      zk_check(zkhip_keypair_crs_desc(kp.host(), &d), "zkhip_keypair_crs_desc");
      zk_check(zkhip_dispatcher_new(c.agg_, &d, opts, devices.data(), (int)devices.size(), gpu_slots, witness_workers,
                                    witness_on_gpu ? ZKHIP_PIPELINE_GPU_WITNESS : 0u, &d_), "zkhip_dispatcher_new");
    }
    node_stream(const node_stream&) = delete;
    node_stream& operator=(const node_stream&) = delete;
    ~node_stream() { zkhip_dispatcher_free(d_); }
    uint64_t submit(const nested_verification_key& nested_vk, const std::array<const nested_extended_proof*, NumProofs>& nested_proofs) {
      std::vector<uint64_t> vk, proofs, inputs;
      c_.flatten(nested_vk, nested_proofs, vk, proofs, inputs);
      uint64_t r[6], s[6], ticket = 0;
      random_scalars(r, s);
      zk_check(zkhip_dispatcher_submit(d_, vk.data(), proofs.data(), inputs.data(), r, s, &ticket), "zkhip_dispatcher_submit");
      return ticket;
    }
    extended_proof wait(uint64_t ticket) {
      const size_t np = c_.num_primary_inputs();
      std::vector<uint64_t> prim(np * 6);
      uint64_t out[72];
      zk_check(zkhip_dispatcher_wait(d_, ticket, prim.data(), out), "zkhip_dispatcher_wait");
      extended_proof ep;
      std::memcpy(ep.proof.a.data(), out, 192); std::memcpy(ep.proof.b.data(), out + 24, 192); std::memcpy(ep.proof.c.data(), out + 48, 192);
      for (size_t i = 0; i < np; i++) {
        std::array<uint64_t, 6> x;
        std::memcpy(x.data(), &prim[i * 6], 48);
        ep.primary_inputs.push_back(x);
      }
      return ep;
    }
    std::vector<size_t> batches_per_device() const {          // how many batches each entry of the device list has been given
      std::vector<size_t> v((size_t)zkhip_dispatcher_size(d_));
      zk_check(zkhip_dispatcher_stats(d_, v.data()), "zkhip_dispatcher_stats");
      return v;
    }
   private:
    aggregator_circuit& c_;
    zkhip_dispatcher* d_ = nullptr;
  };
  std::unique_ptr<node_stream> open_node_stream(const keypair& kp, const std::vector<int>& devices, int gpu_slots = 32, int witness_workers = 10,
                                           bool witness_on_gpu = false, const zkhip_key_opts* opts = nullptr) {
    return std::unique_ptr<node_stream>(new node_stream(*this, kp, devices, gpu_slots, witness_workers, witness_on_gpu, opts));
  }

  // Non-const like the reference (it fills its protoboard); not re-entrant.
  extended_proof prove(const nested_verification_key& nested_vk,
                       const std::array<const nested_extended_proof*, NumProofs>& nested_proofs, const keypair& kp) {
    std::vector<uint64_t> vk, proofs, inputs, z(cs_.n_vars * 6);
    flatten(nested_vk, nested_proofs, vk, proofs, inputs);
    int well_formed = 0;                   // libsnark: proof.is_well_formed(); in the circuit: the proof variables' curve checks
    zk_check(zkhip_aggregator_check_inputs(agg_, vk.data(), proofs.data(), &well_formed), "zkhip_aggregator_check_inputs");
    if (!well_formed) throw std::runtime_error("nested proof or verification key has a point that is not on its curve");
    zk_check(zkhip_aggregator_witness(agg_, vk.data(), proofs.data(), inputs.data(), z.data()), "zkhip_aggregator_witness");
    if (!r1cs_) zk_check(zkhip_r1cs_upload_ex(&cs_, kp.domain_size(), &r1cs_), "zkhip_r1cs_upload_ex");   // (zkhip_groth16_prove follows the key from then on)
    uint64_t r[6], s[6];
    random_scalars(r, s);
    uint64_t out[72];
    zk_check(zkhip_groth16_prove(kp.pk(), r1cs_, z.data(), r, s, out), "zkhip_groth16_prove");
    extended_proof ep;
    std::memcpy(ep.proof.a.data(), out, 192); std::memcpy(ep.proof.b.data(), out + 24, 192); std::memcpy(ep.proof.c.data(), out + 48, 192);
    for (size_t i = 0; i < num_primary_inputs(); i++) {
      std::array<uint64_t, 6> x;
      std::memcpy(x.data(), &z[(i + 1) * 6], 48);
      ep.primary_inputs.push_back(x);
    }
    return ep;
  }

 private:
  // argument checks of the reference's prove (tcc:138-141) + the flat limb arrays the C ABI takes
  void flatten(const nested_verification_key& nested_vk, const std::array<const nested_extended_proof*, NumProofs>& nested_proofs,
               std::vector<uint64_t>& vk, std::vector<uint64_t>& proofs, std::vector<uint64_t>& inputs) const {
    for (size_t i = 0; i < NumProofs; i++) {
      const auto& in = nested_proofs[i]->get_primary_inputs();
      if (in.size() != inputs_per_nested_proof_)
        throw std::runtime_error("attempt to aggregate proof with invalid number of inputs");            // tcc:138-141, same text
      const nested_proof& p = nested_proofs[i]->get_proof();
      proofs.insert(proofs.end(), p.a.begin(), p.a.end());
      proofs.insert(proofs.end(), p.b.begin(), p.b.end());
      proofs.insert(proofs.end(), p.c.begin(), p.c.end());
      for (const auto& x : in) inputs.insert(inputs.end(), x.begin(), x.end());
    }
    if (nested_vk.abc_g1.size() != inputs_per_nested_proof_ + 1) throw std::runtime_error("nested verification key has the wrong size");
    vk = nested_vk.flat();
  }
  static void random_scalars(uint64_t r[6], uint64_t s[6]) {      // uniform in Fr (libff::Fr::random_element())
    zk_check(zkhip_fr_random(r), "zkhip_fr_random");
    zk_check(zkhip_fr_random(s), "zkhip_fr_random");
  }
  size_t inputs_per_nested_proof_;
  zkhip_aggregator* agg_ = nullptr;
  zkhip_r1cs_desc cs_;
  zkhip_r1cs* r1cs_ = nullptr;
};

}  // namespace zecale_amd
