// zkhip_aggregator: the wrapping circuit's constraint system (host) and, built on first use, the GPU witness program.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <mutex>
#include <vector>

struct zkhip_aggregator {
  size_t num_proofs, inputs_per_proof;
  size_t n_vars = 0, n_primary = 0, n_constraints = 0;
  // where the sections of the assignment begin (circuit/sections.hpp): [0, sec_hash) the constant ONE, the primary inputs, the
  // nested key's and the nested proofs' variables; [sec_hash, sec_key) the MiMC hash of the key; [sec_key, sec_proofs) the lines
  // of -beta and -delta; [sec_proofs, n_vars) the proof sections
  size_t sec_hash = 0, sec_key = 0, sec_proofs = 0;
  std::vector<uint32_t> rp[3], col[3];
  std::vector<uint64_t> val[3];
  // GPU witness generator (witness.hip): the tape and its per-device uploads; owned through these two opaque members so that
  // the host-only translation unit needs no HIP types
  std::mutex gpu_mu;
  void* gpu_state = nullptr;
  void (*gpu_release)(zkhip_aggregator*) = nullptr;
};

// ---- per-application constants (the reference registers a nested key once: aggregator_server.cpp:170-235) ----
// Host part (aggregator.cpp): the nested key with the lines of -beta and -delta computed once; a witness generator that runs the
// proof sections only and leaves ZERO at the application's constant positions.
//   zk_app_host_new       runs the key section for `nested_vk`; *state is freed by zk_app_host_free
//   zk_app_host_witness   z_out: n_vars x 6 limbs; z[1] = vk_hash, z[2] = packed results, zeros at s_idx (sorted auxiliary positions,
//                         all above n_primary) and over the hash and key sections, everything else as zkhip_aggregator_witness
extern "C" int zk_app_host_new(const zkhip_aggregator* a, const uint64_t* nested_vk, void** state);
extern "C" void zk_app_host_free(void* state);
extern "C" int zk_app_host_witness(const zkhip_aggregator* a, const void* state, const uint64_t* nested_vk, const uint64_t* nested_proofs,
                                   const uint64_t* nested_inputs, const uint32_t* s_idx, size_t n_s, const uint64_t vk_hash[6], uint64_t* z_out);
