// zkhip_aggregator: the wrapping circuit's constraint system (host) and, built on first use, the GPU witness program.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <mutex>
#include <vector>

struct zkhip_aggregator {
  size_t num_proofs, inputs_per_proof;
  size_t n_vars = 0, n_primary = 0, n_constraints = 0;
  std::vector<uint32_t> rp[3], col[3];
  std::vector<uint64_t> val[3];
  // GPU witness generator (witness.hip): the tape and its per-device uploads; owned through these two opaque members so that
  // the host-only translation unit needs no HIP types
  std::mutex gpu_mu;
  void* gpu_state = nullptr;
  void (*gpu_release)(zkhip_aggregator*) = nullptr;
};
