// Internal interface of the GPU witness generator (witness.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "aggregator_internal.h"
#include "witness_tape.h"

namespace zkhip {
constexpr int WT_SUBK_LEVELS = 12;        // K = 2^0 .. 2^11
struct WitnessProg {
  const uint8_t* code;
  const int32_t *a, *b;
  const uint32_t* level_start;
  const int32_t* out_ref;
  const uint32_t* consts;          // a value slot each: 14 limbs of the device form in 16 words
  uint32_t n_levels, n_pos, n_vars, n_inputs;
  uint32_t chain_start;            // positions [chain_start, n_pos): the key-hash chain (k_witness_chain)
  const uint32_t* subk;            // WT_SUBK_LEVELS value slots: 2^k r in subtraction-safe limbs
  uint32_t mu;                     // floor(2^390 / r) or one less (w_reduce)
};
// a program's device copy (seven buffers) and the argument block the kernels take
struct WitnessProgDev {
  WitnessProg prog;
  void* bufs[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};
// uploads a recorded program to the calling thread's current device / frees it (the device must be current)
int witness_prog_upload(const WitnessTape& T, WitnessProgDev* out, char* err, size_t errlen);
void witness_prog_free(WitnessProgDev* pd);
// the program of `a` on the calling thread's current device (recorded and uploaded on first use)
int witness_prog(zkhip_aggregator* a, WitnessProg* out, const WitnessTape** tape, char* err, size_t errlen);
// one workgroup per batch; inputs: batches x n_inputs x 6 u64 (ABI form: nested key | proofs | inputs); values: batches x n_pos x 16 u32 (witness_value_bytes);
// z: batches x n_vars x 6 u64 (ABI form); flags: one word per batch, set when an inversion met zero (cleared by the caller)
constexpr size_t witness_value_bytes = 64;      // one value slot
void witness_launch(const WitnessProg& P, const uint64_t* d_inputs, uint32_t* d_values, uint64_t* d_z, uint32_t* d_flags, uint32_t batches,
                    hipStream_t st, hipStream_t st_chain, hipEvent_t ev_fork, hipEvent_t ev_join);
}  // namespace zkhip
