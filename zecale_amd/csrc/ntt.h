// Internal interface of the NTT engine (ntt.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/zkhip.h"

namespace zkhip {
// packed device form: 12 u32 per element (value * 2^406 mod r, < 2^384)
int ntt_dev_packed(uint32_t* d_data, uint32_t* d_tmp, int log_d, int inverse, int coset, hipStream_t st, char* err, size_t errlen);
int ntt_dev_abi(uint64_t* d_data, int log_d, int inverse, int coset, char* err, size_t errlen);
void fr_abi_to_dev(const uint64_t* d_in, uint32_t* d_out, size_t n, hipStream_t st);
void fr_dev_to_abi(const uint32_t* d_in, uint64_t* d_out, size_t n, hipStream_t st);
}  // namespace zkhip
