// Internal interface of the NTT engine (ntt.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/zkhip.h"

namespace zkhip {
// packed device form: 12 u32 per element (value * 2^406 mod r, < 2^384)
// In place.  A vector of 2^12 elements or more is in natural order on one side of a transform and in TRANSPOSED order on the
// other: with K = 2^ntt_layout_logk(log_d) and N2 = 2^log_d / K, element k1 + K*k2 lives at k1*N2 + k2.  in_transposed says on
// which side the input is (smaller vectors: natural on both sides, ntt_layout_logk = 0).
int ntt_dev_packed(uint32_t* d_data, int log_d, int inverse, int coset, int in_transposed, hipStream_t st, char* err, size_t errlen);
int ntt_dev_packed_batch(uint32_t* const* d_bufs, int nbuf, int log_d, int inverse, int coset, int in_transposed, hipStream_t st, char* err, size_t errlen);   // up to 3 vectors, same launches
int ntt_layout_logk(int log_d);
int ntt_measure(int log_d, int inverse, int coset, int batch, int reps, double* ms_per_transform, char* err, size_t errlen);   // the passes alone, HIP events
int ntt_dev_abi(uint64_t* d_data, int log_d, int inverse, int coset, char* err, size_t errlen);
void fr_abi_to_dev(const uint64_t* d_in, uint32_t* d_out, size_t n, hipStream_t st);
void fr_abi_to_dev_merge(const uint64_t* d_in, const uint64_t* d_in2, uint32_t* d_out, size_t n, hipStream_t st);   // limb-wise OR of two disjoint parts
void fr_dev_to_abi(const uint32_t* d_in, uint64_t* d_out, size_t n, hipStream_t st);
}  // namespace zkhip
