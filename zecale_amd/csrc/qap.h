// Internal interface of the R1CS -> QAP witness map (qap.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/zkhip.h"

namespace zkhip {

struct CsrDev {
  uint32_t* row_ptr = nullptr;   // n + 1
  uint32_t* col = nullptr;       // nnz
  uint32_t* val = nullptr;       // nnz x 12 packed words (device form)
  size_t nnz = 0;
};

struct R1csDev {
  int spmv_log_lanes = 4;      // lanes per row of the sparse products: 2^4 (a proof alone), 2^2 (a prover that shares the chip), 2^0 (short rows)
  int spmv_log_lanes_alone = 4;   // what r1cs_upload chose for a proof alone (by the average row length)
  size_t n_constraints = 0, n_vars = 0, n_primary = 0;   // n_vars counts the constant ONE
  // The evaluation domain (domain.hpp): d points - a power of two (small == 0, big == d; the reference's forced choice and the
  // default), or big + small with both powers of two (libfqfft's step_radix2_domain, by option or by the key).  log_d = ceil(log2 d).
  size_t d = 0, big = 0, small = 0;
  int log_d = 0, log_big = 0, log_small = 0;
  CsrDev A, B, C;
  // work buffers (packed device form, d elements each)
  uint32_t *bufA = nullptr, *bufB = nullptr, *bufC = nullptr, *tmp = nullptr, *z = nullptr;
  uint32_t* zinv = nullptr;   // 1 / Z on the coset g x, 14 limbs per class: one class for a radix-2 domain (g^d - 1); a step domain has
                              // big / small classes on its big part (by i mod (big / small)) and one more for its small part
  // step domains: powers in packed device form - w^i (i < big), w^-i (i < small), g^i and g^-i (i < d) - and 1/2 (14 limbs)
  uint32_t *pw_w = nullptr, *pw_winv = nullptr, *pw_g = nullptr, *pw_ginv = nullptr, *half = nullptr;
};

// domain_size: 0 = the reference's forced power of two for n + l + 1 points (default); (size_t)-1 = libfqfft's unforced choice
// (ZKHIP_DOMAIN_STEP); else a valid domain size >= n + l + 1 (what a proving key was generated on) - domain.hpp
int r1cs_upload(const zkhip_r1cs_desc* d, size_t domain_size, R1csDev** out, char* err, size_t errlen);
// moves the handle to another domain (no-op if it is there already); nothing may be in flight on it
int r1cs_set_domain(R1csDev* r, size_t domain_size, char* err, size_t errlen);
void r1cs_free(R1csDev* r);
// z: device pointer, ABI form (n_vars x 6 u64).  Leaves h (packed device form, d elements) in r->bufA.
// d_z_app != null: d_z_abi is MASKED (zeros at an application's constant positions) and d_z_app (n_vars x 6, zeros everywhere else)
// holds those constants: the map runs on their union
int qap_h_dev(R1csDev* r, const uint64_t* d_z_abi, hipStream_t st, char* err, size_t errlen, const uint64_t* d_z_app = nullptr);
// satisfiability check <A_i,z><B_i,z> = <C_i,z> for all i (reference: _pb.is_satisfied(), aggregator_circuit.tcc:159-164)
int r1cs_is_satisfied_dev(R1csDev* r, const uint64_t* d_z_abi, hipStream_t st, int* ok, char* err, size_t errlen);

}  // namespace zkhip
