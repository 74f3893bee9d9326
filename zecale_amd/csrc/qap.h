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
  int spmv_log_lanes = 4;      // lanes per row of the sparse products: 2^4 (a proof alone) or 2^2 (a prover that shares the chip)
  size_t n_constraints = 0, n_vars = 0, n_primary = 0;   // n_vars counts the constant ONE
  int log_d = 0;
  CsrDev A, B, C;
  // work buffers (packed device form, d elements each)
  uint32_t *bufA = nullptr, *bufB = nullptr, *bufC = nullptr, *tmp = nullptr, *z = nullptr;
  uint32_t* zinv = nullptr;   // 1 / (g^d - 1), 14 limbs
};

int r1cs_upload(const zkhip_r1cs_desc* d, R1csDev** out, char* err, size_t errlen);
void r1cs_free(R1csDev* r);
// z: device pointer, ABI form (n_vars x 6 u64).  Leaves h (packed device form, d elements) in r->bufA.
int qap_h_dev(R1csDev* r, const uint64_t* d_z_abi, hipStream_t st, char* err, size_t errlen);
// satisfiability check <A_i,z><B_i,z> = <C_i,z> for all i (reference: _pb.is_satisfied(), aggregator_circuit.tcc:159-164)
int r1cs_is_satisfied_dev(R1csDev* r, const uint64_t* d_z_abi, hipStream_t st, int* ok, char* err, size_t errlen);

}  // namespace zkhip
