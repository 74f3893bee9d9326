// r1cs_to_qap_witness_map on gfx950: three sparse matrix-vector
// products over Fr, 3 iFFT + 3 cosetFFT, H = (A o B - C) / Z on the coset, 1 icosetFFT.
// Reached in the reference from wsnarkT::generate_proof (libzecale/circuits/aggregator_circuit.tcc:168);
// the optional satisfiability check mirrors _pb.is_satisfied() (aggregator_circuit.tcc:159-164).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "fp29.cuh"
#include "host_field.hpp"
#include "ntt.h"
#include "qap.h"

namespace zkhip {

typedef Fp<FrParams> FrD;

__device__ __forceinline__ FrD q_load12(const uint32_t* p) {
  uint32_t w[12];
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1], c = q[2];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
  return fp_unpack32<FrParams>(w);
}
__device__ __forceinline__ void q_store12(uint32_t* p, const FrD& v) {
  uint32_t w[12];
  fp_pack32<FrParams>(v, w);
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]);
  q[1] = make_uint4(w[4], w[5], w[6], w[7]);
  q[2] = make_uint4(w[8], w[9], w[10], w[11]);
}

// out[i] = <M_i, z> for i < n ; for the A matrix also out[n + k] = z_k, k <= n_primary ; zero up to d.
// SIXTEEN lanes per row: lane j takes the row's terms j, j+16, ...; the partial sums are folded with DPP row
// shifts.  (One lane per row is latency-bound by the longest rows - a packing constraint has 253 terms, an Fq12
// multiplication constraint 47 - and a wrapping circuit has fewer rows than the chip has lanes.)
// Products are accumulated lazily (each < 2r) and folded every 32 terms.
template <int LANES>
__device__ __forceinline__ FrD fr_shfl_down(const FrD& v, int delta) {
  FrD r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.l[i] = (uint32_t)__shfl_down((int)v.l[i], delta, LANES);
  return r;
}

// LG: log2 of the lanes per row - 4 for a proof alone (latency), 2 for a prover that shares the chip (a quarter of the lane-cycles:
// every lane of a row's group runs the fold and the final normalisations whether it had terms or not).
// the three matrices of a system in one launch: blockIdx.y selects A, B or C
struct Spmv3 {
  const uint32_t *row_ptr[3], *col[3], *val[3];
  uint32_t* out[3];
  uint32_t extra[3];        // n_primary + 1 for A, else 0
};
template <int LG>
__global__ void __launch_bounds__(256) k_spmv(Spmv3 m, const uint32_t* __restrict__ z, uint32_t n, uint32_t d,
                                               int log_k, int log_n2 /* log_k != 0: out in the NTT's transposed order */) {
  const uint32_t* __restrict__ row_ptr = m.row_ptr[blockIdx.y];
  const uint32_t* __restrict__ col = m.col[blockIdx.y];
  const uint32_t* __restrict__ val = m.val[blockIdx.y];
  uint32_t* __restrict__ out = m.out[blockIdx.y];
  const uint32_t extra = m.extra[blockIdx.y];
  uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr uint32_t LANES = 1u << LG;
  uint32_t i = gt >> LG, sub = gt & (LANES - 1);
  if (i >= d) return;                       // whole 16-lane groups leave together
  FrD acc = fp_zero<FrParams>();
  if (i < n) {
    uint32_t k0 = row_ptr[i], k1 = row_ptr[i + 1];
    uint32_t cnt = 0;
    for (uint32_t k = k0 + sub; k < k1; k += LANES) {
      FrD p = fp_mul(q_load12(val + (size_t)k * 12), q_load12(z + (size_t)col[k] * 12));
      acc = fp_add(acc, p);
      if ((++cnt & 31u) == 0) acc = fp_mul(acc, fp_one<FrParams>());   // back to < 2r
    }
    if (cnt > 1) acc = fp_mul(acc, fp_one<FrParams>());                 // every partial sum < 2r
    // fold the 16 partial sums (each < 2r; the total < 32r fits the lazy bound; one final fold below)
    for (int delta = LANES / 2; delta >= 1; delta >>= 1) acc = fp_add(acc, fr_shfl_down<LANES>(acc, delta));
    if (sub == 0) acc = fp_mul(acc, fp_one<FrParams>());                // stored values are always < 2r
  } else if (i < n + extra) {
    acc = q_load12(z + (size_t)(i - n) * 12);
  }
  const size_t loc = log_k ? ((size_t)(i & ((1u << log_k) - 1)) << log_n2) + (i >> log_k) : i;
  if (sub == 0) q_store12(out + loc * 12, acc);
}

// H[i] = (A[i] B[i] - C[i]) * zinv   (in place into A)
__global__ void __launch_bounds__(256) k_h_pointwise(uint32_t* __restrict__ A, const uint32_t* __restrict__ B,
                                                      const uint32_t* __restrict__ C, const uint32_t* __restrict__ zinv, uint32_t d) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d) return;
  FrD ab = fp_mul(q_load12(A + (size_t)i * 12), q_load12(B + (size_t)i * 12));
  FrD c = fp_mul(q_load12(C + (size_t)i * 12), fp_one<FrParams>());   // < 2r whatever the input bound
  FrD t = fp_sub<FrParams, 2>(ab, c);
  FrD zi;
#pragma unroll
  for (int k = 0; k < 14; k++) zi.l[k] = zinv[k];
  q_store12(A + (size_t)i * 12, fp_mul(t, zi));
}

// flag = 1 if some row has <A,z><B,z> != <C,z>   (inputs: the three SpMV outputs)
__global__ void __launch_bounds__(256) k_check_sat(const uint32_t* __restrict__ A, const uint32_t* __restrict__ B,
                                                    const uint32_t* __restrict__ C, uint32_t n, uint32_t* __restrict__ flag) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  FrD ab = fp_mul(q_load12(A + (size_t)i * 12), q_load12(B + (size_t)i * 12));
  FrD c = fp_mul(q_load12(C + (size_t)i * 12), fp_one<FrParams>());
  FrD t = fp_mul(fp_sub<FrParams, 2>(ab, c), fp_one<FrParams>());     // < 2r
  if (!fp_is_zero_2p(t)) atomicOr(flag, 1u);
}

// ------------------------------------------------------------------------------------------
#define Q_HIP(x)                                                                                   \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) { snprintf(err, errlen, "%s: %s", #x, hipGetErrorString(e_)); return ZKHIP_ERR_HIP; } \
  } while (0)

static int csr_upload(const uint32_t* row_ptr, const uint32_t* col, const uint64_t* val, size_t n, CsrDev* out, char* err, size_t errlen) {
  size_t nnz = row_ptr[n];
  out->nnz = nnz;
  Q_HIP(hipMalloc(&out->row_ptr, (n + 1) * 4));
  Q_HIP(hipMemcpy(out->row_ptr, row_ptr, (n + 1) * 4, hipMemcpyHostToDevice));
  Q_HIP(hipMalloc(&out->col, (nnz ? nnz : 1) * 4));
  Q_HIP(hipMalloc(&out->val, (nnz ? nnz : 1) * 48));
  if (nnz) {
    Q_HIP(hipMemcpy(out->col, col, nnz * 4, hipMemcpyHostToDevice));
    uint64_t* tmp;
    Q_HIP(hipMalloc(&tmp, nnz * 48));
    hipError_t e = hipMemcpy(tmp, val, nnz * 48, hipMemcpyHostToDevice);
    if (e == hipSuccess) { fr_abi_to_dev(tmp, out->val, nnz, 0); e = hipDeviceSynchronize(); }
    (void)hipFree(tmp);
    Q_HIP(e);
  }
  return ZKHIP_OK;
}

static int r1cs_upload_impl(const zkhip_r1cs_desc* d, R1csDev* r, char* err, size_t errlen);
int r1cs_upload(const zkhip_r1cs_desc* d, R1csDev** out, char* err, size_t errlen) {
  if (!d || !out || d->n_vars < d->n_primary + 1) { snprintf(err, errlen, "r1cs_upload: bad descriptor"); return ZKHIP_ERR_ARG; }
  for (size_t k = 0; k < 3; k++) {
    const uint32_t* rp = k == 0 ? d->a_row_ptr : k == 1 ? d->b_row_ptr : d->c_row_ptr;
    const uint32_t* cl = k == 0 ? d->a_col : k == 1 ? d->b_col : d->c_col;
    if (!rp) { snprintf(err, errlen, "r1cs_upload: null row_ptr"); return ZKHIP_ERR_ARG; }
    for (size_t i = 0; i < d->n_constraints; i++)
      if (rp[i] > rp[i + 1]) { snprintf(err, errlen, "r1cs_upload: row_ptr not monotone"); return ZKHIP_ERR_ARG; }
    for (size_t j = 0; j < rp[d->n_constraints]; j++)
      if (cl[j] >= d->n_vars) { snprintf(err, errlen, "r1cs_upload: column index out of range"); return ZKHIP_ERR_ARG; }
  }
  R1csDev* r = new R1csDev();
  int rc = r1cs_upload_impl(d, r, err, errlen);
  if (rc != ZKHIP_OK) { r1cs_free(r); return rc; }        // frees what was uploaded before the failure
  *out = r;
  return ZKHIP_OK;
}

static int r1cs_upload_impl(const zkhip_r1cs_desc* d, R1csDev* r, char* err, size_t errlen) {
  using host::HFr;
  r->n_constraints = d->n_constraints; r->n_vars = d->n_vars; r->n_primary = d->n_primary;
  size_t need = d->n_constraints + d->n_primary + 1;
  int lg = 0;
  while (((size_t)1 << lg) < need) lg++;
  if (lg > 22) { snprintf(err, errlen, "r1cs_upload: domain larger than 2^22"); return ZKHIP_ERR_ARG; }
  r->log_d = lg;
  size_t dd = (size_t)1 << lg;
  int rc;
  if ((rc = csr_upload(d->a_row_ptr, d->a_col, d->a_val, d->n_constraints, &r->A, err, errlen)) != ZKHIP_OK) return rc;
  if ((rc = csr_upload(d->b_row_ptr, d->b_col, d->b_val, d->n_constraints, &r->B, err, errlen)) != ZKHIP_OK) return rc;
  if ((rc = csr_upload(d->c_row_ptr, d->c_col, d->c_val, d->n_constraints, &r->C, err, errlen)) != ZKHIP_OK) return rc;
  Q_HIP(hipMalloc(&r->bufA, dd * 48)); Q_HIP(hipMalloc(&r->bufB, dd * 48)); Q_HIP(hipMalloc(&r->bufC, dd * 48));
  Q_HIP(hipMalloc(&r->tmp, 256)); Q_HIP(hipMalloc(&r->z, d->n_vars * 48));     // tmp: the satisfiability flag
  // 1 / (g^d - 1): Z is constant on the coset g<omega>
  HFr g = HFr::from_limbs(FrParams::GEN64);
  uint64_t e[1] = {(uint64_t)dd};
  HFr zc = (g.pow_limbs(e, 1) - HFr::one()).inv();
  uint64_t l[6];
  zc.to_limbs(l);
  FrD zd = fp_cond_sub_p(fp_from_abi<FrParams>(l));
  Q_HIP(hipMalloc(&r->zinv, 14 * 4));
  Q_HIP(hipMemcpy(r->zinv, zd.l, 14 * 4, hipMemcpyHostToDevice));
  return ZKHIP_OK;
}

void r1cs_free(R1csDev* r) {
  if (!r) return;
  void* ptrs[] = {r->A.row_ptr, r->A.col, r->A.val, r->B.row_ptr, r->B.col, r->B.val, r->C.row_ptr, r->C.col, r->C.val,
                  r->bufA, r->bufB, r->bufC, r->tmp, r->z, r->zinv};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  delete r;
}

// the three products, in the order the first transform wants (transposed for 2^12 rows and more, see ntt.h)
static void spmv3(R1csDev* r, hipStream_t st) {
  uint32_t n = (uint32_t)r->n_constraints, d = 1u << r->log_d;
  const int lk = ntt_layout_logk(r->log_d), ln = lk ? r->log_d - lk : 0;
  const CsrDev* M[3] = {&r->A, &r->B, &r->C};
  Spmv3 m;
  for (int k = 0; k < 3; k++) { m.row_ptr[k] = M[k]->row_ptr; m.col[k] = M[k]->col; m.val[k] = M[k]->val; m.extra[k] = k == 0 ? (uint32_t)r->n_primary + 1 : 0u; }
  m.out[0] = r->bufA; m.out[1] = r->bufB; m.out[2] = r->bufC;
  if (r->spmv_log_lanes == 2)
    hipLaunchKernelGGL(k_spmv<2>, dim3((unsigned)(((size_t)d * 4 + 255) / 256), 3), dim3(256), 0, st, m, r->z, n, d, lk, ln);
  else
    hipLaunchKernelGGL(k_spmv<4>, dim3((unsigned)(((size_t)d * 16 + 255) / 256), 3), dim3(256), 0, st, m, r->z, n, d, lk, ln);
}

int qap_h_dev(R1csDev* r, const uint64_t* d_z_abi, hipStream_t st, char* err, size_t errlen) {
  const int lg = r->log_d;
  const uint32_t d = 1u << lg;
  fr_abi_to_dev(d_z_abi, r->z, r->n_vars, st);
  spmv3(r, st);
  int rc;
  uint32_t* bufs[3] = {r->bufA, r->bufB, r->bufC};
  if ((rc = ntt_dev_packed_batch(bufs, 3, lg, 1, 0, 1, st, err, errlen)) != ZKHIP_OK) return rc;   // iFFT of A, B, C: transposed -> natural
  if ((rc = ntt_dev_packed_batch(bufs, 3, lg, 0, 1, 0, st, err, errlen)) != ZKHIP_OK) return rc;   // cosetFFT: natural -> transposed
  hipLaunchKernelGGL(k_h_pointwise, dim3((d + 255) / 256), dim3(256), 0, st, r->bufA, r->bufB, r->bufC, r->zinv, d);   // (any order)
  if ((rc = ntt_dev_packed(r->bufA, lg, 1, 1, 1, st, err, errlen)) != ZKHIP_OK) return rc;       // icosetFFT: transposed -> natural
  Q_HIP(hipGetLastError());
  return ZKHIP_OK;
}

int r1cs_is_satisfied_dev(R1csDev* r, const uint64_t* d_z_abi, hipStream_t st, int* ok, char* err, size_t errlen) {
  fr_abi_to_dev(d_z_abi, r->z, r->n_vars, st);
  spmv3(r, st);
  uint32_t* flag = r->tmp;   // first word of the scratch buffer
  Q_HIP(hipMemsetAsync(flag, 0, 4, st));
  // every place of the three vectors (whatever their order): the rows past the constraints have B = C = 0
  uint32_t n = 1u << r->log_d;
  hipLaunchKernelGGL(k_check_sat, dim3((n + 255) / 256), dim3(256), 0, st, r->bufA, r->bufB, r->bufC, n, flag);
  uint32_t h = 0;
  Q_HIP(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, st));
  Q_HIP(hipStreamSynchronize(st));
  *ok = (h == 0);
  return ZKHIP_OK;
}

}  // namespace zkhip
