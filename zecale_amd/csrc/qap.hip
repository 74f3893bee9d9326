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
#include "domain.hpp"
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
// Where element i of a vector lives when the transform that reads it next wants its input TRANSPOSED (ntt.h): a radix-2 domain
// is one part of d points; a step domain is a part of `big` points followed by a part of d - big points, each in the order of
// its own transform.
struct PartLayout { uint32_t big; int log_k0, log_n20, log_k1, log_n21; };
__device__ __forceinline__ size_t part_loc(const PartLayout& L, uint32_t i) {
  if (i < L.big) return L.log_k0 ? ((size_t)(i & ((1u << L.log_k0) - 1)) << L.log_n20) + (i >> L.log_k0) : i;
  const uint32_t j = i - L.big;
  return (size_t)L.big + (L.log_k1 ? ((size_t)(j & ((1u << L.log_k1) - 1)) << L.log_n21) + (j >> L.log_k1) : j);
}
// the inverse: which element lives at location loc
__device__ __forceinline__ uint32_t part_index(const PartLayout& L, uint32_t loc) {
  if (loc < L.big) return L.log_k0 ? ((loc & ((1u << L.log_n20) - 1)) << L.log_k0) + (loc >> L.log_n20) : loc;
  const uint32_t j = loc - L.big;
  return L.big + (L.log_k1 ? ((j & ((1u << L.log_n21) - 1)) << L.log_k1) + (j >> L.log_n21) : j);
}

template <int LG>
__global__ void __launch_bounds__(256) k_spmv(Spmv3 m, const uint32_t* __restrict__ z, uint32_t n, uint32_t d, PartLayout lay) {
#ifndef ZK_SHORT_KERNEL_PRIO_LEVEL
#define ZK_SHORT_KERNEL_PRIO_LEVEL 3
#endif
  if (ZK_SHORT_KERNEL_PRIO_LEVEL) __builtin_amdgcn_s_setprio(ZK_SHORT_KERNEL_PRIO_LEVEL);      // (ntt.hip: a short kernel beside the provers asks for the SIMD)
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
  if (sub == 0) q_store12(out + part_loc(lay, i) * 12, acc);
}

// H = (A B - C) / Z on the coset, in place into A, at every LOCATION (the vectors are in their transforms' output order).  Z is
// constant on the coset of a radix-2 domain (compr == 0: one class); on a step domain it depends on the element: class i mod compr
// on the big part, class compr on the small part (domain.hpp).
__global__ void __launch_bounds__(256) k_h_pointwise(uint32_t* __restrict__ A, const uint32_t* __restrict__ B,
                                                      const uint32_t* __restrict__ C, const uint32_t* __restrict__ zinv, uint32_t d,
                                                      PartLayout lay, uint32_t compr) {
  uint32_t loc = blockIdx.x * blockDim.x + threadIdx.x;
  if (loc >= d) return;
  FrD ab = fp_mul(q_load12(A + (size_t)loc * 12), q_load12(B + (size_t)loc * 12));
  FrD c = fp_mul(q_load12(C + (size_t)loc * 12), fp_one<FrParams>());   // < 2r whatever the input bound
  FrD t = fp_sub<FrParams, 2>(ab, c);
  uint32_t cls = 0;
  if (compr) {
    const uint32_t i = part_index(lay, loc);
    cls = i < lay.big ? (i & (compr - 1)) : compr;
  }
  FrD zi;
#pragma unroll
  for (int k = 0; k < 14; k++) zi.l[k] = zinv[(size_t)cls * 14 + k];
  q_store12(A + (size_t)loc * 12, fp_mul(t, zi));
}

// ---- step domains (libfqfft's step_radix2_domain; domain.hpp): the element-wise halves of its transforms ------------------------
// With B = big, S = small, compr = B / S, w of order 2 B: a polynomial a of d = B + S coefficients is evaluated on the big part
// through c[i] = a[i] + a[i + B] (i < S), a[i] otherwise (x^B = 1 there), and on the small part through d[i] = w^i (a[i] - a[i + B])
// or w^i a[i] (x^B = -1 there), folded to e[i] = sum_j d[i + j S] (the points' S-th powers agree); c and e then go through plain
// transforms of B and S points.  The inverse: U0 = c and U1 = e come out of the two inverse transforms, d[i] = e[i] - sum_{j >= 1}
// w^(i + j S) c[i + j S] for i < S, a[i] - a[i + B] = w^-i d[i], a[i] + a[i + B] = c[i].
// One COLUMN i < S (the elements i, i + S, ..., i + (compr - 1) S) per group of W lanes (W a power of two <= 256; the lanes of a
// group stride over the rows and meet in LDS): compr = 2 for the wrapping circuit, but a domain 2^20 + 2^3 has 2^17 rows a column.
//   MID  : inverse half, twist by g^i (cosetFFT), forward half:  U0, U1 (coefficient side of the two inverse transforms, natural
//          order) -> c, e (input of the two forward transforms, natural order), in place.
//   FINAL: inverse half, twist by g^-i (icosetFFT): U0, U1 -> the d coefficients, natural order, in place.
struct StepArgs {
  uint32_t* buf[3];        // up to three vectors of d elements (big part first)
  const uint32_t *pw_w, *pw_winv, *pw_g, *pw_ginv, *half;
  uint32_t big, small, log_small, compr, W, log_w;
};
__device__ __forceinline__ FrD fr_ld14(const uint32_t* p) {
  FrD v;
#pragma unroll
  for (int k = 0; k < 14; k++) v.l[k] = p[k];
  return v;
}
__device__ __forceinline__ FrD fr_red(const FrD& x) { return fp_mul(x, fp_one<FrParams>()); }      // any lazy bound -> < 2r
// sum over the W lanes of a group (W <= 256 consecutive lanes of the block); the result is valid in every lane
__device__ __forceinline__ FrD group_sum(FrD v, uint32_t* lds /* [14][256] */, uint32_t W) {
  if (W == 1) return v;
  const uint32_t tid = threadIdx.x, g0 = tid & ~(W - 1);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 14; k++) lds[k * 256 + tid] = v.l[k];
  __syncthreads();
  FrD acc = fp_zero<FrParams>();
  for (uint32_t u = 0; u < W; u++) {
    FrD t;
#pragma unroll
    for (int k = 0; k < 14; k++) t.l[k] = lds[k * 256 + g0 + u];
    acc = fp_add(acc, t);                                       // each term < 2r: 256 of them stay far inside the lazy bound (2^10 r)
  }
  return fr_red(acc);
}
template <bool FINAL>
__global__ void __launch_bounds__(256) k_step_half(StepArgs a) {
  __shared__ uint32_t lds[14 * 256];
  uint32_t* __restrict__ buf = a.buf[blockIdx.y];
  const uint32_t B = a.big, S = a.small, compr = a.compr, W = a.W;
  const uint32_t col = (blockIdx.x * 256u + threadIdx.x) >> a.log_w, lane = threadIdx.x & (W - 1);
  const bool live = col < S;                                     // (whole groups leave the arithmetic together; every lane reaches the barriers)
  const uint32_t i = live ? col : 0;
  const FrD half = fr_ld14(a.half);
  // ---- inverse half: column sum of w^(i + j S) U0[i + j S] over the rows j >= 1
  FrD part = fp_zero<FrParams>();
  uint32_t cnt = 0;
  for (uint32_t j = lane ? lane : W; j < compr; j += W) {        // (lane 0 starts at row W: row 0 is not part of this sum)
    const uint32_t k = i + j * S;
    part = fp_add(part, fp_mul(q_load12(buf + (size_t)k * 12), q_load12(a.pw_w + (size_t)k * 12)));
    if ((++cnt & 63u) == 0) part = fr_red(part);
  }
  part = fr_red(part);
  const FrD colsum = group_sum(part, lds, W);
  FrD a_lo = fp_zero<FrParams>(), a_hi = a_lo;
  if (live && lane == 0) {
    const FrD u0 = q_load12(buf + (size_t)i * 12), u1 = q_load12(buf + (size_t)(B + i) * 12);
    const FrD diff = fp_mul(fp_sub<FrParams, 2>(u1, colsum), q_load12(a.pw_winv + (size_t)i * 12));       // a[i] - a[i + B]  (< 2r)
    a_lo = fp_mul(fp_add(u0, diff), half);
    a_hi = fp_mul(fp_sub<FrParams, 2>(u0, diff), half);
  }
  if constexpr (FINAL) {
    // coefficients times g^-i: rows j >= 1 of the column are U0 itself
    for (uint32_t j = lane; j < compr; j += W) {
      const uint32_t k = i + j * S;
      if (!live) break;
      const FrD v = (j == 0) ? a_lo : q_load12(buf + (size_t)k * 12);
      q_store12(buf + (size_t)k * 12, fp_cond_sub_p(fp_mul(v, q_load12(a.pw_ginv + (size_t)k * 12))));
    }
    if (live && lane == 0) q_store12(buf + (size_t)(B + i) * 12, fp_cond_sub_p(fp_mul(a_hi, q_load12(a.pw_ginv + (size_t)(B + i) * 12))));
    return;
  }
  // ---- twist by g^k (cosetFFT), forward half: c[k] and d[k] for every row of the column, e[i] = sum of the d's
  FrD dpart = fp_zero<FrParams>();
  cnt = 0;
  for (uint32_t j = lane; j < compr; j += W) {
    const uint32_t k = i + j * S;
    FrD lo = (j == 0) ? a_lo : q_load12(buf + (size_t)k * 12);
    lo = fp_mul(lo, q_load12(a.pw_g + (size_t)k * 12));                                                       // a[k] g^k
    FrD c = lo, dd = lo;
    if (j == 0) {
      const FrD hi = fp_mul(a_hi, q_load12(a.pw_g + (size_t)(B + i) * 12));                                   // a[B + i] g^(B + i)
      c = fp_add(lo, hi);
      dd = fp_sub<FrParams, 2>(lo, hi);
    }
    if (live) q_store12(buf + (size_t)k * 12, fp_cond_sub_p(fr_red(c)));
    dpart = fp_add(dpart, fp_mul(dd, q_load12(a.pw_w + (size_t)k * 12)));
    if ((++cnt & 63u) == 0) dpart = fr_red(dpart);
  }
  dpart = fr_red(dpart);
  const FrD e = group_sum(dpart, lds, W);
  if (live && lane == 0) q_store12(buf + (size_t)(B + i) * 12, fp_cond_sub_p(e));
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

static int r1cs_upload_impl(const zkhip_r1cs_desc* d, size_t domain_size, R1csDev* r, char* err, size_t errlen);
int r1cs_upload(const zkhip_r1cs_desc* d, size_t domain_size, R1csDev** out, char* err, size_t errlen) {
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
  int rc = r1cs_upload_impl(d, domain_size, r, err, errlen);
  if (rc != ZKHIP_OK) { r1cs_free(r); return rc; }        // frees what was uploaded before the failure
  *out = r;
  return ZKHIP_OK;
}

static void free_domain_buffers(R1csDev* r) {
  void** ptrs[] = {(void**)&r->bufA, (void**)&r->bufB, (void**)&r->bufC, (void**)&r->zinv, (void**)&r->pw_w, (void**)&r->pw_winv,
                   (void**)&r->pw_g, (void**)&r->pw_ginv, (void**)&r->half};
  for (void** p : ptrs) if (*p) { (void)hipFree(*p); *p = nullptr; }
}

// Everything of a constraint system that depends on its evaluation domain: the three work vectors, 1 / Z on the coset and, for a
// step domain, the powers of its element-wise halves.  The matrices do not.  A handle can be moved to another domain (the proving
// key decides: zkhip_groth16_prove with a key of another valid domain size); the caller makes sure nothing is in flight on it.
static int r1cs_build_domain(R1csDev* r, size_t want, char* err, size_t errlen);
int r1cs_set_domain(R1csDev* r, size_t domain_size, char* err, size_t errlen) {
  const size_t points = r->n_constraints + r->n_primary + 1;
  const size_t want = host::resolve_domain(points, domain_size);
  if (!want) {
    snprintf(err, errlen, "evaluation domain of %zu points: not a power of two or 2^k + 2^r, or smaller than the system's %zu points", domain_size, points);
    return ZKHIP_ERR_ARG;
  }
  if (want > ((size_t)1 << 22)) { snprintf(err, errlen, "r1cs: domain larger than 2^22"); return ZKHIP_ERR_ARG; }
  if (want == r->d && r->bufA) return ZKHIP_OK;           // (r->d != 0 only after EVERY buffer of that domain was built: below)
  free_domain_buffers(r);
  const int rc = r1cs_build_domain(r, want, err, errlen);
  if (rc != ZKHIP_OK) {
    // A failure half-way (an allocation refused while a 32-instance pipeline holds the HBM) must not leave the handle NAMING the new
    // domain over freed / null buffers - the next prove through it would launch the QAP kernels on null pointers.  The
    // handle is left on NO domain: r->d = 0 matches no key, so follow_key_domain and this function rebuild on the next call.
    free_domain_buffers(r);
    r->d = r->big = r->small = 0;
    r->log_d = r->log_big = r->log_small = 0;
  }
  return rc;
}
static int r1cs_build_domain(R1csDev* r, size_t want, char* err, size_t errlen) {
  using host::HFr;
  const host::EvalDomain dom(want);
  r->d = dom.m; r->big = dom.big; r->small = dom.small;
  r->log_d = host::ceil_log2(dom.m); r->log_big = dom.log_big; r->log_small = dom.log_small;
  const size_t dd = dom.m;
  int rc;
  Q_HIP(hipMalloc(&r->bufA, dd * 48)); Q_HIP(hipMalloc(&r->bufB, dd * 48)); Q_HIP(hipMalloc(&r->bufC, dd * 48));
  // 1 / Z on the coset g x, per class (see qap.h)
  const HFr g = HFr::from_limbs(FrParams::GEN64);
  const size_t classes = dom.is_step() ? dom.compr() + 1 : 1;
  std::vector<uint32_t> zi(classes * 14);
  {
    std::vector<HFr> zs(classes), pref(classes);
    HFr acc = HFr::one(), x = g;                       // big part: x = g big_omega^k, k < compr (radix-2: any point of the coset)
    for (size_t k = 0; k < classes; k++) {
      if (dom.is_step() && k == dom.compr()) x = g * dom.omega;                   // the small part
      zs[k] = dom.vanishing(x);
      pref[k] = acc; acc = acc * zs[k];
      x = x * dom.big_omega;
    }
    HFr inv_all = acc.inv();
    for (size_t k = classes; k-- > 0;) {
      const HFr zinv = inv_all * pref[k];
      inv_all = inv_all * zs[k];
      uint64_t l[6];
      zinv.to_limbs(l);
      FrD zd = fp_cond_sub_p(fp_from_abi<FrParams>(l));
      memcpy(&zi[k * 14], zd.l, 14 * 4);
    }
  }
  Q_HIP(hipMalloc(&r->zinv, zi.size() * 4));
  Q_HIP(hipMemcpy(r->zinv, zi.data(), zi.size() * 4, hipMemcpyHostToDevice));
  if (dom.is_step()) {
    // powers for the element-wise halves of the step domain's transforms, packed device form
    auto upload_powers = [&](const HFr& base, size_t n, uint32_t** out) -> int {
      std::vector<uint64_t> h(n * 6);
      HFr p = HFr::one();
      for (size_t i = 0; i < n; i++) { p.to_limbs(&h[i * 6]); p = p * base; }
      uint64_t* tmp = nullptr;
      Q_HIP(hipMalloc(out, n * 48));
      Q_HIP(hipMalloc(&tmp, n * 48));
      hipError_t e = hipMemcpy(tmp, h.data(), n * 48, hipMemcpyHostToDevice);
      if (e == hipSuccess) { fr_abi_to_dev(tmp, *out, n, 0); e = hipDeviceSynchronize(); }
      (void)hipFree(tmp);
      Q_HIP(e);
      return ZKHIP_OK;
    };
    if ((rc = upload_powers(dom.omega, dom.big, &r->pw_w)) != ZKHIP_OK || (rc = upload_powers(dom.omega.inv(), dom.small, &r->pw_winv)) != ZKHIP_OK ||
        (rc = upload_powers(g, dom.m, &r->pw_g)) != ZKHIP_OK || (rc = upload_powers(g.inv(), dom.m, &r->pw_ginv)) != ZKHIP_OK)
      return rc;
    uint64_t l[6];
    HFr::from_u64(2).inv().to_limbs(l);
    FrD hd = fp_cond_sub_p(fp_from_abi<FrParams>(l));
    Q_HIP(hipMalloc(&r->half, 14 * 4));
    Q_HIP(hipMemcpy(r->half, hd.l, 14 * 4, hipMemcpyHostToDevice));
  }
  return ZKHIP_OK;
}

static int r1cs_upload_impl(const zkhip_r1cs_desc* d, size_t domain_size, R1csDev* r, char* err, size_t errlen) {
  r->n_constraints = d->n_constraints; r->n_vars = d->n_vars; r->n_primary = d->n_primary;
  // lanes per row of the sparse products: sixteen for a proof alone (a wrapping circuit has rows of 47 and 253 terms and fewer rows than the
  // chip has lanes), ONE where the rows are short (a system of two-term rows, 2^20 of them: sixteen lanes a row made 1.7 ms of folding)
  {
    const size_t nnz = (size_t)d->a_row_ptr[d->n_constraints] + d->b_row_ptr[d->n_constraints] + d->c_row_ptr[d->n_constraints];
    r->spmv_log_lanes_alone = (d->n_constraints && nnz <= 9 * d->n_constraints) ? 0 : 4;
    r->spmv_log_lanes = r->spmv_log_lanes_alone;
  }
  // the evaluation domain: the reference's forced power of two unless the caller (a proving key) names another one (domain.hpp)
  int rc;
  if ((rc = r1cs_set_domain(r, domain_size, err, errlen)) != ZKHIP_OK) return rc;
  if ((rc = csr_upload(d->a_row_ptr, d->a_col, d->a_val, d->n_constraints, &r->A, err, errlen)) != ZKHIP_OK) return rc;
  if ((rc = csr_upload(d->b_row_ptr, d->b_col, d->b_val, d->n_constraints, &r->B, err, errlen)) != ZKHIP_OK) return rc;
  if ((rc = csr_upload(d->c_row_ptr, d->c_col, d->c_val, d->n_constraints, &r->C, err, errlen)) != ZKHIP_OK) return rc;
  Q_HIP(hipMalloc(&r->tmp, 256)); Q_HIP(hipMalloc(&r->z, d->n_vars * 48));     // tmp: the satisfiability flag
  return ZKHIP_OK;
}

void r1cs_free(R1csDev* r) {
  if (!r) return;
  void* ptrs[] = {r->A.row_ptr, r->A.col, r->A.val, r->B.row_ptr, r->B.col, r->B.val, r->C.row_ptr, r->C.col, r->C.val,
                  r->bufA, r->bufB, r->bufC, r->tmp, r->z, r->zinv, r->pw_w, r->pw_winv, r->pw_g, r->pw_ginv, r->half};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  delete r;
}

// the order the transforms of this domain read their input in when it is TRANSPOSED (per part; see ntt.h)
static PartLayout part_layout(const R1csDev* r) {
  PartLayout L;
  L.big = (uint32_t)r->big;
  L.log_k0 = ntt_layout_logk(r->log_big); L.log_n20 = L.log_k0 ? r->log_big - L.log_k0 : 0;
  L.log_k1 = r->small ? ntt_layout_logk(r->log_small) : 0; L.log_n21 = L.log_k1 ? r->log_small - L.log_k1 : 0;
  return L;
}

// the three products, in the order the first transform wants (transposed for 2^12 rows and more, see ntt.h)
static void spmv3(R1csDev* r, hipStream_t st) {
  uint32_t n = (uint32_t)r->n_constraints, d = (uint32_t)r->d;
  const CsrDev* M[3] = {&r->A, &r->B, &r->C};
  Spmv3 m;
  for (int k = 0; k < 3; k++) { m.row_ptr[k] = M[k]->row_ptr; m.col[k] = M[k]->col; m.val[k] = M[k]->val; m.extra[k] = k == 0 ? (uint32_t)r->n_primary + 1 : 0u; }
  m.out[0] = r->bufA; m.out[1] = r->bufB; m.out[2] = r->bufC;
  const PartLayout lay = part_layout(r);
  if (r->spmv_log_lanes == 0)
    hipLaunchKernelGGL(k_spmv<0>, dim3((unsigned)(((size_t)d + 255) / 256), 3), dim3(256), 0, st, m, r->z, n, d, lay);
  else if (r->spmv_log_lanes == 2)
    hipLaunchKernelGGL(k_spmv<2>, dim3((unsigned)(((size_t)d * 4 + 255) / 256), 3), dim3(256), 0, st, m, r->z, n, d, lay);
  else
    hipLaunchKernelGGL(k_spmv<4>, dim3((unsigned)(((size_t)d * 16 + 255) / 256), 3), dim3(256), 0, st, m, r->z, n, d, lay);
}

// the plain transforms of a step domain's two parts: nbuf vectors, big part at buf, small part at buf + big
static int step_ntts(R1csDev* r, uint32_t* const* bufs, int nbuf, int inverse, int in_transposed, hipStream_t st, char* err, size_t errlen) {
  uint32_t* small_bufs[3];
  for (int k = 0; k < nbuf; k++) small_bufs[k] = bufs[k] + r->big * 12;
  int rc = ntt_dev_packed_batch(bufs, nbuf, r->log_big, inverse, 0, in_transposed, st, err, errlen);
  if (rc != ZKHIP_OK) return rc;
  return ntt_dev_packed_batch(small_bufs, nbuf, r->log_small, inverse, 0, in_transposed, st, err, errlen);
}

int qap_h_dev(R1csDev* r, const uint64_t* d_z_abi, hipStream_t st, char* err, size_t errlen, const uint64_t* d_z_app) {
  if (!r->d || !r->bufA || !r->bufB || !r->bufC || !r->zinv) {           // (a domain switch that failed: r1cs_set_domain)
    snprintf(err, errlen, "constraint system handle has no evaluation domain (an earlier domain switch failed)");
    return ZKHIP_ERR_ARG;
  }
  const int lg = r->log_d;
  const uint32_t d = (uint32_t)r->d;
  if (d_z_app) fr_abi_to_dev_merge(d_z_abi, d_z_app, r->z, r->n_vars, st);       // masked assignment | the application's constants
  else fr_abi_to_dev(d_z_abi, r->z, r->n_vars, st);
  spmv3(r, st);
  int rc;
  uint32_t* bufs[3] = {r->bufA, r->bufB, r->bufC};
  const PartLayout lay = part_layout(r);
  if (!r->small) {
    if ((rc = ntt_dev_packed_batch(bufs, 3, lg, 1, 0, 1, st, err, errlen)) != ZKHIP_OK) return rc;   // iFFT of A, B, C: transposed -> natural
    if ((rc = ntt_dev_packed_batch(bufs, 3, lg, 0, 1, 0, st, err, errlen)) != ZKHIP_OK) return rc;   // cosetFFT: natural -> transposed
    hipLaunchKernelGGL(k_h_pointwise, dim3((d + 255) / 256), dim3(256), 0, st, r->bufA, r->bufB, r->bufC, r->zinv, d, lay, 0u);   // (any order)
    if ((rc = ntt_dev_packed(r->bufA, lg, 1, 1, 1, st, err, errlen)) != ZKHIP_OK) return rc;       // icosetFFT: transposed -> natural
    Q_HIP(hipGetLastError());
    return ZKHIP_OK;
  }
  // step domain: every transform = the plain transforms of its two parts around an element-wise half (k_step_half)
  StepArgs sa;
  sa.buf[0] = r->bufA; sa.buf[1] = r->bufB; sa.buf[2] = r->bufC;
  sa.pw_w = r->pw_w; sa.pw_winv = r->pw_winv; sa.pw_g = r->pw_g; sa.pw_ginv = r->pw_ginv; sa.half = r->half;
  sa.big = (uint32_t)r->big; sa.small = (uint32_t)r->small; sa.log_small = (uint32_t)r->log_small;
  sa.compr = (uint32_t)(r->big / r->small);
  sa.W = sa.compr < 256u ? sa.compr : 256u;
  sa.log_w = 0; while ((1u << sa.log_w) < sa.W) sa.log_w++;
  const unsigned sblocks = (unsigned)(((size_t)sa.small * sa.W + 255) / 256);
  if ((rc = step_ntts(r, bufs, 3, 1, 1, st, err, errlen)) != ZKHIP_OK) return rc;                     // iFFT, transform halves: transposed -> natural
  hipLaunchKernelGGL(k_step_half<false>, dim3(sblocks, 3), dim3(256), 0, st, sa);                      // ... its element-wise half, x g^i, the cosetFFT's element-wise half
  if ((rc = step_ntts(r, bufs, 3, 0, 0, st, err, errlen)) != ZKHIP_OK) return rc;                     // cosetFFT, transform halves: natural -> transposed
  hipLaunchKernelGGL(k_h_pointwise, dim3((d + 255) / 256), dim3(256), 0, st, r->bufA, r->bufB, r->bufC, r->zinv, d, lay, sa.compr);
  if ((rc = step_ntts(r, bufs, 1, 1, 1, st, err, errlen)) != ZKHIP_OK) return rc;                     // icosetFFT of H: transposed -> natural
  hipLaunchKernelGGL(k_step_half<true>, dim3(sblocks, 1), dim3(256), 0, st, sa);                       // ... element-wise half, x g^-i
  Q_HIP(hipGetLastError());
  return ZKHIP_OK;
}

int r1cs_is_satisfied_dev(R1csDev* r, const uint64_t* d_z_abi, hipStream_t st, int* ok, char* err, size_t errlen) {
  if (!r->d || !r->bufA || !r->bufB || !r->bufC) {
    snprintf(err, errlen, "constraint system handle has no evaluation domain (an earlier domain switch failed)");
    return ZKHIP_ERR_ARG;
  }
  fr_abi_to_dev(d_z_abi, r->z, r->n_vars, st);
  spmv3(r, st);
  uint32_t* flag = r->tmp;   // first word of the scratch buffer
  Q_HIP(hipMemsetAsync(flag, 0, 4, st));
  // every place of the three vectors (whatever their order): the rows past the constraints have B = C = 0
  uint32_t n = (uint32_t)r->d;
  hipLaunchKernelGGL(k_check_sat, dim3((n + 255) / 256), dim3(256), 0, st, r->bufA, r->bufB, r->bufC, n, flag);
  uint32_t h = 0;
  Q_HIP(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, st));
  Q_HIP(hipStreamSynchronize(st));
  *ok = (h == 0);
  return ZKHIP_OK;
}

}  // namespace zkhip
