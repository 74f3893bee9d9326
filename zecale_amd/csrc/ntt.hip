// Radix-2 number-theoretic transforms over Fr of BW6-761 on gfx950: FFT / iFFT / cosetFFT /
// icosetFFT on the domain <omega>, omega = 15^((r-1)/2^log_d) - the seven size-d transforms of
// r1cs_to_qap_witness_map that the reference reaches through wsnarkT::generate_proof
// (libzecale/circuits/aggregator_circuit.tcc:168; libfqfft basic_radix2_domain).
//
// Structure: "four-step" decomposition d = K * N2 with both factors <= 2^11 so that every
// sub-transform runs entirely in LDS (limb-major [14][K] image, conflict-free: 2^11 x 56 B =
// 112 KiB of the CU's 160 KiB):
//   pass A   N2 column transforms of size K (stride N2), root omega^N2, then the inter-step
//            twiddle omega^(c*k1); the forward coset shift g^i is folded into the load
//   pass B   K row transforms of size N2 (contiguous), root omega^K; the result X[k1 + K*k2] is
//            written in natural order; 1/d and the inverse coset shift are folded into the store
// Two HBM round trips per transform (algorithmic bytes 2 * d * 48 B; here 4 * d * 48 B).
// Elements travel between passes as 12 packed words (48 B, device Montgomery form, value < 2^384:
// lazily reduced, no canonicalisation between passes).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include <vector>

#include "fp29.cuh"
#include "host_field.hpp"
#include "ntt.h"

namespace zkhip {

typedef Fp<FrParams> FrD;

// ---- element I/O -------------------------------------------------------------------------
__device__ __forceinline__ FrD fr_load12(const uint32_t* p) {
  uint32_t w[12];
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1], c = q[2];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
  return fp_unpack32<FrParams>(w);
}
__device__ __forceinline__ void fr_store12(uint32_t* p, const FrD& v) {   // v < 2^384, limbs normalised
  uint32_t w[12];
  fp_pack32<FrParams>(v, w);
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]);
  q[1] = make_uint4(w[4], w[5], w[6], w[7]);
  q[2] = make_uint4(w[8], w[9], w[10], w[11]);
}
__device__ __forceinline__ FrD fr_load14(const uint32_t* p) {
  FrD v;
#pragma unroll
  for (int i = 0; i < 14; i++) v.l[i] = p[i];
  return v;
}

// ABI (6 x u64, Montgomery 2^384, canonical) <-> packed device form
__global__ void __launch_bounds__(256) k_fr_abi_to_dev(const uint64_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x[6];
#pragma unroll
  for (int k = 0; k < 6; k++) x[k] = in[i * 6 + k];
  fr_store12(out + i * 12, fp_from_abi<FrParams>(x));
}
__global__ void __launch_bounds__(256) k_fr_dev_to_abi(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x[6];
  fp_to_abi<FrParams>(fr_load12(in + i * 12), x);
#pragma unroll
  for (int k = 0; k < 6; k++) out[i * 6 + k] = x[k];
}

// factor^e from a two-level table: lo[e & 1023] * hi[e >> 10]  (14-limb entries, device form)
__device__ __forceinline__ FrD pow_tab(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi, uint32_t e) {
  return fp_mul(fr_load14(lo + (size_t)(e & 1023u) * 14), fr_load14(hi + (size_t)(e >> 10) * 14));
}

struct TileArgs {
  const uint32_t* src;      // packed elements
  uint32_t* dst;
  uint32_t in_b, in_j;      // element j of tile b is src[b*in_b + j*in_j]
  uint32_t out_b, out_j;    // ... and goes to dst[b*out_b + j*out_j]
  const uint32_t* tw;       // K/2 twiddles of the size-K transform (14 limbs each)
  const uint32_t* pre_lo;   // optional: multiply input element with global index i by pre^i
  const uint32_t* pre_hi;
  const uint32_t* mid_lo;   // optional: multiply output j of tile b by mid^(b*j)
  const uint32_t* mid_hi;
  const uint32_t* post_lo;  // optional: multiply output with global index o by post^o (constant folded into lo)
  const uint32_t* post_hi;
  const uint32_t* post_const;   // optional: multiply every output by a constant (14 limbs)
};

constexpr int tile_threads(int logk) { return logk <= 6 ? 64 : (1 << (logk - 1)); }

template <int LOGK>
__global__ void __launch_bounds__(tile_threads(LOGK)) k_ntt_tile(TileArgs a) {
  constexpr int K = 1 << LOGK;
  constexpr int NT = (K / 2) < 1 ? 1 : (K / 2);
  __shared__ uint32_t lds[14 * K];
  const uint32_t tid = threadIdx.x;
  const uint32_t b = blockIdx.x;
  if (tid < (uint32_t)NT) {
    // load two elements, (optionally) pre-multiply, store bit-reversed into LDS
#pragma unroll
    for (int h = 0; h < 2; h++) {
      uint32_t j = tid + h * NT;
      if (j >= (uint32_t)K) break;
      uint32_t gi = b * a.in_b + j * a.in_j;
      FrD v = fr_load12(a.src + (size_t)gi * 12);
      if (a.pre_lo) v = fp_mul(v, pow_tab(a.pre_lo, a.pre_hi, gi));
      uint32_t rj = LOGK ? (__brev(j) >> ((32 - LOGK) & 31)) : 0u;     // (LOGK = 0: a transform of size 1)
#pragma unroll
      for (int i = 0; i < 14; i++) lds[i * K + rj] = v.l[i];
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int s = 1; s <= LOGK; s++) {
    if (tid < (uint32_t)NT) {
      const uint32_t half = 1u << (s - 1);
      const uint32_t jj = tid & (half - 1);
      const uint32_t p0 = ((tid >> (s - 1)) << s) + jj, p1 = p0 + half;
      FrD u, v;
#pragma unroll
      for (int i = 0; i < 14; i++) { u.l[i] = lds[i * K + p0]; v.l[i] = lds[i * K + p1]; }
      FrD t = (s == 1) ? v : fp_mul(v, fr_load14(a.tw + (size_t)(jj << (LOGK - s)) * 14));   // first stage: w = 1
      FrD x = fp_add(u, t);
      FrD y = fp_sub<FrParams, 2>(u, t);
      if (s == 1) y = fp_sub<FrParams, 16>(u, t);   // t = v may be lazily bounded (< 16 r) here
#pragma unroll
      for (int i = 0; i < 14; i++) { lds[i * K + p0] = x.l[i]; lds[i * K + p1] = y.l[i]; }
    }
    __syncthreads();
  }
  if (tid < (uint32_t)NT) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      uint32_t j = tid + h * NT;
      if (j >= (uint32_t)K) break;
      FrD v;
#pragma unroll
      for (int i = 0; i < 14; i++) v.l[i] = lds[i * K + j];
      uint32_t go = b * a.out_b + j * a.out_j;
      if (a.mid_lo) v = fp_mul(v, pow_tab(a.mid_lo, a.mid_hi, b * j));
      if (a.post_lo) v = fp_mul(v, pow_tab(a.post_lo, a.post_hi, go));
      if (a.post_const) v = fp_mul(v, fr_load14(a.post_const));
      fr_store12(a.dst + (size_t)go * 12, v);
    }
  }
}

// ------------------------------------------------------------------------------------------
// host: tables and plans
// ------------------------------------------------------------------------------------------
using host::HFr;

static void push_dev_limbs(std::vector<uint32_t>& out, const HFr& x) {
  uint64_t l[6];
  x.to_limbs(l);
  FrD d = fp_cond_sub_p(fp_from_abi<FrParams>(l));   // host-compiled fp29 arithmetic
  for (int i = 0; i < 14; i++) out.push_back(d.l[i]);
}

static HFr hfr_pow_u64(const HFr& b, uint64_t e) {
  uint64_t l[1] = {e};
  return b.pow_limbs(l, 1);
}

struct PowTable { uint32_t *lo = nullptr, *hi = nullptr; };

struct NttTables {
  int log_d = 0, log_k = 0, log_n2 = 0;
  uint32_t *twA = nullptr, *twB = nullptr;   // size-K and size-N2 twiddles for this direction
  PowTable mid;                              // omega^(+-1) powers (inter-step twiddle)
  PowTable coset;                            // forward: g^i ; inverse: g^-i * d^-1
  uint32_t* inv_d = nullptr;                 // inverse only: d^-1
};

static hipError_t upload(const std::vector<uint32_t>& v, uint32_t** d) {
  hipError_t e = hipMalloc(d, v.size() * 4);
  if (e != hipSuccess) return e;
  return hipMemcpy(*d, v.data(), v.size() * 4, hipMemcpyHostToDevice);
}

static hipError_t make_pow_table(const HFr& base, const HFr& constant, uint32_t max_e, PowTable* t) {
  std::vector<uint32_t> lo, hi;
  HFr acc = constant;
  for (int j = 0; j < 1024; j++) { push_dev_limbs(lo, acc); acc = acc * base; }
  HFr step = hfr_pow_u64(base, 1024), h = HFr::one();
  for (uint32_t j = 0; j <= (max_e >> 10); j++) { push_dev_limbs(hi, h); h = h * step; }
  hipError_t e = upload(lo, &t->lo);
  if (e != hipSuccess) return e;
  return upload(hi, &t->hi);
}

static hipError_t make_twiddles(const HFr& root /* primitive 2^log_k-th root */, int log_k, uint32_t** d) {
  std::vector<uint32_t> tw;
  HFr acc = HFr::one();
  size_t half = log_k == 0 ? 1 : ((size_t)1 << (log_k - 1));
  for (size_t j = 0; j < half; j++) { push_dev_limbs(tw, acc); acc = acc * root; }
  return upload(tw, d);
}

static std::map<int, NttTables>& table_cache() {
  static std::map<int, NttTables> c;
  return c;
}

static int get_tables(int log_d, int inverse, NttTables** out, char* err, size_t errlen) {
  int dev = 0;
  (void)hipGetDevice(&dev);                    // tables live in the memory of the calling thread's current device
  int key = (dev * 64 + log_d) * 2 + (inverse ? 1 : 0);
  static std::mutex mu;                        // prover instances call in from several host threads
  std::lock_guard<std::mutex> lk(mu);
  auto& c = table_cache();
  auto it = c.find(key);
  if (it != c.end()) { *out = &it->second; return ZKHIP_OK; }
  NttTables t;
  t.log_d = log_d;
  t.log_k = (log_d + 1) / 2;
  if (log_d <= 11) t.log_k = log_d;
  t.log_n2 = log_d - t.log_k;
  // omega = g^((r-1)/2^log_d) = (2^46-th root)^(2^(46-log_d))
  HFr omega = HFr::from_limbs(FrParams::ROOT_2_46_64);
  for (int i = 0; i < FrParams::TWO_ADICITY - log_d; i++) omega = omega.sqr();
  if (inverse) omega = omega.inv();
  HFr g = HFr::from_limbs(FrParams::GEN64);
  const uint32_t d = 1u << log_d;
  hipError_t e = hipSuccess;
  do {
    // root of the size-K column transforms: omega^N2 ; of the size-N2 row transforms: omega^K
    if ((e = make_twiddles(hfr_pow_u64(omega, (uint64_t)1 << t.log_n2), t.log_k, &t.twA)) != hipSuccess) break;
    if (t.log_n2 > 0) {
      if ((e = make_twiddles(hfr_pow_u64(omega, (uint64_t)1 << t.log_k), t.log_n2, &t.twB)) != hipSuccess) break;
      if ((e = make_pow_table(omega, HFr::one(), d, &t.mid)) != hipSuccess) break;
    }
    if (!inverse) {
      if ((e = make_pow_table(g, HFr::one(), d, &t.coset)) != hipSuccess) break;
    } else {
      HFr dinv = HFr::from_u64(d).inv();
      if ((e = make_pow_table(g.inv(), dinv, d, &t.coset)) != hipSuccess) break;
      std::vector<uint32_t> v;
      push_dev_limbs(v, dinv);
      if ((e = upload(v, &t.inv_d)) != hipSuccess) break;
    }
  } while (0);
  if (e != hipSuccess) { snprintf(err, errlen, "ntt tables: %s", hipGetErrorString(e)); return ZKHIP_ERR_HIP; }
  c[key] = t;
  *out = &c[key];
  return ZKHIP_OK;
}

template <int LOGK>
static void launch_tile(const TileArgs& a, uint32_t tiles, hipStream_t st) {
  hipLaunchKernelGGL(k_ntt_tile<LOGK>, dim3(tiles), dim3(tile_threads(LOGK)), 0, st, a);
}
static void launch_tile_dyn(int log_k, const TileArgs& a, uint32_t tiles, hipStream_t st) {
  switch (log_k) {
    case 0: launch_tile<0>(a, tiles, st); break;
    case 1: launch_tile<1>(a, tiles, st); break;
    case 2: launch_tile<2>(a, tiles, st); break;
    case 3: launch_tile<3>(a, tiles, st); break;
    case 4: launch_tile<4>(a, tiles, st); break;
    case 5: launch_tile<5>(a, tiles, st); break;
    case 6: launch_tile<6>(a, tiles, st); break;
    case 7: launch_tile<7>(a, tiles, st); break;
    case 8: launch_tile<8>(a, tiles, st); break;
    case 9: launch_tile<9>(a, tiles, st); break;
    case 10: launch_tile<10>(a, tiles, st); break;
    default: launch_tile<11>(a, tiles, st); break;
  }
}

// Transform `d_data` (packed device form, 2^log_d elements) in place, using `d_tmp` (same size).
int ntt_dev_packed(uint32_t* d_data, uint32_t* d_tmp, int log_d, int inverse, int coset, hipStream_t st, char* err, size_t errlen) {
  if (log_d < 0 || log_d > 22) { snprintf(err, errlen, "ntt: log_d must be in [0, 22]"); return ZKHIP_ERR_ARG; }
  NttTables* t;
  int rc = get_tables(log_d, inverse, &t, err, errlen);
  if (rc != ZKHIP_OK) return rc;
  const uint32_t K = 1u << t->log_k, N2 = 1u << t->log_n2;
  TileArgs a;
  memset(&a, 0, sizeof a);
  if (t->log_n2 == 0) {
    // single tile: whole transform in LDS
    a.src = d_data; a.dst = d_data; a.in_b = 0; a.in_j = 1; a.out_b = 0; a.out_j = 1; a.tw = t->twA;
    if (!inverse && coset) { a.pre_lo = t->coset.lo; a.pre_hi = t->coset.hi; }
    if (inverse && coset) { a.post_lo = t->coset.lo; a.post_hi = t->coset.hi; }
    if (inverse && !coset) a.post_const = t->inv_d;
    launch_tile_dyn(t->log_k, a, 1, st);
  } else {
    // pass A: columns c = tile index, elements c + N2*j; in place; inter-step twiddle omega^(c*k1)
    a.src = d_data; a.dst = d_data; a.in_b = 1; a.in_j = N2; a.out_b = 1; a.out_j = N2; a.tw = t->twA;
    a.mid_lo = t->mid.lo; a.mid_hi = t->mid.hi;
    if (!inverse && coset) { a.pre_lo = t->coset.lo; a.pre_hi = t->coset.hi; }
    launch_tile_dyn(t->log_k, a, N2, st);
    // pass B: rows k1 = tile index, elements N2*k1 + c (contiguous); X[k1 + K*k2] -> tmp
    memset(&a, 0, sizeof a);
    a.src = d_data; a.dst = d_tmp; a.in_b = N2; a.in_j = 1; a.out_b = 1; a.out_j = K; a.tw = t->twB;
    if (inverse && coset) { a.post_lo = t->coset.lo; a.post_hi = t->coset.hi; }
    if (inverse && !coset) a.post_const = t->inv_d;
    launch_tile_dyn(t->log_n2, a, K, st);
    hipError_t e = hipMemcpyAsync(d_data, d_tmp, ((size_t)12 * 4) << log_d, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { snprintf(err, errlen, "ntt: %s", hipGetErrorString(e)); return ZKHIP_ERR_HIP; }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(err, errlen, "ntt launch: %s", hipGetErrorString(e)); return ZKHIP_ERR_HIP; }
  return ZKHIP_OK;
}

void fr_abi_to_dev(const uint64_t* d_in, uint32_t* d_out, size_t n, hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_fr_abi_to_dev, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_in, d_out, n);
}
void fr_dev_to_abi(const uint32_t* d_in, uint64_t* d_out, size_t n, hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_fr_dev_to_abi, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_in, d_out, n);
}

// ABI-form device buffer (6 u64 per element), in place
int ntt_dev_abi(uint64_t* d_data, int log_d, int inverse, int coset, char* err, size_t errlen) {
  size_t d = (size_t)1 << log_d;
  uint32_t *p = nullptr, *tmp = nullptr;
  int rc = ZKHIP_OK;
  hipError_t e;
  if ((e = hipMalloc(&p, d * 48)) != hipSuccess || (e = hipMalloc(&tmp, d * 48)) != hipSuccess) {
    snprintf(err, errlen, "ntt: %s", hipGetErrorString(e));
    if (p) (void)hipFree(p);
    return ZKHIP_ERR_HIP;
  }
  fr_abi_to_dev(d_data, p, d, 0);
  rc = ntt_dev_packed(p, tmp, log_d, inverse, coset, 0, err, errlen);
  if (rc == ZKHIP_OK) {
    fr_dev_to_abi(p, d_data, d, 0);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { snprintf(err, errlen, "ntt: %s", hipGetErrorString(e)); rc = ZKHIP_ERR_HIP; }
  }
  (void)hipFree(p);
  (void)hipFree(tmp);
  return rc;
}

}  // namespace zkhip
