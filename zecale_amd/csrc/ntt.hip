// Radix-2 number-theoretic transforms over Fr of BW6-761 on gfx950: FFT / iFFT / cosetFFT /
// icosetFFT on the domain <omega>, omega = 15^((r-1)/2^log_d) - the seven size-d transforms of
// r1cs_to_qap_witness_map that the reference reaches through wsnarkT::generate_proof
// (libzecale/circuits/aggregator_circuit.tcc:168; libfqfft basic_radix2_domain).
//
// Structure: "four-step" decomposition d = K * N2 with both factors <= 2^11, every sub-transform entirely in LDS
// (limb-major image, 14 x 2048 words = 112 KiB of the CU's 160 KiB), BOTH passes in place - no second buffer:
//   a vector v of length d is kept either in natural order or in TRANSPOSED order, v[k1 + K*k2] at k1*N2 + k2.
//   natural -> transposed   pass 1: the N2 column transforms of size K (stride N2), then the inter-step twiddle
//                           omega^(c*k1); pass 2: the K row transforms of size N2 (contiguous), result left in the row
//   transposed -> natural   pass 1: the K row transforms of size N2 (contiguous: element i1*K + i2 lives at i2*N2 + i1),
//                           twiddle omega^(i2*k1'); pass 2: the N2 column transforms of size K; X[k1' + N2*k2'] lands
//                           at k2'*N2 + k1' - natural order
//   r1cs_to_qap_witness_map chains them so that nothing is ever reordered: SpMV writes transposed, iFFT (t->n),
//   cosetFFT (n->t), the pointwise H (order-blind), icosetFFT (t->n).
//   Two HBM round trips per transform (2 * 2 * d * 48 B), the algorithmic minimum of a two-pass transform.
// A workgroup takes C adjacent sub-transforms (C * K = 2048 elements when there are that many): the strided pass then
// moves C * 48 contiguous bytes per row instead of 48, and a thread carries FOUR elements through two butterfly stages
// per LDS round trip (radix 4): half the barriers and half the LDS traffic of one butterfly per thread per stage.
// Multiplications per element of a 2^20 transform: 10 in the butterflies, ONE for everything else.  The inter-step
// twiddle comes from a full table laid out like the data (one 48-byte load, no two-level product), and the scalings ride
// along: 1/d and the column part of a coset shift are folded into that table; the row part of the forward shift g^i is
// folded into the first pass's stage twiddles (a DIT transform of x_j*c^j is the plain one with stage-s twiddles times
// c^(K/2^s)); only the inverse coset shift needs a second table multiplication (K entries, by output index).
// Elements travel as 12 packed words (48 B, device Montgomery form, value < 2^384: lazily reduced).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
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
// the same for an assignment in two parts (per-application constants: zkhip_aggregator_app): `in` holds zeros where `in2` holds the
// application's constants and the other way round, so the whole assignment is their limb-wise OR
__global__ void __launch_bounds__(256) k_fr_abi_to_dev_merge(const uint64_t* __restrict__ in, const uint64_t* __restrict__ in2, uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x[6];
#pragma unroll
  for (int k = 0; k < 6; k++) x[k] = in[i * 6 + k] | in2[i * 6 + k];
  fr_store12(out + i * 12, fp_from_abi<FrParams>(x));
}
// (log_k != 0: `in` is in transposed order, element i at (i mod 2^log_k) * 2^log_n2 + (i >> log_k))
__global__ void __launch_bounds__(256) k_fr_dev_to_abi(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, size_t n, int log_k, int log_n2) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x[6];
  const size_t loc = log_k ? ((i & (((size_t)1 << log_k) - 1)) << log_n2) + (i >> log_k) : i;
  fp_to_abi<FrParams>(fr_load12(in + loc * 12), x);
#pragma unroll
  for (int k = 0; k < 6; k++) out[i * 6 + k] = x[k];
}

// factor^e from a two-level table: lo[e & 1023] * hi[e >> 10]  (14-limb entries, device form)
__device__ __forceinline__ FrD pow_tab(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi, uint32_t e) {
  return fp_mul(fr_load14(lo + (size_t)(e & 1023u) * 14), fr_load14(hi + (size_t)(e >> 10) * 14));
}

constexpr int NTT_MAX_BATCH = 3;
struct PassArgs {
  uint32_t* data[NTT_MAX_BATCH];   // packed elements, transformed in place; blockIdx.y picks the vector (same transform on each)
  uint32_t strided;         // 1: sub-transform q is column q (element j at q + j*S); 0: row q (element j at q*K + j)
  uint32_t S;               // row length of the strided pass
  const uint32_t* tw;       // stage twiddles, heap order: stage s (butterflies of span 2^s) entry jj at 2^(s-1) + jj (14 limbs each)
  uint32_t tw_scaled;       // the stage twiddles carry a coset shift: the first stage multiplies too
  const uint32_t* mid;      // optional: the output at data location loc is multiplied by mid[loc] (12 packed words each)
  const uint32_t* post_k;   // optional: output k of every sub-transform is multiplied by post_k[k] (14 limbs each)
  const uint32_t* post_const;   // optional: every output multiplied by a constant (14 limbs)
};

constexpr int pass_threads(int logk, int logc) { return (logk + logc) <= 8 ? 64 : (1 << (logk + logc - 2)); }

// LDS bank swizzle (round 5; profiles/r05_ntt_counters_before.csv: 79 % of the kernel's LDS-array cycles were bank-conflict cycles).
// The tile is limb-major, lds[limb * E + p]; a radix-4 round of span q = 2^(s-1) reads p0 + t q (t = 0..3) with p0 running over the
// indices whose bits s-1 and s are clear: for q = 1, 4, 16 the 32 lanes of a half-wave then fall on 8, 8 and 16 of the 32 banks
// (ds_read_b32 / ds_write_b32: bank = dword address mod 32, lanes conflict within a 32-lane half), and the bit-reversed placement
// of the load phase is a power-of-two stride too.  Element p lives at p ^ m(p >> 5), m linear over GF(2) with the columns below -
// chosen so that in EVERY phase (load natural / strided, rounds s = 1, 3, 5, 7, 9, the odd last stage, store) the 32 lanes of a
// half-wave hit 32 different banks: the address bits that vary across a half-wave map onto the five bank bits with full rank.
//   bit 5 -> banks {0,2}   bit 6 -> {1,3,4}   bit 7 -> {2}   bit 8 -> {3}   bit 9 -> {3,4}   bit 10 -> {0,4}
__device__ __forceinline__ uint32_t lds_sw(uint32_t p) {
  const uint32_t h = p >> 5;
  return p ^ ((h & 1u) * 5u) ^ (((h >> 1) & 1u) * 26u) ^ (h & 0xcu) ^ (((h >> 4) & 1u) * 24u) ^ (((h >> 5) & 1u) * 17u);
}

// Inputs < 8r (every producer here stores < 2r); values grow by at most 2r per stage (< 2^7 r = 2^384 at the end).
// A workgroup of the QAP map's kernels that becomes resident beside the provers takes the place of one of a CU's two k_accumulate
// workgroups (displacement), whose waves raise their own priority as they go: at the default priority a short
// kernel's waves get the issue slots the accumulation leaves and hold their CU slot several times longer than their work takes.
// They ask for the SIMD instead (-DZK_SHORT_KERNEL_PRIO=0: off).
#ifndef ZK_SHORT_KERNEL_PRIO_LEVEL
#define ZK_SHORT_KERNEL_PRIO_LEVEL 3
#endif
#define ZK_SHORT_KERNEL_PRIO() do { if (ZK_SHORT_KERNEL_PRIO_LEVEL) __builtin_amdgcn_s_setprio(ZK_SHORT_KERNEL_PRIO_LEVEL); } while (0)
template <int LOGK, int LOGC>
__global__ void __launch_bounds__(pass_threads(LOGK, LOGC)) k_ntt_pass(PassArgs a) {
  constexpr uint32_t K = 1u << LOGK, C = 1u << LOGC, E = K * C, T = pass_threads(LOGK, LOGC);
  __shared__ uint32_t lds[14 * E];
  ZK_SHORT_KERNEL_PRIO();
  const uint32_t tid = threadIdx.x;
  const uint32_t q0 = blockIdx.x * C;
  uint32_t* const data = a.data[blockIdx.y];
  constexpr uint32_t EPT = (E + T - 1) / T;
  // ---- load: C*48 contiguous bytes per row in the strided pass, the whole tile contiguous in the other
#pragma unroll
  for (uint32_t h = 0; h < EPT; h++) {
    const uint32_t e = tid + h * T;
    if (E < T && e >= E) break;
    uint32_t cc, j;
    if (a.strided) { cc = e & (C - 1); j = e >> LOGC; } else { j = e & (K - 1); cc = e >> LOGK; }
    const uint32_t q = q0 + cc;
    const size_t loc = a.strided ? (size_t)q + (size_t)j * a.S : ((size_t)q << LOGK) + j;
    const FrD v = fr_load12(data + loc * 12);
    const uint32_t rj = LOGK ? (__brev(j) >> ((32 - LOGK) & 31)) : 0u;     // (LOGK = 0: a transform of size 1)
    const uint32_t pw = lds_sw(cc * K + rj);
#pragma unroll
    for (int i = 0; i < 14; i++) lds[i * E + pw] = v.l[i];
  }
  __syncthreads();
  // ---- two stages per round trip through LDS
  int s = 1;
#pragma unroll 1
  for (; s + 1 <= LOGK; s += 2) {
    for (uint32_t g = tid; g < E / 4; g += T) {
      const uint32_t cc = LOGK >= 2 ? (g >> ((LOGK - 2) & 31)) : 0u, gi = g & (K / 4 - 1);
      const uint32_t qq = 1u << (s - 1);
      const uint32_t jj = gi & (qq - 1), blk = gi >> (s - 1);
      const uint32_t p0 = cc * K + (blk << (s + 1)) + jj;
      const uint32_t a0 = lds_sw(p0), a1 = lds_sw(p0 + qq), a2 = lds_sw(p0 + 2 * qq), a3 = lds_sw(p0 + 3 * qq);
      FrD x0, x1, x2, x3;
#pragma unroll
      for (int i = 0; i < 14; i++) {
        x0.l[i] = lds[i * E + a0]; x1.l[i] = lds[i * E + a1];
        x2.l[i] = lds[i * E + a2]; x3.l[i] = lds[i * E + a3];
      }
      FrD y0, y1, y2, y3, z0, z1, z2, z3;
      if (s == 1 && !a.tw_scaled) {                     // twiddles 1, 1 | 1, omega^(K/4): one multiplication
        y0 = fp_add(x0, x1); y1 = fp_sub<FrParams, 16>(x0, x1);
        y2 = fp_add(x2, x3); y3 = fp_sub<FrParams, 16>(x2, x3);
        FrD u3 = fp_mul(y3, fr_load14(a.tw + (size_t)3 * 14));
        z0 = fp_add(y0, y2); z2 = fp_sub<FrParams, 16>(y0, y2);
        z1 = fp_add(y1, u3); z3 = fp_sub<FrParams, 2>(y1, u3);
      } else {
        const FrD w = fr_load14(a.tw + (size_t)(qq + jj) * 14);
        const FrD wa = fr_load14(a.tw + (size_t)(2 * qq + jj) * 14);
        const FrD wb = fr_load14(a.tw + (size_t)(3 * qq + jj) * 14);
        FrD t1 = fp_mul(x1, w), t3 = fp_mul(x3, w);
        y0 = fp_add(x0, t1); y1 = fp_sub<FrParams, 2>(x0, t1);
        y2 = fp_add(x2, t3); y3 = fp_sub<FrParams, 2>(x2, t3);
        FrD u2 = fp_mul(y2, wa), u3 = fp_mul(y3, wb);
        z0 = fp_add(y0, u2); z2 = fp_sub<FrParams, 2>(y0, u2);
        z1 = fp_add(y1, u3); z3 = fp_sub<FrParams, 2>(y1, u3);
      }
#pragma unroll
      for (int i = 0; i < 14; i++) {
        lds[i * E + a0] = z0.l[i]; lds[i * E + a1] = z1.l[i];
        lds[i * E + a2] = z2.l[i]; lds[i * E + a3] = z3.l[i];
      }
    }
    __syncthreads();
  }
  if (s == LOGK) {                                      // an odd number of stages: the last one alone
    for (uint32_t g = tid; g < E / 2; g += T) {
      const uint32_t cc = LOGK >= 1 ? (g >> ((LOGK - 1) & 31)) : 0u, gi = g & (K / 2 - 1);
      const uint32_t half = 1u << (s - 1);
      const uint32_t jj = gi & (half - 1);
      const uint32_t p0 = lds_sw(cc * K + ((gi >> (s - 1)) << s) + jj), p1 = lds_sw(cc * K + ((gi >> (s - 1)) << s) + jj + half);
      FrD u, v;
#pragma unroll
      for (int i = 0; i < 14; i++) { u.l[i] = lds[i * E + p0]; v.l[i] = lds[i * E + p1]; }
      FrD x, y;
      if (s == 1 && !a.tw_scaled) { x = fp_add(u, v); y = fp_sub<FrParams, 16>(u, v); }
      else {
        FrD t = fp_mul(v, fr_load14(a.tw + (size_t)(half + jj) * 14));
        x = fp_add(u, t); y = fp_sub<FrParams, 2>(u, t);
      }
#pragma unroll
      for (int i = 0; i < 14; i++) { lds[i * E + p0] = x.l[i]; lds[i * E + p1] = y.l[i]; }
    }
    __syncthreads();
  }
  // ---- store (same places)
#pragma unroll
  for (uint32_t h = 0; h < EPT; h++) {
    const uint32_t e = tid + h * T;
    if (E < T && e >= E) break;
    uint32_t cc, k;
    if (a.strided) { cc = e & (C - 1); k = e >> LOGC; } else { k = e & (K - 1); cc = e >> LOGK; }
    const uint32_t q = q0 + cc;
    const size_t loc = a.strided ? (size_t)q + (size_t)k * a.S : ((size_t)q << LOGK) + k;
    FrD v;
    const uint32_t pr = lds_sw(cc * K + k);
#pragma unroll
    for (int i = 0; i < 14; i++) v.l[i] = lds[i * E + pr];
    if (a.mid) v = fp_mul(v, fr_load12(a.mid + loc * 12));
    if (a.post_k) v = fp_mul(v, fr_load14(a.post_k + (size_t)k * 14));
    if (a.post_const) v = fp_mul(v, fr_load14(a.post_const));
    fr_store12(data + loc * 12, v);
  }
}

// out[loc] = mid^(q*k) * ext^(q*ext_q + k*ext_k)   for the element (sub-transform q, output k) that lives at loc:
// loc = q + k*S (strided first pass) or q*K1 + k (contiguous first pass).  ext_lo == nullptr: no second factor.
__global__ void __launch_bounds__(256) k_mid_table(uint32_t* __restrict__ out, uint32_t d, uint32_t strided, uint32_t log_s /* log2 of S or K1 */,
                                                    const uint32_t* __restrict__ mid_lo, const uint32_t* __restrict__ mid_hi,
                                                    const uint32_t* __restrict__ ext_lo, const uint32_t* __restrict__ ext_hi, uint32_t ext_q, uint32_t ext_k) {
  const uint32_t loc = blockIdx.x * blockDim.x + threadIdx.x;
  if (loc >= d) return;
  const uint32_t lo_part = loc & ((1u << log_s) - 1), hi_part = loc >> log_s;
  const uint32_t q = strided ? lo_part : hi_part, k = strided ? hi_part : lo_part;
  FrD v = pow_tab(mid_lo, mid_hi, q * k);
  if (ext_lo) v = fp_mul(v, pow_tab(ext_lo, ext_hi, q * ext_q + k * ext_k));
  fr_store12(out + (size_t)loc * 12, v);
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
  uint32_t *twA = nullptr, *twB = nullptr;   // stage twiddles (heap order) of the size-K and size-N2 transforms of this direction
  uint32_t* twA_coset = nullptr;             // forward: those of size K times (g^N2)^(K/2^s) - the row part of the shift g^i
  PowTable mid;                              // omega^(+-1) powers (the tables below are made from these two)
  PowTable coset;                            // forward: g^i ; inverse: g^-i * d^-1
  uint32_t* inv_d = nullptr;                 // inverse only: d^-1
  uint32_t* mid_full[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [coset][input transposed]: inter-step twiddle with the scalings folded in
  uint32_t* post_k[2] = {nullptr, nullptr};  // inverse coset, [input transposed]: the part of g^-o the table above cannot hold
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

// stage twiddles of a size-2^log_k DIT transform with root `root`, in heap order: entry 2^(s-1) + jj = root^(jj * 2^(log_k-s))
// times shift^(2^(log_k-s))  (shift = 1: the plain transform; entry 0 unused)
static hipError_t make_twiddles(const HFr& root, const HFr& shift, int log_k, uint32_t** d) {
  std::vector<uint32_t> tw;
  push_dev_limbs(tw, HFr::one());
  for (int s = 1; s <= log_k; s++) {
    const HFr step = hfr_pow_u64(root, (uint64_t)1 << (log_k - s)), scale = hfr_pow_u64(shift, (uint64_t)1 << (log_k - s));
    HFr acc = scale;
    for (size_t jj = 0; jj < ((size_t)1 << (s - 1)); jj++) { push_dev_limbs(tw, acc); acc = acc * step; }
  }
  return upload(tw, d);
}

static hipError_t make_powers(const HFr& base, const HFr& constant, size_t n, uint32_t** d) {
  std::vector<uint32_t> v;
  HFr acc = constant;
  for (size_t j = 0; j < n; j++) { push_dev_limbs(v, acc); acc = acc * base; }
  return upload(v, d);
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
    const HFr rootA = hfr_pow_u64(omega, (uint64_t)1 << t.log_n2);
    if ((e = make_twiddles(rootA, HFr::one(), t.log_k, &t.twA)) != hipSuccess) break;
    if (t.log_n2 > 0) {
      if ((e = make_twiddles(hfr_pow_u64(omega, (uint64_t)1 << t.log_k), HFr::one(), t.log_n2, &t.twB)) != hipSuccess) break;
      if ((e = make_pow_table(omega, HFr::one(), d, &t.mid)) != hipSuccess) break;
    }
    const uint64_t K = (uint64_t)1 << t.log_k, N2 = (uint64_t)1 << t.log_n2;
    if (!inverse) {
      if ((e = make_pow_table(g, HFr::one(), d, &t.coset)) != hipSuccess) break;
      if ((e = make_twiddles(rootA, hfr_pow_u64(g, N2), t.log_k, &t.twA_coset)) != hipSuccess) break;
    } else {
      const HFr dinv = HFr::from_u64(d).inv(), h = g.inv();
      if ((e = make_pow_table(h, dinv, d, &t.coset)) != hipSuccess) break;
      std::vector<uint32_t> v;
      push_dev_limbs(v, dinv);
      if ((e = upload(v, &t.inv_d)) != hipSuccess) break;
      if (t.log_n2 == 0) {
        if ((e = make_powers(h, dinv, K, &t.post_k[0])) != hipSuccess) break;                  // one tile: g^-k / d
      } else {
        // output o = k1 + K*k2 (natural input) or k1' + N2*k2' (transposed input): the first part is in the full table
        if ((e = make_powers(hfr_pow_u64(h, K), HFr::one(), N2, &t.post_k[0])) != hipSuccess) break;
        if ((e = make_powers(hfr_pow_u64(h, N2), HFr::one(), K, &t.post_k[1])) != hipSuccess) break;
      }
    }
  } while (0);
  if (e != hipSuccess) { snprintf(err, errlen, "ntt tables: %s", hipGetErrorString(e)); return ZKHIP_ERR_HIP; }
  c[key] = t;
  *out = &c[key];
  return ZKHIP_OK;
}

template <int LOGK, int LOGC>
static void launch_pass(const PassArgs& a, uint32_t groups, hipStream_t st) {
  uint32_t nbuf = 1;
  while (nbuf < NTT_MAX_BATCH && a.data[nbuf]) nbuf++;
  hipLaunchKernelGGL((k_ntt_pass<LOGK, LOGC>), dim3(groups, nbuf), dim3(pass_threads(LOGK, LOGC)), 0, st, a);
}
static int one_each_max() { static const int v = [] { const char* e = getenv("ZKHIP_NTT_ONE_EACH_MAX"); int x = e ? atoi(e) : 22; return x < 0 || x > 22 ? 22 : x; }(); return v; }
// One workgroup per sub-transform at every size (round 5; rounds 2-4: up to 2^18, beyond that 2048 elements per workgroup - adjacent
// sub-transforms, C * 48 contiguous bytes per row of the strided pass).  A 2048-element tile is 112 KiB of LDS: ONE workgroup per CU,
// whose load and store phases and seven barriers leave the CU's SIMDs idle (SQ_WAIT_ANY 35 % of the wave-cycles); two independent
// 1024-element workgroups per CU overlap each other's phases: 2^20, three vectors: 0.260 -> 0.236 ms per transform, 2^21: 0.555 ->
// 0.527 (profiles/r05_ntt_one_each.txt; a 2^11-point sub-transform needs the whole tile either way: 2^22 unchanged).  The strided
// pass then reads 48 contiguous bytes per row instead of 96 - it is not what the pass waits for.  ZKHIP_NTT_ONE_EACH_MAX=18 restores
// the wide tiles.
static void launch_pass_dyn(int log_k, bool one_each, const PassArgs& a, uint32_t n_sub, hipStream_t st) {
  if (one_each) {
    switch (log_k) {
      case 0: launch_pass<0, 0>(a, n_sub, st); break;
      case 1: launch_pass<1, 0>(a, n_sub, st); break;
      case 2: launch_pass<2, 0>(a, n_sub, st); break;
      case 3: launch_pass<3, 0>(a, n_sub, st); break;
      case 4: launch_pass<4, 0>(a, n_sub, st); break;
      case 5: launch_pass<5, 0>(a, n_sub, st); break;
      case 6: launch_pass<6, 0>(a, n_sub, st); break;
      case 7: launch_pass<7, 0>(a, n_sub, st); break;
      case 8: launch_pass<8, 0>(a, n_sub, st); break;
      case 9: launch_pass<9, 0>(a, n_sub, st); break;
      case 10: launch_pass<10, 0>(a, n_sub, st); break;
      default: launch_pass<11, 0>(a, n_sub, st); break;
    }
    return;
  }
  switch (log_k) {
    case 9: launch_pass<9, 2>(a, n_sub >> 2, st); break;
    case 10: launch_pass<10, 1>(a, n_sub >> 1, st); break;
    default: launch_pass<11, 0>(a, n_sub, st); break;
  }
}

int ntt_layout_logk(int log_d) { return log_d <= 11 ? 0 : (log_d + 1) / 2; }

// the inter-step twiddle table of one kind of transform, made on first use (on the caller's stream)
static int get_mid_full(NttTables* t, int inverse, int coset, int in_transposed, hipStream_t st, uint32_t** out, char* err, size_t errlen) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  uint32_t*& m = t->mid_full[coset ? 1 : 0][in_transposed ? 1 : 0];
  if (!m) {
    const uint32_t d = 1u << t->log_d;
    uint32_t* p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)d * 48);
    if (e != hipSuccess) { snprintf(err, errlen, "ntt tables: %s", hipGetErrorString(e)); return ZKHIP_ERR_HIP; }
    // first pass: strided columns q (loc = q + k*N2) for natural input, contiguous rows q (loc = q*N2 + k) for transposed input
    const uint32_t *ext_lo = nullptr, *ext_hi = nullptr;
    uint32_t ext_q = 0, ext_k = 0;
    if (!inverse && coset) { ext_lo = t->coset.lo; ext_hi = t->coset.hi; ext_q = 1; }             // g^q (the rest of g^i is in the stage twiddles)
    if (inverse) { ext_lo = t->coset.lo; ext_hi = t->coset.hi; ext_k = coset ? 1 : 0; }           // g^-k / d, or 1/d alone
    hipLaunchKernelGGL(k_mid_table, dim3((d + 255) / 256), dim3(256), 0, st, p, d, in_transposed ? 0u : 1u, (uint32_t)t->log_n2,
                       t->mid.lo, t->mid.hi, ext_lo, ext_hi, ext_q, ext_k);
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipFree(p); snprintf(err, errlen, "ntt tables: %s", hipGetErrorString(e)); return ZKHIP_ERR_HIP; }
    m = p;
  }
  *out = m;
  return ZKHIP_OK;
}

// Transform `d_data` (packed device form, 2^log_d elements) in place.  Vectors of 2^12 elements and more are in natural order on
// one side and in transposed order on the other (ntt_layout_logk): in_transposed says which side the input is.
int ntt_dev_packed(uint32_t* d_data, int log_d, int inverse, int coset, int in_transposed, hipStream_t st, char* err, size_t errlen) {
  uint32_t* one[1] = {d_data};
  return ntt_dev_packed_batch(one, 1, log_d, inverse, coset, in_transposed, st, err, errlen);
}

// the same transform on up to NTT_MAX_BATCH vectors in the same launches (the A, B, C vectors of the QAP map)
int ntt_dev_packed_batch(uint32_t* const* d_bufs, int nbuf, int log_d, int inverse, int coset, int in_transposed, hipStream_t st, char* err, size_t errlen) {
  if (nbuf < 1 || nbuf > NTT_MAX_BATCH) { snprintf(err, errlen, "ntt: 1 to %d vectors per call", NTT_MAX_BATCH); return ZKHIP_ERR_ARG; }
  if (log_d < 0 || log_d > 22) { snprintf(err, errlen, "ntt: log_d must be in [0, 22]"); return ZKHIP_ERR_ARG; }
  NttTables* t;
  int rc = get_tables(log_d, inverse, &t, err, errlen);
  if (rc != ZKHIP_OK) return rc;
  const uint32_t K = 1u << t->log_k, N2 = 1u << t->log_n2;
  PassArgs a;
  auto set_data = [&](PassArgs& p) { for (int i = 0; i < nbuf; i++) p.data[i] = d_bufs[i]; };
  memset(&a, 0, sizeof a);
  set_data(a);
  if (t->log_n2 == 0) {
    // one workgroup: the whole transform in LDS
    a.strided = 0; a.tw = t->twA;
    if (!inverse && coset) { a.tw = t->twA_coset; a.tw_scaled = 1; }
    if (inverse && coset) a.post_k = t->post_k[0];
    if (inverse && !coset) a.post_const = t->inv_d;
    launch_pass_dyn(t->log_k, true, a, 1, st);
  } else {
    if (!inverse && coset && in_transposed) { snprintf(err, errlen, "ntt: a forward coset transform takes its input in natural order"); return ZKHIP_ERR_ARG; }
    uint32_t* mid = nullptr;
    if ((rc = get_mid_full(t, inverse, coset, in_transposed, st, &mid, err, errlen)) != ZKHIP_OK) return rc;
    a.mid = mid;
    if (!in_transposed) {
      // pass 1: columns c (elements c + N2*j), twiddle omega^(c*k1)
      a.strided = 1; a.S = N2; a.tw = t->twA;
      if (!inverse && coset) { a.tw = t->twA_coset; a.tw_scaled = 1; }
      launch_pass_dyn(t->log_k, log_d <= one_each_max(), a, N2, st);
      // pass 2: rows k1 (contiguous); output k2 is X[k1 + K*k2], left in the row
      memset(&a, 0, sizeof a);
      set_data(a); a.strided = 0; a.tw = t->twB;
      if (inverse && coset) a.post_k = t->post_k[0];
      launch_pass_dyn(t->log_n2, log_d <= one_each_max(), a, K, st);
    } else {
      // pass 1: rows i2 (element i1*K + i2 of the input lives at i2*N2 + i1), twiddle omega^(i2*k1')
      a.strided = 0; a.tw = t->twB;
      launch_pass_dyn(t->log_n2, log_d <= one_each_max(), a, K, st);
      // pass 2: columns k1' (stride N2); output k2' is X[k1' + N2*k2'] at k2'*N2 + k1'
      memset(&a, 0, sizeof a);
      set_data(a); a.strided = 1; a.S = N2; a.tw = t->twA;
      if (inverse && coset) a.post_k = t->post_k[1];
      launch_pass_dyn(t->log_k, log_d <= one_each_max(), a, N2, st);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(err, errlen, "ntt launch: %s", hipGetErrorString(e)); return ZKHIP_ERR_HIP; }
  return ZKHIP_OK;
}

// Time the transform kernels alone (bench.py's ntt_2_20 roofline): `batch` resident vectors (1 .. 3: the QAP map transforms A, B, C
// in the same launches) of 2^log_d packed elements with pseudo-random contents, `reps` transforms after one untimed, HIP events on
// the stream the passes are launched on.  *ms_per_transform = elapsed / (reps * batch): the time one size-d transform costs.
// A transform leaves its output in the other order (natural <-> transposed): successive repetitions alternate the side, as the
// QAP map's chain does; forward coset transforms (natural input only) are put back by an untimed inverse in between.
int ntt_measure(int log_d, int inverse, int coset, int batch, int reps, double* ms_per_transform, char* err, size_t errlen) {
  if (log_d < 1 || log_d > 22 || batch < 1 || batch > NTT_MAX_BATCH || reps < 1 || reps > 1000 || !ms_per_transform) { snprintf(err, errlen, "ntt_measure: bad argument"); return ZKHIP_ERR_ARG; }
  const size_t d = (size_t)1 << log_d;
  uint32_t* bufs[NTT_MAX_BATCH] = {nullptr, nullptr, nullptr};
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = ZKHIP_OK;
  hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  std::vector<uint32_t> h(d * 12);
  uint64_t x = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < h.size(); i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)x; }
  for (size_t i = 0; i < d; i++) h[i * 12 + 11] &= 0x00ffffffu;                  // values below 2^376 < r: valid lazily-reduced elements
  for (int b = 0; b < batch && e == hipSuccess; b++) {
    e = hipMalloc(&bufs[b], d * 48);
    if (e == hipSuccess) e = hipMemcpy(bufs[b], h.data(), d * 48, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) { snprintf(err, errlen, "ntt_measure: %s", hipGetErrorString(e)); rc = ZKHIP_ERR_HIP; }
  float total = 0.f;
  const bool fwd_coset = !inverse && coset;
  int side = 0;                                       // 0: the vectors are in natural order
  for (int r = -1; r < reps && rc == ZKHIP_OK; r++) {
    if (fwd_coset && side) {                          // back to natural order, untimed
      rc = ntt_dev_packed_batch(bufs, batch, log_d, 1, 0, 1, st, err, errlen);
      side = 0;
      if (rc != ZKHIP_OK) break;
    }
    if (r >= 0) (void)hipEventRecord(e0, st);
    rc = ntt_dev_packed_batch(bufs, batch, log_d, inverse, coset, log_d >= 12 ? side : 0, st, err, errlen);
    if (r >= 0) {
      (void)hipEventRecord(e1, st);
      if (hipEventSynchronize(e1) != hipSuccess) { snprintf(err, errlen, "ntt_measure: event"); rc = ZKHIP_ERR_HIP; break; }
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      total += ms;
    } else (void)hipStreamSynchronize(st);
    if (log_d >= 12) side ^= 1;
  }
  if (rc == ZKHIP_OK) *ms_per_transform = (double)total / (double)(reps * batch);
  for (uint32_t* b : bufs) if (b) (void)hipFree(b);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}

void fr_abi_to_dev(const uint64_t* d_in, uint32_t* d_out, size_t n, hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_fr_abi_to_dev, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_in, d_out, n);
}
void fr_abi_to_dev_merge(const uint64_t* d_in, const uint64_t* d_in2, uint32_t* d_out, size_t n, hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_fr_abi_to_dev_merge, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_in, d_in2, d_out, n);
}
void fr_dev_to_abi(const uint32_t* d_in, uint64_t* d_out, size_t n, hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_fr_dev_to_abi, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_in, d_out, n, 0, 0);
}

// ABI-form device buffer (6 u64 per element), in place, natural order on both sides
int ntt_dev_abi(uint64_t* d_data, int log_d, int inverse, int coset, char* err, size_t errlen) {
  size_t d = (size_t)1 << log_d;
  uint32_t* p = nullptr;
  int rc = ZKHIP_OK;
  hipError_t e;
  if ((e = hipMalloc(&p, d * 48)) != hipSuccess) { snprintf(err, errlen, "ntt: %s", hipGetErrorString(e)); return ZKHIP_ERR_HIP; }
  fr_abi_to_dev(d_data, p, d, 0);
  rc = ntt_dev_packed(p, log_d, inverse, coset, 0, 0, err, errlen);
  if (rc == ZKHIP_OK) {
    const int lk = ntt_layout_logk(log_d);                 // the result is transposed: put back in order on the way out
    hipLaunchKernelGGL(k_fr_dev_to_abi, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, 0, p, d_data, d, lk, lk ? log_d - lk : 0);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { snprintf(err, errlen, "ntt: %s", hipGetErrorString(e)); rc = ZKHIP_ERR_HIP; }
  }
  (void)hipFree(p);
  return rc;
}

}  // namespace zkhip
