// Micro-benchmark: how well does ONE dependent chain of Fq multiplications per lane (what k_accumulate runs) feed the
// v_mad_u64_u32 pipe at a GIVEN occupancy?  Occupancy is forced through the LDS a block asks for (k_accumulate: 78 KiB per 256-lane
// block = two waves per SIMD).  Reported: wave-mads per second chip-wide.  Result (profiles/r03_ubench_multiplier.txt): the fp_mul / fp_sqr
// chains reach 473-485 G wave-mads/s from two waves per SIMD on (370-390 with one) - MORE than tools/ubench/valu_rates.hip saw with eight
// independent mads per wave (415 / 448 G/s at 2 / 4 waves), so this, not that, is the pipe's peak: 476 G wave-mads/s = 20.9 G Fq-mul/s;
// the dual product fp_mul2 runs at 427-431 (the split of its long columns costs feeding).
#include "../../zecale_amd/csrc/fp29.cuh"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace zkhip;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// One level of Karatsuba on the OPERAND half of the Montgomery product: a = a0 + a1 B^14, b likewise;
// z0 = a0 b0 (14 x 14), z2 = a1 b1 (13 x 13), zm = (a0 + a1)(b0 + b1) (14 x 14, limbs < 2^30: a column of 14 products < 2^64);
// product column t[k] = z0[k] + (zm - z0 - z2)[k - 14] + z2[k - 28] - 561 products instead of 729 - fed to the same interleaved
// reduction as fp_mul (729 reduction terms).  A COMPLETE multiplier (same result as fp_mul, checked below): the column sums of z0
// and z2 are each used twice, fourteen columns apart, so 2 x 14 of them stay live as 64-bit values.
template <class PR>
__device__ __forceinline__ Fp<PR> fp_mul_kara(Fp<PR> a, Fp<PR> b) {
  constexpr int N = PR::NL, H = 14, L = N - H;
  static_assert(N == 27, "");
  Fp<PR> r;
  uint32_t m[N], sa[H], sb[H];
#pragma unroll
  for (int i = 0; i < H; i++) { sa[i] = a.l[i] + (i < L ? a.l[H + i] : 0u); sb[i] = b.l[i] + (i < L ? b.l[H + i] : 0u); }
  uint64_t z0[2 * H - 1], z2[2 * L - 1];
  uint64_t acc = 0;
  // (exact loop bounds everywhere, as in fp_mul: with guarded bodies the compiler leaves the loops rolled and the arrays in scratch)
#pragma unroll
  for (int k = 0; k < 2 * N - 1; k++) {
    if (k <= 2 * H - 2) {                          // z0[k]
      uint64_t t = 0;
#pragma unroll
      for (int i = (k < H ? 0 : k - H + 1); i <= (k < H ? k : H - 1); i++) t += (uint64_t)a.l[i] * b.l[k - i];
      z0[k] = t;
      acc += t;
    }
    if (k >= H && k - H <= 2 * H - 2) {            // middle term, column c = k - H
      const int c = k - H;
      uint64_t t = 0;
#pragma unroll
      for (int i = (c < H ? 0 : c - H + 1); i <= (c < H ? c : H - 1); i++) t += (uint64_t)sa[i] * sb[c - i];
      if (c <= 2 * L - 2) {                        // z2[c] is first needed here
        uint64_t u = 0;
#pragma unroll
        for (int i = (c < L ? 0 : c - L + 1); i <= (c < L ? c : L - 1); i++) u += (uint64_t)a.l[H + i] * b.l[H + c - i];
        z2[c] = u;
        t -= u;
      }
      acc += t - z0[c];
    }
    if (k >= 2 * H && k - 2 * H <= 2 * L - 2) acc += z2[k - 2 * H];
    if (k < N) {
#pragma unroll
      for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * PR::P[k - i];
      m[k] = ((uint32_t)acc * PR::PINV) & M29;
      acc += (uint64_t)m[k] * PR::P[0];
    } else {
#pragma unroll
      for (int i = k - N + 1; i < N; i++) acc += (uint64_t)m[i] * PR::P[k - i];
      r.l[k - N] = (uint32_t)acc & M29;
    }
    acc >>= 29;
  }
  r.l[N - 1] = (uint32_t)acc;
  return r;
}

// fp_mul with ONE accumulator carried through all columns (the source's own shape): hipcc re-associates the column sums of fp_mul
// (the carry of the previous column is added LAST), runs eight columns side by side on eight register pairs and joins every column
// with a 64-bit addition (v_lshl_add_u64: as dear as a mad).  KIND 4: every mad its own asm statement - the compiler then puts an
// s_nop between any two of them (an inline asm's VGPR result read by the next instruction: gfx950's dst_sel forwarding hazard, assumed
// for asm), 1,458 per product.  KIND 5 - 7 (fp29_chain.cuh, what the library now uses): asm BLOCKS of up to 13 mads, ~150 nops per
// product.  profiles/r04_ubench_multiplier.txt.
#define ZK_MAD(acc, x, y) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc")
#define ZK_MAD_S(acc, x, c) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "s"(c) : "vcc")
template <class PR>
__device__ __forceinline__ Fp<PR> fp_mul_chain(Fp<PR> a, Fp<PR> b) {
  constexpr int N = PR::NL;
  Fp<PR> r;
  uint32_t m[N];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) ZK_MAD(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) ZK_MAD_S(acc, m[i], PR::P[k - i]);
    m[k] = ((uint32_t)acc * PR::PINV) & M29;
    ZK_MAD_S(acc, m[k], PR::P[0]);
    acc >>= 29;
  }
#pragma unroll
  for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
    for (int i = k - N + 1; i < N; i++) ZK_MAD(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - N + 1; i < N; i++) ZK_MAD_S(acc, m[i], PR::P[k - i]);
    r.l[k - N] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  r.l[N - 1] = (uint32_t)acc;
  return r;
}

#include "../../zecale_amd/csrc/fp29_chain.cuh"
template <int KIND>   // 5: fp_mul_chain2 (asm blocks of up to 13 mads), 0: fp_mul chain, 1: fp_sqr chain, 2: fp_mul2 chain, 3: Karatsuba fp_mul chain, 4: single-accumulator fp_mul
__global__ void __launch_bounds__(256) k_chain(const uint32_t* in, uint32_t* out, int iters) {
  extern __shared__ uint32_t lds[];
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq x, y, z;
  for (int i = 0; i < 27; i++) { x.l[i] = in[i] ^ (tid & 0xff); y.l[i] = in[27 + i] ^ ((tid >> 8) & 0xff); z.l[i] = in[i] ^ 0x55; }
  x.l[26] &= 0x3f; y.l[26] &= 0x3f; z.l[26] &= 0x3f;
  if (iters < 0) lds[threadIdx.x] = x.l[0];       // (keeps the allocation)
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) { x = fp_mul(x, y); y = fp_mul(y, x); }
    if (KIND == 1) { x = fp_sqr(x); y = fp_sqr(y); }
    if (KIND == 2) { x = fp_mul2(x, y, z, x); y = fp_mul2(y, x, z, y); }
    if (KIND == 3) { x = fp_mul_kara(x, y); y = fp_mul_kara(y, x); }
    if (KIND == 4) { x = fp_mul_chain(x, y); y = fp_mul_chain(y, x); }
    if (KIND == 5) { x = fp_mul_chain2(x, y); y = fp_mul_chain2(y, x); }
    if (KIND == 6) { x = fp_sqr_chain(x); y = fp_sqr_chain(y); }
    if (KIND == 7) { x = fp_mul2_chain(x, y, z, x); y = fp_mul2_chain(y, x, z, y); }
  }
  uint32_t s = 0;
  for (int i = 0; i < 27; i++) s ^= x.l[i] + y.l[i];
  out[tid] = s;
  if (iters == 3 && tid < 64) for (int i = 0; i < 27; i++) out[64 + tid * 54 + i] = x.l[i], out[64 + tid * 54 + 27 + i] = y.l[i];     // (the correctness check)
}

template <int KIND>
void run(const char* name, int mads_per_op, size_t lds_bytes, int waves, double peak) {
  const int iters = 200;
  int blocks = 256 * 2 * 4;                     // several rounds of the machine at any occupancy up to 8 blocks per CU
  int nthreads = blocks * 256;
  std::vector<uint32_t> h(54);
  for (auto& v : h) v = (uint32_t)rand() & M29;
  uint32_t *in, *out;
  CHECK(hipMalloc(&in, h.size() * 4)); CHECK(hipMalloc(&out, (size_t)nthreads * 4));
  CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipFuncSetAttribute((const void*)k_chain<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k_chain<KIND><<<blocks, 256, lds_bytes>>>(in, out, 2);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k_chain<KIND><<<blocks, 256, lds_bytes>>>(in, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double wave_mads = (double)nthreads / 64 * iters * 2 * mads_per_op;
  double rate = wave_mads / (ms * 1e-3) / 1e9;
  printf("%-8s %d wave(s)/SIMD (LDS %3zu KiB/block)  %.3f ms  %.1f G wave-mads/s = %.2f G %s/s\n", name, waves,
         lds_bytes / 1024, ms, rate, rate * 64 / mads_per_op, name);
  (void)peak;
  CHECK(hipFree(in)); CHECK(hipFree(out));
}

// the Karatsuba multiplier returns what fp_mul returns (the column sums differ, the reduced limbs do not: same carries, same quotient digits)
static bool check_kara() {
  std::vector<uint32_t> h(54);
  for (auto& v : h) v = (uint32_t)rand() & M29;
  uint32_t *in, *o0, *o3;
  const size_t words = 64 + 64 * 54;
  CHECK(hipMalloc(&in, h.size() * 4)); CHECK(hipMalloc(&o0, words * 4 + 256 * 4)); CHECK(hipMalloc(&o3, words * 4 + 256 * 4));
  CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipFuncSetAttribute((const void*)k_chain<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute((const void*)k_chain<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute((const void*)k_chain<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  k_chain<0><<<1, 64, 1024>>>(in, o0, 3);
  k_chain<3><<<1, 64, 1024>>>(in, o3, 3);
  CHECK(hipDeviceSynchronize());
  std::vector<uint32_t> a(words), b(words);
  CHECK(hipMemcpy(a.data(), o0, words * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b.data(), o3, words * 4, hipMemcpyDeviceToHost));
  bool ok = true;
  for (size_t i = 64; i < words; i++) ok = ok && a[i] == b[i];
  k_chain<4><<<1, 64, 1024>>>(in, o3, 3);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(b.data(), o3, words * 4, hipMemcpyDeviceToHost));
  for (size_t i = 64; i < words; i++) ok = ok && a[i] == b[i];
  CHECK(hipFuncSetAttribute((const void*)k_chain<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  k_chain<5><<<1, 64, 1024>>>(in, o3, 3);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(b.data(), o3, words * 4, hipMemcpyDeviceToHost));
  for (size_t i = 64; i < words; i++) ok = ok && a[i] == b[i];
  const int pairs[2][2] = {{1, 6}, {2, 7}};
  for (auto& pr : pairs) {
    if (pr[0] == 1) { k_chain<1><<<1, 64, 1024>>>(in, o0, 3); k_chain<6><<<1, 64, 1024>>>(in, o3, 3); }
    else { k_chain<2><<<1, 64, 1024>>>(in, o0, 3); k_chain<7><<<1, 64, 1024>>>(in, o3, 3); }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(a.data(), o0, words * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b.data(), o3, words * 4, hipMemcpyDeviceToHost));
    for (size_t i = 64; i < words; i++) ok = ok && a[i] == b[i];
  }
  CHECK(hipFree(in)); CHECK(hipFree(o0)); CHECK(hipFree(o3));
  return ok;
}

int main() {
  printf("Karatsuba and single-accumulator multipliers equal fp_mul on 64 lanes x 6 chained products: %s\n", check_kara() ? "yes" : "NO");
  struct { size_t lds; int waves; double peak; } occ[] = {{150 * 1024, 1, 224.0}, {78 * 1024, 2, 415.0}, {38 * 1024, 4, 448.0}};
  for (auto& o : occ) {
    run<0>("fp_mul", 1458, o.lds, o.waves, o.peak);
    run<1>("fp_sqr", 1107, o.lds, o.waves, o.peak);
    run<2>("fp_mul2", 2187, o.lds, o.waves, o.peak);
    run<3>("fp_mul_kara (1,290 mads; rate in fp_mul-equivalents of 1,458)", 1458, o.lds, o.waves, o.peak);
    run<4>("fp_mul_chain (one accumulator)", 1458, o.lds, o.waves, o.peak);
    run<5>("fp_mul_chain2 (one accumulator, asm blocks of <= 13 mads)", 1458, o.lds, o.waves, o.peak);
    run<6>("fp_sqr_chain", 1107, o.lds, o.waves, o.peak);
    run<7>("fp_mul2_chain", 2187, o.lds, o.waves, o.peak);
  }
  return 0;
}
