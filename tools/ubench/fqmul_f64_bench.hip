// Micro-benchmark: could an Fq Montgomery product built on the FP64 FMA pipe beat the v_mad_u64_u32 one?
// The FP64 construction (Emmart / Zheng / Weems: "Faster modular exponentiation using double precision floating point
// arithmetic on the GPU"): limbs of 52 bits in doubles, 15 limbs for 761 bits (780-bit radix); a limb product a_i b_j < 2^104 is
// split by two FMAs in round-toward-zero,
//     hi = fma(a_i, b_j, 2^104)            (its mantissa holds floor(a_i b_j / 2^52))
//     lo = fma(a_i, b_j, (2^104 + 2^52) - hi)   (its mantissa holds a_i b_j mod 2^52)
// and the two raw bit patterns are added into 64-bit integer column accumulators (the constant offsets are subtracted once per
// column).  Per limb product: 2 x v_fma_f64 + 1 x v_add_f64 + 2 x v_lshl_add_u64 = 5 VALU instructions for 52 x 52 bits; the
// integer multiplier spends ONE v_mad_u64_u32 on 29 x 29 bits.  One Fq product = 15 x 15 limb products + as many reduction
// terms = 450 of those 5-instruction steps (2,250 instructions) against 1,458 mads.
// This benchmark times exactly that inner step, 450 per "multiplication", with nothing else (no carries, no quotient digits, no
// conversions): an UPPER BOUND on what an FP64 multiplier could reach.  The gate was 1.25 x 19.5 G Fq-mul/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int NL = 15;
constexpr int MULS_PER_THREAD = 64;

__global__ void __launch_bounds__(256) k_f64(double* out, unsigned long long* outi, unsigned seed) {
  // 15-limb operands: integers below 2^52 held in doubles
  double a[NL], b[NL], p[NL];
  for (int i = 0; i < NL; i++) {
    a[i] = (double)(((unsigned long long)(threadIdx.x * 2654435761u + seed + i) << 20) | 12345u);
    b[i] = (double)(((unsigned long long)(threadIdx.x * 40503u + seed * 7 + i) << 19) | 54321u);
    p[i] = (double)(((unsigned long long)(seed * 977 + i) << 21) | 999u);
  }
  const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
  unsigned long long acc_lo = 0, acc_hi = 0;
  __builtin_amdgcn_s_setreg(1 | (2 << 6) | (1 << 11), 3);          // MODE.FP_ROUND[3:2] (f64 / f16): 3 = toward zero
  for (int it = 0; it < MULS_PER_THREAD; it++) {
    // product columns (225 limb products) and reduction columns (225 more: the quotient digits m_i are stand-ins, the
    // accumulator's own low bits, since only the instruction mix matters here)
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
#pragma unroll
      for (int i = (k < NL ? 0 : k - NL + 1); i <= (k < NL ? k : NL - 1); i++) {
        double hi = __builtin_fma(a[i], b[k - i], C1);
        double lo = __builtin_fma(a[i], b[k - i], C2 - hi);
        acc_hi += (unsigned long long)__double_as_longlong(hi);
        acc_lo += (unsigned long long)__double_as_longlong(lo);
      }
#pragma unroll
      for (int i = (k < NL ? 0 : k - NL + 1); i <= (k < NL ? k : NL - 1); i++) {
        double hi = __builtin_fma(b[i], p[k - i], C1);
        double lo = __builtin_fma(b[i], p[k - i], C2 - hi);
        acc_hi += (unsigned long long)__double_as_longlong(hi);
        acc_lo += (unsigned long long)__double_as_longlong(lo);
      }
    }
    a[it % NL] = (double)(acc_lo & 0xFFFFFFFFFFFFFull);             // feed a result back so that nothing is hoisted
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a[0] + a[1];
  outi[blockIdx.x * blockDim.x + threadIdx.x] = acc_lo ^ acc_hi;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs=%d\n", prop.gcnArchName, prop.multiProcessorCount);
  for (int waves : {1, 2, 4}) {
    const int blocks = 256 * waves;                                   // 256 threads = one wave per SIMD of a CU
    double* out; unsigned long long* outi;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8)); CHECK(hipMalloc(&outi, (size_t)blocks * 256 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_f64<<<blocks, 256>>>(out, outi, 1u); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_f64<<<blocks, 256>>>(out, outi, 2u);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double muls = (double)blocks * 256 * MULS_PER_THREAD;
    printf("fp64 split-product inner loop: waves/SIMD=%d  %.3f ms  => %.2f G Fq-mul-equivalents/s (upper bound; integer multiplier: 19.5 measured)\n",
           waves, ms, muls / ms / 1e6);
    CHECK(hipFree(out)); CHECK(hipFree(outi));
  }
  return 0;
}
