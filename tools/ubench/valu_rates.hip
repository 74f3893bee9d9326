// Micro-benchmark: issue rate of the integer/fp64 VALU ops a 768-bit Montgomery
// multiplier can be built from, on gfx950.  Prints cycles per wave-instruction
// per SIMD (assuming the shader clock reported by s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;   // independent chains

template <int OP>
__global__ void __launch_bounds__(256) k(unsigned* out, unsigned long long* cyc, unsigned seed) {
  unsigned a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  unsigned long long acc[UNROLL];
  double dacc[UNROLL];
  for (int i = 0; i < UNROLL; i++) { acc[i] = a + i; dacc[i] = (double)(a + i); }
  double da = (double)a * 1e-9, db = (double)b * 1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) {
      if constexpr (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      if constexpr (OP == 1) { unsigned lo = (unsigned)acc[i]; asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(lo) : "v"(a)); acc[i] = lo; }
      if constexpr (OP == 2) { unsigned lo = (unsigned)acc[i]; asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(lo) : "v"(a)); acc[i] = lo; }
      if constexpr (OP == 3) { unsigned lo = (unsigned)acc[i]; asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b)); acc[i] = lo; }
      if constexpr (OP == 4) { unsigned lo = (unsigned)acc[i]; asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(lo) : "v"(a)); acc[i] = lo; }
      if constexpr (OP == 5) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dacc[i]) : "v"(da), "v"(db));
      if constexpr (OP == 6) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % UNROLL]));
      if constexpr (OP == 7) { unsigned lo = (unsigned)acc[i]; asm volatile("v_add_co_u32 %0, vcc, %1, %0\n\tv_addc_co_u32 %0, vcc, %2, %0, vcc" : "+v"(lo) : "v"(a), "v"(b) : "vcc"); acc[i] = lo; }
      if constexpr (OP == 8) { unsigned lo = (unsigned)acc[i]; asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b)); acc[i] = lo; }
      if constexpr (OP == 9) { float f = __uint_as_float((unsigned)acc[i]); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f) : "v"(__uint_as_float(a)), "v"(__uint_as_float(b))); acc[i] = __float_as_uint(f); }
      if constexpr (OP == 10) { unsigned lo = (unsigned)acc[i]; asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b)); acc[i] = lo; }
      if constexpr (OP == 11) { unsigned lo = (unsigned)acc[i]; asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(acc[i])); }
      if constexpr (OP == 12) { unsigned lo = (unsigned)acc[i]; asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b)); acc[i] = lo; }
      if constexpr (OP == 13) { unsigned lo = (unsigned)acc[i]; asm volatile("v_and_b32 %0, %1, %0" : "+v"(lo) : "v"(a)); acc[i] = lo; }
      if constexpr (OP == 14) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "s"(seed) : "vcc");
      if constexpr (OP == 15) asm volatile("v_add_f64 %0, %1, %0" : "+v"(dacc[i]) : "v"(da));
      if constexpr (OP == 16) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(dacc[i]) : "v"(da));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long s = 0; double ds = 0;
  for (int i = 0; i < UNROLL; i++) { s += acc[i]; ds += dacc[i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)s + (unsigned)ds;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int waves_per_simd) {
  int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
  unsigned* out; unsigned long long* cyc;
  CHECK(hipMalloc(&out, blocks * 256 * 4)); CHECK(hipMalloc(&cyc, blocks * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<OP><<<blocks, 256>>>(out, cyc, 12345u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k<OP><<<blocks, 256>>>(out, cyc, 12345u);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks);
  CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
  double avg = 0; for (auto c : h) avg += (double)c; avg /= blocks;
  double ninstr = (double)ITERS * UNROLL;             // per wave
  // s_memtime ticks at 100 MHz on some parts; report both interpretations
  double total_wave_instr = ninstr * blocks * 4;
  double per_simd_ns = (double)ms * 1e6 / (ninstr * waves_per_simd);
  printf("%-22s waves/SIMD=%d  wall=%.3f ms  ns/instr/SIMD=%.3f  (=> %.2f cyc @2.4GHz)  memtime_ticks/instr/wave=%.3f  Ginstr/s=%.1f\n",
         name, waves_per_simd, ms, per_simd_ns, per_simd_ns * 2.4, avg / ninstr, total_wave_instr / ms / 1e6);
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  for (int w : {1, 2, 4}) {
    run<9>("v_fma_f32", w);
    run<0>("v_mad_u64_u32", w);
    run<14>("v_mad_u64_u32(sgpr)", w);
    run<1>("v_mul_lo_u32", w);
    run<2>("v_mul_hi_u32", w);
    run<3>("v_mad_u32_u24", w);
    run<12>("v_mad_i32_i24", w);
    run<4>("v_mul_hi_u32_u24", w);
    run<10>("v_mad_u32_u16", w);
    run<5>("v_fma_f64", w);
    run<15>("v_add_f64", w);
    run<16>("v_mul_f64", w);
    run<6>("v_lshl_add_u64", w);
    run<7>("v_add_co+v_addc_co", w);
    run<8>("v_add3_u32", w);
    run<11>("v_lshrrev_b64", w);
    run<13>("v_and_b32", w);
  }
  return 0;
}
