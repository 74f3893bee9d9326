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

template <int KIND>   // 0: fp_mul chain, 1: fp_sqr chain, 2: fp_mul2 chain
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
  }
  uint32_t s = 0;
  for (int i = 0; i < 27; i++) s ^= x.l[i] + y.l[i];
  out[tid] = s;
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

int main() {
  struct { size_t lds; int waves; double peak; } occ[] = {{150 * 1024, 1, 224.0}, {78 * 1024, 2, 415.0}, {38 * 1024, 4, 448.0}};
  for (auto& o : occ) {
    run<0>("fp_mul", 1458, o.lds, o.waves, o.peak);
    run<1>("fp_sqr", 1107, o.lds, o.waves, o.peak);
    run<2>("fp_mul2", 2187, o.lds, o.waves, o.peak);
  }
  return 0;
}
