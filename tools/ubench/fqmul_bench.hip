// Micro-benchmark: throughput of the product's Fq / Fr Montgomery multiplication (fp29.cuh)
// per lane, at 1 and 2 waves per SIMD, inline vs out-of-line.  Build twice:
//   hipcc --offload-arch=gfx950 -O3 -DZK_MUL_INLINE=0 -o fqmul_bench_noinl fqmul_bench.hip
//   hipcc --offload-arch=gfx950 -O3 -DZK_MUL_INLINE=1 -o fqmul_bench_inl  fqmul_bench.hip
#include "../../zecale_amd/csrc/fp29.cuh"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace zkhip;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <class PR, int WPS, bool SQR>
__global__ void __launch_bounds__(256, WPS) k_chain(const uint32_t* in, uint32_t* out, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fp<PR> x, y;
  for (int i = 0; i < PR::NL; i++) { x.l[i] = in[i] ^ (tid & 0xff); y.l[i] = in[PR::NL + i] ^ ((tid >> 8) & 0xff); }
  x.l[PR::NL - 1] &= 0x3f; y.l[PR::NL - 1] &= 0x3f;
  for (int it = 0; it < iters; it++) {
#pragma unroll 1
    for (int j = 0; j < 5; j++) {
      if (SQR) { x = fp_sqr(x); y = fp_sqr(y); }
      else { x = fp_mul(x, y); y = fp_mul(y, x); }
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < PR::NL; i++) s ^= x.l[i] + y.l[i];
  out[tid] = s;
}

template <class PR, int WPS, bool SQR>
void run(const char* name, int iters) {
  int blocks = 256 * WPS * 4;  // 4 full rounds of the machine
  int nthreads = blocks * 256;
  std::vector<uint32_t> h(2 * PR::NL);
  for (auto& v : h) v = (uint32_t)rand() & M29;
  uint32_t *in, *out;
  CHECK(hipMalloc(&in, h.size() * 4)); CHECK(hipMalloc(&out, nthreads * 4));
  CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k_chain<PR, WPS, SQR><<<blocks, 256>>>(in, out, 2);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k_chain<PR, WPS, SQR><<<blocks, 256>>>(in, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double muls = (double)nthreads * iters * 10;
  printf("%-28s waves/SIMD<=%d  %.2f ms  %.2f Gmul/s  (%.0f cycles@2.4GHz per wave-mul per SIMD)\n", name, WPS, ms, muls / ms / 1e6,
         2.4e9 / (muls / (ms * 1e-3) / 1024 / 64));
  CHECK(hipFree(in)); CHECK(hipFree(out));
}

int main() {
  printf("ZK_MUL_INLINE=%d\n", ZK_MUL_INLINE);
  run<FqParams, 1, false>("Fq mul", 40);
  run<FqParams, 2, false>("Fq mul", 40);
  run<FqParams, 1, true>("Fq sqr", 40);
  run<FqParams, 2, true>("Fq sqr", 40);
  run<FrParams, 2, false>("Fr mul", 150);
  run<FrParams, 4, false>("Fr mul", 150);
  run<FrParams, 8, false>("Fr mul", 150);
  return 0;
}
