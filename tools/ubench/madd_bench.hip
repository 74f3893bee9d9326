// Micro-benchmark of the memory-resident mixed addition (ec_mem.cuh madd_mem) without the MSM's
// gathers: every lane adds points from a small L2-resident table into its own accumulator.
#include "../../zecale_amd/csrc/ec_mem.cuh"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace zkhip;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(256, 2) k(const AffPacked* __restrict__ tab, int ntab, uint32_t* __restrict__ work, uint32_t n, int iters) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  XyzzRef acc = make_ref(work, n, t);
  const AffPacked* p0 = &tab[t % ntab];
  mem_st(acc, CX, aff_ld_x(p0)); mem_st(acc, CY, aff_ld_y(p0, false));
  mem_st(acc, CZZ, fp_one<FqParams>()); mem_st(acc, CZZZ, fp_one<FqParams>());
  for (int it = 0; it < iters; it++) {
    const AffPacked* p = &tab[(MODE == 0) ? ((t * 7 + it * 13 + 1) % ntab) : ((t * 2654435761u + it * 40503u) % (uint32_t)ntab)];
    madd_mem(acc, p, (it & 1) != 0);
  }
}

int main(int argc, char** argv) {
  int iters = 32;
  uint32_t n = 786432;
  for (int mode = 0; mode < 2; mode++) {
    int ntab = mode == 0 ? 1024 : (1 << 20);
    // table of pseudo points (not on the curve; the formulas do not care for timing)
    std::vector<uint32_t> h((size_t)ntab * 48);
    for (auto& v : h) v = (uint32_t)rand() * 2654435761u;
    for (int i = 0; i < ntab; i++) { h[(size_t)i * 48 + 23] &= 0x00ffffff; h[(size_t)i * 48 + 47] &= 0x00ffffff; }
    AffPacked* tab; uint32_t* work;
    CHECK(hipMalloc(&tab, h.size() * 4)); CHECK(hipMalloc(&work, (size_t)n * 108 * 4));
    CHECK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
      CHECK(hipEventRecord(e0));
      if (mode == 0) k<0><<<(n + 255) / 256, 256>>>(tab, ntab, work, n, iters);
      else k<1><<<(n + 255) / 256, 256>>>(tab, ntab, work, n, iters);
      CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("mode %d (table %d pts): %d madds x %u lanes: %.2f ms  -> %.2f Gmul-eq/s\n", mode, ntab, iters, n, ms, (double)n * iters * 10 / ms / 1e6);
    }
    CHECK(hipFree(tab)); CHECK(hipFree(work));
  }
  return 0;
}
