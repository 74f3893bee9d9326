// What would witness generation cost on the GPU?  (measure, don't estimate.)
// The wrapping witness is a dependency chain: ~375 rounds in which up to four independent Fr inversions (the slopes of one Miller
// step) are followed by the Fq12 arithmetic that consumes them - ~44k Fr multiplications in total, of which
// the ~120 of one round are independent of each other but every round needs the previous one.  This benchmark runs exactly that
// shape with the product's own device arithmetic (fp29.cuh Fr, fp_inv.cuh): per lane, ROUNDS x [ one shared inversion for four
// values (Montgomery's trick: 1 inversion + 9 multiplications) + MULS dependent-on-the-round multiplications ], and reports
//   - the latency of one chain (one witness on one lane), with 1 wave on the chip and with the chip full,
//   - witnesses per second when every lane carries its own batch (the only parallelism a lane-per-witness design has),
//   - the same with a whole wave sharing one witness's independent multiplications (64 lanes split the MULS of a round; the
//     inversions are computed redundantly by all lanes, as SIMD lanes do).
// Build: hipcc --offload-arch=gfx950 -O3 -DZK_MUL_INLINE=1 -o build/witness_chain_bench tools/ubench/witness_chain_bench.hip
#include "../../zecale_amd/csrc/fp_inv.cuh"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace zkhip;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef Fp<FrParams> FrD;

constexpr int ROUNDS = 375, MULS = 120;

template <int SHARE>   // SHARE lanes cooperate on the independent multiplications of a round (1: lane per witness, 64: wave per witness)
__global__ void __launch_bounds__(256) k_chain(const uint32_t* in, uint32_t* out, int rounds) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  FrD x, y;
  for (int i = 0; i < 14; i++) { x.l[i] = in[i] ^ (SHARE == 1 ? (tid & 0xff) : (blockIdx.x & 0xff)); y.l[i] = in[14 + i]; }
  x.l[13] &= 0x3f; y.l[13] &= 0x3f;
  FrD acc = x;
#pragma unroll 1
  for (int r = 0; r < rounds; r++) {
    // four denominators derived from the running value, inverted together
    FrD d0 = fp_add(acc, y), d1 = fp_add(d0, y), d2 = fp_add(d1, y), d3 = fp_add(d2, y);
    FrD p1 = fp_mul(d0, d1), p2 = fp_mul(p1, d2), p3 = fp_mul(p2, d3);
    FrD inv = fp_inv<FrParams>(p3);
    FrD i3 = fp_mul(inv, p2); inv = fp_mul(inv, d3);
    FrD i2 = fp_mul(inv, p1); inv = fp_mul(inv, d2);
    FrD i1 = fp_mul(inv, d0); FrD i0 = fp_mul(inv, d1);
    acc = fp_add(fp_add(i0, i1), fp_add(i2, i3));
    // the round's multiplications (independent of each other, dependent on the inverses): a lane does its share
    FrD s = acc;
#pragma unroll 1
    for (int k = 0; k < (MULS + SHARE - 1) / SHARE; k++) s = fp_mul(s, fp_add(acc, y));
    if (SHARE > 1) {   // the partial results meet again (an exchange through LDS in a real kernel): model it with one cross-lane reduction
#pragma unroll
      for (int i = 0; i < 14; i++) s.l[i] ^= (uint32_t)__shfl_xor((int)s.l[i], 1);
    }
    acc = s;
  }
  uint32_t h = 0;
  for (int i = 0; i < 14; i++) h ^= acc.l[i];
  out[tid] = h;
}

template <int SHARE>
double run(int blocks, int rounds) {
  std::vector<uint32_t> h(28);
  for (auto& v : h) v = (uint32_t)rand() & M29;
  uint32_t *in, *out;
  CHECK(hipMalloc(&in, h.size() * 4)); CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k_chain<SHARE><<<blocks, 256>>>(in, out, 2);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k_chain<SHARE><<<blocks, 256>>>(in, out, rounds);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipFree(in)); CHECK(hipFree(out));
  return ms;
}

int main() {
  printf("witness-shaped chain: %d rounds x [4 inversions sharing one safegcd + %d multiplications], Fr of BW6-761\n", ROUNDS, MULS);
  double a = run<1>(1, ROUNDS);
  printf("lane per witness, ONE workgroup on the chip : %8.2f ms per chain (latency of one witness; 256 witnesses in flight)\n", a);
  double b = run<1>(256 * 8, ROUNDS);
  printf("lane per witness, chip full (524,288 lanes)  : %8.2f ms per chain -> %.0f witnesses/s IF 524,288 batches were in flight\n", b, 524288.0 / b * 1e3);
  double c = run<64>(1, ROUNDS);
  printf("wave per witness, ONE workgroup              : %8.2f ms per chain (4 witnesses in flight)\n", c);
  double d = run<64>(256 * 8, ROUNDS);
  printf("wave per witness, chip full (8,192 waves)    : %8.2f ms per chain -> %.0f witnesses/s IF 8,192 batches were in flight\n", d, 8192.0 / d * 1e3);
  double e = run<64>(4, ROUNDS);
  printf("wave per witness, 16 batches in flight       : %8.2f ms per chain -> %.0f witnesses/s (the streaming prover keeps ~14 batches in flight)\n", e, 16.0 / e * 1e3);
  return 0;
}
