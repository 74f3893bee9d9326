// Witness generation of the wrapping circuit ON THE GPU (BASELINE's north star lists it among the kernels).
// Replaces the generate_r1cs_witness calls of aggregator_circuit::prove (libzecale/circuits/aggregator_circuit.tcc:136-157,
// aggregator_gadget.tcc:87-112) for a server that keeps many batches in flight: the host generator (aggregator.cpp) takes 8 ms on
// three cores per batch - 25 core-ms, five busy cores at 200 proofs/s, and a one-GPU job gets sixteen.
//
// The assignment is a straight-line program over Fr recorded from the circuit's own template code (witness_tape.cpp): 327 k field
// operations for a batch of two, 144 k of them multiplications and 326 inversions, laid out in levels of ONE KIND of instruction
// each.  A level holds a few dozen independent instructions - nothing to fill a chip with, hardly a wave: k_witness interprets the
// program with ONE WAVE PER BATCH, 64 instructions (a "chunk") at a time, no barriers; instruction words and old operands are
// prefetched, recent results wait in an LDS ring; values are lazily reduced under bounds the tape builder tracks; every lane
// inverts with fp_inv (division steps), all lanes of a wave at once.  The key hash (a strictly sequential MiMC chain) is a second
// program run by a second wave on a second stream (k_witness_chain).  The parallelism is ACROSS batches: each batch in flight costs
// two waves out of the chip's 2,048+ wave slots, so witness generation rides along under the provers' kernels and the host cores
// are free for the tails.  27.7 ms per witness, sixteen witnesses per launch in the streaming prover.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>

#include "aggregator_internal.h"
#include "fp_inv.cuh"
#include "witness.h"
#include "../../include/zkhip.h"

namespace zkhip {

typedef Fp<FrParams> FrD;

// A value slot: the 14 limbs of the device form as they are (no packing: an addition is then 14 adds and a carry pass), padded
// to 16 words = 64 bytes.  Values are LAZILY reduced: the tape builder tracks an upper bound (a multiple of r) for every value,
// picks the subtraction's K r accordingly and inserts a reduction where a bound would pass 256 r (witness_tape.cpp).
constexpr int WSLOT = 16;
struct WVal { uint4 q[4]; };
__device__ __forceinline__ FrD w_from(const WVal& v) {
  FrD r;
  r.l[0] = v.q[0].x; r.l[1] = v.q[0].y; r.l[2] = v.q[0].z; r.l[3] = v.q[0].w; r.l[4] = v.q[1].x; r.l[5] = v.q[1].y; r.l[6] = v.q[1].z;
  r.l[7] = v.q[1].w; r.l[8] = v.q[2].x; r.l[9] = v.q[2].y; r.l[10] = v.q[2].z; r.l[11] = v.q[2].w; r.l[12] = v.q[3].x; r.l[13] = v.q[3].y;
  return r;
}
__device__ __forceinline__ WVal w_to(const FrD& r) {
  WVal v;
  v.q[0] = make_uint4(r.l[0], r.l[1], r.l[2], r.l[3]); v.q[1] = make_uint4(r.l[4], r.l[5], r.l[6], r.l[7]);
  v.q[2] = make_uint4(r.l[8], r.l[9], r.l[10], r.l[11]); v.q[3] = make_uint4(r.l[12], r.l[13], 0u, 0u);
  return v;
}
__device__ __forceinline__ WVal w_ld(const uint4* q) { WVal v; v.q[0] = q[0]; v.q[1] = q[1]; v.q[2] = q[2]; v.q[3] = q[3]; return v; }
__device__ __forceinline__ void w_st(uint4* q, const WVal& v) { q[0] = v.q[0]; q[1] = v.q[1]; q[2] = v.q[2]; q[3] = v.q[3]; }

// constants: ABI form -> a value slot (once per upload)
__global__ void __launch_bounds__(256) k_witness_consts(const uint64_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x[6];
#pragma unroll
  for (int k = 0; k < 6; k++) x[k] = in[(size_t)i * 6 + k];
  w_st(reinterpret_cast<uint4*>(out + (size_t)i * WSLOT), w_to(fp_cond_sub_p(fp_from_abi<FrParams>(x))));
}

__device__ __forceinline__ bool w_binary(uint32_t c) { return c == WT_ADD || c == WT_MUL || c >= WT_SUBK; }

// x (below 2^12 r) minus an estimated multiple of r: below 3r.  With t = floor(x / 2^358) (31 bits) and mu <= floor(2^390 / r)
// (off by one at most), q = floor(t mu / 2^32) is never above floor(x / r) and at most 2 below it.  Fourteen small products and a
// borrow chain - the price of an addition, where a multiplication by one would cost a level of the dear kind.
__device__ __forceinline__ FrD w_reduce(const FrD& x, uint32_t mu) {
  const uint32_t t = (x.l[13] << 19) | (x.l[12] >> 10);
  const uint32_t q = (uint32_t)(((uint64_t)t * mu) >> 32);
  FrD r;
  int64_t carry = 0;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    const int64_t v = (int64_t)x.l[i] - (int64_t)((uint64_t)q * FrParams::P[i]) + carry;
    r.l[i] = (i < 13) ? ((uint32_t)v & M29) : (uint32_t)v;
    carry = v >> 29;
  }
  return r;
}

// one instruction: operands x (and y), result r.  Returns true when an inversion that must not meet zero did.
// subk: the table of K r in subtraction-safe limbs (LDS), a value slot per log2 K.
__device__ __forceinline__ bool w_exec(uint32_t c, int32_t rb, const FrD& x, const FrD& y, const uint64_t* __restrict__ in, int32_t ra, FrD& r,
                                       const uint4* subk, uint32_t mu) {
  bool bad = false;
  if (c >= WT_SUBK) {                                                                  // a - b + 2^k r
    const FrD kp = w_from(w_ld(&subk[(c - WT_SUBK) * 4]));
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = x.l[i] + kp.l[i] - y.l[i];
    fp_normalise(r);
    return false;
  }
  switch (c) {
    case WT_INPUT: {
      uint64_t w[6];
#pragma unroll
      for (int k = 0; k < 6; k++) w[k] = in[(size_t)ra * 6 + k];
      r = fp_cond_sub_p(fp_from_abi<FrParams>(w));
      break;
    }
    case WT_ADD: r = fp_add(x, y); break;
    case WT_RED: r = w_reduce(x, mu); break;
    case WT_MUL: r = fp_mul(x, y); break;
    case WT_INV:
    case WT_INV0:
      r = fp_inv<FrParams>(fp_cond_sub_kp<FrParams, 2>(x));                           // (its operand is below 4r: the builder sees to it)
      bad = (c == WT_INV) && fp_is_zero_2p(r);                                        // the host generator would have taken another path
      break;
    default: {                                                                         // WT_BIT
      FrD one_raw = fp_zero<FrParams>();
      one_raw.l[0] = 1;
      uint32_t w[12];
      fp_pack32<FrParams>(fp_cond_sub_p(fp_mul(x, one_raw)), w);                        // the canonical integer
      r = ((w[rb >> 5] >> (rb & 31)) & 1u) ? fp_one<FrParams>() : fp_zero<FrParams>();
      break;
    }
  }
  return bad;
}

// Chunks [c0, c1) of the levelled program, a chunk being 64 consecutive positions (a level is padded to whole chunks, so a chunk
// never straddles two levels and running the chunks in order respects every dependency).  ONE WAVE PER BATCH: a level of this
// program holds a few dozen independent instructions - there is nothing for a second wave to do, and with one wave there is no
// barrier and no waiting for stores between levels.  What a level costs is then the latency of its operands, and the
// program being static, almost all of it is taken off the critical path:
//   * the instruction words are fetched three chunks ahead;
//   * the last RING results wait in an LDS ring (slot = position mod RING; 1,024 = the last 16 chunks held nine operands in ten);
//   * an older operand is loaded from the value array one chunk ahead (its store is many chunks old).
// (A witness is cut into several launches of a few milliseconds so that the kernels of the provers that share a hardware queue
// with it are not held up for its whole duration.)
struct WIns { uint32_t code; int32_t a, b; };
__device__ __forceinline__ WIns w_fetch(const WitnessProg& P, uint32_t chunk, uint32_t n_chunks, uint32_t lane) {
  // (no branch around the loads: the compiler counts outstanding loads exactly only in straight-line code; past the end the
  //  last chunk is fetched again and turned into no-ops)
  const uint32_t p = min(chunk, n_chunks - 1) * 64 + lane;
  WIns r{P.code[p], P.a[p], P.b[p]};
  if (chunk >= n_chunks) r.code = WT_NOP;
  return r;
}
// issue the loads of the operands of `in` that are complete in memory (constants; positions below lim).  ALWAYS two loads per lane,
// from a harmless address (constant 0) when there is nothing to fetch: a branch here would make the compiler wait for every
// outstanding memory operation - the previous chunk's stores included - at its join (measured: 1.2 us per chunk, whatever it held).
__device__ __forceinline__ void w_prefetch(const WitnessProg& P, const uint32_t* __restrict__ vals, const WIns& in, uint32_t lim, WVal& px, WVal& py) {
  const bool has_a = in.code != WT_NOP && in.code != WT_INPUT, has_b = w_binary(in.code);
  const uint32_t* pa = P.consts;
  const uint32_t* pb = P.consts;
  if (has_a && in.a < 0) pa = P.consts + (size_t)(-1 - in.a) * WSLOT;
  if (has_a && in.a >= 0 && (uint32_t)in.a < lim) pa = vals + (size_t)in.a * WSLOT;
  if (has_b && in.b < 0) pb = P.consts + (size_t)(-1 - in.b) * WSLOT;
  if (has_b && in.b >= 0 && (uint32_t)in.b < lim) pb = vals + (size_t)in.b * WSLOT;
  px = w_ld(reinterpret_cast<const uint4*>(pa));
  py = w_ld(reinterpret_cast<const uint4*>(pb));
}

// WPG witnesses per WORKGROUP, one wave each, RING results per wave in LDS: WPG x RING x 64 B = 64 KiB of the CU's 160 whatever the
// split.  Why the split matters (round 6): a witness wave cannot share a CU with TWO workgroups of k_accumulate (78 KiB of LDS and
// all 512 VGPRs of every SIMD each pair), so every CU that hosts a witness workgroup runs ONE accumulation workgroup instead of two -
// one wave per SIMD, 78 % of the multiplier's rate - for as long as the witness launch lasts.  With one witness per workgroup and
// its 64 KiB ring (rounds 3-5) a launch of sixteen witnesses took a slot on SIXTEEN CUs and eight launches in flight on half the
// chip: the 10 % by which the GPU generator trailed the device's own limit.  Four witnesses per workgroup - one wave per SIMD, a
// quarter of the ring each - take the same slot on FOUR CUs.  The shorter ring (the last 4 chunks instead of 16) turns some ring hits
// into prefetches from the value array, which the chunk-ahead prefetch already covers (it always issues its two loads per lane).
#ifndef ZK_WITNESS_PRIO
#define ZK_WITNESS_PRIO 1
#endif
constexpr int WIT_PRIO = ZK_WITNESS_PRIO;
template <uint32_t WPG, uint32_t WIT_RING>
__global__ void __launch_bounds__(64 * WPG) k_witness(WitnessProg P, uint32_t c0, uint32_t c1, const uint64_t* __restrict__ inputs /* batches x n_inputs x 6, ABI */,
                                                       uint32_t* __restrict__ values /* batches x n_pos x 16 */, uint32_t* __restrict__ flags, uint32_t n_batches) {
  __shared__ uint4 ring_all[WPG * WIT_RING * 4];
  __shared__ uint4 subk[WT_SUBK_LEVELS * 4];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t batch = blockIdx.x * WPG + wave;
  uint4* ring = ring_all + (size_t)wave * WIT_RING * 4;
  if (lane < WT_SUBK_LEVELS * 4) subk[lane] = reinterpret_cast<const uint4*>(P.subk)[lane];      // (every wave writes the same words, and reads what IT wrote)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (batch >= n_batches) return;                                   // (whole waves: nothing below synchronises across waves)
  // A witness wave shares its SIMD with a wave of k_accumulate, which raises its own priority as it goes (s_setprio 1 .. 3): at the
  // default priority the witness wave got the issue slots the other one left - a launch of sixteen witnesses lasted ~190 ms beside the
  // provers against ~25 ms alone, holding its CUs' second workgroup slot all that time.  Its instruction stream is a trickle next to
  // the accumulation's (a dependent chain that waits on LDS and memory most of the time), so it takes the SIMD whenever it can issue.
  if (WIT_PRIO) __builtin_amdgcn_s_setprio(3);
  const uint64_t* in = inputs + (size_t)batch * P.n_inputs * 6;
  uint32_t* vals = values + (size_t)batch * P.n_pos * WSLOT;
  uint32_t bad = 0;
  // one chunk: `cur` with its prefetched operands (cx, cy); `nxt` is the chunk after it, whose operands are requested here
  auto step = [&](uint32_t j, const WIns& cur, WVal& cx, WVal& cy, const WIns& nxt, WVal& nx, WVal& ny) {
    // chunk i finds in the ring what this launch wrote and chunk i itself does not overwrite: positions from ring_lo(i) on
    auto ring_lo = [&](uint32_t i) { const uint32_t w = (i + 1) * 64; return max(c0 * 64, w > WIT_RING ? w - WIT_RING : 0u); };
    w_prefetch(P, vals, nxt, ring_lo(j + 1), nx, ny);
    const uint32_t lim = ring_lo(j);
    const uint32_t c = cur.code;
    if (c != WT_NOP) {
      FrD x = fp_zero<FrParams>(), y = x, r = x;
      if (c != WT_INPUT) {
        x = w_from((cur.a < 0 || (uint32_t)cur.a < lim) ? cx : w_ld(&ring[((uint32_t)cur.a % WIT_RING) * 4]));
        if (w_binary(c)) y = w_from((cur.b < 0 || (uint32_t)cur.b < lim) ? cy : w_ld(&ring[((uint32_t)cur.b % WIT_RING) * 4]));
      }
      if (w_exec(c, cur.b, x, y, in, cur.a, r, subk, P.mu)) bad = 1;
      const WVal v = w_to(r);
      w_st(reinterpret_cast<uint4*>(vals + ((size_t)j * 64 + lane) * WSLOT), v);
      w_st(&ring[((j * 64 + lane) % WIT_RING) * 4], v);
    }
    // the ring: this chunk's reads above come before its writes in program order, the next chunk's reads after them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  const uint32_t n_chunks = c1;
  WIns i0 = w_fetch(P, c0, n_chunks, lane), i1 = w_fetch(P, c0 + 1, n_chunks, lane), i2 = w_fetch(P, c0 + 2, n_chunks, lane), i3 = w_fetch(P, c0 + 3, n_chunks, lane);
  WVal ax, ay, bx, by;
  for (int k = 0; k < 4; k++) ax.q[k] = ay.q[k] = bx.q[k] = by.q[k] = make_uint4(0, 0, 0, 0);
  w_prefetch(P, vals, i0, c0 * 64, ax, ay);                            // everything before this launch is in memory (ring_lo(c0))
#pragma unroll 1
  for (uint32_t j = c0; j < c1; j += 2) {
    const WIns n4 = w_fetch(P, j + 4, n_chunks, lane), n5 = w_fetch(P, j + 5, n_chunks, lane);
    step(j, i0, ax, ay, i1, bx, by);
    if (j + 1 < c1) step(j + 1, i1, bx, by, i2, ax, ay);
    i0 = i2; i1 = i3; i2 = n4; i3 = n5;
  }
  if (bad) atomicOr(&flags[batch], 1u);
}

// The key-hash chain ([chain_start, n_pos) of the program, in execution order) by ONE wave per batch: every lane computes the
// same instruction (a chain has nothing to spread), lane 0 stores.  The last 64 results wait in an LDS ring; the instruction
// words are fetched 64 at a time, a lane each, and broadcast.
__global__ void __launch_bounds__(64) k_witness_chain(WitnessProg P, uint32_t chain_start, const uint64_t* __restrict__ inputs,
                                                       uint32_t* __restrict__ values, uint32_t* __restrict__ flags) {
  __shared__ uint4 ring[64 * 4];
  __shared__ uint4 subk[WT_SUBK_LEVELS * 4];
  const uint32_t batch = blockIdx.x, lane = threadIdx.x;
  if (lane < WT_SUBK_LEVELS * 4) subk[lane] = reinterpret_cast<const uint4*>(P.subk)[lane];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const uint64_t* in = inputs + (size_t)batch * P.n_inputs * 6;
  uint32_t* vals = values + (size_t)batch * P.n_pos * WSLOT;
  uint32_t bad = 0;
#pragma unroll 1
  for (uint32_t base = chain_start; base < P.n_pos; base += 64) {
    const uint32_t mine = base + lane;
    const uint32_t my_c = mine < P.n_pos ? P.code[mine] : (uint32_t)WT_NOP;
    const int32_t my_a = mine < P.n_pos ? P.a[mine] : 0, my_b = mine < P.n_pos ? P.b[mine] : 0;
    const uint32_t n = min(64u, P.n_pos - base);
#pragma unroll 1
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t p = base + i;
      const uint32_t c = (uint32_t)__shfl((int)my_c, (int)i);
      const int32_t ra = __shfl(my_a, (int)i), rb = __shfl(my_b, (int)i);
      if (c == WT_NOP) continue;
      auto load = [&](int32_t ref) -> FrD {
        if (ref < 0) return w_from(w_ld(reinterpret_cast<const uint4*>(P.consts + (size_t)(-1 - ref) * WSLOT)));
        if ((uint32_t)ref + 64 > p) return w_from(w_ld(&ring[((uint32_t)ref % 64) * 4]));     // one of the last 64 results
        return w_from(w_ld(reinterpret_cast<const uint4*>(vals + (size_t)ref * WSLOT)));       // older: its store is long complete
      };
      FrD x = fp_zero<FrParams>(), y = x, r = x;
      if (c != WT_INPUT) x = load(ra);
      if (w_binary(c)) y = load(rb);
      if (w_exec(c, rb, x, y, in, ra, r, subk, P.mu)) bad = 1;
      const WVal v = w_to(r);
      if (lane == 0) {
        w_st(reinterpret_cast<uint4*>(vals + (size_t)p * WSLOT), v);
        w_st(&ring[(p % 64) * 4], v);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // LDS write visible to the wave's next read
      __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");          // global stores of this chunk complete before older values are re-read
  }
  if (bad && lane == 0) atomicOr(&flags[batch], 1u);
}

// the assignment, in ABI form (after both programs)
__global__ void __launch_bounds__(256) k_witness_out(WitnessProg P, const uint32_t* __restrict__ values, uint64_t* __restrict__ z_out) {
  const uint32_t batch = blockIdx.y;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_vars) return;
  const uint32_t* vals = values + (size_t)batch * P.n_pos * WSLOT;
  const int32_t ref = P.out_ref[i];
  FrD v = w_from(w_ld(reinterpret_cast<const uint4*>(ref >= 0 ? vals + (size_t)ref * WSLOT : P.consts + (size_t)(-1 - ref) * WSLOT)));
  uint64_t w[6];
  fp_to_abi<FrParams>(v, w);                                       // (a full reduction: any bound below 2^10 r)
  uint64_t* z = z_out + ((size_t)batch * P.n_vars + i) * 6;
#pragma unroll
  for (int k = 0; k < 6; k++) z[k] = w[k];
}

// ------------------------------------------------------------------------------------------------ host side
struct GpuWitnessState {
  WitnessTape tape;
  bool built = false;
  std::map<int, WitnessProgDev> dev;        // per device
};

void witness_prog_free(WitnessProgDev* pd) {
  for (void*& p : pd->bufs) if (p) { (void)hipFree(p); p = nullptr; }
}

static void gpu_release(zkhip_aggregator* a) {
  GpuWitnessState* st = (GpuWitnessState*)a->gpu_state;
  if (!st) return;
  for (auto& kv : st->dev) {
    if (hipSetDevice(kv.first) != hipSuccess) continue;
    witness_prog_free(&kv.second);
  }
  delete st;
  a->gpu_state = nullptr;
}

// a recorded program on the calling thread's current device
int witness_prog_upload(const WitnessTape& T, WitnessProgDev* out, char* err, size_t errlen) {
  WitnessProgDev pd;
  const size_t n = T.code.size(), nc = T.consts.size() / 6;
  uint64_t* d_c64 = nullptr;
  hipError_t e = hipSuccess;
  auto up = [&](int slot, const void* src, size_t bytes) {
    if (e != hipSuccess) return;
    e = hipMalloc(&pd.bufs[slot], bytes ? bytes : 4);
    if (e == hipSuccess && bytes) e = hipMemcpy(pd.bufs[slot], src, bytes, hipMemcpyHostToDevice);
  };
  up(0, T.code.data(), n); up(1, T.a.data(), n * 4); up(2, T.b.data(), n * 4);
  up(3, T.level_start.data(), T.level_start.size() * 4); up(4, T.out_ref.data(), T.out_ref.size() * 4);
  if (e == hipSuccess) e = hipMalloc(&pd.bufs[5], nc * 64 + 64);
  // K r (K = 2^k, k < WT_SUBK_LEVELS) in subtraction-safe limbs: every limb but the top raised by 2^29 at its upper neighbour's expense
  uint32_t subk[WT_SUBK_LEVELS][16];
  memset(subk, 0, sizeof subk);
  for (int k = 0; k < WT_SUBK_LEVELS; k++) {
    uint64_t c = 0;
    for (int i = 0; i < 14; i++) {
      c += (uint64_t)FrParams::P[i] << k;
      subk[k][i] = (i < 13) ? (uint32_t)(c & M29) : (uint32_t)c;
      c >>= 29;
    }
    for (int i = 0; i < 13; i++) { subk[k][i] += 1u << 29; subk[k][i + 1] -= 1u; }
  }
  up(6, subk, sizeof subk);
  // mu <= floor(2^390 / r), from the top 64 bits of r (bits 313 .. 376)
  uint32_t mu = 0;
  {
    unsigned __int128 top = 0;                 // r >> 290 (limbs 10 .. 13), then >> 23
    for (int i = 13; i >= 10; i--) top = (top << 29) | FrParams::P[i];
    const uint64_t p_hi = (uint64_t)(top >> 23);
    mu = (uint32_t)((((unsigned __int128)1) << 77) / ((unsigned __int128)p_hi + 1));
  }
  if (e == hipSuccess) e = hipMalloc(&d_c64, nc * 48 + 48);
  if (e == hipSuccess) e = hipMemcpy(d_c64, T.consts.data(), nc * 48, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_witness_consts, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, 0, d_c64, (uint32_t*)pd.bufs[5], (uint32_t)nc);
    e = hipDeviceSynchronize();
  }
  if (d_c64) (void)hipFree(d_c64);
  if (e != hipSuccess) {
    witness_prog_free(&pd);
    snprintf(err, errlen, "witness program upload: %s", hipGetErrorString(e));
    return ZKHIP_ERR_HIP;
  }
  pd.prog = WitnessProg{(const uint8_t*)pd.bufs[0], (const int32_t*)pd.bufs[1], (const int32_t*)pd.bufs[2], (const uint32_t*)pd.bufs[3],
                        (const int32_t*)pd.bufs[4], (const uint32_t*)pd.bufs[5], (uint32_t)(T.level_start.size() - 1), (uint32_t)n,
                        (uint32_t)T.n_vars, (uint32_t)T.n_inputs, T.chain_start, (const uint32_t*)pd.bufs[6], mu};
  *out = pd;
  return ZKHIP_OK;
}

// the program of `a` on the calling thread's current device (built and uploaded on first use)
int witness_prog(zkhip_aggregator* a, WitnessProg* out, const WitnessTape** tape, char* err, size_t errlen) {
  std::lock_guard<std::mutex> lk(a->gpu_mu);
  if (!a->gpu_state) { a->gpu_state = new GpuWitnessState(); a->gpu_release = gpu_release; }
  GpuWitnessState* st = (GpuWitnessState*)a->gpu_state;
  if (!st->built) {
    std::string e;
    if (witness_tape_build(a->num_proofs, a->inputs_per_proof, &st->tape, &e) != 0) { snprintf(err, errlen, "witness tape: %s", e.c_str()); return ZKHIP_ERR_STATE; }
    if (st->tape.n_vars != a->n_vars) { snprintf(err, errlen, "witness tape: %zu variables, the circuit has %zu", st->tape.n_vars, a->n_vars); return ZKHIP_ERR_STATE; }
    st->built = true;
  }
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) { snprintf(err, errlen, "hipGetDevice failed"); return ZKHIP_ERR_HIP; }
  auto it = st->dev.find(device);
  if (it == st->dev.end()) {
    WitnessProgDev pd;
    int rc = witness_prog_upload(st->tape, &pd, err, errlen);
    if (rc != ZKHIP_OK) return rc;
    it = st->dev.emplace(device, pd).first;
  }
  *out = it->second.prog;
  if (tape) *tape = &st->tape;
  return ZKHIP_OK;
}

void witness_launch(const WitnessProg& P, const uint64_t* d_inputs, uint32_t* d_values, uint64_t* d_z, uint32_t* d_flags, uint32_t batches,
                    hipStream_t st, hipStream_t st_chain, hipEvent_t ev_fork, hipEvent_t ev_join) {
  // the key-hash chain on a second stream, next to the levelled program; the assignment is gathered when both are done
  (void)hipEventRecord(ev_fork, st);
  (void)hipStreamWaitEvent(st_chain, ev_fork, 0);
  if (P.chain_start < P.n_pos) hipLaunchKernelGGL(k_witness_chain, dim3(batches), dim3(64), 0, st_chain, P, P.chain_start, d_inputs, d_values, d_flags);
  (void)hipEventRecord(ev_join, st_chain);
  static const uint32_t seg_env = [] { const char* e = getenv("ZKHIP_WITNESS_SEGMENT"); int v = e ? atoi(e) : 0; return (uint32_t)(v >= 64 && v <= (1 << 20) ? v : 2048); }();
  const uint32_t n_chunks = P.chain_start / 64, seg = seg_env;       // chunks per launch: a few milliseconds (tuning knob: ZKHIP_WITNESS_SEGMENT)
  // witnesses per workgroup (see k_witness): 4 by default; ZKHIP_WITNESS_WPG = 1 | 2 | 4 (1: rounds 3-5's form, a 64 KiB ring per witness)
  static const int wpg = [] { const char* e = getenv("ZKHIP_WITNESS_WPG"); const int v = e ? atoi(e) : 4; return (v == 1 || v == 2 || v == 4) ? v : 4; }();
  for (uint32_t c0 = 0; c0 < n_chunks; c0 += seg) {
    const uint32_t c1 = c0 + seg < n_chunks ? c0 + seg : n_chunks;
    if (wpg == 4) hipLaunchKernelGGL((k_witness<4, 256>), dim3((batches + 3) / 4), dim3(256), 0, st, P, c0, c1, d_inputs, d_values, d_flags, batches);
    else if (wpg == 2) hipLaunchKernelGGL((k_witness<2, 512>), dim3((batches + 1) / 2), dim3(128), 0, st, P, c0, c1, d_inputs, d_values, d_flags, batches);
    else hipLaunchKernelGGL((k_witness<1, 1024>), dim3(batches), dim3(64), 0, st, P, c0, c1, d_inputs, d_values, d_flags, batches);
  }
  (void)hipStreamWaitEvent(st, ev_join, 0);
  hipLaunchKernelGGL(k_witness_out, dim3((P.n_vars + 255) / 256, batches), dim3(256), 0, st, P, d_values, d_z);
}

}  // namespace zkhip
