// Witness generation of the wrapping circuit ON THE GPU (BASELINE's north star lists it among the kernels).
// Replaces the generate_r1cs_witness calls of aggregator_circuit::prove (libzecale/circuits/aggregator_circuit.tcc:136-157,
// aggregator_gadget.tcc:87-112) for a server that keeps many batches in flight: the host generator (aggregator.cpp) takes 8 ms on
// three cores per batch - 25 core-ms, five busy cores at 200 proofs/s, and a one-GPU job gets sixteen.
//
// The assignment is a straight-line program over Fr recorded from the circuit's own template code (witness_tape.cpp): ~370 k
// field operations, 1,334 of them inversions, ~9,800 dependent levels deep (the MiMC chain of the key hash is the longest path,
// the 253-step input accumulator with one inversion per step the slowest).  k_witness interprets it with ONE WORKGROUP PER BATCH:
// the instructions of a level are independent and spread over the workgroup's 256 lanes; a level ends with __syncthreads (results
// travel through the value array in global memory, which is coherent inside a CU); every lane inverts with fp_inv (division
// steps), all lanes of a wave at once.  There is nothing to fill a chip with inside one witness - the parallelism is ACROSS
// batches: each batch in flight costs four waves out of the chip's 2,048+ wave slots, so witness generation rides along under
// the provers' kernels and the host cores are free for the tails.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>

#include "aggregator_internal.h"
#include "fp_inv.cuh"
#include "witness.h"
#include "../../include/zkhip.h"

namespace zkhip {

typedef Fp<FrParams> FrD;

__device__ __forceinline__ FrD w_load12(const uint32_t* p) {
  uint32_t w[12];
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 x = q[0], y = q[1], z = q[2];
  w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w; w[4] = y.x; w[5] = y.y; w[6] = y.z; w[7] = y.w;
  w[8] = z.x; w[9] = z.y; w[10] = z.z; w[11] = z.w;
  return fp_unpack32<FrParams>(w);
}
__device__ __forceinline__ void w_store12(uint32_t* p, const FrD& v) {     // v < 2^384, limbs normalised
  uint32_t w[12];
  fp_pack32<FrParams>(v, w);
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]);
  q[1] = make_uint4(w[4], w[5], w[6], w[7]);
  q[2] = make_uint4(w[8], w[9], w[10], w[11]);
}

// constants: ABI form -> packed device form (once per upload)
__global__ void __launch_bounds__(256) k_witness_consts(const uint64_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x[6];
#pragma unroll
  for (int k = 0; k < 6; k++) x[k] = in[(size_t)i * 6 + k];
  w_store12(out + (size_t)i * 12, fp_cond_sub_p(fp_from_abi<FrParams>(x)));
}

// one instruction: operands x (and y), result r.  Returns true when an inversion that must not meet zero did.
__device__ __forceinline__ bool w_exec(uint32_t c, int32_t rb, const FrD& x, const FrD& y, const uint64_t* __restrict__ in, int32_t ra, FrD& r) {
  bool bad = false;
  switch (c) {
    case WT_INPUT: {
      uint64_t w[6];
#pragma unroll
      for (int k = 0; k < 6; k++) w[k] = in[(size_t)ra * 6 + k];
      r = fp_cond_sub_p(fp_from_abi<FrParams>(w));
      break;
    }
    case WT_ADD: r = fp_cond_sub_kp<FrParams, 2>(fp_add(x, y)); break;                  // stored values stay below 2p
    case WT_SUB: r = fp_cond_sub_kp<FrParams, 2>(fp_sub<FrParams, 2>(x, y)); break;
    case WT_MUL: r = fp_mul(x, y); break;
    case WT_INV:
    case WT_INV0:
      r = fp_inv<FrParams>(x);
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
// program holds fifteen multiplications on average - there is nothing for a second wave to do, and with one wave there is no
// barrier and no waiting for stores between levels.  What a level costs is then the latency of its operands, and the
// program being static, almost all of it is taken off the critical path:
//   * the instruction words are fetched three chunks ahead;
//   * the last 2,048 results wait in an LDS ring (slot = position mod 2,048: the last 32 chunks) - nine operands in ten;
//   * an older operand is loaded from the value array one chunk ahead (its store is tens of chunks old).
// (A witness is cut into several launches of a few milliseconds so that the kernels of the provers that share a hardware queue
// with it are not held up for its whole duration.)
struct WIns { uint32_t code; int32_t a, b; };
__device__ __forceinline__ WIns w_fetch(const WitnessProg& P, uint32_t chunk, uint32_t n_chunks, uint32_t lane) {
  WIns r{(uint32_t)WT_NOP, 0, 0};
  if (chunk < n_chunks) {
    const uint32_t p = chunk * 64 + lane;
    r.code = P.code[p]; r.a = P.a[p]; r.b = P.b[p];
  }
  return r;
}
__device__ __forceinline__ bool w_binary(uint32_t c) { return c == WT_ADD || c == WT_SUB || c == WT_MUL; }
// issue the loads of the operands of `in` that are complete in memory (constants; positions below lim)
__device__ __forceinline__ void w_prefetch(const WitnessProg& P, const uint32_t* __restrict__ vals, const WIns& in, uint32_t lim, uint4 (&px)[3], uint4 (&py)[3]) {
  if (in.code == WT_NOP || in.code == WT_INPUT) return;
  if (in.a < 0 || (uint32_t)in.a < lim) {
    const uint4* q = reinterpret_cast<const uint4*>(in.a < 0 ? P.consts + (size_t)(-1 - in.a) * 12 : vals + (size_t)in.a * 12);
    px[0] = q[0]; px[1] = q[1]; px[2] = q[2];
  }
  if (w_binary(in.code) && (in.b < 0 || (uint32_t)in.b < lim)) {
    const uint4* q = reinterpret_cast<const uint4*>(in.b < 0 ? P.consts + (size_t)(-1 - in.b) * 12 : vals + (size_t)in.b * 12);
    py[0] = q[0]; py[1] = q[1]; py[2] = q[2];
  }
}
__device__ __forceinline__ FrD w_from3(const uint4 (&v)[3]) {
  uint32_t w[12] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w, v[2].x, v[2].y, v[2].z, v[2].w};
  return fp_unpack32<FrParams>(w);
}

constexpr uint32_t WIT_RING = 2048;      // 96 KiB of the CU's 160
__global__ void __launch_bounds__(64) k_witness(WitnessProg P, uint32_t c0, uint32_t c1, const uint64_t* __restrict__ inputs /* batches x n_inputs x 6, ABI */,
                                                 uint32_t* __restrict__ values /* batches x n_pos x 12 */, uint32_t* __restrict__ flags) {
  __shared__ uint4 ring[WIT_RING * 3];
  const uint32_t batch = blockIdx.x, lane = threadIdx.x;
  const uint64_t* in = inputs + (size_t)batch * P.n_inputs * 6;
  uint32_t* vals = values + (size_t)batch * P.n_pos * 12;
  uint32_t bad = 0;
  // one chunk: `cur` with its prefetched operands (cx, cy); `nxt` is the chunk after it, whose operands are requested here
  auto step = [&](uint32_t j, const WIns& cur, uint4 (&cx)[3], uint4 (&cy)[3], const WIns& nxt, uint4 (&nx)[3], uint4 (&ny)[3]) {
    // chunk i finds in the ring what this launch wrote and chunk i itself does not overwrite: positions from ring_lo(i) on
    auto ring_lo = [&](uint32_t i) { const uint32_t w = (i + 1) * 64; return max(c0 * 64, w > WIT_RING ? w - WIT_RING : 0u); };
    w_prefetch(P, vals, nxt, ring_lo(j + 1), nx, ny);
    const uint32_t lim = ring_lo(j);
    const uint32_t c = cur.code;
    if (c != WT_NOP) {
      FrD x = fp_zero<FrParams>(), y = x, r = x;
      if (c != WT_INPUT) {
        if (cur.a < 0 || (uint32_t)cur.a < lim) x = w_from3(cx);
        else { const uint4* q = &ring[((uint32_t)cur.a % WIT_RING) * 3]; uint4 t[3] = {q[0], q[1], q[2]}; x = w_from3(t); }
        if (w_binary(c)) {
          if (cur.b < 0 || (uint32_t)cur.b < lim) y = w_from3(cy);
          else { const uint4* q = &ring[((uint32_t)cur.b % WIT_RING) * 3]; uint4 t[3] = {q[0], q[1], q[2]}; y = w_from3(t); }
        }
      }
      if (w_exec(c, cur.b, x, y, in, cur.a, r)) bad = 1;
      uint32_t w[12];
      fp_pack32<FrParams>(r, w);
      const uint4 v0 = make_uint4(w[0], w[1], w[2], w[3]), v1 = make_uint4(w[4], w[5], w[6], w[7]), v2 = make_uint4(w[8], w[9], w[10], w[11]);
      uint4* g = reinterpret_cast<uint4*>(vals + ((size_t)j * 64 + lane) * 12);
      g[0] = v0; g[1] = v1; g[2] = v2;
      uint4* q = &ring[((j * 64 + lane) % WIT_RING) * 3];
      q[0] = v0; q[1] = v1; q[2] = v2;
    }
    // the ring: this chunk's reads above come before its writes in program order, the next chunk's reads after them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  const uint32_t n_chunks = c1;
  WIns i0 = w_fetch(P, c0, n_chunks, lane), i1 = w_fetch(P, c0 + 1, n_chunks, lane), i2 = w_fetch(P, c0 + 2, n_chunks, lane), i3 = w_fetch(P, c0 + 3, n_chunks, lane);
  uint4 ax[3], ay[3], bx[3], by[3];
  for (int k = 0; k < 3; k++) ax[k] = ay[k] = bx[k] = by[k] = make_uint4(0, 0, 0, 0);
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
  __shared__ uint4 ring[64 * 3];
  const uint32_t batch = blockIdx.x, lane = threadIdx.x;
  const uint64_t* in = inputs + (size_t)batch * P.n_inputs * 6;
  uint32_t* vals = values + (size_t)batch * P.n_pos * 12;
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
        if (ref < 0) return w_load12(P.consts + (size_t)(-1 - ref) * 12);
        if ((uint32_t)ref + 64 > p) {                     // one of the last 64 results
          const uint4* q = &ring[((uint32_t)ref % 64) * 3];
          uint4 x = q[0], y = q[1], z = q[2];
          uint32_t w[12] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w, z.x, z.y, z.z, z.w};
          return fp_unpack32<FrParams>(w);
        }
        return w_load12(vals + (size_t)ref * 12);            // older: its store is long complete (fenced below every 64 instructions)
      };
      FrD x = fp_zero<FrParams>(), y = x, r = x;
      if (c != WT_INPUT) x = load(ra);
      if (c == WT_ADD || c == WT_SUB || c == WT_MUL) y = load(rb);
      if (w_exec(c, rb, x, y, in, ra, r)) bad = 1;
      uint32_t w[12];
      fp_pack32<FrParams>(r, w);
      const uint4 v0 = make_uint4(w[0], w[1], w[2], w[3]), v1 = make_uint4(w[4], w[5], w[6], w[7]), v2 = make_uint4(w[8], w[9], w[10], w[11]);
      if (lane == 0) {
        uint4* g = reinterpret_cast<uint4*>(vals + (size_t)p * 12);
        g[0] = v0; g[1] = v1; g[2] = v2;
        uint4* q = &ring[(p % 64) * 3];
        q[0] = v0; q[1] = v1; q[2] = v2;
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
  const uint32_t* vals = values + (size_t)batch * P.n_pos * 12;
  const int32_t ref = P.out_ref[i];
  FrD v = w_load12(ref >= 0 ? vals + (size_t)ref * 12 : P.consts + (size_t)(-1 - ref) * 12);
  uint64_t w[6];
  fp_to_abi<FrParams>(v, w);
  uint64_t* z = z_out + ((size_t)batch * P.n_vars + i) * 6;
#pragma unroll
  for (int k = 0; k < 6; k++) z[k] = w[k];
}

// ------------------------------------------------------------------------------------------------ host side
struct ProgDev {
  WitnessProg prog;
  void* bufs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};
struct GpuWitnessState {
  WitnessTape tape;
  bool built = false;
  std::map<int, ProgDev> dev;        // per device
};

static void gpu_release(zkhip_aggregator* a) {
  GpuWitnessState* st = (GpuWitnessState*)a->gpu_state;
  if (!st) return;
  for (auto& kv : st->dev) {
    if (hipSetDevice(kv.first) != hipSuccess) continue;
    for (void* p : kv.second.bufs) if (p) (void)hipFree(p);
  }
  delete st;
  a->gpu_state = nullptr;
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
    const WitnessTape& T = st->tape;
    ProgDev pd;
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
    if (e == hipSuccess) e = hipMalloc(&pd.bufs[5], nc * 48 + 48);
    if (e == hipSuccess) e = hipMalloc(&d_c64, nc * 48 + 48);
    if (e == hipSuccess) e = hipMemcpy(d_c64, T.consts.data(), nc * 48, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_witness_consts, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, 0, d_c64, (uint32_t*)pd.bufs[5], (uint32_t)nc);
      e = hipDeviceSynchronize();
    }
    if (d_c64) (void)hipFree(d_c64);
    if (e != hipSuccess) {
      for (void* p : pd.bufs) if (p) (void)hipFree(p);
      snprintf(err, errlen, "witness program upload: %s", hipGetErrorString(e));
      return ZKHIP_ERR_HIP;
    }
    pd.prog = WitnessProg{(const uint8_t*)pd.bufs[0], (const int32_t*)pd.bufs[1], (const int32_t*)pd.bufs[2], (const uint32_t*)pd.bufs[3],
                          (const int32_t*)pd.bufs[4], (const uint32_t*)pd.bufs[5], (uint32_t)(T.level_start.size() - 1), (uint32_t)n,
                          (uint32_t)T.n_vars, (uint32_t)T.n_inputs, T.chain_start};
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
  const uint32_t n_chunks = P.chain_start / 64, seg = 2048;       // chunks per launch: a few milliseconds
  for (uint32_t c0 = 0; c0 < n_chunks; c0 += seg)
    hipLaunchKernelGGL(k_witness, dim3(batches), dim3(64), 0, st, P, c0, (c0 + seg < n_chunks ? c0 + seg : n_chunks), d_inputs, d_values, d_flags);
  (void)hipStreamWaitEvent(st, ev_join, 0);
  hipLaunchKernelGGL(k_witness_out, dim3((P.n_vars + 255) / 256, batches), dim3(256), 0, st, P, d_values, d_z);
}

}  // namespace zkhip
