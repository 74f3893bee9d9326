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

// levels [l0, l1) of the program; the launch that reaches the last level also writes the assignment.  (A witness is cut into
// several launches of a few milliseconds so that the kernels of the provers that share a hardware queue with it are not held up
// for the whole 60 ms.)
// WIT_TPW lanes per batch, WIT_WPB batches per workgroup.  (Measured: eight batches of 128 lanes in one 1,024-lane workgroup - fewer
// CUs shared with the provers' 256-VGPR accumulation waves - lose more to the slower level loop than they win: 114-170 proofs/s
// against 170-185 with one 256-lane workgroup per batch.)
constexpr uint32_t WIT_TPW = 256, WIT_WPB = 1;
__global__ void __launch_bounds__(WIT_TPW * WIT_WPB) k_witness(WitnessProg P, uint32_t l0, uint32_t l1, uint32_t n_batches,
                                                  const uint64_t* __restrict__ inputs /* batches x n_inputs x 6, ABI */,
                                                  uint32_t* __restrict__ values /* batches x n_pos x 12 */,
                                                  uint64_t* __restrict__ z_out /* batches x n_vars x 6, ABI */, uint32_t* __restrict__ flags) {
  const uint32_t tid = threadIdx.x % WIT_TPW;
  uint32_t batch = blockIdx.x * WIT_WPB + threadIdx.x / WIT_TPW;
  const bool active = batch < n_batches;
  if (!active) batch = 0;               // (idle lane groups keep walking the levels for the barrier; they touch nothing)
  const uint64_t* in = inputs + (size_t)batch * P.n_inputs * 6;
  uint32_t* vals = values + (size_t)batch * P.n_pos * 12;
  uint32_t bad = 0;
#pragma unroll 1
  for (uint32_t l = l0; l < l1; l++) {
    const uint32_t p1 = P.level_start[l + 1];
#pragma unroll 1
    for (uint32_t p = P.level_start[l] + (active ? tid : 0x7fffffffu); p < p1; p += WIT_TPW) {
      const uint32_t c = P.code[p];
      if (c == WT_NOP) continue;
      const int32_t ra = P.a[p], rb = P.b[p];
      FrD x = fp_zero<FrParams>(), y = x, r = x;
      if (c != WT_INPUT) x = w_load12(ra >= 0 ? vals + (size_t)ra * 12 : P.consts + (size_t)(-1 - ra) * 12);
      if (c == WT_ADD || c == WT_SUB || c == WT_MUL) y = w_load12(rb >= 0 ? vals + (size_t)rb * 12 : P.consts + (size_t)(-1 - rb) * 12);
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
        case WT_INV0: {
          r = fp_inv<FrParams>(x);
          if (c == WT_INV && fp_is_zero_2p(r)) bad = 1;                            // the host generator would have taken another path
          break;
        }
        default: {                                                                 // WT_BIT
          FrD one_raw = fp_zero<FrParams>();
          one_raw.l[0] = 1;
          uint32_t w[12];
          fp_pack32<FrParams>(fp_cond_sub_p(fp_mul(x, one_raw)), w);                // the canonical integer
          r = ((w[rb >> 5] >> (rb & 31)) & 1u) ? fp_one<FrParams>() : fp_zero<FrParams>();
          break;
        }
      }
      w_store12(vals + (size_t)p * 12, r);
    }
    __syncthreads();
  }
  if (bad) atomicOr(&flags[batch], 1u);
  if (l1 < P.n_levels || !active) return;
  // the assignment, in ABI form
  uint64_t* z = z_out + (size_t)batch * P.n_vars * 6;
  for (uint32_t i = tid; i < P.n_vars; i += WIT_TPW) {
    const int32_t ref = P.out_ref[i];
    FrD v = w_load12(ref >= 0 ? vals + (size_t)ref * 12 : P.consts + (size_t)(-1 - ref) * 12);
    uint64_t w[6];
    fp_to_abi<FrParams>(v, w);
#pragma unroll
    for (int k = 0; k < 6; k++) z[(size_t)i * 6 + k] = w[k];
  }
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
                          (uint32_t)T.n_vars, (uint32_t)T.n_inputs};
    it = st->dev.emplace(device, pd).first;
  }
  *out = it->second.prog;
  if (tape) *tape = &st->tape;
  return ZKHIP_OK;
}

void witness_launch(const WitnessProg& P, const uint64_t* d_inputs, uint32_t* d_values, uint64_t* d_z, uint32_t* d_flags, uint32_t batches, hipStream_t st) {
  const uint32_t seg = 1024;       // levels per launch: ~6 ms
  for (uint32_t l0 = 0; l0 < P.n_levels; l0 += seg)
    hipLaunchKernelGGL(k_witness, dim3((batches + WIT_WPB - 1) / WIT_WPB), dim3(WIT_TPW * WIT_WPB), 0, st, P, l0,
                       (l0 + seg < P.n_levels ? l0 + seg : P.n_levels), batches, d_inputs, d_values, d_z, d_flags);
}

}  // namespace zkhip
