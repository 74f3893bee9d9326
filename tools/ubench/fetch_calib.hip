// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS library's access patterns (the guide calibrates only
// 16 B/lane streaming; "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   k_row4      4 B per lane, limb-major rows through a buffer descriptor (the accumulators' Y / slot traffic): 1 GiB read
//   k_gather16  16 B per lane x 12 = one 192-byte packed affine point per lane at a scattered index: 768 MiB read
//   k_store4    4 B per lane buffer stores, limb-major rows: 1 GiB written
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ; run under rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __amdgpu_buffer_rsrc_t rsrc_t;

__global__ void __launch_bounds__(256) k_row4(uint32_t* base, uint32_t stride, uint32_t* sink) {
  uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(64u * stride * 4u), 0x00020000);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 64; k++) acc += __builtin_amdgcn_raw_buffer_load_b32(rs, gid * 4u, (uint32_t)k * stride * 4u, 0);
  if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void __launch_bounds__(256) k_store4(uint32_t* base, uint32_t stride) {
  uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(64u * stride * 4u), 0x00020000);
#pragma unroll
  for (int k = 0; k < 64; k++) __builtin_amdgcn_raw_buffer_store_b32(gid + k, rs, gid * 4u, (uint32_t)k * stride * 4u, 0);
}

struct Rec { uint4 w[12]; };
__global__ void __launch_bounds__(256) k_gather16(const Rec* recs, uint32_t n_mask, uint32_t* sink) {
  uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t idx = (gid * 2654435761u) & n_mask;       // odd multiplier: a permutation of [0, 2^k)
  const Rec* r = &recs[idx];
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 12; k++) { uint4 v = r->w[k]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const uint32_t stride = 1u << 22;                   // 4M lanes x 64 rows x 4 B = 1 GiB
  uint32_t *buf = nullptr, *sink = nullptr;
  Rec* recs = nullptr;
  CK(hipMalloc(&buf, (size_t)64 * stride * 4));
  CK(hipMalloc(&sink, 64));
  const uint32_t nrec = 1u << 22;                     // 4M records x 192 B = 768 MiB
  CK(hipMalloc(&recs, (size_t)nrec * sizeof(Rec)));
  CK(hipMemset(buf, 1, (size_t)64 * stride * 4));
  CK(hipMemset(recs, 1, (size_t)nrec * sizeof(Rec)));
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_row4, dim3(stride / 256), dim3(256), 0, 0, buf, stride, sink);
    hipLaunchKernelGGL(k_gather16, dim3(nrec / 256), dim3(256), 0, 0, recs, nrec - 1, sink);
    hipLaunchKernelGGL(k_store4, dim3(stride / 256), dim3(256), 0, 0, buf, stride);
    CK(hipDeviceSynchronize());
  }
  printf("k_row4 reads %zu bytes, k_gather16 reads %zu bytes, k_store4 writes %zu bytes per launch\n", (size_t)64 * stride * 4,
         (size_t)nrec * sizeof(Rec), (size_t)64 * stride * 4);
  return 0;
}
