// C ABI (include/zkhip.h) over the HIP engines.  No torch types, plain pointers and sizes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <mutex>

#include "ec.cuh"
#include "host_field.hpp"
#include "msm.h"
#include "ntt.h"

using namespace zkhip;

struct zkhip_bases {
  AffPacked* d_pts;
  size_t len;
};

namespace {
struct Lib {
  bool inited = false;
  int device = -1;
  int forced_c = 0;
  MsmCtx msm;
  bool msm_ready = false;
  char err[512] = {0};
  std::mutex mu;
} g;

int fail(int code, const char* msg) {
  snprintf(g.err, sizeof g.err, "%s", msg);
  return code;
}
#define API_HIP(x)                                                                           \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      snprintf(g.err, sizeof g.err, "%s: %s", #x, hipGetErrorString(e_));                    \
      return ZKHIP_ERR_HIP;                                                                  \
    }                                                                                        \
  } while (0)

int auto_window(size_t n) {
  if (g.forced_c) return g.forced_c;
  if (n <= (1u << 10)) return 8;
  if (n <= (1u << 13)) return 10;
  if (n <= (1u << 16)) return 12;
  if (n <= (1u << 18)) return 14;
  return 16;
}

int ensure_msm(size_t n) {
  int c = auto_window(n);
  if (g.msm_ready && g.msm.max_n >= n && g.msm.c == c) return ZKHIP_OK;
  if (g.msm_ready) { msm_plan_free(&g.msm); g.msm_ready = false; }
  int rc = msm_plan_init(&g.msm, n, c);
  if (rc != ZKHIP_OK) { snprintf(g.err, sizeof g.err, "msm_plan_init: %s", g.msm.errbuf); return rc; }
  g.msm_ready = true;
  return ZKHIP_OK;
}
}  // namespace

extern "C" {

int zkhip_init(int device) {
  std::lock_guard<std::mutex> lk(g.mu);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) return fail(ZKHIP_ERR_NO_DEVICE, "no HIP device (the gfx950 kernels are the only compute path)");
  if (device < 0 || device >= count) return fail(ZKHIP_ERR_ARG, "device index out of range");
  API_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  API_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    snprintf(g.err, sizeof g.err, "device %d is %s, this library contains gfx950 code only", device, prop.gcnArchName);
    return ZKHIP_ERR_NO_DEVICE;
  }
  g.device = device;
  g.inited = true;
  return ZKHIP_OK;
}

void zkhip_shutdown(void) {
  std::lock_guard<std::mutex> lk(g.mu);
  if (g.msm_ready) { msm_plan_free(&g.msm); g.msm_ready = false; }
  g.inited = false;
}

const char* zkhip_strerror(int code) {
  switch (code) {
    case ZKHIP_OK: return "ok";
    case ZKHIP_ERR_ARG: return "bad argument";
    case ZKHIP_ERR_NO_DEVICE: return "no gfx950 device";
    case ZKHIP_ERR_HIP: return "HIP runtime error";
    case ZKHIP_ERR_STATE: return "library not initialised";
    default: return "unknown error";
  }
}
const char* zkhip_last_error(void) { return g.err; }

int zkhip_set_msm_window(int c) {
  if (c != 0 && (c < 4 || c > 18)) return fail(ZKHIP_ERR_ARG, "window must be 0 or in [4, 18]");
  g.forced_c = c;
  return ZKHIP_OK;
}

int zkhip_bases_upload_dev(const void* d_bases_affine, size_t len, zkhip_bases** out) {
  std::lock_guard<std::mutex> lk(g.mu);
  if (!g.inited) return fail(ZKHIP_ERR_STATE, "zkhip_init not called");
  if (!out || (len && !d_bases_affine)) return fail(ZKHIP_ERR_ARG, "null pointer");
  zkhip_bases* b = new zkhip_bases{nullptr, len};
  if (len) {
    API_HIP(hipMalloc(&b->d_pts, len * sizeof(AffPacked)));
    int rc = ensure_msm(len);
    if (rc != ZKHIP_OK) return rc;
    rc = msm_bases_convert(&g.msm, (const uint64_t*)d_bases_affine, len, b->d_pts);
    if (rc != ZKHIP_OK) { snprintf(g.err, sizeof g.err, "%s", g.msm.errbuf); return rc; }
  }
  *out = b;
  return ZKHIP_OK;
}

int zkhip_bases_upload(const uint64_t* bases_affine, size_t len, zkhip_bases** out) {
  if (!g.inited) return fail(ZKHIP_ERR_STATE, "zkhip_init not called");
  if (!out || (len && !bases_affine)) return fail(ZKHIP_ERR_ARG, "null pointer");
  void* d = nullptr;
  if (len) {
    API_HIP(hipMalloc(&d, len * 192));
    API_HIP(hipMemcpy(d, bases_affine, len * 192, hipMemcpyHostToDevice));
  }
  int rc = zkhip_bases_upload_dev(d, len, out);
  if (d) (void)hipFree(d);
  return rc;
}

size_t zkhip_bases_len(const zkhip_bases* b) { return b ? b->len : 0; }

void zkhip_bases_free(zkhip_bases* b) {
  if (!b) return;
  if (b->d_pts) (void)hipFree(b->d_pts);
  delete b;
}

int zkhip_msm_dev(const zkhip_bases* bases, size_t offset, const void* d_scalars, size_t len, int scalars_montgomery,
                  uint64_t out_jac[36]) {
  std::lock_guard<std::mutex> lk(g.mu);
  if (!g.inited) return fail(ZKHIP_ERR_STATE, "zkhip_init not called");
  if (!bases || !out_jac || (len && !d_scalars)) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (offset > bases->len || len > bases->len - offset) return fail(ZKHIP_ERR_ARG, "offset + len exceeds the base set");
  int rc = ensure_msm(len ? len : 1);
  if (rc != ZKHIP_OK) return rc;
  rc = msm_run(&g.msm, bases->d_pts + offset, (const uint64_t*)d_scalars, len, scalars_montgomery, out_jac);
  if (rc != ZKHIP_OK) snprintf(g.err, sizeof g.err, "%s", g.msm.errbuf);
  return rc;
}

int zkhip_msm(const zkhip_bases* bases, size_t offset, const uint64_t* scalars, size_t len, int scalars_montgomery,
              uint64_t out_jac[36]) {
  if (!g.inited) return fail(ZKHIP_ERR_STATE, "zkhip_init not called");
  if (len && !scalars) return fail(ZKHIP_ERR_ARG, "null pointer");
  void* d = nullptr;
  if (len) {
    API_HIP(hipMalloc(&d, len * 48));
    API_HIP(hipMemcpy(d, scalars, len * 48, hipMemcpyHostToDevice));
  }
  int rc = zkhip_msm_dev(bases, offset, d, len, scalars_montgomery, out_jac);
  if (d) (void)hipFree(d);
  return rc;
}

int zkhip_msm_raw(const uint64_t* bases_affine, const uint64_t* scalars, size_t len, int scalars_montgomery,
                  uint64_t out_jac[36]) {
  zkhip_bases* b = nullptr;
  int rc = zkhip_bases_upload(bases_affine, len, &b);
  if (rc != ZKHIP_OK) return rc;
  rc = zkhip_msm(b, 0, scalars, len, scalars_montgomery, out_jac);
  zkhip_bases_free(b);
  return rc;
}

int zkhip_fixed_base_mul_dev(const uint64_t base_affine[24], const void* d_scalars, size_t len, int scalars_montgomery,
                             void* d_out_affine) {
  std::lock_guard<std::mutex> lk(g.mu);
  if (!g.inited) return fail(ZKHIP_ERR_STATE, "zkhip_init not called");
  if (!base_affine || (len && (!d_scalars || !d_out_affine))) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (len == 0) return ZKHIP_OK;
  return fixed_base_mul(base_affine, (const uint64_t*)d_scalars, len, scalars_montgomery, (uint64_t*)d_out_affine, g.err, sizeof g.err);
}

int zkhip_fixed_base_mul(const uint64_t base_affine[24], const uint64_t* scalars, size_t len, int scalars_montgomery,
                         uint64_t* out_affine) {
  if (!g.inited) return fail(ZKHIP_ERR_STATE, "zkhip_init not called");
  if (len && (!scalars || !out_affine)) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (len == 0) return ZKHIP_OK;
  void *ds = nullptr, *dp = nullptr;
  API_HIP(hipMalloc(&ds, len * 48));
  API_HIP(hipMalloc(&dp, len * 192));
  API_HIP(hipMemcpy(ds, scalars, len * 48, hipMemcpyHostToDevice));
  int rc = zkhip_fixed_base_mul_dev(base_affine, ds, len, scalars_montgomery, dp);
  if (rc == ZKHIP_OK) API_HIP(hipMemcpy(out_affine, dp, len * 192, hipMemcpyDeviceToHost));
  (void)hipFree(ds); (void)hipFree(dp);
  return rc;
}

int zkhip_ntt_dev(void* d_data, unsigned log_d, int dir, int coset) {
  std::lock_guard<std::mutex> lk(g.mu);
  if (!g.inited) return fail(ZKHIP_ERR_STATE, "zkhip_init not called");
  if (!d_data) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (log_d > 22) return fail(ZKHIP_ERR_ARG, "log_d must be <= 22");
  return ntt_dev_abi((uint64_t*)d_data, (int)log_d, dir != 0, coset != 0, g.err, sizeof g.err);
}

int zkhip_ntt(uint64_t* data, unsigned log_d, int dir, int coset) {
  if (!g.inited) return fail(ZKHIP_ERR_STATE, "zkhip_init not called");
  if (!data) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (log_d > 22) return fail(ZKHIP_ERR_ARG, "log_d must be <= 22");
  size_t bytes = ((size_t)48) << log_d;
  void* d = nullptr;
  API_HIP(hipMalloc(&d, bytes));
  API_HIP(hipMemcpy(d, data, bytes, hipMemcpyHostToDevice));
  int rc = zkhip_ntt_dev(d, log_d, dir, coset);
  if (rc == ZKHIP_OK) API_HIP(hipMemcpy(data, d, bytes, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return rc;
}

float zkhip_last_accumulate_ms(void) { return g.msm_ready ? g.msm.last_accumulate_ms : 0.f; }

int zkhip_jac_to_affine(const uint64_t jac[36], uint64_t aff[24]) {
  using namespace host;
  if (!jac || !aff) return ZKHIP_ERR_ARG;
  HJac p;
  p.X = HFq::from_limbs(jac); p.Y = HFq::from_limbs(jac + 12); p.Z = HFq::from_limbs(jac + 24);
  HFq x, y;
  p.to_affine(x, y);
  x.to_limbs(aff); y.to_limbs(aff + 12);
  return ZKHIP_OK;
}

int zkhip_jac_add(const uint64_t a[36], const uint64_t b[36], uint64_t out[36]) {
  using namespace host;
  if (!a || !b || !out) return ZKHIP_ERR_ARG;
  HJac p, q;
  p.X = HFq::from_limbs(a); p.Y = HFq::from_limbs(a + 12); p.Z = HFq::from_limbs(a + 24);
  q.X = HFq::from_limbs(b); q.Y = HFq::from_limbs(b + 12); q.Z = HFq::from_limbs(b + 24);
  HJac r = p.add(q);
  r.X.to_limbs(out); r.Y.to_limbs(out + 12); r.Z.to_limbs(out + 24);
  return ZKHIP_OK;
}

}  // extern "C"
