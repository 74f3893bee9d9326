// C ABI (include/zkhip.h) over the HIP engines.  No torch types, plain pointers and sizes.
#include <math.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <memory>
#include <mutex>

#include "ec.cuh"
#include "host_field.hpp"
#include "domain.hpp"
#include "msm.h"
#include "wait.h"
#include "ntt.h"
#include "qap.h"
#include "pairing_host.hpp"
#include "witness.h"
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <thread>
#include <vector>
#include <pthread.h>

using namespace zkhip;

struct zkhip_bases {
  AffPacked* d_pts;  // len points; after zkhip_bases_precompute: table_c > 0 and levels x len points (level w = 2^(table_c w) P)
  uint8_t* d_inf;    // 1 where the base (or its table level) is the point at infinity; same shape as d_pts
  size_t len;
  int table_c;       // 0: plain base set
  int table_naf;     // 1: the table holds EVERY bit position (378 levels): scalars are recoded in non-adjacent form (msm.h, merged == 2)
  size_t n_finite;   // bases that are not the point at infinity (counted at upload)
  int device;        // the GPU that holds them
  int plain_c = 0;   // plain base set: window of the MSMs over it (zkhip_bases_set_window; 0: by the number of terms)
};

struct zkhip_r1cs {
  R1csDev* dev;
  int device;
};

struct TailTables { host::FixedBase8 d1, d2; };      // fixed-base tables of delta_1 and delta_2 for the prover's tail
struct zkhip_crs {
  size_t n_vars, n_primary, domain_size;
  zkhip_bases *A, *B2, *B1, *H, *L;
  uint64_t alpha_g1[24], beta_g1[24], beta_g2[24], delta_g1[24], delta_g2[24];
  int device;
  int batch_msms = 1;     // the five MSMs of a proof in one launch sequence (zkhip_key_opts; the key carries its own choice)
  mutable std::once_flag tail_once;               // built by the key's first proof (~0.2 s of host time)
  mutable std::unique_ptr<TailTables> tail;
};

struct zkhip_keypair {
  size_t n_vars, n_primary, domain_size;
  std::vector<uint64_t> alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2, A, B2, B1, H, L, ABC;
};

namespace {
// Everything one proof in flight needs on the device: a stream for the QAP map, the witness buffer, five MSM contexts
// (the prover keeps 2 (large) or 5 (small circuits) MSMs in flight).  The library owns one (the plain entry points,
// serialised by g.mu); every zkhip_prover owns another, so several host threads can keep several proofs in flight.
constexpr int ZK_MSM_SLOTS = 8;
constexpr int ZK_CTX_Z = ZK_MSM_SLOTS + 1, ZK_CTX_H = ZK_MSM_SLOTS + 2, ZK_CTX_TOTAL = ZK_MSM_SLOTS + 3;
struct ProveState {
  // MSM contexts: [0, ZK_MSM_SLOTS) the slots of zkhip_msm_submit / collect (the first five also serve a proof whose five MSMs run as
  // separate launch sequences), [ZK_MSM_SLOTS] the context of a proof's five MSMs in ONE launch sequence, [ZK_CTX_Z] / [ZK_CTX_H] the
  // two sequences of a proof ALONE (round 6): the four MSMs over the assignment, and the H MSM behind the QAP map
  MsmCtx ctx[ZK_CTX_TOTAL];
  bool ready[ZK_CTX_TOTAL] = {};
  hipStream_t st = nullptr;
  hipEvent_t ev_st = nullptr;      // blocking-sync event for waits on st (the waiting host thread sleeps)
  hipEvent_t ev_up = nullptr, ev_qap = nullptr;   // split proofs: the assignment is on the device / the QAP map has finished (stream-to-stream)
  bool split_last = false;         // the last proof ran as two launch sequences (A, B-G2, B-G1, L beside the QAP map; then H)
  MsmCtx* last_acc_ctx2 = nullptr; // ... whose second accumulation launch ran on this plan
  uint64_t* dz = nullptr;
  size_t dz_cap = 0;
  double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool chained_last = false;       // the last proof put upload, QAP map and MSMs on ONE stream without host waits: ms[0..1] are enqueueing times
  float last_accumulate_ms = 0.f;
  MsmCtx* last_acc_ctx = nullptr;  // the plan last_accumulate_ms was read from (zkhip_last_accumulate_entries)
  float last_acc_interval[2] = {0.f, 0.f};   // begin / end of that launch on the device's time base
  int last_submit_slot = -1;       // zkhip_msm_submit: the slot of the previous submission (its accumulation gates the next one's)
  uint32_t quad_below = 0;         // 0: the engine's default; else the MSM contexts' quad_below (zkhip_prover_set_streaming)
  hipStream_t pre[2] = {nullptr, nullptr};   // streams made ahead of the first proof (zkhip_prover_create_streams): the launch sequence's plan adopts them
  void release() {
    for (int k = 0; k < 2; k++) if (pre[k]) { (void)hipStreamDestroy(pre[k]); pre[k] = nullptr; }
    for (int k = 0; k < ZK_CTX_TOTAL; k++) if (ready[k]) { msm_plan_free(&ctx[k]); ready[k] = false; }
    if (st) { (void)hipStreamDestroy(st); st = nullptr; }
    if (ev_st) { (void)hipEventDestroy(ev_st); ev_st = nullptr; }
    if (ev_up) { (void)hipEventDestroy(ev_up); ev_up = nullptr; }
    if (ev_qap) { (void)hipEventDestroy(ev_qap); ev_qap = nullptr; }
    last_acc_ctx = last_acc_ctx2 = nullptr;
    if (dz) { (void)hipFree(dz); dz = nullptr; dz_cap = 0; }
  }
};

// One of these per GPU the process has initialised (zkhip_init(device), once per device).  HIP's current device is a
// property of the calling HOST THREAD, so every entry point that touches the device binds its thread first: to the device of
// the handle it is given (bases, key, constraint system, prover), or - for the entry points without a handle - to the
// thread's current library device (zkhip_set_device; defaults to the first initialised device).
constexpr int ZK_MAX_DEVICES = 16;
struct DevState {
  bool inited = false;
  ProveState ps;            // work space of the plain (handle-less / library-serialised) entry points on this device
  std::mutex mu;            // serialises them
};
struct Lib {
  DevState dev[ZK_MAX_DEVICES];
  int default_device = -1;
  // process-wide DEFAULTS of the key options (deprecated setters zkhip_set_*; a key's own options travel in zkhip_key_opts and
  // are resolved once at upload): atomics, so that a setter racing an upload is at least a clean read of one value or the other
  std::atomic<int> forced_c{0};
  std::atomic<int> crs_tables{1};       // zkhip_crs_upload builds window tables (zkhip_set_crs_precompute)
  std::atomic<int> batch_msms{1};       // table-backed keys: the five MSMs of a proof in one launch sequence
  std::mutex mu;            // guards inited / default_device
} g;
thread_local char t_err[512] = {0};   // zkhip_last_error(): the calling thread's last failure
thread_local int t_dev = -1;          // this thread's library device (-1: the default device); changed by zkhip_init / zkhip_set_device ONLY
thread_local int t_slot_dev[8] = {-1, -1, -1, -1, -1, -1, -1, -1};   // device of this thread's last zkhip_msm_submit per slot (zkhip_msm_collect has no handle)
thread_local int t_prove_dev = -1;    // device of this thread's last MSM or proof through a handle (zkhip_last_prove_timings / _accumulate_ms)

int fail(int code, const char* msg) {
  snprintf(t_err, sizeof t_err, "%s", msg);
  return code;
}
int cur_dev() { return t_dev >= 0 ? t_dev : g.default_device; }
// bind the calling thread to device d (it must have been initialised)
int bind_dev(int d) {
  if (d < 0 || d >= ZK_MAX_DEVICES || !g.dev[d].inited) return fail(ZKHIP_ERR_STATE, "zkhip_init not called (for this device)");
  hipError_t e = hipSetDevice(d);
  if (e != hipSuccess) { snprintf(t_err, sizeof t_err, "hipSetDevice(%d): %s", d, hipGetErrorString(e)); return ZKHIP_ERR_HIP; }
  return ZKHIP_OK;
}
#define BIND_CUR()  do { int rc_ = bind_dev(cur_dev()); if (rc_ != ZKHIP_OK) return rc_; } while (0)
#define BIND(h)     do { int rc_ = bind_dev((h)->device); if (rc_ != ZKHIP_OK) return rc_; } while (0)
// frees device / host allocations of an entry point on every exit path
struct Scratch {
  std::vector<void*> dev;
  ~Scratch() { for (void* p : dev) if (p) (void)hipFree(p); }
  hipError_t alloc(void** out, size_t bytes) { hipError_t e = hipMalloc(out, bytes ? bytes : 1); if (e == hipSuccess) dev.push_back(*out); return e; }
};
#define API_HIP(x)                                                                           \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      snprintf(t_err, sizeof t_err, "%s: %s", #x, hipGetErrorString(e_));                    \
      return ZKHIP_ERR_HIP;                                                                  \
    }                                                                                        \
  } while (0)

int auto_window(size_t n) {
  if (g.forced_c.load()) return g.forced_c.load();
  if (n <= (1u << 10)) return 8;
  if (n <= (1u << 13)) return 10;
  if (n <= (1u << 16)) return 12;
  if (n <= (1u << 18)) return 14;
  return 16;
}

// window size of a table (one shared bucket window: the reduction is W times cheaper, so c is larger than auto_window's)
int auto_table_window(size_t n) {
  if (n <= (1u << 10)) return 9;
  if (n <= (1u << 13)) return 12;
  if (n <= (1u << 15)) return 14;
  if (n <= (1u << 17)) return 16;
  if (n <= (1u << 19)) return 18;
  if (n <= (1u << 21)) return 20;
  return 21;
}

// table_c = 0: plain plan with the automatic window;  > 0: merged plan (bases are a window table built for table_c)
// total: bound on the terms of all K jobs of a launch together (0: K * n)
// would ensure_ctx keep this context as it is (same plan, nothing pending)?
static bool ctx_reusable(const MsmCtx* cx, bool ready, size_t n, int table_c, int K, int naf, size_t total, int plain_c = 0) {
  const int c = table_c ? table_c : (plain_c ? plain_c : auto_window(n)), merged = table_c ? (naf ? 2 : 1) : 0;
  if (total == 0 || total > (size_t)K * n) total = (size_t)K * n;
  return ready && !cx->pending && cx->max_n >= n && cx->total_terms >= total && cx->c == c && cx->merged == merged && cx->K == K &&
         cx->aff_forced == msm_forced_aff_levels();
}
int ensure_ctx(MsmCtx* cx, bool* ready, size_t n, int table_c, int K = 1, int naf = 0, size_t total = 0, int plain_c = 0, hipStream_t* adopt = nullptr) {
  const int c = table_c ? table_c : (plain_c ? plain_c : auto_window(n)), merged = table_c ? (naf ? 2 : 1) : 0;
  if (total == 0 || total > (size_t)K * n) total = (size_t)K * n;
  if (*ready && cx->pending) return fail(ZKHIP_ERR_STATE, "an MSM submitted on this context has not been collected (zkhip_msm_collect)");
  if (*ready && cx->max_n >= n && cx->total_terms >= total && cx->c == c && cx->merged == merged && cx->K == K && cx->aff_forced == msm_forced_aff_levels()) return ZKHIP_OK;
  if (*ready) { msm_plan_free(cx); *ready = false; }
  int rc = msm_plan_init(cx, n, c, merged, K, total, adopt);
  if (rc != ZKHIP_OK) {
    snprintf(t_err, sizeof t_err, "msm_plan_init: %s", cx->errbuf);
    msm_plan_free(cx);          // whatever was allocated before the failure
    return rc;
  }
  *ready = true;
  return ZKHIP_OK;
}
}  // namespace

extern "C" {

int zkhip_init(int device) {
  std::lock_guard<std::mutex> lk(g.mu);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) return fail(ZKHIP_ERR_NO_DEVICE, "no HIP device (the gfx950 kernels are the only compute path)");
  if (device < 0 || device >= count || device >= ZK_MAX_DEVICES) return fail(ZKHIP_ERR_ARG, "device index out of range");
  API_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  API_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    snprintf(t_err, sizeof t_err, "device %d is %s, this library contains gfx950 code only", device, prop.gcnArchName);
    return ZKHIP_ERR_NO_DEVICE;
  }
  g.dev[device].inited = true;
  if (g.default_device < 0) g.default_device = device;
  t_dev = device;
  return ZKHIP_OK;
}

int zkhip_set_device(int device) {
  int rc = bind_dev(device);
  if (rc == ZKHIP_OK) t_dev = device;
  return rc;
}
int zkhip_get_device(void) { return cur_dev(); }

void zkhip_shutdown(void) {
  std::lock_guard<std::mutex> lk(g.mu);
  for (int d = 0; d < ZK_MAX_DEVICES; d++) {
    if (!g.dev[d].inited) continue;
    std::lock_guard<std::mutex> lkd(g.dev[d].mu);
    if (hipSetDevice(d) == hipSuccess) g.dev[d].ps.release();
    g.dev[d].inited = false;
  }
  g.default_device = -1;
  t_dev = -1;
}

const char* zkhip_strerror(int code) {
  switch (code) {
    case ZKHIP_OK: return "ok";
    case ZKHIP_ERR_ARG: return "bad argument";
    case ZKHIP_ERR_NO_DEVICE: return "no gfx950 device";
    case ZKHIP_ERR_HIP: return "HIP runtime error";
    case ZKHIP_ERR_STATE: return "library not initialised";
    case ZKHIP_ERR_NO_TICKET: return "no such ticket";
    default: return "unknown error";
  }
}
const char* zkhip_last_error(void) { return t_err; }

int zkhip_set_msm_window(int c) {
  if (c != 0 && (c < 4 || c > 18)) return fail(ZKHIP_ERR_ARG, "window must be 0 or in [4, 18]");   // (tables: zkhip_bases_precompute takes up to 22)
  g.forced_c.store(c);
  return ZKHIP_OK;
}

// device memory for callers without a HIP runtime of their own (the *_dev entry points take such pointers)
int zkhip_device_alloc(size_t bytes, void** out) {
  BIND_CUR();
  if (!out) return fail(ZKHIP_ERR_ARG, "null pointer");
  API_HIP(hipMalloc(out, bytes ? bytes : 1));
  return ZKHIP_OK;
}
int zkhip_device_free(void* p) {
  if (p) API_HIP(hipFree(p));
  return ZKHIP_OK;
}
int zkhip_device_copy_in(void* dst, const void* src, size_t bytes) {
  BIND_CUR();
  if (bytes && (!dst || !src)) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (bytes) {
    API_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    API_HIP(hipStreamSynchronize(0));
  }
  return ZKHIP_OK;
}

int zkhip_device_copy_out(void* dst_host, const void* src_device, size_t bytes) {
  BIND_CUR();
  if (bytes && (!dst_host || !src_device)) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (bytes) API_HIP(hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost));
  return ZKHIP_OK;
}

static int bases_upload_dev_impl(const void* d_bases_affine, size_t len, zkhip_bases* b) {
  if (!len) return ZKHIP_OK;
  API_HIP(hipMalloc(&b->d_pts, len * sizeof(AffPacked)));
  API_HIP(hipMalloc(&b->d_inf, len));
  int rc = msm_bases_convert((const uint64_t*)d_bases_affine, len, b->d_pts, b->d_inf, t_err, sizeof t_err);
  if (rc != ZKHIP_OK) return rc;
  std::vector<uint8_t> flags(len);
  API_HIP(hipMemcpy(flags.data(), b->d_inf, len, hipMemcpyDeviceToHost));
  size_t inf = 0;
  for (uint8_t f : flags) inf += f;
  b->n_finite = len - inf;
  return ZKHIP_OK;
}

int zkhip_bases_upload_dev(const void* d_bases_affine, size_t len, zkhip_bases** out) {
  BIND_CUR();
  if (!out || (len && !d_bases_affine)) return fail(ZKHIP_ERR_ARG, "null pointer");
  const int dev = cur_dev();
  std::lock_guard<std::mutex> lk(g.dev[dev].mu);
  zkhip_bases* b = new zkhip_bases{nullptr, nullptr, len, 0, 0, len, dev, 0};
  int rc = bases_upload_dev_impl(d_bases_affine, len, b);
  if (rc != ZKHIP_OK) { zkhip_bases_free(b); return rc; }      // whatever was allocated before the failure
  *out = b;
  return ZKHIP_OK;
}

int zkhip_bases_upload(const uint64_t* bases_affine, size_t len, zkhip_bases** out) {
  BIND_CUR();
  if (!out || (len && !bases_affine)) return fail(ZKHIP_ERR_ARG, "null pointer");
  Scratch sc;
  void* d = nullptr;
  if (len) {
    API_HIP(sc.alloc(&d, len * 192));
    API_HIP(hipMemcpy(d, bases_affine, len * 192, hipMemcpyHostToDevice));
  }
  return zkhip_bases_upload_dev(d, len, out);
}

size_t zkhip_bases_len(const zkhip_bases* b) { return b ? b->len : 0; }

int zkhip_set_affine_levels(int levels) {
  if (levels < -1 || levels > MSM_MAX_AFF_LEVELS) return fail(ZKHIP_ERR_ARG, "levels must be -1 (automatic) or in [0, 4]");
  msm_force_aff_levels(levels);
  return ZKHIP_OK;
}
int zkhip_set_crs_precompute(int on) { g.crs_tables.store(on ? 1 : 0); return ZKHIP_OK; }
static std::atomic<int> g_table_naf{-1};       // -1: the environment decides (default off); see naf_tables_wanted
int zkhip_set_table_naf(int on) { g_table_naf.store(on < 0 ? -1 : (on ? 1 : 0)); return ZKHIP_OK; }
int zkhip_set_batch_msms(int on) { g.batch_msms.store(on ? 1 : 0); return ZKHIP_OK; }

// Which kind of table: one level per window (default), or every bit position (378 levels) with the scalars recoded in width-(c+1)
// non-adjacent form - an eighth fewer additions per scalar over the same buckets, sixteen times the table.  MEASURED (DESIGN.md
// section 5): the wrapping key (17.6 GB of tables instead of 1.2) gains 4.5 % in a stream of proofs and loses 5 % alone - its
// accumulation launch shrinks by 2 %, not 12: every addition gathers its point from a table that no longer fits the TLB's reach;
// a 2^20-point set (76 GB) LOSES 24 %.  So it is an option (zkhip_set_table_naf / ZKHIP_TABLE_NAF=1, within ZKHIP_NAF_TABLE_GB,
// default 48 GB per base set or proving key), off by default in the library (the streaming bench and the gRPC server
// switch it on for their key), tested like the default.
static bool naf_tables_fit(size_t total_points) {
  static const double cap_gb = [] { const char* e = getenv("ZKHIP_NAF_TABLE_GB"); double v = e ? atof(e) : 48.0; return v > 0 ? v : 48.0; }();
  return (double)total_points * 378.0 * (double)(sizeof(AffPacked) + 1) <= cap_gb * 1e9 && total_points * 378 < ((size_t)1 << 31);
}
static bool naf_tables_wanted(size_t total_points) {
  static const int env_on = [] { const char* e = getenv("ZKHIP_TABLE_NAF"); return e ? atoi(e) : 0; }();
  const int cur = g_table_naf.load();
  const int on = cur >= 0 ? cur : env_on;
  return on && naf_tables_fit(total_points);
}
static int bases_precompute_mode(zkhip_bases* b, int c, int naf);
int zkhip_bases_precompute(zkhip_bases* b, int c) {
  if (!b) return fail(ZKHIP_ERR_ARG, "null pointer");
  return bases_precompute_mode(b, c, naf_tables_wanted(b->len) ? 1 : 0);
}
int zkhip_bases_precompute_ex(zkhip_bases* b, int c, int table_naf) {
  if (!b) return fail(ZKHIP_ERR_ARG, "null pointer");
  const int naf = table_naf < 0 ? (naf_tables_wanted(b->len) ? 1 : 0) : ((table_naf && naf_tables_fit(b->len)) ? 1 : 0);
  return bases_precompute_mode(b, c, naf);
}
static int bases_precompute_mode(zkhip_bases* b, int c, int naf) {
  if (!b) return fail(ZKHIP_ERR_ARG, "null pointer");
  BIND(b);
  std::lock_guard<std::mutex> lk(g.dev[b->device].mu);
  if (b->table_c) return fail(ZKHIP_ERR_STATE, "base set already has a window table");
  if (c == 0) c = auto_table_window(b->len);
  if (c < 4 || c > 22) return fail(ZKHIP_ERR_ARG, "table window must be 0 (automatic) or in [4, 22]");
  if (b->len == 0) { b->table_c = c; b->table_naf = naf; return ZKHIP_OK; }
  const size_t levels = (size_t)msm_table_levels(c, naf);
  if (levels * b->len >= ((size_t)1 << 31)) return fail(ZKHIP_ERR_ARG, "table too large (levels * len must stay below 2^31)");
  AffPacked* tab = nullptr;
  uint8_t* tinf = nullptr;
  hipError_t e = hipMalloc(&tab, levels * b->len * sizeof(AffPacked));
  if (e == hipSuccess) e = hipMalloc(&tinf, levels * b->len);
  if (e == hipSuccess) e = hipMemcpy(tab, b->d_pts, b->len * sizeof(AffPacked), hipMemcpyDeviceToDevice);
  if (e == hipSuccess) e = hipMemcpy(tinf, b->d_inf, b->len, hipMemcpyDeviceToDevice);
  int rc = ZKHIP_OK;
  if (e != hipSuccess) { snprintf(t_err, sizeof t_err, "window table allocation: %s", hipGetErrorString(e)); rc = ZKHIP_ERR_HIP; }
  else rc = msm_table_build(tab, tinf, b->len, c, naf, t_err, sizeof t_err);
  if (rc != ZKHIP_OK) { if (tab) (void)hipFree(tab); if (tinf) (void)hipFree(tinf); return rc; }
  (void)hipFree(b->d_pts); (void)hipFree(b->d_inf);
  b->d_pts = tab; b->d_inf = tinf; b->table_c = c; b->table_naf = naf;
  return ZKHIP_OK;
}
int zkhip_bases_table_window(const zkhip_bases* b) { return b ? b->table_c : 0; }
int zkhip_bases_set_window(zkhip_bases* b, int c) {
  if (!b) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (c != 0 && (c < 4 || c > 18)) return fail(ZKHIP_ERR_ARG, "window must be 0 (automatic) or in [4, 18]");
  std::lock_guard<std::mutex> lk(g.dev[b->device].mu);
  b->plain_c = c;
  return ZKHIP_OK;
}

void zkhip_bases_free(zkhip_bases* b) {
  if (!b) return;
  (void)bind_dev(b->device);
  if (b->d_pts) (void)hipFree(b->d_pts);
  if (b->d_inf) (void)hipFree(b->d_inf);
  delete b;
}

int zkhip_msm_dev(const zkhip_bases* bases, size_t offset, const void* d_scalars, size_t len, int scalars_montgomery,
                  uint64_t out_jac[36]) {
  if (!bases || !out_jac || (len && !d_scalars)) return fail(ZKHIP_ERR_ARG, "null pointer");
  BIND(bases);
  ProveState& ps = g.dev[bases->device].ps;
  std::lock_guard<std::mutex> lk(g.dev[bases->device].mu);
  if (offset > bases->len || len > bases->len - offset) return fail(ZKHIP_ERR_ARG, "offset + len exceeds the base set");
  int rc = ensure_ctx(&ps.ctx[0], &ps.ready[0], len ? len : 1, bases->table_c, 1, bases->table_naf, 0, bases->plain_c);
  if (rc != ZKHIP_OK) return rc;
  rc = msm_run(&ps.ctx[0], bases->d_pts + offset, bases->d_inf ? bases->d_inf + offset : nullptr, (const uint64_t*)d_scalars, len,
               scalars_montgomery, bases->len, out_jac);
  if (rc != ZKHIP_OK) snprintf(t_err, sizeof t_err, "%s", ps.ctx[0].errbuf);
  else { ps.last_acc_ctx = &ps.ctx[0]; ps.last_accumulate_ms = ps.ctx[0].last_accumulate_ms; ps.last_acc_interval[0] = ps.ctx[0].last_acc_begin_ms; ps.last_acc_interval[1] = ps.ctx[0].last_acc_end_ms; t_prove_dev = bases->device; }
  return rc;
}

// Asynchronous form: enqueue on one of the library's MSM contexts and return; collect later.  Two MSMs in flight overlap
// the latency-bound bucket reduction of one with the accumulation of the other (what the prover does between its own MSMs).
int zkhip_msm_submit(const zkhip_bases* bases, size_t offset, const void* d_scalars, size_t len, int scalars_montgomery, int slot) {
  if (!bases || (len && !d_scalars)) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (slot < 0 || slot >= ZK_MSM_SLOTS) return fail(ZKHIP_ERR_ARG, "slot must be in [0, 7]");
  BIND(bases);
  ProveState& ps = g.dev[bases->device].ps;
  std::lock_guard<std::mutex> lk(g.dev[bases->device].mu);
  if (offset > bases->len || len > bases->len - offset) return fail(ZKHIP_ERR_ARG, "offset + len exceeds the base set");
  MsmCtx* cx = &ps.ctx[slot];
  if (ps.ready[slot] && cx->pending) return fail(ZKHIP_ERR_STATE, "slot busy: collect its result first");
  int rc = ensure_ctx(cx, &ps.ready[slot], len ? len : 1, bases->table_c, 1, bases->table_naf, 0, bases->plain_c);
  if (rc != ZKHIP_OK) return rc;
  // a stream of MSMs, optionally GATED (ZKHIP_MSM_GATE=1): the accumulation of this one waits for the end of the accumulation
  // submitted before it on another slot, so that two accumulations never share the chip.  Measured (tools/gate_ab.sh, 2^20 terms,
  // eight in flight): 79.6 Mscalar/s gated against 83.7 free-running - the free overlap fills the tail of one accumulation with the
  // head of the next - so the default is off; the gate gives event-timed kernel durations that are per-launch costs.
  static const bool gate = getenv("ZKHIP_MSM_GATE") ? atoi(getenv("ZKHIP_MSM_GATE")) != 0 : false;
  const int prev = ps.last_submit_slot;
  cx->acc_gate = (gate && prev >= 0 && prev != slot && ps.ready[prev]) ? ps.ctx[prev].ev_acc1 : nullptr;
  rc = msm_launch(cx, bases->d_pts + offset, bases->d_inf ? bases->d_inf + offset : nullptr, (const uint64_t*)d_scalars, len,
                  scalars_montgomery, bases->len);
  cx->acc_gate = nullptr;                   // (the event belongs to another context: never kept beyond this launch)
  if (rc != ZKHIP_OK) snprintf(t_err, sizeof t_err, "%s", cx->errbuf);
  else if (len) ps.last_submit_slot = slot;
  if (rc == ZKHIP_OK) t_slot_dev[slot] = bases->device;      // zkhip_msm_collect(slot) has no handle: it collects where this thread submitted
  return rc;
}

int zkhip_msm_collect(int slot, uint64_t out_jac[36]) {
  if (slot < 0 || slot >= ZK_MSM_SLOTS || !out_jac) return fail(ZKHIP_ERR_ARG, "bad slot or null pointer");
  const int dev = t_slot_dev[slot] >= 0 ? t_slot_dev[slot] : cur_dev();     // the device this thread submitted the slot on
  { int rc_ = bind_dev(dev); if (rc_ != ZKHIP_OK) return rc_; }
  ProveState& ps = g.dev[dev].ps;
  std::lock_guard<std::mutex> lk(g.dev[dev].mu);
  MsmCtx* cx = &ps.ctx[slot];
  if (!ps.ready[slot] || !cx->pending) return fail(ZKHIP_ERR_STATE, "nothing submitted on this slot");
  int rc = msm_finish(cx, out_jac);
  if (rc != ZKHIP_OK) snprintf(t_err, sizeof t_err, "%s", cx->errbuf);
  else { ps.last_acc_ctx = cx; ps.last_accumulate_ms = cx->last_accumulate_ms; ps.last_acc_interval[0] = cx->last_acc_begin_ms; ps.last_acc_interval[1] = cx->last_acc_end_ms; t_prove_dev = dev; }
  return rc;
}

int zkhip_msm(const zkhip_bases* bases, size_t offset, const uint64_t* scalars, size_t len, int scalars_montgomery,
              uint64_t out_jac[36]) {
  if (!bases || (len && !scalars)) return fail(ZKHIP_ERR_ARG, "null pointer");
  BIND(bases);
  Scratch sc;
  void* d = nullptr;
  if (len) {
    API_HIP(sc.alloc(&d, len * 48));
    API_HIP(hipMemcpy(d, scalars, len * 48, hipMemcpyHostToDevice));
    API_HIP(hipStreamSynchronize(0));     // the MSM streams are not ordered against the null stream
  }
  return zkhip_msm_dev(bases, offset, d, len, scalars_montgomery, out_jac);
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
  BIND_CUR();
  std::lock_guard<std::mutex> lk(g.dev[cur_dev()].mu);
  if (!base_affine || (len && (!d_scalars || !d_out_affine))) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (len == 0) return ZKHIP_OK;
  return fixed_base_mul(base_affine, (const uint64_t*)d_scalars, len, scalars_montgomery, (uint64_t*)d_out_affine, t_err, sizeof t_err);
}

int zkhip_fixed_base_mul(const uint64_t base_affine[24], const uint64_t* scalars, size_t len, int scalars_montgomery,
                         uint64_t* out_affine) {
  BIND_CUR();
  if (len && (!scalars || !out_affine)) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (len == 0) return ZKHIP_OK;
  Scratch sc;
  void *ds = nullptr, *dp = nullptr;
  API_HIP(sc.alloc(&ds, len * 48));
  API_HIP(sc.alloc(&dp, len * 192));
  API_HIP(hipMemcpy(ds, scalars, len * 48, hipMemcpyHostToDevice));
  int rc = zkhip_fixed_base_mul_dev(base_affine, ds, len, scalars_montgomery, dp);
  if (rc == ZKHIP_OK) API_HIP(hipMemcpy(out_affine, dp, len * 192, hipMemcpyDeviceToHost));
  return rc;
}

int zkhip_ntt_dev(void* d_data, unsigned log_d, int dir, int coset) {
  BIND_CUR();
  std::lock_guard<std::mutex> lk(g.dev[cur_dev()].mu);
  if (!d_data) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (log_d > 22) return fail(ZKHIP_ERR_ARG, "log_d must be <= 22");
  return ntt_dev_abi((uint64_t*)d_data, (int)log_d, dir != 0, coset != 0, t_err, sizeof t_err);
}

int zkhip_ntt(uint64_t* data, unsigned log_d, int dir, int coset) {
  BIND_CUR();
  if (!data) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (log_d > 22) return fail(ZKHIP_ERR_ARG, "log_d must be <= 22");
  size_t bytes = ((size_t)48) << log_d;
  Scratch sc;
  void* d = nullptr;
  API_HIP(sc.alloc(&d, bytes));
  API_HIP(hipMemcpy(d, data, bytes, hipMemcpyHostToDevice));
  int rc = zkhip_ntt_dev(d, log_d, dir, coset);
  if (rc == ZKHIP_OK) API_HIP(hipMemcpy(data, d, bytes, hipMemcpyDeviceToHost));
  return rc;
}

int zkhip_r1cs_upload_ex(const zkhip_r1cs_desc* d, size_t domain_size, zkhip_r1cs** out) {
  BIND_CUR();
  std::lock_guard<std::mutex> lk(g.dev[cur_dev()].mu);
  if (!d || !out) return fail(ZKHIP_ERR_ARG, "null pointer");
  R1csDev* dev = nullptr;
  int rc = r1cs_upload(d, domain_size, &dev, t_err, sizeof t_err);
  if (rc != ZKHIP_OK) return rc;
  *out = new zkhip_r1cs{dev, cur_dev()};
  return ZKHIP_OK;
}
int zkhip_r1cs_upload(const zkhip_r1cs_desc* d, zkhip_r1cs** out) { return zkhip_r1cs_upload_ex(d, 0, out); }   // the reference's forced power of two

int zkhip_r1cs_set_domain(zkhip_r1cs* r, size_t domain_size) {
  if (!r) return fail(ZKHIP_ERR_ARG, "null pointer");
  BIND(r);
  std::lock_guard<std::mutex> lk(g.dev[r->device].mu);
  return r1cs_set_domain(r->dev, domain_size, t_err, sizeof t_err);
}

void zkhip_r1cs_free(zkhip_r1cs* r) {
  if (!r) return;
  (void)bind_dev(r->device);
  r1cs_free(r->dev);
  delete r;
}

unsigned zkhip_r1cs_log_domain(const zkhip_r1cs* r) { return r ? (unsigned)r->dev->log_d : 0; }
size_t zkhip_r1cs_domain_size(const zkhip_r1cs* r) { return r ? r->dev->d : 0; }
size_t zkhip_domain_size(size_t min_size) { return host::forced_domain_size(min_size); }
size_t zkhip_step_domain_size(size_t min_size) { return host::eval_domain_size(min_size); }
int zkhip_domain_is_valid(size_t domain_size) { return host::is_valid_domain(domain_size) ? 1 : 0; }

int zkhip_r1cs_is_satisfied(zkhip_r1cs* r, const uint64_t* z, int* ok) {
  if (!r || !z || !ok) return fail(ZKHIP_ERR_ARG, "null pointer");
  BIND(r);
  std::lock_guard<std::mutex> lk(g.dev[r->device].mu);
  Scratch sc;
  uint64_t* dz = nullptr;
  API_HIP(sc.alloc((void**)&dz, r->dev->n_vars * 48));
  API_HIP(hipMemcpy(dz, z, r->dev->n_vars * 48, hipMemcpyHostToDevice));
  return r1cs_is_satisfied_dev(r->dev, dz, 0, ok, t_err, sizeof t_err);
}

int zkhip_qap_h(zkhip_r1cs* r, const uint64_t* z, uint64_t* h_out) {
  if (!r || !z || !h_out) return fail(ZKHIP_ERR_ARG, "null pointer");
  BIND(r);
  std::lock_guard<std::mutex> lk(g.dev[r->device].mu);
  size_t d = r->dev->d;
  Scratch sc;
  uint64_t *dz = nullptr, *dh = nullptr;
  API_HIP(sc.alloc((void**)&dz, r->dev->n_vars * 48));
  API_HIP(sc.alloc((void**)&dh, d * 48));
  API_HIP(hipMemcpy(dz, z, r->dev->n_vars * 48, hipMemcpyHostToDevice));
  int rc = qap_h_dev(r->dev, dz, 0, t_err, sizeof t_err);
  if (rc == ZKHIP_OK) {
    fr_dev_to_abi(r->dev->bufA, dh, d, 0);
    API_HIP(hipMemcpy(h_out, dh, d * 48, hipMemcpyDeviceToHost));
  }
  return rc;
}

// The options of a key, resolved once at upload against the process-wide defaults (zkhip_set_*), then carried by the handle:
// two threads loading keys with different options never see each other's choice.
struct ResolvedOpts { int precompute, naf, window, batch; };
static ResolvedOpts resolve_opts(const zkhip_key_opts* o) {
  ResolvedOpts r;
  r.precompute = (o && o->precompute >= 0) ? (o->precompute ? 1 : 0) : g.crs_tables.load();
  r.naf = (o && o->table_naf >= 0) ? (o->table_naf ? 1 : 0) : -1;                  // -1: naf_tables_wanted() decides
  r.window = (o && o->window > 0) ? o->window : g.forced_c.load();
  r.batch = (o && o->batch_msms >= 0) ? (o->batch_msms ? 1 : 0) : g.batch_msms.load();
  return r;
}

// one window size for the whole key (the prover's MSM contexts are shared by the five query vectors)
static int crs_build_tables(zkhip_crs* c, const ResolvedOpts& o) {
  c->batch_msms = o.batch;
  if (!o.precompute) return ZKHIP_OK;
  zkhip_bases* all[5] = {c->A, c->B2, c->B1, c->H, c->L};
  size_t total = 0, finite = 0, maxlen = 0;
  // (the window follows the longest vector's FINITE bases: a slice cut by finite terms - zkhip_key_partition - can be long and
  //  mostly infinity; a whole key's longest vector is H or L, all finite: unchanged)
  for (zkhip_bases* b : all) { total += b->len; finite += b->n_finite; if (b->n_finite > maxlen) maxlen = b->n_finite; }
  if (maxlen == 0) maxlen = c->A->len > c->H->len ? c->A->len : c->H->len;
  // one kind of table for the whole key; an explicit request for the larger kind still respects the memory guard
  const int naf = (o.naf >= 0 ? (o.naf && naf_tables_fit(total)) : naf_tables_wanted(total)) ? 1 : 0;
  int tc = o.window ? o.window : auto_table_window(maxlen);
  if (!o.window && naf && finite > 0) {
    // every-bit-position tables leave the window free of the table's layout: the five MSMs of a proof share one launch sequence,
    // so what counts is the key's FINITE bases together - additions ~ finite * 378 / (c + 2), bucket reduction ~ 5 * 2^(c-1).
    // Measured in a stream of proofs (tools/acc_probe.py --prove-stream): 176 k bases (the wrapping key) c = 14 / 15 / 16 ->
    // 363 / 372-378 / 367 proofs/s; 370 k bases (nine inputs per nested proof) c = 15 / 16 / 17 -> 182 / 192 / 190.5.
    const double copt = log2((double)finite) - 2.45;
    tc = (int)(copt + 0.5);
    if (tc < 9) tc = 9;
    if (tc > 20) tc = 20;
  }
  for (zkhip_bases* b : all) {
    int rc = bases_precompute_mode(b, tc, naf);
    if (rc != ZKHIP_OK) return rc;
  }
  return ZKHIP_OK;
}

static int crs_upload_impl(const zkhip_crs_desc* d, size_t a_lo, size_t a_len, size_t h_lo, size_t h_len, size_t l_lo, size_t l_len,
                           const zkhip_key_opts* opts, zkhip_crs** out) {
  if (opts && (opts->window < 0 || (opts->window > 0 && (opts->window < 4 || opts->window > 22))))
    return fail(ZKHIP_ERR_ARG, "zkhip_key_opts.window must be 0 (automatic) or in [4, 22]");
  zkhip_crs* c = new zkhip_crs();
  c->device = cur_dev();
  c->n_vars = d->n_vars; c->n_primary = d->n_primary; c->domain_size = d->domain_size;
  memcpy(c->alpha_g1, d->alpha_g1, 192); memcpy(c->beta_g1, d->beta_g1, 192); memcpy(c->beta_g2, d->beta_g2, 192);
  memcpy(c->delta_g1, d->delta_g1, 192); memcpy(c->delta_g2, d->delta_g2, 192);
  int rc;
  if ((rc = zkhip_bases_upload(d->a_query + a_lo * 24, a_len, &c->A)) == ZKHIP_OK &&
      (rc = zkhip_bases_upload(d->b_g2_query + a_lo * 24, a_len, &c->B2)) == ZKHIP_OK &&
      (rc = zkhip_bases_upload(d->b_g1_query + a_lo * 24, a_len, &c->B1)) == ZKHIP_OK &&
      (rc = zkhip_bases_upload(d->h_query + h_lo * 24, h_len, &c->H)) == ZKHIP_OK &&
      (rc = zkhip_bases_upload(d->l_query + l_lo * 24, l_len, &c->L)) == ZKHIP_OK)
    rc = crs_build_tables(c, resolve_opts(opts));
  if (rc != ZKHIP_OK) { zkhip_crs_free(c); return rc; }     // frees what was uploaded so far (null members are skipped)
  *out = c;
  return ZKHIP_OK;
}

int zkhip_crs_upload_ex(const zkhip_crs_desc* d, const zkhip_key_opts* opts, zkhip_crs** out) {
  BIND_CUR();
  if (!d || !out || !d->alpha_g1 || !d->beta_g1 || !d->beta_g2 || !d->delta_g1 || !d->delta_g2)
    return fail(ZKHIP_ERR_ARG, "null pointer");
  if (d->n_vars < d->n_primary + 1 || d->domain_size < 1) return fail(ZKHIP_ERR_ARG, "bad sizes");
  if (!host::is_valid_domain(d->domain_size)) return fail(ZKHIP_ERR_ARG, "proving key: domain_size is neither a power of two nor 2^k + 2^r");
  return crs_upload_impl(d, 0, d->n_vars, 0, d->domain_size - 1, 0, d->n_vars - d->n_primary - 1, opts, out);
}
int zkhip_crs_upload(const zkhip_crs_desc* d, zkhip_crs** out) { return zkhip_crs_upload_ex(d, nullptr, out); }

int zkhip_crs_upload_slice_ex(const zkhip_crs_desc* d, size_t a_lo, size_t a_len, size_t h_lo, size_t h_len, size_t l_lo, size_t l_len,
                              const zkhip_key_opts* opts, zkhip_crs** out) {
  BIND_CUR();
  if (!d || !out || !d->alpha_g1 || !d->beta_g1 || !d->beta_g2 || !d->delta_g1 || !d->delta_g2) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (d->n_vars < d->n_primary + 1 || d->domain_size < 1) return fail(ZKHIP_ERR_ARG, "bad sizes");
  if (!host::is_valid_domain(d->domain_size)) return fail(ZKHIP_ERR_ARG, "proving key: domain_size is neither a power of two nor 2^k + 2^r");
  if (a_lo + a_len > d->n_vars || h_lo + h_len > d->domain_size - 1 || l_lo + l_len > d->n_vars - d->n_primary - 1)
    return fail(ZKHIP_ERR_ARG, "slice out of range");
  return crs_upload_impl(d, a_lo, a_len, h_lo, h_len, l_lo, l_len, opts, out);
}
int zkhip_crs_upload_slice(const zkhip_crs_desc* d, size_t a_lo, size_t a_len, size_t h_lo, size_t h_len, size_t l_lo, size_t l_len,
                           zkhip_crs** out) {
  return zkhip_crs_upload_slice_ex(d, a_lo, a_len, h_lo, h_len, l_lo, l_len, nullptr, out);
}

int zkhip_crs_table_window(const zkhip_crs* c) { return (c && c->A) ? c->A->table_c : 0; }
int zkhip_crs_device(const zkhip_crs* c) { return c ? c->device : -1; }
int zkhip_crs_table_kind(const zkhip_crs* c) { return (c && c->A && c->A->table_c) ? (c->A->table_naf ? 2 : 1) : 0; }
int zkhip_crs_finite_terms(const zkhip_crs* c, size_t out[5]) {
  if (!c || !out) return fail(ZKHIP_ERR_ARG, "null pointer");
  const zkhip_bases* all[5] = {c->A, c->B2, c->B1, c->H, c->L};
  for (int k = 0; k < 5; k++) out[k] = all[k] ? all[k]->n_finite : 0;
  return ZKHIP_OK;
}

void zkhip_crs_free(zkhip_crs* c) {
  if (!c) return;
  (void)bind_dev(c->device);
  zkhip_bases_free(c->A); zkhip_bases_free(c->B2); zkhip_bases_free(c->B1); zkhip_bases_free(c->H); zkhip_bases_free(c->L);
  delete c;
}

int zkhip_last_prove_timings(double out_ms[8]) {
  const int dev = t_prove_dev >= 0 ? t_prove_dev : cur_dev();
  if (!out_ms || dev < 0) return ZKHIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(g.dev[dev].mu);
  memcpy(out_ms, g.dev[dev].ps.ms, sizeof g.dev[0].ps.ms);
  return ZKHIP_OK;
}

// The five MSMs of a proof over THIS process's slice of the proving key (the key is partitioned across the
// GPUs of a node; each rank computes partial sums, the ranks exchange 5 x 288 bytes).  Slice = [a_lo, a_lo + a_len) of
// the A / B queries (indices into z), [h_lo, h_lo + h_len) of the H query (indices into h), [l_lo, l_lo + l_len) of the
// L query (indices into z[n_primary+1 ..]).  The whole key is the slice (0, n_vars), (0, d - 1), (0, n_vars - l - 1).
// z: host assignment (uploaded here) - or d_z_ready: the assignment already in device memory, ABI form, complete (GPU witness)
// the cheap argument and size checks of a proof, before anything is started for it (the tail's scalar multiplications used to
// be spawned before them)
// ---- per-application constants (zkhip.h: zkhip_aggregator_app) ------------------------------------------------------------------
struct zkhip_aggregator_app {
  zkhip_aggregator* agg = nullptr;
  const zkhip_crs* crs = nullptr;          // the key the cached points belong to
  int device = 0;
  std::vector<uint64_t> vk;                // the nested key (identity of the application)
  std::vector<uint32_t> s_idx;             // the constant positions: auxiliary variables only, sorted
  std::vector<uint64_t> s_val;             // their values, 6 limbs each
  uint64_t vk_hash[6];                     // primary input 0
  uint64_t points[4 * 36];                 // sum over s_idx of z_i Base_i for the A, B-G2, B-G1 and L queries (Jacobian)
  void* host_state = nullptr;              // aggregator.cpp: the key with its lines
  uint64_t* d_z_app = nullptr;             // n_vars x 6 limbs on the device: the constants at their positions, zero elsewhere
  std::mutex mu;                           // the GPU witness program of this application, uploaded on first use
  WitnessTape tape;
  bool prog_ready = false;
  WitnessProgDev prog;
};

// the proof's key and (for a host assignment) its masking
static int app_check(const zkhip_aggregator_app* app, const zkhip_crs* crs, const uint64_t* z_host) {
  if (app->crs != crs) return fail(ZKHIP_ERR_ARG, "this application handle was made for another proving key");
  if (z_host)
    for (uint32_t i : app->s_idx) {
      const uint64_t* w = z_host + (size_t)i * 6;
      if (w[0] | w[1] | w[2] | w[3] | w[4] | w[5])
        return fail(ZKHIP_ERR_ARG, "assignment is not masked: a non-zero value at one of the application's constant positions (zkhip_aggregator_witness_app / zkhip_aggregator_app_mask)");
    }
  return ZKHIP_OK;
}
// A, B-G2, B-G1 and L sums of a masked proof += the application's cached points (H is untouched)
static void app_add_points(const zkhip_aggregator_app* app, uint64_t sums[180]) {
  using namespace host;
  auto jac = [](const uint64_t* p) { HJac q; q.X = HFq::from_limbs(p); q.Y = HFq::from_limbs(p + 12); q.Z = HFq::from_limbs(p + 24); return q; };
  static const int slot[4] = {0, 1, 2, 4};
  for (int k = 0; k < 4; k++) {
    uint64_t* s = sums + 36 * slot[k];
    const HJac t = jac(s).add(jac(app->points + 36 * k));
    t.X.to_limbs(s); t.Y.to_limbs(s + 12); t.Z.to_limbs(s + 24);
  }
}

// The proving key is authoritative for the evaluation domain (a reference key of the wrapping circuit says 65,536, a key generated
// here with ZKHIP_DOMAIN_STEP says 49,152): a shared constraint-system handle that sits on another domain is moved to the key's -
// its matrices stay, the domain's buffers are rebuilt (once: the next proof with this key finds it there).  A key whose domain is
// too small for the system, or not a domain at all, is refused.  Caller holds the device's mutex.
static int follow_key_domain(const zkhip_crs* crs, R1csDev* rd) {
  if (crs->n_vars != rd->n_vars || crs->n_primary != rd->n_primary) return fail(ZKHIP_ERR_ARG, "proving key and constraint system do not match");
  if (crs->domain_size == rd->d) return ZKHIP_OK;
  if (crs->domain_size == 0 || crs->domain_size == (size_t)-1) return fail(ZKHIP_ERR_ARG, "proving key without a domain size");
  return r1cs_set_domain(rd, crs->domain_size, t_err, sizeof t_err);
}

// 0 (default): a proof's five MSMs are one launch sequence; 1 / 2: the split form (see prove_partial).  ZKHIP_PROVE_SPLIT / zkhip_set_prove_split.
static std::atomic<int> g_prove_split{[] { const char* e = getenv("ZKHIP_PROVE_SPLIT"); const int v = e ? atoi(e) : 0; return v < 0 || v > 2 ? 0 : v; }()};
extern "C" int zkhip_set_prove_split(int mode) {
  if (mode < 0 || mode > 2) return fail(ZKHIP_ERR_ARG, "prove split: 0 (one launch sequence), 1 (two, gated) or 2 (two, not gated)");
  g_prove_split.store(mode);
  return ZKHIP_OK;
}
static int prove_check(const zkhip_crs* crs, const R1csDev* rd, size_t a_lo, size_t h_lo, size_t l_lo) {
  const size_t m = rd->n_vars, l = rd->n_primary, d = rd->d;
  const size_t a_len = crs->A->len, h_len = crs->H->len, l_len = crs->L->len;
  if (crs->n_vars != m || crs->n_primary != l) return fail(ZKHIP_ERR_ARG, "proving key and constraint system do not match");
  if (crs->domain_size != d) return fail(ZKHIP_ERR_ARG, "the constraint system handle is not on the proving key's evaluation domain");
  if (crs->B2->len != a_len || crs->B1->len != a_len || a_lo + a_len > m || h_lo + h_len > d - 1 || l_lo + l_len > m - l - 1)
    return fail(ZKHIP_ERR_ARG, "key slice out of range");
  const int tc = crs->A->table_c;
  if (crs->B2->table_naf != crs->A->table_naf || crs->B1->table_naf != crs->A->table_naf || crs->H->table_naf != crs->A->table_naf ||
      crs->L->table_naf != crs->A->table_naf)
    return fail(ZKHIP_ERR_ARG, "the five query vectors of a proving key must share one kind of table");
  if (crs->B2->table_c != tc || crs->B1->table_c != tc || crs->H->table_c != tc || crs->L->table_c != tc)
    return fail(ZKHIP_ERR_ARG, "the five query vectors of a proving key must share one table window");
  return ZKHIP_OK;
}

static int prove_partial(ProveState& ps, const zkhip_crs* crs, R1csDev* rd, const uint64_t* z, size_t a_lo, size_t h_lo, size_t l_lo,
                         uint64_t sums[5 * 36], const uint64_t* d_z_ready = nullptr, const uint64_t* d_z_app = nullptr) {
  using clk = std::chrono::steady_clock;
  auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
  const size_t m = rd->n_vars, l = rd->n_primary;
  const size_t a_len = crs->A->len, h_len = crs->H->len, l_len = crs->L->len;
  { int rc_ = prove_check(crs, rd, a_lo, h_lo, l_lo); if (rc_ != ZKHIP_OK) return rc_; }
  const int tc = crs->A->table_c;
  auto t0 = clk::now();
  if (!ps.st) {
    // (ZKHIP_QAP_STREAM_PRIO=1: the QAP map's stream at the highest priority - an experiment of the split proof, see below)
    static const int qprio = getenv("ZKHIP_QAP_STREAM_PRIO") ? atoi(getenv("ZKHIP_QAP_STREAM_PRIO")) : 0;
    int lo = 0, hi = 0;
    if (qprio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) API_HIP(hipStreamCreateWithPriority(&ps.st, hipStreamNonBlocking, hi));
    else API_HIP(hipStreamCreateWithFlags(&ps.st, hipStreamNonBlocking));
  }
  if (!ps.ev_st) API_HIP(hipEventCreateWithFlags(&ps.ev_st, hipEventBlockingSync | hipEventDisableTiming));
  if (ps.dz_cap < m) {
    if (ps.dz) { (void)hipFree(ps.dz); ps.dz = nullptr; ps.dz_cap = 0; }
    API_HIP(hipMalloc(&ps.dz, m * 48));
    ps.dz_cap = m;
  }
  const uint64_t* dz = d_z_ready ? d_z_ready : ps.dz;
  // a streaming prover whose five MSMs go through one launch sequence on ONE stream puts the upload and the QAP map on that stream
  // too: the host thread enqueues a whole proof without waiting between the phases (device side 376.8 -> 387.5 proofs/s in three
  // A/B pairs; ZKHIP_STREAM_CHAIN=0 restores the two waits, =2 chains every prover instance: +0.8 % for four 2^20 provers, not
  // adopted - the plain entry point's phase timings are the bench's one-proof-alone figures).  A chained proof's phase timings measure
  // enqueueing, not execution.
  static const int chain_env = getenv("ZKHIP_STREAM_CHAIN") ? atoi(getenv("ZKHIP_STREAM_CHAIN")) : 1;
  size_t maxlen = a_len > h_len ? a_len : h_len;
  if (maxlen < 1) maxlen = 1;
  size_t total_finite = 0;                          // (the rule of msm_launch_multi: a base at infinity never produces an entry)
  {
    const zkhip_bases* qs[5] = {crs->A, crs->B2, crs->B1, crs->H, crs->L};
    const size_t lens[5] = {a_len, a_len, a_len, h_len, l_len};
    for (int j = 0; j < 5; j++) {
      const size_t nf = lens[j] == qs[j]->len ? qs[j]->n_finite : 0;
      total_finite += (nf && nf < lens[j]) ? nf : lens[j];
    }
  }
  // (only when the launch sequence's context will be used as it stands: a context that ensure_ctx is about to rebuild gets a new stream)
  const bool chain = chain_env && (ps.quad_below || chain_env == 2) && crs->A->table_c > 0 && crs->batch_msms && ps.ctx[ZK_MSM_SLOTS].stream &&
                     ctx_reusable(&ps.ctx[ZK_MSM_SLOTS], ps.ready[ZK_MSM_SLOTS], maxlen, tc, 5, crs->A->table_naf, total_finite ? total_finite : 1);
  hipStream_t qst = chain ? ps.ctx[ZK_MSM_SLOTS].stream : ps.st;
  ps.chained_last = chain;
  ps.split_last = false;
  ps.last_acc_ctx2 = nullptr;
  // OPTION, off by default (ZKHIP_PROVE_SPLIT=1; =2: without the gate between the two accumulations) - a proof that is alone on the
  // chip as TWO launch sequences (round 6): four of the five MSMs - A, B-G2, B-G1, L - need only the assignment, so
  // their sort and accumulation start as soon as z is on the device, on their own plan, while the QAP map runs beside them on the
  // prover's stream; the H MSM follows on a second plan behind an event on h, and the bucket reduction of the first sequence runs
  // under its accumulation.  Same group elements, bit-identical proofs (the partitioned and serial tests pass with it on).
  // MEASURED AND NOT ADOPTED (profiles/r06_split_proof_ab.txt, one box, tools/partition_model.py): one 2^20 proof alone 50.0 ms as one
  // sequence, 52.0 split (49.8-50.5 without the gate); one rank of an eight-way 2^22 key 38.6 -> 39.8 ms; an eighth of a 2^20 key
  // 12.4 -> 13.5-14.8 ms.  The map is not idle time to hide: at 2^22 its seven FFTs and three SpMVs fill the chip (9.4 ms of
  // multiplier and HBM work) and share the accumulation's issue slots when run beside it - the two accumulation launches together
  // take 44 ms where the single one takes 37 - and every launch sequence brings its own one-fill tail, stitching and reduction chain.
  const int split_env = g_prove_split.load();
  if (split_env && !chain && !ps.quad_below && tc > 0 && crs->batch_msms && h_len > 0 && a_len > 0) {
    const zkhip_bases* qz[4] = {crs->A, crs->B2, crs->B1, crs->L};
    const size_t lz[4] = {a_len, a_len, a_len, l_len};
    size_t tot_z = 0, max_z = a_len > l_len ? a_len : l_len;
    for (int j = 0; j < 4; j++) { const size_t nf = lz[j] == qz[j]->len ? qz[j]->n_finite : 0; tot_z += (nf && nf < lz[j]) ? nf : lz[j]; }
    const size_t nf_h = h_len == crs->H->len ? crs->H->n_finite : 0, tot_h = (nf_h && nf_h < h_len) ? nf_h : h_len;
    MsmCtx *cz = &ps.ctx[ZK_CTX_Z], *ch = &ps.ctx[ZK_CTX_H];
    int rc = ensure_ctx(cz, &ps.ready[ZK_CTX_Z], max_z ? max_z : 1, tc, 4, crs->A->table_naf, tot_z ? tot_z : 1, 0, ps.pre);
    if (rc == ZKHIP_OK) rc = ensure_ctx(ch, &ps.ready[ZK_CTX_H], h_len, tc, 1, crs->A->table_naf, tot_h ? tot_h : 1);
    if (rc == ZKHIP_OK) {
      if (!ps.ev_up) API_HIP(hipEventCreateWithFlags(&ps.ev_up, hipEventDisableTiming));
      if (!ps.ev_qap) API_HIP(hipEventCreateWithFlags(&ps.ev_qap, hipEventDisableTiming));
      if (!d_z_ready) API_HIP(hipMemcpyAsync(ps.dz, z, m * 48, hipMemcpyHostToDevice, ps.st));
      API_HIP(hipEventRecord(ps.ev_up, ps.st));
      API_HIP(hipStreamWaitEvent(cz->stream, ps.ev_up, 0));
      ps.ms[0] = ms_since(t0);
      t0 = clk::now();
      MsmJob mz[4];
      const uint64_t* scz[4] = {dz + a_lo * 6, dz + a_lo * 6, dz + a_lo * 6, dz + (l + 1 + l_lo) * 6};
      for (int j = 0; j < 4; j++)
        mz[j] = MsmJob{qz[j]->d_pts, qz[j]->d_inf, scz[j], lz[j], 1, qz[j]->len, lz[j] == qz[j]->len ? qz[j]->n_finite : 0};
      const MsmJob mh{crs->H->d_pts, crs->H->d_inf, (const uint64_t*)rd->bufA + h_lo * 6, h_len, 2, crs->H->len, nf_h};
      auto tl0 = clk::now();
      rc = msm_launch_multi(cz, 4, mz);
      if (rc != ZKHIP_OK) { snprintf(t_err, sizeof t_err, "%s", cz->errbuf); return rc; }
      rc = qap_h_dev(rd, dz, ps.st, t_err, sizeof t_err, d_z_app);
      if (rc == ZKHIP_OK) {
        hipError_t e = hipEventRecord(ps.ev_qap, ps.st);
        if (e == hipSuccess) e = hipStreamWaitEvent(ch->stream, ps.ev_qap, 0);
        if (e != hipSuccess) { snprintf(t_err, sizeof t_err, "split proof: %s", hipGetErrorString(e)); rc = ZKHIP_ERR_HIP; }
      }
      if (rc == ZKHIP_OK) {
        ch->acc_gate = split_env == 2 ? nullptr : cz->ev_acc1;       // H's accumulation starts when the first one has ended (2: no gate)
        rc = msm_launch_multi(ch, 1, &mh);
        ch->acc_gate = nullptr;
        if (rc != ZKHIP_OK) snprintf(t_err, sizeof t_err, "%s", ch->errbuf);
      }
      ps.ms[1] = ms_since(t0);                      // (enqueueing the QAP map and both sequences: the phases overlap on the device)
      uint64_t sz[4 * 36];
      const int rc_z = msm_finish_multi(cz, 4, sz);           // (always collected: a pending plan would refuse the next proof)
      if (rc_z != ZKHIP_OK && rc == ZKHIP_OK) { snprintf(t_err, sizeof t_err, "%s", cz->errbuf); rc = rc_z; }
      if (rc != ZKHIP_OK) { (void)hipStreamSynchronize(ps.st); return rc; }
      if ((rc = msm_finish_multi(ch, 1, sums + 108)) != ZKHIP_OK) { snprintf(t_err, sizeof t_err, "%s", ch->errbuf); return rc; }
      memcpy(sums, sz, 3 * 36 * 8);
      memcpy(sums + 144, sz + 108, 36 * 8);
      for (int j = 0; j < 5; j++) ps.ms[2 + j] = ms_since(tl0);
      ps.last_accumulate_ms = cz->last_accumulate_ms + ch->last_accumulate_ms;
      ps.last_acc_ctx = cz; ps.last_acc_ctx2 = ch;
      ps.split_last = true;
      return ZKHIP_OK;
    }
    if (rc != ZKHIP_ERR_ARG) return rc;             // (a plan that does not fit 32-bit entry positions: the single-sequence path below decides)
  }
  if (!d_z_ready) {
    API_HIP(hipMemcpyAsync(ps.dz, z, m * 48, hipMemcpyHostToDevice, qst));
    if (!chain) {
      API_HIP(hipEventRecord(ps.ev_st, qst));
      API_HIP(zk_event_wait(ps.ev_st));
    }
  }
  ps.ms[0] = ms_since(t0);
  t0 = clk::now();
  // (an application's constants - d_z_app - are part of the assignment the QAP map sees; the MSMs over z below run on the masked
  //  vector: a zero scalar produces no bucket entry, the constants' share of A, B and L is the application's four cached points)
  int rc = qap_h_dev(rd, dz, qst, t_err, sizeof t_err, d_z_app);
  if (rc != ZKHIP_OK) return rc;
  if (!chain) {
    API_HIP(hipEventRecord(ps.ev_st, qst));
    API_HIP(zk_event_wait(ps.ev_st));   // the MSM contexts run on their own streams
  }
  ps.ms[1] = ms_since(t0);
  struct { const zkhip_bases* b; const uint64_t* sc; size_t len; int mode; uint64_t* out; } jobs[5] = {
      {crs->A, dz + a_lo * 6, a_len, 1, sums}, {crs->B2, dz + a_lo * 6, a_len, 1, sums + 36}, {crs->B1, dz + a_lo * 6, a_len, 1, sums + 72},
      {crs->H, (const uint64_t*)rd->bufA + h_lo * 6, h_len, 2, sums + 108}, {crs->L, dz + (l + 1 + l_lo) * 6, l_len, 1, sums + 144}};
  bool batched = tc > 0 && crs->batch_msms;
  if (batched) {
    // table-backed key: the five MSMs share ONE launch sequence (one sort, one accumulation launch over all five entry
    // lists, one reduction chain with a bucket window per MSM) - a fifth of the launches, five times the lanes in each.
    // (A plan that does not fit the engine's 32-bit entry positions is refused with ZKHIP_ERR_ARG: one sequence per MSM then.)
    const size_t total = total_finite;
    rc = ensure_ctx(&ps.ctx[ZK_MSM_SLOTS], &ps.ready[ZK_MSM_SLOTS], maxlen, tc, 5, crs->A->table_naf, total ? total : 1, 0, ps.pre);
    if (rc == ZKHIP_OK && ps.quad_below) { ps.ctx[ZK_MSM_SLOTS].quad_below = ps.quad_below; ps.ctx[ZK_MSM_SLOTS].one_stream = 1; }
    if (rc == ZKHIP_ERR_ARG) {
      // (a silent performance cliff) the plan's 32-bit entry positions / slice weights do not hold this key's terms at
      // this window: said once per process, then one launch sequence per MSM
      static std::atomic<bool> told{false};
      if (!told.exchange(true))
        fprintf(stderr, "zkhip: the five MSMs of this key do not fit ONE launch sequence (window %d, %zu terms: 32-bit slice weights); "
                        "using one sequence per MSM - a larger table window (zkhip_key_opts.window) avoids this\n", tc, total);
      batched = false;
    }
    else if (rc != ZKHIP_OK) return rc;
  }
  if (batched) {
    MsmCtx* cx = &ps.ctx[ZK_MSM_SLOTS];
    MsmJob mj[5];
    for (int j = 0; j < 5; j++)
      mj[j] = MsmJob{jobs[j].b->d_pts, jobs[j].b->d_inf, jobs[j].sc, jobs[j].len, jobs[j].mode, jobs[j].b->len,
                     jobs[j].len == jobs[j].b->len ? jobs[j].b->n_finite : 0};
    auto tl0 = clk::now();
    if ((rc = msm_launch_multi(cx, 5, mj)) == ZKHIP_OK) rc = msm_finish_multi(cx, 5, sums);
    if (rc != ZKHIP_OK) {
      // (a chained proof's stream also carried the upload and the QAP map: their failure surfaces here)
      snprintf(t_err, sizeof t_err, "%s%s", cx->errbuf, chain ? " [chained proof: this stream also carried the upload of the assignment and the QAP map]" : "");
      return rc;
    }
    for (int j = 0; j < 5; j++) ps.ms[2 + j] = ms_since(tl0);
    ps.last_accumulate_ms = cx->last_accumulate_ms;
    ps.last_acc_ctx = cx;
    return ZKHIP_OK;
  }
  // MSMs in flight: while MSM j reduces its buckets (latency-bound, few lanes), MSM j+1 accumulates.  Large
  // circuits keep two contexts (each holds ~1.3 GB of work space at 2^20); small ones (every phase is
  // latency-bound and the chip is mostly idle) run all five MSMs side by side.
  const int nctx = (maxlen <= ((size_t)1 << 18)) ? 5 : 2;
  MsmCtx* ctxs[5];
  for (int k = 0; k < nctx; k++) {
    if ((rc = ensure_ctx(&ps.ctx[k], &ps.ready[k], maxlen, tc, 1, crs->A->table_naf)) != ZKHIP_OK) return rc;
    if (ps.quad_below) { ps.ctx[k].quad_below = ps.quad_below; ps.ctx[k].one_stream = 1; }
    ctxs[k] = &ps.ctx[k];
  }
  clk::time_point tl[5];
  for (int j = 0; j < 5; j++) {
    MsmCtx* cx = ctxs[j % nctx];
    if (j >= nctx) {
      rc = msm_finish(cx, jobs[j - nctx].out);
      ps.ms[2 + j - nctx] = ms_since(tl[j - nctx]);
      if (rc != ZKHIP_OK) { snprintf(t_err, sizeof t_err, "%s", cx->errbuf); return rc; }
    }
    tl[j] = clk::now();
    rc = msm_launch(cx, jobs[j].b->d_pts, jobs[j].b->d_inf, jobs[j].sc, jobs[j].len, jobs[j].mode, jobs[j].b->len);
    if (rc != ZKHIP_OK) { snprintf(t_err, sizeof t_err, "%s", cx->errbuf); return rc; }
  }
  for (int j = 5 - nctx; j < 5; j++) {
    rc = msm_finish(ctxs[j % nctx], jobs[j].out);
    ps.ms[2 + j] = ms_since(tl[j]);
    if (rc != ZKHIP_OK) { snprintf(t_err, sizeof t_err, "%s", ctxs[j % nctx]->errbuf); return rc; }
  }
  ps.last_accumulate_ms = ctxs[4 % nctx]->last_accumulate_ms;
  ps.last_acc_ctx = ctxs[4 % nctx];
  return ZKHIP_OK;
}

int zkhip_groth16_prove_partial(const zkhip_crs* crs_slice, zkhip_r1cs* r1cs, const uint64_t* z, size_t a_lo, size_t h_lo, size_t l_lo,
                                uint64_t sums_jac[180]) {
  if (!crs_slice || !r1cs || !z || !sums_jac) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (crs_slice->device != r1cs->device) return fail(ZKHIP_ERR_ARG, "proving key and constraint system live on different devices");
  BIND(crs_slice);
  std::lock_guard<std::mutex> lk(g.dev[crs_slice->device].mu);
  { int rc_ = follow_key_domain(crs_slice, r1cs->dev); if (rc_ != ZKHIP_OK) return rc_; }
  return prove_partial(g.dev[crs_slice->device].ps, crs_slice, r1cs->dev, z, a_lo, h_lo, l_lo, sums_jac);
}

// tail: A = alpha + evA + r delta1;  B = beta + evB + s delta;  C = evH + evL + s A + r B1 - rs delta1
// The part of the prover's tail that depends on the key and on (r, s) only - r delta1, s delta2, s delta1, rs delta1: four 377-bit
// scalar multiplications of single points - is started BEFORE the device work of a proof and collected after it (one host thread
// each, asleep on the GPU otherwise): a proof on its own then waits for ONE scalar multiplication after its MSMs (s A, r B1 side
// by side) instead of two in a row.
struct TailPre {
  bool started = false;
  uint64_t rc_[6], sc_[6], rsc_[6];
  std::future<host::HJac> rd1, sd2, sd1, rsd1;
  // A packaged_task's future does not block in its destructor (std::async's did): a proof that fails between tail_begin and
  // finish_impl must not return while a zk-tail worker still holds the key's tables.  The tasks themselves capture
  // their scalars BY VALUE, so nothing in the pool points into this object either.
  ~TailPre() {
    for (std::future<host::HJac>* f : {&rd1, &sd2, &sd1, &rsd1})
      if (f->valid()) f->wait();
  }
};
// The tail's scalar multiplications run on a small pool of host threads ("zk-tail"; round 5: rounds 2-4 started a std::async thread
// per multiplication - six thread creations per proof, 2,400 a second in a stream).  Tasks are ~0.15 ms (fixed-base, with the key's
// tables) or ~0.5 ms (s A, r B1: variable base); six threads serve the ~1.5 ms of them a proof needs at any rate one GPU sustains.
namespace {
class TailPool {
 public:
  // (never destroyed: its workers sleep on the condition variable for the life of the process, and destroying a condition variable
  //  that has waiters - what a static object's destructor would do at exit - blocks)
  static TailPool& get() { static TailPool* p = new TailPool(); return *p; }
  std::future<host::HJac> run(std::function<host::HJac()> fn) {
    auto task = std::make_shared<std::packaged_task<host::HJac()>>(std::move(fn));
    std::future<host::HJac> f = task->get_future();
    {
      std::lock_guard<std::mutex> lk(mu_);
      q_.push_back([task] { (*task)(); });
    }
    cv_.notify_one();
    return f;
  }
 private:
  TailPool() {
    const char* e = getenv("ZKHIP_TAIL_THREADS");
    int n = e ? atoi(e) : 6;
    if (n < 1 || n > 64) n = 6;
    for (int i = 0; i < n; i++) {
      std::thread([this] {
        pthread_setname_np(pthread_self(), "zk-tail");
        for (;;) {
          std::function<void()> job;
          {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return !q_.empty(); });
            job = std::move(q_.front());
            q_.pop_front();
          }
          job();
        }
      }).detach();                 // (process-lifetime workers: they sleep on the queue)
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> q_;
};
}  // namespace
struct TailScalar { uint64_t v[6]; };
static TailScalar tail_scalar(const uint64_t* k) { TailScalar t; memcpy(t.v, k, sizeof t.v); return t; }
static std::future<host::HJac> tail_smul(host::HJac p, const uint64_t* k) {
  const TailScalar kv = tail_scalar(k);
  return TailPool::get().run([p, kv]() { return p.mul_canonical(kv.v, 6); });
}
// tab: the key's fixed-base tables (null: the plain zkhip_groth16_finish, which is handed raw points: variable-base products)
static void tail_begin(TailPre& tp, const uint64_t delta_g1[24], const uint64_t delta_g2[24], const uint64_t r_m[6], const uint64_t s_m[6],
                       const TailTables* tab = nullptr) {
  using namespace host;
  HFr r = HFr::from_limbs(r_m), s = HFr::from_limbs(s_m), rs = r * s;
  r.to_canonical(tp.rc_); s.to_canonical(tp.sc_); rs.to_canonical(tp.rsc_);
  if (tab) {
    const TailScalar rc = tail_scalar(tp.rc_), sc = tail_scalar(tp.sc_), rsc = tail_scalar(tp.rsc_);
    tp.rd1 = TailPool::get().run([tab, rc] { return tab->d1.mul(rc.v); });
    tp.sd2 = TailPool::get().run([tab, sc] { return tab->d2.mul(sc.v); });
    tp.sd1 = TailPool::get().run([tab, sc] { return tab->d1.mul(sc.v); });
    tp.rsd1 = TailPool::get().run([tab, rsc] { return tab->d1.mul(rsc.v); });
  } else {
    auto aff = [](const uint64_t* p) { return HJac::from_affine(HFq::from_limbs(p), HFq::from_limbs(p + 12)); };
    const HJac d1 = aff(delta_g1), d2 = aff(delta_g2);
    tp.rd1 = tail_smul(d1, tp.rc_); tp.sd2 = tail_smul(d2, tp.sc_); tp.sd1 = tail_smul(d1, tp.sc_); tp.rsd1 = tail_smul(d1, tp.rsc_);
  }
  tp.started = true;
}
// the key's tables for r delta_1, s delta_1, r s delta_1, s delta_2 (built once, by the key's first proof)
static const TailTables* crs_tail_tables(const zkhip_crs* c) {
  std::call_once(c->tail_once, [c] {
    using namespace host;
    auto aff = [](const uint64_t* p) { return HJac::from_affine(HFq::from_limbs(p), HFq::from_limbs(p + 12)); };
    std::unique_ptr<TailTables> t(new TailTables());
    t->d1.build(aff(c->delta_g1));
    t->d2.build(aff(c->delta_g2));
    c->tail = std::move(t);
  });
  return (c->tail->d1.ok && c->tail->d2.ok) ? c->tail.get() : nullptr;       // (not ok: variable-base products in tail_begin)
}

static int finish_impl(const uint64_t alpha_g1[24], const uint64_t beta_g1[24], const uint64_t beta_g2[24], const uint64_t delta_g1[24],
                       const uint64_t delta_g2[24], const uint64_t sums_jac[180], const uint64_t r_m[6], const uint64_t s_m[6],
                       uint64_t proof_affine[72], double* tail_ms, TailPre* pre = nullptr) {
  using namespace host;
  using clk = std::chrono::steady_clock;
  if (!alpha_g1 || !beta_g1 || !beta_g2 || !delta_g1 || !delta_g2 || !sums_jac || !r_m || !s_m || !proof_affine) return fail(ZKHIP_ERR_ARG, "null pointer");
  auto t0 = clk::now();
  const uint64_t *evA = sums_jac, *evB2 = sums_jac + 36, *evB1 = sums_jac + 72, *evH = sums_jac + 108, *evL = sums_jac + 144;
  auto jac = [](const uint64_t* p) { HJac q; q.X = HFq::from_limbs(p); q.Y = HFq::from_limbs(p + 12); q.Z = HFq::from_limbs(p + 24); return q; };
  auto aff = [](const uint64_t* p) { return HJac::from_affine(HFq::from_limbs(p), HFq::from_limbs(p + 12)); };
  TailPre local;
  TailPre& tp = (pre && pre->started) ? *pre : local;
  if (!tp.started) tail_begin(tp, delta_g1, delta_g2, r_m, s_m);       // (the plain zkhip_groth16_finish: nothing was started ahead)
  HJac gA = jac(evA).add(aff(alpha_g1)).add(tp.rd1.get());
  auto f_sA = tail_smul(gA, tp.sc_);
  HJac gB1 = jac(evB1).add(aff(beta_g1)).add(tp.sd1.get());
  auto f_rB1 = tail_smul(gB1, tp.rc_);
  HJac gB2 = jac(evB2).add(aff(beta_g2)).add(tp.sd2.get());
  HJac gC = jac(evH).add(jac(evL)).add(f_sA.get()).add(f_rB1.get()).add(tp.rsd1.get().neg());
  HFq x, y;
  gA.to_affine(x, y); x.to_limbs(proof_affine); y.to_limbs(proof_affine + 12);
  gB2.to_affine(x, y); x.to_limbs(proof_affine + 24); y.to_limbs(proof_affine + 36);
  gC.to_affine(x, y); x.to_limbs(proof_affine + 48); y.to_limbs(proof_affine + 60);
  if (tail_ms) *tail_ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
  return ZKHIP_OK;
}

int zkhip_groth16_finish(const uint64_t alpha_g1[24], const uint64_t beta_g1[24], const uint64_t beta_g2[24], const uint64_t delta_g1[24],
                         const uint64_t delta_g2[24], const uint64_t sums_jac[180], const uint64_t r_m[6], const uint64_t s_m[6],
                         uint64_t proof_affine[72]) {
  double tail_ms = 0;
  const int rc = finish_impl(alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2, sums_jac, r_m, s_m, proof_affine, &tail_ms);
  const int dev = cur_dev();
  if (rc == ZKHIP_OK && dev >= 0) { std::lock_guard<std::mutex> lk(g.dev[dev].mu); g.dev[dev].ps.ms[7] = tail_ms; }
  return rc;
}

int zkhip_groth16_prove(const zkhip_crs* crs, zkhip_r1cs* r1cs, const uint64_t* z, const uint64_t r_m[6], const uint64_t s_m[6],
                        uint64_t proof_affine[72]) {
  uint64_t sums[180];
  if (!crs || !r1cs || !z || !r_m || !s_m || !proof_affine) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (crs->device != r1cs->device) return fail(ZKHIP_ERR_ARG, "proving key and constraint system live on different devices");
  BIND(crs);
  TailPre pre;
  {
    std::lock_guard<std::mutex> lk(g.dev[crs->device].mu);
    { int rc_ = follow_key_domain(crs, r1cs->dev); if (rc_ != ZKHIP_OK) return rc_; }
    const size_t m = r1cs->dev->n_vars, l = r1cs->dev->n_primary, d = r1cs->dev->d;
    if (crs->A->len != m || crs->H->len != d - 1 || crs->L->len != m - l - 1) return fail(ZKHIP_ERR_ARG, "proving key and constraint system do not match");
    { int rc_ = prove_check(crs, r1cs->dev, 0, 0, 0); if (rc_ != ZKHIP_OK) return rc_; }
    tail_begin(pre, crs->delta_g1, crs->delta_g2, r_m, s_m, crs_tail_tables(crs));       // the key-only part of the tail runs under the device work
    int rc = prove_partial(g.dev[crs->device].ps, crs, r1cs->dev, z, 0, 0, 0, sums);
    if (rc != ZKHIP_OK) return rc;
  }
  t_prove_dev = crs->device;      // zkhip_last_prove_timings reads the device of this thread's last proof
  double tail_ms = 0;
  const int rc = finish_impl(crs->alpha_g1, crs->beta_g1, crs->beta_g2, crs->delta_g1, crs->delta_g2, sums, r_m, s_m, proof_affine, &tail_ms, &pre);
  if (rc == ZKHIP_OK) { std::lock_guard<std::mutex> lk(g.dev[crs->device].mu); g.dev[crs->device].ps.ms[7] = tail_ms; }   // (two threads may prove through this entry point)
  return rc;
}

// ---- prover instances: one proof in flight each, several instances per GPU -------------------------------------------
struct zkhip_prover {
  int device;
  const zkhip_crs* crs;
  R1csDev* rd;          // own copy of the constraint system + QAP work buffers
  ProveState ps;
  std::mutex mu;
  bool slice = false;                    // crs is a slice of the key (zkhip_prover_new_slice): partial sums only
  size_t a_lo = 0, h_lo = 0, l_lo = 0;   // where the slice starts in the A / B, H and L queries
};

static int prover_new_impl(const zkhip_crs* crs, const zkhip_r1cs_desc* cs, bool slice, size_t a_lo, size_t h_lo, size_t l_lo, zkhip_prover** out) {
  if (!crs || !cs || !out) return fail(ZKHIP_ERR_ARG, "null pointer");
  BIND(crs);
  R1csDev* rd = nullptr;
  {
    std::lock_guard<std::mutex> lk(g.dev[crs->device].mu);
    if (crs->n_vars != cs->n_vars || crs->n_primary != cs->n_primary) return fail(ZKHIP_ERR_ARG, "proving key and constraint system do not match");
    int rc = r1cs_upload(cs, crs->domain_size, &rd, t_err, sizeof t_err);     // the key's domain, whichever kind (domain.hpp)
    if (rc != ZKHIP_OK) return rc;
  }
  const size_t m = rd->n_vars, l = rd->n_primary, d = rd->d;
  bool ok = crs->n_vars == m && crs->n_primary == l && crs->domain_size == d;
  if (ok && !slice) ok = crs->A->len == m && crs->H->len == d - 1 && crs->L->len == m - l - 1;
  if (ok && slice) ok = a_lo + crs->A->len <= m && h_lo + crs->H->len <= d - 1 && l_lo + crs->L->len <= m - l - 1;
  if (!ok) {
    r1cs_free(rd);
    return fail(ZKHIP_ERR_ARG, slice ? "key slice out of range for this constraint system" : "proving key and constraint system do not match");
  }
  zkhip_prover* p = new zkhip_prover();
  p->device = crs->device;
  p->crs = crs; p->rd = rd;
  p->slice = slice; p->a_lo = a_lo; p->h_lo = h_lo; p->l_lo = l_lo;
  *out = p;
  return ZKHIP_OK;
}

int zkhip_prover_new(const zkhip_crs* crs, const zkhip_r1cs_desc* cs, zkhip_prover** out) { return prover_new_impl(crs, cs, false, 0, 0, 0, out); }
int zkhip_prover_new_slice(const zkhip_crs* crs_slice, const zkhip_r1cs_desc* cs, size_t a_lo, size_t h_lo, size_t l_lo, zkhip_prover** out) {
  return prover_new_impl(crs_slice, cs, true, a_lo, h_lo, l_lo, out);
}

void zkhip_prover_free(zkhip_prover* p) {
  if (!p) return;
  (void)bind_dev(p->device);
  p->ps.release();
  r1cs_free(p->rd);
  delete p;
}

static int prover_prove_impl(zkhip_prover* p, const uint64_t* z, const uint64_t* d_z, const uint64_t r_m[6], const uint64_t s_m[6], uint64_t proof_affine[72],
                             const zkhip_aggregator_app* app = nullptr);
int zkhip_prover_prove(zkhip_prover* p, const uint64_t* z, const uint64_t r_m[6], const uint64_t s_m[6], uint64_t proof_affine[72]) {
  if (!z) return fail(ZKHIP_ERR_ARG, "null pointer");
  return prover_prove_impl(p, z, nullptr, r_m, s_m, proof_affine);
}
int zkhip_prover_prove_dev(zkhip_prover* p, const void* d_z, const uint64_t r_m[6], const uint64_t s_m[6], uint64_t proof_affine[72]) {
  if (!d_z) return fail(ZKHIP_ERR_ARG, "null pointer");
  return prover_prove_impl(p, nullptr, (const uint64_t*)d_z, r_m, s_m, proof_affine);
}
int zkhip_prover_prove_partial(zkhip_prover* p, const uint64_t* z, uint64_t sums_jac[180]) {
  if (!p || !z || !sums_jac) return fail(ZKHIP_ERR_ARG, "null pointer");
  BIND(p);
  std::lock_guard<std::mutex> lk(p->mu);
  return prove_partial(p->ps, p->crs, p->rd, z, p->a_lo, p->h_lo, p->l_lo, sums_jac);
}
static int prover_prove_impl(zkhip_prover* p, const uint64_t* z, const uint64_t* d_z, const uint64_t r_m[6], const uint64_t s_m[6], uint64_t proof_affine[72],
                             const zkhip_aggregator_app* app) {
  if (!p || !r_m || !s_m || !proof_affine) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (p->slice) return fail(ZKHIP_ERR_STATE, "this prover holds a slice of the key: zkhip_prover_prove_partial");
  BIND(p);                                    // called from pipeline / application threads that never ran zkhip_init
  std::lock_guard<std::mutex> lk(p->mu);
  uint64_t sums[180];
  const zkhip_crs* c = p->crs;
  { int rc_ = prove_check(c, p->rd, 0, 0, 0); if (rc_ != ZKHIP_OK) return rc_; }
  if (app) { int rc_ = app_check(app, c, z); if (rc_ != ZKHIP_OK) return rc_; }
  TailPre pre;
  tail_begin(pre, c->delta_g1, c->delta_g2, r_m, s_m, crs_tail_tables(c));             // the key-only part of the tail runs under the device work
  int rc = prove_partial(p->ps, p->crs, p->rd, z, 0, 0, 0, sums, d_z, app ? app->d_z_app : nullptr);
  if (rc != ZKHIP_OK) return rc;
  if (app) app_add_points(app, sums);
  return finish_impl(c->alpha_g1, c->beta_g1, c->beta_g2, c->delta_g1, c->delta_g2, sums, r_m, s_m, proof_affine, &p->ps.ms[7], &pre);
}

// ---- per-application constants: the public entry points ---------------------------------------------------------------------
// (RegisterApplication fixes a nested verification key, aggregator_server.cpp:170-235; everything the wrapping circuit derives from
// the key alone - its variables, its MiMC hash chain, the lines of -beta and -delta, the doubling chains of ABC_i - is the same in
// every batch of that application: a quarter of the assignment.)
int zkhip_aggregator_app_new(zkhip_aggregator* a, const zkhip_crs* crs, const uint64_t* nested_vk, zkhip_aggregator_app** out) {
  if (!a || !crs || !nested_vk || !out) return fail(ZKHIP_ERR_ARG, "null pointer");
  BIND(crs);
  const size_t m = a->n_vars, l = a->n_primary;
  if (crs->n_vars != m || crs->n_primary != l || crs->A->len != m || crs->L->len != m - l - 1)
    return fail(ZKHIP_ERR_ARG, "zkhip_aggregator_app_new: the proving key is not the whole key of this circuit");
  std::unique_ptr<zkhip_aggregator_app> app(new zkhip_aggregator_app());
  app->agg = a; app->crs = crs; app->device = crs->device;
  const size_t vk_w = 60 + 12 * (a->inputs_per_proof + 1);
  app->vk.assign(nested_vk, nested_vk + vk_w);
  {
    // the key's own points must lie on their curves (the proofs of a batch are checked per batch): checked with the key's points
    // standing in for a proof
    std::vector<uint64_t> stand_in(48 * a->num_proofs);
    for (size_t q = 0; q < a->num_proofs; q++) {
      memcpy(&stand_in[q * 48], nested_vk, 96); memcpy(&stand_in[q * 48 + 12], nested_vk + 12, 192); memcpy(&stand_in[q * 48 + 36], nested_vk, 96);
    }
    int ok = 0;
    if (zkhip_aggregator_check_inputs(a, nested_vk, stand_in.data(), &ok) != ZKHIP_OK || !ok)
      return fail(ZKHIP_ERR_ARG, "zkhip_aggregator_app_new: the nested verification key has a point that is not on its curve");
  }
  // the program of this application: the circuit recorded with the key as CONSTANTS (witness_tape.cpp) - what folds is constant
  std::string terr;
  if (witness_tape_build(a->num_proofs, a->inputs_per_proof, &app->tape, &terr, nested_vk) != 0) {
    snprintf(t_err, sizeof t_err, "application program: %s", terr.c_str());
    return ZKHIP_ERR_ARG;                            // (a key with a point off its curve / a degenerate key meets an inversion of zero here)
  }
  WitnessTape& T = app->tape;
  if (T.n_vars != m) return fail(ZKHIP_ERR_STATE, "application program does not match the circuit");
  if (T.out_ref[1] >= 0) return fail(ZKHIP_ERR_STATE, "application program: the key hash did not fold into a constant");
  memcpy(app->vk_hash, &T.consts[(size_t)(-1 - T.out_ref[1]) * 6], 48);
  // the zero constant every masked position is pointed at
  int32_t zero_ref = 0;
  {
    size_t zi = T.consts.size() / 6;
    for (size_t i = 0; i < T.consts.size() / 6; i++) {
      const uint64_t* c = &T.consts[i * 6];
      if (!(c[0] | c[1] | c[2] | c[3] | c[4] | c[5])) { zi = i; break; }
    }
    if (zi == T.consts.size() / 6) T.consts.insert(T.consts.end(), 6, 0);
    zero_ref = -1 - (int32_t)zi;
  }
  std::vector<uint64_t> z_app(m * 6, 0);
  for (size_t i = l + 1; i < m; i++) {
    if (T.out_ref[i] >= 0) continue;
    const uint64_t* c = &T.consts[(size_t)(-1 - T.out_ref[i]) * 6];
    app->s_idx.push_back((uint32_t)i);
    app->s_val.insert(app->s_val.end(), c, c + 6);
    memcpy(&z_app[i * 6], c, 48);
    T.out_ref[i] = zero_ref;                         // the application's generator writes the MASKED assignment
  }
  // The masked HOST generator (witness_proofs_only) leaves the whole hash and key sections [sec_hash, sec_proofs) at zero and relies on
  // every one of those positions being a constant of the handle: checked here rather than assumed - a recorder that stopped
  // folding one of them would otherwise give an assignment with a hole, a wrong H and an unverifiable proof with rc OK.
  {
    size_t covered = 0;
    for (uint32_t i : app->s_idx) covered += (i >= a->sec_hash && i < a->sec_proofs) ? 1 : 0;
    if (covered != a->sec_proofs - a->sec_hash)
      return fail(ZKHIP_ERR_STATE, "zkhip_aggregator_app_new: the key's hash and line sections did not fold into constants completely");
  }
  // Self-check: the constants came out of the RECORDING build of the circuit (the route the device takes); the host generator is the
  // reference for parity.  One full host assignment under this key (the key's own points stand in for the proofs: the constant
  // positions do not depend on them) must hold exactly these values there - a degenerate key, where the two routes part, gets no handle.
  {
    std::vector<uint64_t> stand_in(48 * a->num_proofs), zero_inputs(6 * a->inputs_per_proof * a->num_proofs, 0), z_full(m * 6);
    for (size_t q = 0; q < a->num_proofs; q++) {
      memcpy(&stand_in[q * 48], nested_vk, 96); memcpy(&stand_in[q * 48 + 12], nested_vk + 12, 192); memcpy(&stand_in[q * 48 + 36], nested_vk, 96);
    }
    if (zkhip_aggregator_witness(a, nested_vk, stand_in.data(), zero_inputs.data(), z_full.data()) != ZKHIP_OK)
      return fail(ZKHIP_ERR_ARG, "zkhip_aggregator_app_new: the host generator refuses this nested key");
    for (size_t j = 0; j < app->s_idx.size(); j++)
      if (memcmp(&z_full[(size_t)app->s_idx[j] * 6], &app->s_val[j * 6], 48) != 0)
        return fail(ZKHIP_ERR_ARG, "zkhip_aggregator_app_new: degenerate nested key (the recorded constants differ from the host generator's)");
    if (memcmp(&z_full[6], app->vk_hash, 48) != 0) return fail(ZKHIP_ERR_STATE, "zkhip_aggregator_app_new: key hash mismatch");
  }
  int rc = zk_app_host_new(a, nested_vk, &app->host_state);
  if (rc != ZKHIP_OK) return fail(rc, "zkhip_aggregator_app_new: the nested key's lines could not be computed (a degenerate key)");
  hipError_t e = hipMalloc(&app->d_z_app, m * 48);
  if (e == hipSuccess) e = hipMemcpy(app->d_z_app, z_app.data(), m * 48, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipStreamSynchronize(0);
  if (e != hipSuccess) {
    snprintf(t_err, sizeof t_err, "zkhip_aggregator_app_new: %s", hipGetErrorString(e));
    zkhip_aggregator_app_free(app.release());
    return ZKHIP_ERR_HIP;
  }
  // ONE masked MSM per query: sum over the constant positions of z_i Base_i (the positions hold zero in z_app elsewhere)
  struct { const zkhip_bases* b; size_t off, len; } q[4] = {{crs->A, 0, m}, {crs->B2, 0, m}, {crs->B1, 0, m}, {crs->L, l + 1, m - l - 1}};
  for (int k = 0; k < 4 && rc == ZKHIP_OK; k++)
    rc = zkhip_msm_dev(q[k].b, 0, app->d_z_app + q[k].off * 6, q[k].len, 1, app->points + 36 * k);
  if (rc != ZKHIP_OK) { zkhip_aggregator_app_free(app.release()); return rc; }
  *out = app.release();
  return ZKHIP_OK;
}

void zkhip_aggregator_app_free(zkhip_aggregator_app* app) {
  if (!app) return;
  (void)bind_dev(app->device);
  if (app->prog_ready) witness_prog_free(&app->prog);
  if (app->d_z_app) (void)hipFree(app->d_z_app);
  zk_app_host_free(app->host_state);
  delete app;
}

size_t zkhip_aggregator_app_num_constants(const zkhip_aggregator_app* app) { return app ? app->s_idx.size() : 0; }

int zkhip_aggregator_app_constants(const zkhip_aggregator_app* app, uint32_t* positions, uint64_t* values, uint64_t vk_hash[6], uint64_t points_jac[144]) {
  if (!app) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (positions) memcpy(positions, app->s_idx.data(), app->s_idx.size() * 4);
  if (values) memcpy(values, app->s_val.data(), app->s_val.size() * 8);
  if (vk_hash) memcpy(vk_hash, app->vk_hash, 48);
  if (points_jac) memcpy(points_jac, app->points, sizeof app->points);
  return ZKHIP_OK;
}

int zkhip_aggregator_app_mask(const zkhip_aggregator_app* app, uint64_t* z) {
  if (!app || !z) return fail(ZKHIP_ERR_ARG, "null pointer");
  for (size_t j = 0; j < app->s_idx.size(); j++) {
    uint64_t* w = z + (size_t)app->s_idx[j] * 6;
    if (memcmp(w, &app->s_val[j * 6], 48) != 0) return fail(ZKHIP_ERR_ARG, "this assignment was not generated under the application's nested key");
    memset(w, 0, 48);
  }
  return ZKHIP_OK;
}

int zkhip_aggregator_witness_app(const zkhip_aggregator_app* app, const uint64_t* nested_proofs, const uint64_t* nested_inputs, uint64_t* z_out) {
  if (!app || !nested_proofs || !nested_inputs || !z_out) return fail(ZKHIP_ERR_ARG, "null pointer");
  return zk_app_host_witness(app->agg, app->host_state, app->vk.data(), nested_proofs, nested_inputs, app->s_idx.data(), app->s_idx.size(), app->vk_hash, z_out);
}

int zkhip_prover_prove_app(zkhip_prover* p, const zkhip_aggregator_app* app, const uint64_t* z_masked, const uint64_t r_m[6], const uint64_t s_m[6],
                           uint64_t proof_affine[72]) {
  if (!z_masked || !app) return fail(ZKHIP_ERR_ARG, "null pointer");
  return prover_prove_impl(p, z_masked, nullptr, r_m, s_m, proof_affine, app);
}
int zkhip_prover_prove_app_dev(zkhip_prover* p, const zkhip_aggregator_app* app, const void* d_z_masked, const uint64_t r_m[6], const uint64_t s_m[6],
                               uint64_t proof_affine[72]) {
  if (!d_z_masked || !app) return fail(ZKHIP_ERR_ARG, "null pointer");
  return prover_prove_impl(p, nullptr, (const uint64_t*)d_z_masked, r_m, s_m, proof_affine, app);
}

int zkhip_groth16_prove_app(const zkhip_crs* crs, zkhip_r1cs* r1cs, const zkhip_aggregator_app* app, const uint64_t* z_masked, const uint64_t r_m[6],
                            const uint64_t s_m[6], uint64_t proof_affine[72]) {
  uint64_t sums[180];
  if (!crs || !r1cs || !app || !z_masked || !r_m || !s_m || !proof_affine) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (crs->device != r1cs->device) return fail(ZKHIP_ERR_ARG, "proving key and constraint system live on different devices");
  BIND(crs);
  { int rc_ = app_check(app, crs, z_masked); if (rc_ != ZKHIP_OK) return rc_; }
  TailPre pre;
  {
    std::lock_guard<std::mutex> lk(g.dev[crs->device].mu);
    { int rc_ = follow_key_domain(crs, r1cs->dev); if (rc_ != ZKHIP_OK) return rc_; }
    { int rc_ = prove_check(crs, r1cs->dev, 0, 0, 0); if (rc_ != ZKHIP_OK) return rc_; }
    tail_begin(pre, crs->delta_g1, crs->delta_g2, r_m, s_m, crs_tail_tables(crs));
    int rc = prove_partial(g.dev[crs->device].ps, crs, r1cs->dev, z_masked, 0, 0, 0, sums, nullptr, app->d_z_app);
    if (rc != ZKHIP_OK) return rc;
  }
  app_add_points(app, sums);
  t_prove_dev = crs->device;
  double tail_ms = 0;
  const int rc = finish_impl(crs->alpha_g1, crs->beta_g1, crs->beta_g2, crs->delta_g1, crs->delta_g2, sums, r_m, s_m, proof_affine, &tail_ms, &pre);
  if (rc == ZKHIP_OK) { std::lock_guard<std::mutex> lk(g.dev[crs->device].mu); g.dev[crs->device].ps.ms[7] = tail_ms; }
  return rc;
}

// The HIP runtime spreads streams over a fixed number of hardware queues in the order they are CREATED, and kernels of streams that
// share a queue do not overlap.  A streaming prover runs a whole proof on one stream (its launch sequence's main stream); its other two
// streams idle.  Created lazily by the provers' own threads at their first proofs - thirty-two threads at once - the busy streams land
// on the queues as the race decides: a pipeline whose busy streams crowd a few queues proves 8-10 % fewer proofs/s for as long as it
// lives (round 5: 391 against 425-434 proofs/s, instance after instance in one process; same clock, 10 % less power).  A caller
// that owns several instances therefore creates the streams itself, in two passes over the instances: which = 0 the stream that will be
// busy (consecutive creations go to different queues), which = 1 the two that idle.
int zkhip_prover_create_streams(zkhip_prover* p, int which) {
  if (!p || which < 0 || which > 1) return fail(ZKHIP_ERR_ARG, "null prover or unknown pass");
  BIND(p);
  std::lock_guard<std::mutex> lk(p->mu);
  ProveState& ps = p->ps;
  if (which == 0) {
    if (!ps.pre[0] && !ps.ready[ZK_MSM_SLOTS]) API_HIP(hipStreamCreateWithFlags(&ps.pre[0], hipStreamNonBlocking));
  } else {
    if (!ps.pre[1] && !ps.ready[ZK_MSM_SLOTS]) API_HIP(hipStreamCreateWithFlags(&ps.pre[1], hipStreamNonBlocking));
    if (!ps.st) API_HIP(hipStreamCreateWithFlags(&ps.st, hipStreamNonBlocking));
  }
  return ZKHIP_OK;
}

int zkhip_prover_set_streaming(zkhip_prover* p, int on) {
  if (!p) return fail(ZKHIP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> lk(p->mu);
  // measured on the wrapping circuit: 212 -> 228 proofs/s with six provers in flight, 107 -> 98 one at a time
  p->ps.quad_below = on ? 1024u : 0u;
  p->rd->spmv_log_lanes = on ? (p->rd->spmv_log_lanes_alone < 2 ? p->rd->spmv_log_lanes_alone : 2) : p->rd->spmv_log_lanes_alone;
  for (int k = 0; k < ZK_CTX_TOTAL; k++) if (p->ps.ready[k]) { p->ps.ctx[k].quad_below = on ? 1024u : 65536u; p->ps.ctx[k].one_stream = on ? 1 : 0; }
  return ZKHIP_OK;
}

static int last_entries_of(ProveState& ps, uint64_t* out) {
  *out = 0;
  if (!ps.last_acc_ctx) return ZKHIP_OK;
  int rc = msm_last_entries(ps.last_acc_ctx, out);
  if (rc != ZKHIP_OK) snprintf(t_err, sizeof t_err, "%s", ps.last_acc_ctx->errbuf);
  if (rc == ZKHIP_OK && ps.last_acc_ctx2) {          // a split proof: two accumulation launches
    uint64_t more = 0;
    rc = msm_last_entries(ps.last_acc_ctx2, &more);
    if (rc != ZKHIP_OK) snprintf(t_err, sizeof t_err, "%s", ps.last_acc_ctx2->errbuf);
    *out += more;
  }
  return rc;
}
int zkhip_prover_last_accumulate_entries(zkhip_prover* p, uint64_t* out) {
  if (!p || !out) return fail(ZKHIP_ERR_ARG, "null argument");
  BIND(p);
  std::lock_guard<std::mutex> lk(p->mu);
  return last_entries_of(p->ps, out);
}
float zkhip_prover_last_accumulate_ms(zkhip_prover* p) {
  if (!p) return 0.f;
  std::lock_guard<std::mutex> lk(p->mu);
  return p->ps.last_accumulate_ms;
}

int zkhip_prover_timings(zkhip_prover* p, double out_ms[8]) {
  if (!p || !out_ms) return ZKHIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(p->mu);
  memcpy(out_ms, p->ps.ms, sizeof p->ps.ms);
  return ZKHIP_OK;
}

// 1: p's last proof was CHAINED (a streaming prover enqueues upload, QAP map and the five MSMs on one stream and waits once): slots [0]
// and [1] of zkhip_prover_timings then hold the time it took to ENQUEUE those phases, and the MSM slots the whole device time
int zkhip_prover_timings_chained(zkhip_prover* p) {
  if (!p) return 0;
  std::lock_guard<std::mutex> lk(p->mu);
  return p->ps.chained_last ? 1 : (p->ps.split_last ? 2 : 0);
}
// 1: the last proof of the plain entry points on this thread's device ran as two launch sequences (see prove_partial): slot [1] of
// zkhip_last_prove_timings is then the time to ENQUEUE the QAP map and both sequences, the MSM slots hold the device time of all of it
int zkhip_last_prove_split(void) {
  const int dev = t_prove_dev >= 0 ? t_prove_dev : cur_dev();
  if (dev < 0) return 0;
  std::lock_guard<std::mutex> lk(g.dev[dev].mu);
  return g.dev[dev].ps.split_last ? 1 : 0;
}

int zkhip_groth16_verify(const uint64_t vk_alpha_g1[24], const uint64_t vk_beta_g2[24], const uint64_t vk_delta_g2[24],
                         const uint64_t* vk_abc, const uint64_t* inputs, size_t n_inputs, const uint64_t proof_affine[72],
                         int* ok) {
  using namespace host;
  if (!vk_alpha_g1 || !vk_beta_g2 || !vk_delta_g2 || !vk_abc || !proof_affine || !ok || (n_inputs && !inputs))
    return fail(ZKHIP_ERR_ARG, "null pointer");
  auto aff = [](const uint64_t* p) { return HJac::from_affine(HFq::from_limbs(p), HFq::from_limbs(p + 12)); };
  // well-formedness first (libsnark: proof.is_well_formed()): G1 is y^2 = x^3 - 1, G2 is y^2 = x^3 + 4, both over Fq;
  // the all-zero encoding of the point at infinity passes.  An off-curve point would put the pairing in an invalid-curve setting.
  auto on_curve = [](const uint64_t* p, bool g2) {
    HFq x = HFq::from_limbs(p), y = HFq::from_limbs(p + 12);
    if (x.is_zero() && y.is_zero()) return true;
    HFq four = HFq::one().dbl().dbl();
    HFq rhs = x.sqr() * x + (g2 ? four : HFq::one().neg());
    return y.sqr() == rhs;
  };
  bool wf = on_curve(proof_affine, false) && on_curve(proof_affine + 24, true) && on_curve(proof_affine + 48, false) &&
            on_curve(vk_alpha_g1, false) && on_curve(vk_beta_g2, true) && on_curve(vk_delta_g2, true);
  for (size_t i = 0; i <= n_inputs && wf; i++) wf = on_curve(vk_abc + i * 24, false);
  if (!wf) { *ok = 0; return ZKHIP_OK; }
  // acc = ABC_0 + sum x_i ABC_i
  HJac acc = aff(vk_abc);
  for (size_t i = 0; i < n_inputs; i++) {
    uint64_t k[6];
    HFr::from_limbs(inputs + i * 6).to_canonical(k);
    acc = acc.add(aff(vk_abc + (i + 1) * 24).mul_canonical(k, 6));
  }
  uint64_t acc_aff[24], neg_g2[24], neg_beta[24], neg_delta[24];
  HFq x, y;
  acc.to_affine(x, y); x.to_limbs(acc_aff); y.to_limbs(acc_aff + 12);
  auto neg_pt = [](const uint64_t* p, uint64_t* o) {
    memcpy(o, p, 96);
    HFq yy = HFq::from_limbs(p + 12);
    bool inf = HFq::from_limbs(p).is_zero() && yy.is_zero();
    (inf ? yy : yy.neg()).to_limbs(o + 12);
  };
  uint64_t g2[24];
  memcpy(g2, FqParams::G2_GEN_X64, 96); memcpy(g2 + 12, FqParams::G2_GEN_Y64, 96);
  neg_pt(g2, neg_g2); neg_pt(vk_beta_g2, neg_beta); neg_pt(vk_delta_g2, neg_delta);
  std::vector<const uint64_t*> p1 = {proof_affine, acc_aff, vk_alpha_g1, proof_affine + 48};
  std::vector<const uint64_t*> p2 = {proof_affine + 24, neg_g2, neg_beta, neg_delta};
  *ok = pairing_product_is_one(p1, p2) ? 1 : 0;
  return ZKHIP_OK;
}

int zkhip_groth16_setup(const zkhip_r1cs_desc* cs, const uint64_t tau_m[6], const uint64_t alpha_m[6], const uint64_t beta_m[6],
                        const uint64_t delta_m[6], zkhip_keypair** out) {
  return zkhip_groth16_setup_ex(cs, tau_m, alpha_m, beta_m, delta_m, 0, out);     // the reference's forced power-of-two domain
}

extern "C" void zkhip_internal_cuts_by_weight(const uint32_t* w, size_t n, size_t parts, size_t* cuts);      // multi_device.cpp: zkhip_key_partition's rule
// The exponents of a key (host): QAP polynomials at tau through the Lagrange basis of the key's domain (one batch inversion), then
// the five query vectors' scalars.  Shared by the whole-key setup and the slice setup below.
extern "C++" {
namespace {
struct SetupScalars {
  size_t n = 0, m = 0, l = 0, d = 0;
  std::vector<host::HFr> At, Bt, hs, ls, abc, single;      // single: alpha, beta, delta
};
int setup_scalars(const zkhip_r1cs_desc* cs, const uint64_t tau_m[6], const uint64_t alpha_m[6], const uint64_t beta_m[6],
                  const uint64_t delta_m[6], size_t domain_size, SetupScalars& o) {
  using namespace host;
  const size_t n = cs->n_constraints, m = cs->n_vars, l = cs->n_primary;
  if (m < l + 1) return fail(ZKHIP_ERR_ARG, "bad sizes");
  // the evaluation domain (domain.hpp): 0 = the power of two libzeth's groth16_snark forces (the reference's keys), ZKHIP_DOMAIN_STEP =
  // libfqfft's unforced choice (a power of two or 2^k + 2^r), else a valid size at least n + l + 1
  const size_t d = resolve_domain(n + l + 1, domain_size);
  if (!d) return fail(ZKHIP_ERR_ARG, "setup: the evaluation domain is not a power of two or 2^k + 2^r, or smaller than n + l + 1");
  if (d > ((size_t)1 << 22)) return fail(ZKHIP_ERR_ARG, "domain larger than 2^22");
  HFr tau = HFr::from_limbs(tau_m), alpha = HFr::from_limbs(alpha_m), beta = HFr::from_limbs(beta_m), delta = HFr::from_limbs(delta_m);
  if (delta.is_zero()) return fail(ZKHIP_ERR_ARG, "delta must be invertible");
  // Lagrange basis at tau, one batch inversion
  const EvalDomain dom(d);
  const HFr Zt = dom.vanishing(tau);
  std::vector<HFr> Lg;
  if (!dom.lagrange_at(tau, Lg)) return fail(ZKHIP_ERR_ARG, "tau lies in the evaluation domain");
  o.n = n; o.m = m; o.l = l; o.d = d;
  o.At.assign(m, HFr::zero()); o.Bt.assign(m, HFr::zero());
  std::vector<HFr> Ct(m, HFr::zero());
  auto accumulate = [&](const uint32_t* rp, const uint32_t* col, const uint64_t* val, std::vector<HFr>& out_) {
    for (size_t j = 0; j < n; j++)
      for (uint32_t k = rp[j]; k < rp[j + 1]; k++) out_[col[k]] = out_[col[k]] + HFr::from_limbs(val + (size_t)k * 6) * Lg[j];
  };
  accumulate(cs->a_row_ptr, cs->a_col, cs->a_val, o.At);
  accumulate(cs->b_row_ptr, cs->b_col, cs->b_val, o.Bt);
  accumulate(cs->c_row_ptr, cs->c_col, cs->c_val, Ct);
  for (size_t k = 0; k <= l; k++) o.At[k] = o.At[k] + Lg[n + k];          // input-consistency rows
  const HFr delta_inv = delta.inv();
  o.hs.resize(d - 1); o.ls.resize(m - l - 1); o.abc.resize(l + 1); o.single.resize(3);
  HFr t = Zt * delta_inv;
  for (size_t j = 0; j + 1 < d; j++) { o.hs[j] = t; t = t * tau; }
  for (size_t i = l + 1; i < m; i++) o.ls[i - l - 1] = (beta * o.At[i] + alpha * o.Bt[i] + Ct[i]) * delta_inv;
  for (size_t i = 0; i <= l; i++) o.abc[i] = beta * o.At[i] + alpha * o.Bt[i] + Ct[i];
  o.single[0] = alpha; o.single[1] = beta; o.single[2] = delta;
  return ZKHIP_OK;
}
std::vector<uint64_t> pack_scalars(const host::HFr* v, size_t count) {
  std::vector<uint64_t> o(count * 6);
  for (size_t i = 0; i < count; i++) v[i].to_limbs(&o[i * 6]);
  return o;
}
void generators(uint64_t g1[24], uint64_t g2[24]) {
  memcpy(g1, FqParams::G1_GEN_X64, 96); memcpy(g1 + 12, FqParams::G1_GEN_Y64, 96);
  memcpy(g2, FqParams::G2_GEN_X64, 96); memcpy(g2 + 12, FqParams::G2_GEN_Y64, 96);
}
}  // namespace
}  // extern "C++"

int zkhip_groth16_setup_ex(const zkhip_r1cs_desc* cs, const uint64_t tau_m[6], const uint64_t alpha_m[6], const uint64_t beta_m[6],
                           const uint64_t delta_m[6], size_t domain_size, zkhip_keypair** out) {
  using namespace host;
  BIND_CUR();
  if (!cs || !tau_m || !alpha_m || !beta_m || !delta_m || !out) return fail(ZKHIP_ERR_ARG, "null pointer");
  SetupScalars sc;
  int rc = setup_scalars(cs, tau_m, alpha_m, beta_m, delta_m, domain_size, sc);
  if (rc != ZKHIP_OK) return rc;
  uint64_t g1[24], g2[24];
  generators(g1, g2);
  zkhip_keypair* kp = new zkhip_keypair();
  kp->n_vars = sc.m; kp->n_primary = sc.l; kp->domain_size = sc.d;
  auto fb = [&](const uint64_t* base, const std::vector<HFr>& v, std::vector<uint64_t>& dst) -> int {
    dst.assign(v.size() * 24, 0);
    if (v.empty()) return ZKHIP_OK;
    std::vector<uint64_t> s = pack_scalars(v.data(), v.size());
    return zkhip_fixed_base_mul(base, s.data(), v.size(), 1, dst.data());
  };
  std::vector<uint64_t> s1, s2;
  if ((rc = fb(g1, sc.single, s1)) != ZKHIP_OK || (rc = fb(g2, sc.single, s2)) != ZKHIP_OK || (rc = fb(g1, sc.At, kp->A)) != ZKHIP_OK ||
      (rc = fb(g2, sc.Bt, kp->B2)) != ZKHIP_OK || (rc = fb(g1, sc.Bt, kp->B1)) != ZKHIP_OK || (rc = fb(g1, sc.hs, kp->H)) != ZKHIP_OK ||
      (rc = fb(g1, sc.ls, kp->L)) != ZKHIP_OK || (rc = fb(g1, sc.abc, kp->ABC)) != ZKHIP_OK) {
    delete kp;
    return rc;
  }
  kp->alpha_g1.assign(s1.begin(), s1.begin() + 24); kp->beta_g1.assign(s1.begin() + 24, s1.begin() + 48);
  kp->delta_g1.assign(s1.begin() + 48, s1.begin() + 72);
  kp->beta_g2.assign(s2.begin() + 24, s2.begin() + 48); kp->delta_g2.assign(s2.begin() + 48, s2.begin() + 72);
  *out = kp;
  return ZKHIP_OK;
}

// One rank's share of a trusted setup (BASELINE configs[3]: the key pre-partitioned over the GPUs of a node).  Every rank evaluates
// the exponents (host: seconds at 2^22), cuts the three index ranges by FINITE terms exactly as zkhip_key_partition would cut the
// finished key - an exponent of zero IS a base at infinity - and then multiplies ONLY its own slice: an N-th of the fixed-base work,
// the points go from k_fixed_base_mul's output straight into the slice's base sets and window tables without leaving the device
// (round 6, VERDICT r5 item 4: until then every rank generated the whole key, copied its 4 GB to the host twice and uploaded an
// N-th).  *vk_out: a keypair WITHOUT queries (verification half and the five constants a prover's tail needs).
int zkhip_groth16_setup_slice(const zkhip_r1cs_desc* cs, const uint64_t tau_m[6], const uint64_t alpha_m[6], const uint64_t beta_m[6],
                              const uint64_t delta_m[6], size_t domain_size, int parts, int part, const zkhip_key_opts* opts,
                              zkhip_crs** slice_out, size_t ranges[6], zkhip_keypair** vk_out) {
  using namespace host;
  BIND_CUR();
  if (!cs || !tau_m || !alpha_m || !beta_m || !delta_m || !slice_out || !ranges) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (parts < 1 || parts > 64 || part < 0 || part >= parts) return fail(ZKHIP_ERR_ARG, "setup_slice: part must be in [0, parts), parts in [1, 64]");
  if (opts && (opts->window < 0 || (opts->window > 0 && (opts->window < 4 || opts->window > 22))))
    return fail(ZKHIP_ERR_ARG, "zkhip_key_opts.window must be 0 (automatic) or in [4, 22]");
  SetupScalars sc;
  int rc = setup_scalars(cs, tau_m, alpha_m, beta_m, delta_m, domain_size, sc);
  if (rc != ZKHIP_OK) return rc;
  // the cuts of zkhip_key_partition, from the exponents: weight of index i = its finite bases among A, B-G2, B-G1
  std::vector<size_t> ca(parts + 1), chh(parts + 1), cl(parts + 1);
  {
    std::vector<uint32_t> w(sc.m);
    for (size_t i = 0; i < sc.m; i++) w[i] = (sc.At[i].is_zero() ? 0u : 1u) + (sc.Bt[i].is_zero() ? 0u : 2u);
    zkhip_internal_cuts_by_weight(w.data(), w.size(), (size_t)parts, ca.data());
    w.assign(sc.hs.size(), 0);
    for (size_t i = 0; i < sc.hs.size(); i++) w[i] = sc.hs[i].is_zero() ? 0u : 1u;
    zkhip_internal_cuts_by_weight(w.data(), w.size(), (size_t)parts, chh.data());
    w.assign(sc.ls.size(), 0);
    for (size_t i = 0; i < sc.ls.size(); i++) w[i] = sc.ls[i].is_zero() ? 0u : 1u;
    zkhip_internal_cuts_by_weight(w.data(), w.size(), (size_t)parts, cl.data());
  }
  const size_t a_lo = ca[part], a_len = ca[part + 1] - ca[part], h_lo = chh[part], h_len = chh[part + 1] - chh[part],
               l_lo = cl[part], l_len = cl[part + 1] - cl[part];
  ranges[0] = a_lo; ranges[1] = a_lo + a_len; ranges[2] = h_lo; ranges[3] = h_lo + h_len; ranges[4] = l_lo; ranges[5] = l_lo + l_len;
  uint64_t g1[24], g2[24];
  generators(g1, g2);
  // the constants and the verification half (l + 1 + 6 products): through the host, they are needed there
  std::unique_ptr<zkhip_keypair> kp(new zkhip_keypair());
  kp->n_vars = sc.m; kp->n_primary = sc.l; kp->domain_size = sc.d;
  {
    std::vector<uint64_t> s1(72), s2(72), ps = pack_scalars(sc.single.data(), 3), pa = pack_scalars(sc.abc.data(), sc.abc.size());
    kp->ABC.assign(sc.abc.size() * 24, 0);
    if ((rc = zkhip_fixed_base_mul(g1, ps.data(), 3, 1, s1.data())) != ZKHIP_OK || (rc = zkhip_fixed_base_mul(g2, ps.data(), 3, 1, s2.data())) != ZKHIP_OK ||
        (rc = zkhip_fixed_base_mul(g1, pa.data(), sc.abc.size(), 1, kp->ABC.data())) != ZKHIP_OK)
      return rc;
    kp->alpha_g1.assign(s1.begin(), s1.begin() + 24); kp->beta_g1.assign(s1.begin() + 24, s1.begin() + 48);
    kp->delta_g1.assign(s1.begin() + 48, s1.begin() + 72);
    kp->beta_g2.assign(s2.begin() + 24, s2.begin() + 48); kp->delta_g2.assign(s2.begin() + 48, s2.begin() + 72);
  }
  // the slice's five query vectors: exponents up, products on the device, straight into base sets
  zkhip_crs* c = new zkhip_crs();
  c->device = cur_dev();
  c->n_vars = sc.m; c->n_primary = sc.l; c->domain_size = sc.d;
  memcpy(c->alpha_g1, kp->alpha_g1.data(), 192); memcpy(c->beta_g1, kp->beta_g1.data(), 192); memcpy(c->beta_g2, kp->beta_g2.data(), 192);
  memcpy(c->delta_g1, kp->delta_g1.data(), 192); memcpy(c->delta_g2, kp->delta_g2.data(), 192);
  auto query = [&](const uint64_t* base, const HFr* v, size_t count, zkhip_bases** dst) -> int {
    if (!count) return zkhip_bases_upload_dev(nullptr, 0, dst);
    Scratch scr;
    void *ds = nullptr, *dp = nullptr;
    API_HIP(scr.alloc(&ds, count * 48));
    API_HIP(scr.alloc(&dp, count * 192));
    {
      const std::vector<uint64_t> s = pack_scalars(v, count);
      API_HIP(hipMemcpy(ds, s.data(), count * 48, hipMemcpyHostToDevice));
    }
    int r = zkhip_fixed_base_mul_dev(base, ds, count, 1, dp);
    if (r == ZKHIP_OK) r = zkhip_bases_upload_dev(dp, count, dst);
    return r;
  };
  if ((rc = query(g1, sc.At.data() + a_lo, a_len, &c->A)) == ZKHIP_OK && (rc = query(g2, sc.Bt.data() + a_lo, a_len, &c->B2)) == ZKHIP_OK &&
      (rc = query(g1, sc.Bt.data() + a_lo, a_len, &c->B1)) == ZKHIP_OK && (rc = query(g1, sc.hs.data() + h_lo, h_len, &c->H)) == ZKHIP_OK &&
      (rc = query(g1, sc.ls.data() + l_lo, l_len, &c->L)) == ZKHIP_OK)
    rc = crs_build_tables(c, resolve_opts(opts));
  if (rc != ZKHIP_OK) { zkhip_crs_free(c); return rc; }
  *slice_out = c;
  if (vk_out) *vk_out = kp.release();
  return ZKHIP_OK;
}

int zkhip_keypair_crs_desc(const zkhip_keypair* kp, zkhip_crs_desc* o) {
  if (!kp || !o) return ZKHIP_ERR_ARG;
  o->n_vars = kp->n_vars; o->n_primary = kp->n_primary; o->domain_size = kp->domain_size;
  o->alpha_g1 = kp->alpha_g1.data(); o->beta_g1 = kp->beta_g1.data(); o->beta_g2 = kp->beta_g2.data();
  o->delta_g1 = kp->delta_g1.data(); o->delta_g2 = kp->delta_g2.data();
  o->a_query = kp->A.data(); o->b_g2_query = kp->B2.data(); o->b_g1_query = kp->B1.data(); o->h_query = kp->H.data(); o->l_query = kp->L.data();
  return ZKHIP_OK;
}

size_t zkhip_keypair_vk(const zkhip_keypair* kp, uint64_t alpha_g1[24], uint64_t beta_g2[24], uint64_t delta_g2[24], const uint64_t** abc) {
  if (!kp || !alpha_g1 || !beta_g2 || !delta_g2 || !abc) return 0;
  memcpy(alpha_g1, kp->alpha_g1.data(), 192); memcpy(beta_g2, kp->beta_g2.data(), 192); memcpy(delta_g2, kp->delta_g2.data(), 192);
  *abc = kp->ABC.data();
  return kp->n_primary + 1;
}

// ---- keypair file (the role of wsnarkT::keypair_write_bytes / keypair_read_bytes, aggregator_server.cpp:77-94).  The
// reference's byte format is defined in the absent libzeth, so this is the library's own container: a 64-byte header
// ("ZKHIPKP1", sizes) followed by the limb arrays exactly as they cross the C ABI (Montgomery u64 limbs, little-endian),
// and a 64-bit FNV-1a checksum of everything before it.  Host code, no device needed.
namespace {
const char KP_MAGIC[8] = {'Z', 'K', 'H', 'I', 'P', 'K', 'P', '1'};
uint64_t fnv1a(uint64_t h, const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
  return h;
}
}  // namespace

int zkhip_keypair_write(const zkhip_keypair* kp, const char* path) {
  if (!kp || !path) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (kp->A.size() != kp->n_vars * 24 || kp->H.size() != (kp->domain_size - 1) * 24)
    return fail(ZKHIP_ERR_ARG, "this keypair holds no query vectors (zkhip_groth16_setup_slice keeps the slice on the device): nothing to write");
  FILE* f = fopen(path, "wb");
  if (!f) return fail(ZKHIP_ERR_ARG, "cannot open the keypair file for writing");
  uint64_t hdr[8] = {0, (uint64_t)kp->n_vars, (uint64_t)kp->n_primary, (uint64_t)kp->domain_size, 0, 0, 0, 0};
  memcpy(&hdr[0], KP_MAGIC, 8);
  uint64_t h = 0xcbf29ce484222325ull;
  bool ok = fwrite(hdr, 8, 8, f) == 8;
  h = fnv1a(h, hdr, sizeof hdr);
  const std::vector<uint64_t>* parts[] = {&kp->alpha_g1, &kp->beta_g1, &kp->beta_g2, &kp->delta_g1, &kp->delta_g2, &kp->A, &kp->B2, &kp->B1, &kp->H, &kp->L, &kp->ABC};
  for (const auto* v : parts) {
    ok = ok && fwrite(v->data(), 8, v->size(), f) == v->size();
    h = fnv1a(h, v->data(), v->size() * 8);
  }
  ok = ok && fwrite(&h, 8, 1, f) == 1;
  ok = (fclose(f) == 0) && ok;
  return ok ? ZKHIP_OK : fail(ZKHIP_ERR_ARG, "short write on the keypair file");
}

int zkhip_keypair_read(const char* path, zkhip_keypair** out) {
  if (!path || !out) return fail(ZKHIP_ERR_ARG, "null pointer");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(ZKHIP_ERR_ARG, "cannot open the keypair file");
  uint64_t hdr[8];
  if (fread(hdr, 8, 8, f) != 8 || memcmp(&hdr[0], KP_MAGIC, 8) != 0) { fclose(f); return fail(ZKHIP_ERR_ARG, "not a keypair file (bad magic)"); }
  const uint64_t m = hdr[1], l = hdr[2], d = hdr[3];
  // the prover's domains stop at 2^22 points; a key has at most that many variables (one constraint defines at most one)
  // (the domain is one that get_evaluation_domain can return: a power of two or 2^k + 2^r - a fixed point of eval_domain_size)
  if (m < l + 1 || d < 1 || host::eval_domain_size((size_t)d) != d || m > ((uint64_t)1 << 23) || d > ((uint64_t)1 << 22)) { fclose(f); return fail(ZKHIP_ERR_ARG, "keypair file: implausible sizes"); }
  // the header fixes the file's length: check it before any allocation is sized by it
  {
    const uint64_t pts = 5 + 3 * m + (d - 1) + (m - l - 1) + (l + 1), want = 64 + 8 * 24 * pts + 8;
    long here = ftell(f);
    bool okl = here == 64 && fseek(f, 0, SEEK_END) == 0 && (uint64_t)ftell(f) == want && fseek(f, here, SEEK_SET) == 0;
    if (!okl) { fclose(f); return fail(ZKHIP_ERR_ARG, "keypair file truncated or corrupted (length does not match its header)"); }
  }
  zkhip_keypair* kp = nullptr;
  try {
  kp = new zkhip_keypair();
  kp->n_vars = m; kp->n_primary = l; kp->domain_size = d;
  struct { std::vector<uint64_t>* v; size_t pts; } parts[] = {{&kp->alpha_g1, 1}, {&kp->beta_g1, 1}, {&kp->beta_g2, 1}, {&kp->delta_g1, 1}, {&kp->delta_g2, 1},
                                                              {&kp->A, m}, {&kp->B2, m}, {&kp->B1, m}, {&kp->H, d - 1}, {&kp->L, m - l - 1}, {&kp->ABC, l + 1}};
  uint64_t h = fnv1a(0xcbf29ce484222325ull, hdr, sizeof hdr), stored = 0;
  bool ok = true;
  for (auto& p : parts) {
    p.v->resize(p.pts * 24);
    ok = ok && fread(p.v->data(), 8, p.v->size(), f) == p.v->size();
    if (!ok) break;
    h = fnv1a(h, p.v->data(), p.v->size() * 8);
  }
  ok = ok && fread(&stored, 8, 1, f) == 1 && stored == h;
  fclose(f);
  if (!ok) { delete kp; return fail(ZKHIP_ERR_ARG, "keypair file truncated or corrupted (checksum)"); }
  *out = kp;
  return ZKHIP_OK;
  } catch (const std::exception& e) {       // std::bad_alloc: no exception crosses the C ABI
    fclose(f);
    delete kp;
    snprintf(t_err, sizeof t_err, "keypair file: %s", e.what());
    return ZKHIP_ERR_ARG;
  }
}

void zkhip_keypair_free(zkhip_keypair* kp) { delete kp; }

int zkhip_last_accumulate_interval(float out_ms[2]) {
  const int dev = t_prove_dev >= 0 ? t_prove_dev : cur_dev();
  if (!out_ms || dev < 0) return ZKHIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(g.dev[dev].mu);
  out_ms[0] = g.dev[dev].ps.last_acc_interval[0]; out_ms[1] = g.dev[dev].ps.last_acc_interval[1];
  return ZKHIP_OK;
}
int zkhip_last_accumulate_entries(uint64_t* out) {
  if (!out) return fail(ZKHIP_ERR_ARG, "null argument");
  const int dev = t_prove_dev >= 0 ? t_prove_dev : cur_dev();
  if (dev < 0) { *out = 0; return ZKHIP_OK; }
  { int rc_ = bind_dev(dev); if (rc_ != ZKHIP_OK) return rc_; }
  std::lock_guard<std::mutex> lk(g.dev[dev].mu);
  return last_entries_of(g.dev[dev].ps, out);
}
float zkhip_last_accumulate_ms(void) {
  const int dev = t_prove_dev >= 0 ? t_prove_dev : cur_dev();       // where this thread's last MSM / proof ran
  if (dev < 0) return 0.f;
  std::lock_guard<std::mutex> lk(g.dev[dev].mu);
  return g.dev[dev].ps.last_accumulate_ms;
}
// The origin of zkhip_last_accumulate_interval's time base is recorded again, now: the values are FLOAT milliseconds since the origin,
// so a caller that compares intervals (bench.py's union of overlapping launches) re-bases at the start of its timed region.
int zkhip_reset_time_base(void) {
  BIND_CUR();
  return msm_time_base_reset();
}

// ---- witness generation on the GPU (witness.hip) ---------------------------------------------------------------------------
struct zkhip_gpu_witness {
  int device;
  zkhip_aggregator* agg;
  WitnessProg prog;
  size_t in_words;
  size_t max_batches = 1;
  uint64_t* d_in = nullptr;
  uint32_t* d_vals = nullptr;
  uint32_t* d_flag = nullptr;
  uint64_t* h_in = nullptr;       // pinned staging: inputs, then the flag and the primary inputs on the way back
  hipStream_t st = nullptr, st2 = nullptr;
  hipEvent_t ev = nullptr, ev_fork = nullptr, ev_join = nullptr;
};

int zkhip_gpu_witness_new(zkhip_aggregator* a, zkhip_gpu_witness** out) { return zkhip_gpu_witness_new_batched(a, 1, out); }

int zkhip_gpu_witness_new_batched(zkhip_aggregator* a, size_t max_batches, zkhip_gpu_witness** out) {
  BIND_CUR();
  if (!a || !out || max_batches < 1 || max_batches > 256) return fail(ZKHIP_ERR_ARG, "bad argument");
  zkhip_gpu_witness* w = new zkhip_gpu_witness();
  w->device = cur_dev(); w->agg = a; w->max_batches = max_batches;
  const WitnessTape* tape = nullptr;
  int rc = witness_prog(a, &w->prog, &tape, t_err, sizeof t_err);
  if (rc != ZKHIP_OK) { delete w; return rc; }
  w->in_words = (size_t)w->prog.n_inputs * 6;
  hipError_t e = hipMalloc(&w->d_in, max_batches * w->in_words * 8);
  if (e == hipSuccess) e = hipMalloc(&w->d_vals, max_batches * (size_t)w->prog.n_pos * witness_value_bytes);
  if (e == hipSuccess) e = hipMalloc(&w->d_flag, max_batches * 4 + 64);
  if (e == hipSuccess) e = hipHostMalloc(&w->h_in, max_batches * (w->in_words + 1 + a->n_primary * 6) * 8);
  // (High-priority streams for the witness launches - hardware queues of their own instead of a place behind some prover's 2 ms
  //  accumulation launch - were measured in round 6 and LOSE 10 %: 256-265 -> 221-234 proofs/s with nine inputs, 444-449 -> 395-407 with
  //  one, profiles/r06_witness_wpg_and_priority.txt.  A witness wave that is dispatched at once takes its CU slot at once, on whatever
  //  CU frees one first; what the provers need is that those slots are FEW, which is what k_witness's four waves a workgroup do.)
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&w->st, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&w->st2, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&w->ev, hipEventBlockingSync | hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&w->ev_join, hipEventDisableTiming);
  if (e != hipSuccess) {
    snprintf(t_err, sizeof t_err, "zkhip_gpu_witness_new: %s", hipGetErrorString(e));
    zkhip_gpu_witness_free(w);
    return ZKHIP_ERR_HIP;
  }
  *out = w;
  return ZKHIP_OK;
}

void zkhip_gpu_witness_free(zkhip_gpu_witness* w) {
  if (!w) return;
  (void)bind_dev(w->device);
  if (w->d_in) (void)hipFree(w->d_in);
  if (w->d_vals) (void)hipFree(w->d_vals);
  if (w->d_flag) (void)hipFree(w->d_flag);
  if (w->h_in) (void)hipHostFree(w->h_in);
  if (w->st) (void)hipStreamDestroy(w->st);
  if (w->st2) (void)hipStreamDestroy(w->st2);
  if (w->ev) (void)hipEventDestroy(w->ev);
  if (w->ev_fork) (void)hipEventDestroy(w->ev_fork);
  if (w->ev_join) (void)hipEventDestroy(w->ev_join);
  delete w;
}

// n batches in ONE launch sequence (one workgroup each).  vk / proofs / inputs: n pointers; d_z_out: n x n_vars x 6 limbs in device
// memory, contiguous; primary_inputs: n x n_primary x 6 limbs (host, may be null); degenerate[i] = 1 where an inversion met zero
// (that batch's assignment is unusable: use the host generator).
static int gpu_witness_run_impl(zkhip_gpu_witness* w, zkhip_aggregator_app* app, size_t n, const uint64_t* const* nested_vk, const uint64_t* const* nested_proofs,
                                const uint64_t* const* nested_inputs, void* d_z_out, uint64_t* primary_inputs, int* degenerate) {
  if (!w || (!app && !nested_vk) || !nested_proofs || !nested_inputs || !d_z_out || !degenerate) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (n < 1 || n > w->max_batches) return fail(ZKHIP_ERR_ARG, "more batches than the work space holds");
  BIND(w);
  const zkhip_aggregator* a = w->agg;
  const size_t vk_w = 60 + 12 * (a->inputs_per_proof + 1), pr_w = 48 * a->num_proofs, in_w = 6 * a->inputs_per_proof * a->num_proofs;
  if (vk_w + pr_w + in_w != w->in_words) return fail(ZKHIP_ERR_STATE, "witness program does not match the circuit");
  WitnessProg prog = w->prog;
  if (app) {
    // the application's own program (its key folded in: no key-hash chain, a tenth fewer multiplications, zeros at the constant
    // positions of the assignment), uploaded on first use
    if (app->agg != a || app->device != w->device) return fail(ZKHIP_ERR_ARG, "application handle of another circuit or device");
    std::lock_guard<std::mutex> lk(app->mu);
    if (!app->prog_ready) {
      int rc = witness_prog_upload(app->tape, &app->prog, t_err, sizeof t_err);
      if (rc != ZKHIP_OK) return rc;
      app->prog_ready = true;
    }
    if (app->prog.prog.n_pos > w->prog.n_pos || app->prog.prog.n_inputs != w->prog.n_inputs)
      return fail(ZKHIP_ERR_STATE, "the application's program does not fit the generator's work space");
    prog = app->prog.prog;
  }
  for (size_t i = 0; i < n; i++) {
    uint64_t* h = w->h_in + i * w->in_words;
    memcpy(h, app ? app->vk.data() : nested_vk[i], vk_w * 8); memcpy(h + vk_w, nested_proofs[i], pr_w * 8); memcpy(h + vk_w + pr_w, nested_inputs[i], in_w * 8);
  }
  uint64_t* h_flags = w->h_in + w->max_batches * w->in_words;            // [n flags as u32 | primary inputs]
  uint64_t* h_prim = h_flags + w->max_batches;
  API_HIP(hipMemcpyAsync(w->d_in, w->h_in, n * w->in_words * 8, hipMemcpyHostToDevice, w->st));
  API_HIP(hipMemsetAsync(w->d_flag, 0, n * 4, w->st));
  witness_launch(prog, w->d_in, w->d_vals, (uint64_t*)d_z_out, w->d_flag, (uint32_t)n, w->st, w->st2, w->ev_fork, w->ev_join);
  API_HIP(hipGetLastError());
  API_HIP(hipMemcpyAsync(h_flags, w->d_flag, n * 4, hipMemcpyDeviceToHost, w->st));
  for (size_t i = 0; i < n; i++)
    API_HIP(hipMemcpyAsync(h_prim + i * a->n_primary * 6, (const uint64_t*)d_z_out + (i * a->n_vars + 1) * 6, a->n_primary * 48, hipMemcpyDeviceToHost, w->st));
  API_HIP(hipEventRecord(w->ev, w->st));
  API_HIP(zk_event_wait(w->ev));
  for (size_t i = 0; i < n; i++) degenerate[i] = ((const uint32_t*)h_flags)[i] != 0;
  if (primary_inputs) memcpy(primary_inputs, h_prim, n * a->n_primary * 48);
  return ZKHIP_OK;
}

int zkhip_gpu_witness_run_batched(zkhip_gpu_witness* w, size_t n, const uint64_t* const* nested_vk, const uint64_t* const* nested_proofs,
                                  const uint64_t* const* nested_inputs, void* d_z_out, uint64_t* primary_inputs, int* degenerate) {
  return gpu_witness_run_impl(w, nullptr, n, nested_vk, nested_proofs, nested_inputs, d_z_out, primary_inputs, degenerate);
}
// the same for batches of ONE registered application: MASKED assignments (zeros at the application's constant positions), ready for
// zkhip_prover_prove_app_dev
int zkhip_gpu_witness_run_batched_app(zkhip_gpu_witness* w, zkhip_aggregator_app* app, size_t n, const uint64_t* const* nested_proofs,
                                      const uint64_t* const* nested_inputs, void* d_z_out, uint64_t* primary_inputs, int* degenerate) {
  if (!app) return fail(ZKHIP_ERR_ARG, "null pointer");
  return gpu_witness_run_impl(w, app, n, nullptr, nested_proofs, nested_inputs, d_z_out, primary_inputs, degenerate);
}

int zkhip_gpu_witness_run(zkhip_gpu_witness* w, const uint64_t* nested_vk, const uint64_t* nested_proofs, const uint64_t* nested_inputs,
                          void* d_z_out, uint64_t* primary_inputs) {
  int deg = 0;
  int rc = zkhip_gpu_witness_run_batched(w, 1, &nested_vk, &nested_proofs, &nested_inputs, d_z_out, primary_inputs, &deg);
  if (rc == ZKHIP_OK && deg)
    return fail(ZKHIP_ERR_ARG, "GPU witness: an inversion met zero (degenerate input); use the host generator for this batch");
  return rc;
}

int zkhip_aggregator_witness_gpu(zkhip_aggregator* a, const uint64_t* nested_vk, const uint64_t* nested_proofs, const uint64_t* nested_inputs,
                                 uint64_t* z_out) {
  if (!a || !z_out) return fail(ZKHIP_ERR_ARG, "null pointer");
  zkhip_gpu_witness* w = nullptr;
  int rc = zkhip_gpu_witness_new(a, &w);
  if (rc != ZKHIP_OK) return rc;
  void* dz = nullptr;
  hipError_t e = hipMalloc(&dz, a->n_vars * 48);
  if (e != hipSuccess) { zkhip_gpu_witness_free(w); snprintf(t_err, sizeof t_err, "hipMalloc: %s", hipGetErrorString(e)); return ZKHIP_ERR_HIP; }
  rc = zkhip_gpu_witness_run(w, nested_vk, nested_proofs, nested_inputs, dz, nullptr);
  if (rc == ZKHIP_OK && hipMemcpy(z_out, dz, a->n_vars * 48, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(ZKHIP_ERR_HIP, "copy of the assignment failed");
  (void)hipFree(dz);
  zkhip_gpu_witness_free(w);
  return rc;
}

int zkhip_gpu_witness_stats(zkhip_aggregator* a, size_t out[6]) {
  BIND_CUR();
  if (!a || !out) return fail(ZKHIP_ERR_ARG, "null pointer");
  WitnessProg P;
  const WitnessTape* T = nullptr;
  int rc = witness_prog(a, &P, &T, t_err, sizeof t_err);
  if (rc != ZKHIP_OK) return rc;
  out[0] = T->n_recorded; out[1] = T->code.size(); out[2] = T->level_start.size() - 1; out[3] = T->n_mul; out[4] = T->n_inv; out[5] = T->consts.size() / 6;
  return ZKHIP_OK;
}

int zkhip_device_count(void) {
  int count = 0;
  return hipGetDeviceCount(&count) == hipSuccess ? count : 0;
}

// uniform in [0, r): 377-bit draws from the OS, rejected when >= r (a value in [0, r) read as a Montgomery residue is a uniform
// field element either way)
int zkhip_fr_random(uint64_t out[6]) {
  if (!out) return fail(ZKHIP_ERR_ARG, "null pointer");
  FILE* f = fopen("/dev/urandom", "rb");
  if (!f) return fail(ZKHIP_ERR_STATE, "cannot open /dev/urandom");
  for (;;) {
    if (fread(out, 8, 6, f) != 6) { fclose(f); return fail(ZKHIP_ERR_STATE, "short read from /dev/urandom"); }
    out[5] &= ((uint64_t)1 << 57) - 1;               // r has 377 bits
    bool less = false;
    for (int i = 5; i >= 0; i--) {
      if (out[i] != FrParams::P64[i]) { less = out[i] < FrParams::P64[i]; break; }
    }
    if (less) break;
  }
  fclose(f);
  return ZKHIP_OK;
}

int zkhip_to_canonical(int which, const uint64_t* in, uint64_t* out) {
  using namespace host;
  if (!in || !out) return ZKHIP_ERR_ARG;
  if (which == 0) HFq::from_limbs(in).to_canonical(out);
  else if (which == 1) HFr::from_limbs(in).to_canonical(out);
  else return ZKHIP_ERR_ARG;
  return ZKHIP_OK;
}

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

// ---- handle-owned MSM streams ---------------------------------------------------------------------------------------------
// zkhip_msm_submit / collect above address eight PROCESS-WIDE slot numbers: two threads streaming MSMs on one device collide.  A
// zkhip_msm_stream owns its contexts (streams, work space), like a zkhip_prover: any number of them run side by side.
struct zkhip_msm_stream {
  int device = 0;
  const zkhip_bases* bases = nullptr;
  int depth = 0;
  std::vector<MsmCtx> ctx;
  std::vector<char> ready;
  std::vector<uint64_t> ticket_of;       // per slot: the ticket in flight there, 0 = free
  std::vector<void*> d_stage;            // per slot: device copy of host scalars (zkhip_msm_stream_submit_host)
  std::vector<size_t> stage_cap;
  uint64_t next_ticket = 1;
  float last_accumulate_ms = 0.f, last_interval[2] = {0.f, 0.f};
  std::mutex mu;
};

int zkhip_msm_stream_new(const zkhip_bases* bases, int depth, zkhip_msm_stream** out) {
  if (!bases || !out) return fail(ZKHIP_ERR_ARG, "null pointer");
  if (depth < 1 || depth > 16) return fail(ZKHIP_ERR_ARG, "depth must be in [1, 16]");
  BIND(bases);
  zkhip_msm_stream* st = new zkhip_msm_stream();
  st->device = bases->device; st->bases = bases; st->depth = depth;
  st->ctx.resize(depth); st->ready.assign(depth, 0); st->ticket_of.assign(depth, 0);
  st->d_stage.assign(depth, nullptr); st->stage_cap.assign(depth, 0);
  for (auto& c : st->ctx) memset(&c, 0, sizeof c);
  *out = st;
  return ZKHIP_OK;
}

void zkhip_msm_stream_free(zkhip_msm_stream* st) {
  if (!st) return;
  (void)bind_dev(st->device);
  for (int k = 0; k < st->depth; k++) {
    if (st->ready[k]) {
      if (st->ctx[k].pending) { uint64_t drop[36]; (void)msm_finish(&st->ctx[k], drop); }      // never free work space under a running launch
      msm_plan_free(&st->ctx[k]);
    }
    if (st->d_stage[k]) (void)hipFree(st->d_stage[k]);
  }
  delete st;
}

static int msm_stream_submit_impl(zkhip_msm_stream* st, size_t offset, const void* d_scalars, const uint64_t* h_scalars, size_t len,
                                  int scalars_montgomery, uint64_t* ticket) {
  if (!st || !ticket || (len && !d_scalars && !h_scalars)) return fail(ZKHIP_ERR_ARG, "null pointer");
  BIND(st);
  std::lock_guard<std::mutex> lk(st->mu);
  const zkhip_bases* b = st->bases;
  if (offset > b->len || len > b->len - offset) return fail(ZKHIP_ERR_ARG, "offset + len exceeds the base set");
  int slot = -1;
  for (int k = 0; k < st->depth; k++) if (!st->ticket_of[k]) { slot = k; break; }
  if (slot < 0) return fail(ZKHIP_ERR_STATE, "every slot of this stream is in flight: collect a result first");
  MsmCtx* cx = &st->ctx[slot];
  bool rdy = st->ready[slot] != 0;
  int rc = ensure_ctx(cx, &rdy, len ? len : 1, b->table_c, 1, b->table_naf, 0, b->plain_c);
  st->ready[slot] = rdy ? 1 : 0;
  if (rc != ZKHIP_OK) return rc;
  if (h_scalars && len) {
    // host scalars: one asynchronous copy in front of the MSM's kernels, on the context's own stream (truly asynchronous from pinned
    // memory - zkhip_host_alloc -, staged by the runtime from pageable memory); the copies of the MSMs in flight overlap their kernels
    if (st->stage_cap[slot] < len) {
      if (st->d_stage[slot]) { (void)hipFree(st->d_stage[slot]); st->d_stage[slot] = nullptr; st->stage_cap[slot] = 0; }
      API_HIP(hipMalloc(&st->d_stage[slot], len * 48));
      st->stage_cap[slot] = len;
    }
    API_HIP(hipMemcpyAsync(st->d_stage[slot], h_scalars, len * 48, hipMemcpyHostToDevice, cx->stream));
    d_scalars = st->d_stage[slot];
  }
  rc = msm_launch(cx, b->d_pts + offset, b->d_inf ? b->d_inf + offset : nullptr, (const uint64_t*)d_scalars, len, scalars_montgomery, b->len);
  if (rc != ZKHIP_OK) { snprintf(t_err, sizeof t_err, "%s", cx->errbuf); return rc; }
  st->ticket_of[slot] = st->next_ticket;
  *ticket = st->next_ticket++;
  return ZKHIP_OK;
}
int zkhip_msm_stream_submit(zkhip_msm_stream* st, size_t offset, const void* d_scalars, size_t len, int scalars_montgomery, uint64_t* ticket) {
  return msm_stream_submit_impl(st, offset, d_scalars, nullptr, len, scalars_montgomery, ticket);
}
int zkhip_msm_stream_submit_host(zkhip_msm_stream* st, size_t offset, const uint64_t* scalars, size_t len, int scalars_montgomery, uint64_t* ticket) {
  return msm_stream_submit_impl(st, offset, nullptr, scalars, len, scalars_montgomery, ticket);
}

int zkhip_msm_stream_collect(zkhip_msm_stream* st, uint64_t ticket, uint64_t out_jac[36]) {
  if (!st || !out_jac) return fail(ZKHIP_ERR_ARG, "null pointer");
  BIND(st);
  std::lock_guard<std::mutex> lk(st->mu);
  int slot = -1;
  for (int k = 0; k < st->depth; k++) {
    if (!st->ticket_of[k]) continue;
    if (ticket ? st->ticket_of[k] == ticket : (slot < 0 || st->ticket_of[k] < st->ticket_of[slot])) slot = k;    // ticket 0: the oldest
  }
  if (slot < 0) return fail(ZKHIP_ERR_STATE, ticket ? "no such ticket in flight on this stream" : "nothing in flight on this stream");
  MsmCtx* cx = &st->ctx[slot];
  st->ticket_of[slot] = 0;                  // (msm_finish clears `pending` whatever it returns: the slot is free again)
  int rc = msm_finish(cx, out_jac);
  if (rc != ZKHIP_OK) { snprintf(t_err, sizeof t_err, "%s", cx->errbuf); return rc; }
  st->last_accumulate_ms = cx->last_accumulate_ms; st->last_interval[0] = cx->last_acc_begin_ms; st->last_interval[1] = cx->last_acc_end_ms;
  return ZKHIP_OK;
}
float zkhip_msm_stream_last_accumulate_ms(zkhip_msm_stream* st) {
  if (!st) return 0.f;
  std::lock_guard<std::mutex> lk(st->mu);
  return st->last_accumulate_ms;
}
int zkhip_msm_stream_last_accumulate_interval(zkhip_msm_stream* st, float out_ms[2]) {
  if (!st || !out_ms) return fail(ZKHIP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> lk(st->mu);
  out_ms[0] = st->last_interval[0]; out_ms[1] = st->last_interval[1];
  return ZKHIP_OK;
}

int zkhip_measure_fq_mul_rate(double* fq_mul_per_s) {
  BIND_CUR();
  if (!fq_mul_per_s) return fail(ZKHIP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> lk(g.dev[cur_dev()].mu);
  return msm_measure_fqmul_rate(fq_mul_per_s, t_err, sizeof t_err);
}

int zkhip_measure_ntt(unsigned log_d, int dir, int coset, int batch, int reps, double* ms_per_transform) {
  BIND_CUR();
  if (!ms_per_transform) return fail(ZKHIP_ERR_ARG, "null pointer");
  return ntt_measure((int)log_d, dir, coset, batch, reps, ms_per_transform, t_err, sizeof t_err);
}

int zkhip_internal_field_selftest(int field, const uint32_t* limbs_in, size_t n, uint32_t* limbs_out) {
  BIND_CUR();
  if ((field != 0 && field != 1) || (n && (!limbs_in || !limbs_out)) || n > (1u << 20)) return fail(ZKHIP_ERR_ARG, "field 0 (Fq) or 1 (Fr), at most 2^20 cases");
  std::lock_guard<std::mutex> lk(g.dev[cur_dev()].mu);
  return msm_field_selftest(field, limbs_in, n, limbs_out, t_err, sizeof t_err);
}

// Host only (no device needed): the prover's tail on the paths a healthy proof never takes.
//  (1) `rounds` proofs that FAIL between tail_begin and finish_impl: the TailPre goes out of scope with its four tasks queued and the
//      key's tables are freed straight after - under the CPU AddressSanitizer build (tools/sanitize/asan_host_tests.sh) a task that
//      outlives either is a use-after-scope / use-after-free report (packaged_task futures do not wait in ~future);
//  (2) a delta with a small-order component: FixedBase8 must refuse the table (ok == false) and the tail must fall back to
//      variable-base products that equal k P computed directly; a healthy table must equal the variable-base product too.
// g1 / g2: points of order r (the curve generators); small: a point of small order on G1's curve ((1, 0) has order 2).
int zkhip_internal_tail_selftest(const uint64_t g1[24], const uint64_t g2[24], const uint64_t small[24], int rounds) {
  using namespace host;
  if (!g1 || !g2 || !small || rounds < 0 || rounds > 100000) return fail(ZKHIP_ERR_ARG, "tail selftest: arguments");
  auto aff = [](const uint64_t* p) { return HJac::from_affine(HFq::from_limbs(p), HFq::from_limbs(p + 12)); };
  auto same = [](const HJac& a, const HJac& b) {
    if (a.is_inf() || b.is_inf()) return a.is_inf() && b.is_inf();
    HFq ax, ay, bx, by;
    a.to_affine(ax, ay); b.to_affine(bx, by);
    uint64_t l[4][12];
    ax.to_limbs(l[0]); ay.to_limbs(l[1]); bx.to_limbs(l[2]); by.to_limbs(l[3]);
    return !memcmp(l[0], l[2], 96) && !memcmp(l[1], l[3], 96);
  };
  uint64_t r_m[6], s_m[6];
  HFr::from_u64(0x1234567u).to_limbs(r_m);
  (HFr::from_u64(0x89abcdefu) * HFr::from_u64(0xfedcba98u) * HFr::from_u64(0x76543211u)).to_limbs(s_m);
  for (int i = 0; i < rounds; i++) {
    std::unique_ptr<TailTables> tab(new TailTables());
    tab->d1.build(aff(g1));
    tab->d2.build(aff(g2));
    if (!tab->d1.ok || !tab->d2.ok) return fail(ZKHIP_ERR_STATE, "tail selftest: a prime-order table was refused");
    {
      TailPre pre;
      tail_begin(pre, g1, g2, r_m, s_m, tab.get());
    }                                            // "prove_partial failed": nobody collects the futures
    tab.reset();                                 // "the caller frees the key"
    {
      TailPre pre;
      tail_begin(pre, g1, g2, r_m, s_m);         // the variable-base form (zkhip_groth16_finish), abandoned the same way
    }
  }
  // healthy table = variable base, on a full-size scalar
  uint64_t k[6];
  HFr::from_limbs(s_m).to_canonical(k);
  FixedBase8 t1;
  t1.build(aff(g1));
  if (!t1.ok || !same(t1.mul(k), aff(g1).mul_canonical(k, 6))) return fail(ZKHIP_ERR_STATE, "tail selftest: fixed-base table differs from k P");
  // a small-order point: table refused ...
  FixedBase8 t2;
  t2.build(aff(small));
  if (t2.ok) return fail(ZKHIP_ERR_STATE, "tail selftest: the table of a small-order point was accepted");
  // ... and the tail, handed such a delta_1 through a KEY's tables (null: refused), equals the direct computation: with zero sums
  // and alpha = g1 the proof's A is g1 + r delta_1
  TailTables bad;
  bad.d1.build(aff(small)); bad.d2.build(aff(g2));
  const TailTables* usable = (bad.d1.ok && bad.d2.ok) ? &bad : nullptr;           // (what crs_tail_tables returns)
  if (usable) return fail(ZKHIP_ERR_STATE, "tail selftest: unusable tables were offered to the tail");
  uint64_t sums[180] = {0}, proof[72];
  TailPre pre;
  tail_begin(pre, small, g2, r_m, s_m, usable);
  const int rc = finish_impl(g1, g1, g2, small, g2, sums, r_m, s_m, proof, nullptr, &pre);
  if (rc != ZKHIP_OK) return rc;
  uint64_t rc_k[6];
  HFr::from_limbs(r_m).to_canonical(rc_k);
  if (!same(aff(proof), aff(g1).add(aff(small).mul_canonical(rc_k, 6)))) return fail(ZKHIP_ERR_STATE, "tail selftest: g1 + r delta_1 of a small-order delta_1 is wrong");
  return ZKHIP_OK;
}

// pinned host memory for callers without a HIP runtime of their own (source of zkhip_msm_stream_submit_host's asynchronous copies)
int zkhip_host_alloc(size_t bytes, void** out) {
  BIND_CUR();
  if (!out) return fail(ZKHIP_ERR_ARG, "null pointer");
  API_HIP(hipHostMalloc(out, bytes ? bytes : 1));
  return ZKHIP_OK;
}
int zkhip_host_free(void* p) {
  if (p) API_HIP(hipHostFree(p));
  return ZKHIP_OK;
}

// (multi_device.cpp: a worker thread's failure text travels to the thread that called the library)
void zkhip_internal_set_error(const char* msg) { snprintf(t_err, sizeof t_err, "%s", msg ? msg : ""); }

}  // extern "C"
