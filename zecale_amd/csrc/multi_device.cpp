// The GPUs of ONE node behind the library's own surfaces (the reference server is one process that owns the prover:
// aggregator_server/aggregator_server.cpp:106-118, 279-348, 390-416; its only parallelism is OpenMP, CMakeLists.txt:80-84).
//   zkhip_dispatcher     replicas (BASELINE configs[4]): one resident key + one streaming pipeline per entry of a device list; a batch
//                        goes to the pipeline with the fewest batches outstanding.  No collective: whole proofs are independent.
//   zkhip_multi_prover   one proof over a key PARTITIONED across the devices (BASELINE configs[3]): contiguous slices of
//                        the five query vectors, one prover instance and one host thread per slice (QAP map replicated, five MSMs
//                        over the slice), the 5 x 288-byte partial sums added on the host in list order, one host tail.
// A device may appear several times in a list ("0,0": two contexts on GPU 0) - how the tests rehearse N > 1 on a one-GPU box.
// Host code on the C ABI only (no device code, no torch, no collective library): a C++ server links this and nothing else.
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <string.h>

#include "../../include/zkhip.h"

extern "C" void zkhip_internal_set_error(const char* msg);

namespace {
struct DeviceGuard {          // the entry points below move the calling thread's library device while they set things up
  int saved;
  DeviceGuard() : saved(zkhip_get_device()) {}
  ~DeviceGuard() { if (saved >= 0) (void)zkhip_set_device(saved); }
};
int fail(int code, const std::string& msg) {
  zkhip_internal_set_error(msg.c_str());
  return code;
}
int check_devices(const int* devices, int n) {
  if (!devices || n < 1 || n > 64) return fail(ZKHIP_ERR_ARG, "device list: 1 .. 64 entries");
  const int count = zkhip_device_count();
  for (int i = 0; i < n; i++)
    if (devices[i] < 0 || devices[i] >= count) return fail(count ? ZKHIP_ERR_ARG : ZKHIP_ERR_NO_DEVICE, "device list: no such GPU");
  return ZKHIP_OK;
}
bool finite_point(const uint64_t* p) {               // the ABI's point at infinity is the all-zero pattern
  uint64_t acc = 0;
  for (int i = 0; i < 24; i++) acc |= p[i];
  return acc != 0;
}
// cuts[0 .. parts] of n items with weights w (finite bases: a base at infinity produces no bucket entry and costs a slice nothing):
// cut k is the first index at which the running weight reaches k W / parts - contiguous slices of equal WEIGHT, not equal length
// (a third of a real key's B query is the point at infinity).  zecale_amd/dist.py cuts_by_weight is the same rule.
void cuts_by_weight(const std::vector<uint32_t>& w, size_t parts, size_t* cuts) {
  const size_t n = w.size();
  uint64_t total = 0;
  for (uint32_t x : w) total += x;
  cuts[0] = 0; cuts[parts] = n;
  uint64_t run = 0;
  size_t i = 0;
  for (size_t k = 1; k < parts; k++) {
    const uint64_t want = (total * k + parts - 1) / parts;          // ceil(k W / parts)
    while (i < n && run < want) run += w[i++];
    cuts[k] = i;
  }
}
}  // namespace

// (zkhip_groth16_setup_slice cuts by the same rule from the key's exponents, before any point exists)
extern "C" void zkhip_internal_cuts_by_weight(const uint32_t* w, size_t n, size_t parts, size_t* cuts) {
  std::vector<uint32_t> v(w, w + n);
  cuts_by_weight(v, parts, cuts);
}

extern "C" int zkhip_key_partition(const zkhip_crs_desc* key, int parts, size_t* a_cuts, size_t* h_cuts, size_t* l_cuts) {
  if (!key || !a_cuts || !h_cuts || !l_cuts || parts < 1 || parts > 64) return fail(ZKHIP_ERR_ARG, "zkhip_key_partition: bad argument");
  if (key->n_vars < key->n_primary + 1 || key->domain_size < 1 || !key->a_query || !key->b_g2_query || !key->b_g1_query || (key->domain_size > 1 && !key->h_query))
    return fail(ZKHIP_ERR_ARG, "zkhip_key_partition: incomplete key");
  const size_t m = key->n_vars, nl = m - key->n_primary - 1, nh = key->domain_size - 1;
  if (nl && !key->l_query) return fail(ZKHIP_ERR_ARG, "zkhip_key_partition: incomplete key");
  std::vector<uint32_t> w(m);
  for (size_t i = 0; i < m; i++)
    w[i] = (uint32_t)finite_point(key->a_query + i * 24) + (uint32_t)finite_point(key->b_g2_query + i * 24) + (uint32_t)finite_point(key->b_g1_query + i * 24);
  cuts_by_weight(w, (size_t)parts, a_cuts);
  w.assign(nh, 0);
  for (size_t i = 0; i < nh; i++) w[i] = finite_point(key->h_query + i * 24);
  cuts_by_weight(w, (size_t)parts, h_cuts);
  w.assign(nl, 0);
  for (size_t i = 0; i < nl; i++) w[i] = finite_point(key->l_query + i * 24);
  cuts_by_weight(w, (size_t)parts, l_cuts);
  return ZKHIP_OK;
}

// ---------------------------------------------------------------------------------------------------- replicas
struct zkhip_dispatcher {
  std::vector<int> devices;
  std::vector<zkhip_crs*> crs;
  std::vector<zkhip_pipeline*> pipes;
  std::unique_ptr<std::atomic<size_t>[]> outstanding, submitted;
  std::atomic<size_t> rr{0};
};

extern "C" {

void zkhip_dispatcher_free(zkhip_dispatcher* d) {
  if (!d) return;
  for (zkhip_pipeline* p : d->pipes) zkhip_aggregator_pipeline_free(p);
  for (zkhip_crs* c : d->crs) zkhip_crs_free(c);
  delete d;
}

int zkhip_dispatcher_new(zkhip_aggregator* a, const zkhip_crs_desc* key, const zkhip_key_opts* opts, const int* devices, int n_devices,
                         int gpu_slots, int witness_workers, unsigned flags, zkhip_dispatcher** out) {
  if (!a || !key || !out) return fail(ZKHIP_ERR_ARG, "null pointer");
  int rc = check_devices(devices, n_devices);
  if (rc != ZKHIP_OK) return rc;
  DeviceGuard guard;
  zkhip_dispatcher* d = new zkhip_dispatcher();
  d->outstanding.reset(new std::atomic<size_t>[n_devices]);
  d->submitted.reset(new std::atomic<size_t>[n_devices]);
  for (int i = 0; i < n_devices; i++) { d->outstanding[i] = 0; d->submitted[i] = 0; }
  for (int i = 0; i < n_devices && rc == ZKHIP_OK; i++) {
    zkhip_crs* c = nullptr;
    zkhip_pipeline* p = nullptr;
    if ((rc = zkhip_init(devices[i])) == ZKHIP_OK && (rc = zkhip_set_device(devices[i])) == ZKHIP_OK &&
        (rc = zkhip_crs_upload_ex(key, opts, &c)) == ZKHIP_OK) {
      d->crs.push_back(c);
      if ((rc = zkhip_aggregator_pipeline_new_ex(a, c, gpu_slots, witness_workers, flags, &p)) == ZKHIP_OK) d->pipes.push_back(p);
    }
    d->devices.push_back(devices[i]);
  }
  if (rc != ZKHIP_OK) {
    const std::string why = zkhip_last_error();          // (freeing the partial set-up may overwrite the thread's error text)
    zkhip_dispatcher_free(d);
    return fail(rc, why);
  }
  *out = d;
  return ZKHIP_OK;
}

int zkhip_dispatcher_size(const zkhip_dispatcher* d) { return d ? (int)d->pipes.size() : 0; }

int zkhip_dispatcher_submit(zkhip_dispatcher* d, const uint64_t* nested_vk, const uint64_t* nested_proofs, const uint64_t* nested_inputs,
                            const uint64_t r[6], const uint64_t s[6], uint64_t* ticket) {
  if (!d || !ticket) return fail(ZKHIP_ERR_ARG, "null pointer");
  // least loaded first; ties go round the list so that an idle node spreads its first batches over every GPU
  const size_t n = d->pipes.size(), start = d->rr.fetch_add(1) % n;
  size_t best = start, best_load = d->outstanding[start].load();
  for (size_t k = 1; k < n; k++) {
    const size_t i = (start + k) % n, load = d->outstanding[i].load();
    if (load < best_load) { best = i; best_load = load; }
  }
  uint64_t inner = 0;
  d->outstanding[best]++;
  const int rc = zkhip_aggregator_pipeline_submit(d->pipes[best], nested_vk, nested_proofs, nested_inputs, r, s, &inner);   // (blocks while that pipeline is full)
  if (rc != ZKHIP_OK) { d->outstanding[best]--; return rc; }
  if (inner >> 56) { d->outstanding[best]--; return fail(ZKHIP_ERR_STATE, "ticket space exhausted"); }
  d->submitted[best]++;
  *ticket = ((uint64_t)(best + 1) << 56) | inner;
  return ZKHIP_OK;
}

int zkhip_dispatcher_wait(zkhip_dispatcher* d, uint64_t ticket, uint64_t* primary_inputs, uint64_t proof_affine[72]) {
  if (!d || !proof_affine) return fail(ZKHIP_ERR_ARG, "null pointer");
  const size_t idx = (size_t)(ticket >> 56);
  if (idx < 1 || idx > d->pipes.size()) return fail(ZKHIP_ERR_NO_TICKET, "no such ticket");
  const int rc = zkhip_aggregator_pipeline_wait(d->pipes[idx - 1], ticket & (((uint64_t)1 << 56) - 1), primary_inputs, proof_affine);
  // a known ticket is consumed whatever its batch's result was (ZKHIP_ERR_ARG = the batch itself failed: a degenerate or malformed
  // nested proof); only a ticket the pipeline never issued, or one whose pipeline is shutting down, leaves the count alone
  if (rc == ZKHIP_ERR_NO_TICKET) return fail(rc, "no such ticket");
  if (rc != ZKHIP_ERR_STATE) d->outstanding[idx - 1]--;
  return rc;
}

int zkhip_dispatcher_register_app(zkhip_dispatcher* d, const uint64_t* nested_vk) {
  if (!d || !nested_vk) return fail(ZKHIP_ERR_ARG, "null pointer");
  for (zkhip_pipeline* p : d->pipes) {
    const int rc = zkhip_aggregator_pipeline_register_app(p, nested_vk);
    if (rc != ZKHIP_OK) return rc;
  }
  return ZKHIP_OK;
}

// batches each entry of the device list has been given so far (n = zkhip_dispatcher_size values)
int zkhip_dispatcher_stats(const zkhip_dispatcher* d, size_t* submitted_per_entry) {
  if (!d || !submitted_per_entry) return fail(ZKHIP_ERR_ARG, "null pointer");
  for (size_t i = 0; i < d->pipes.size(); i++) submitted_per_entry[i] = d->submitted[i].load();
  return ZKHIP_OK;
}

// batches each entry still owes a collector (what submit's least-loaded routing looks at)
int zkhip_dispatcher_outstanding(const zkhip_dispatcher* d, size_t* per_entry) {
  if (!d || !per_entry) return fail(ZKHIP_ERR_ARG, "null pointer");
  for (size_t i = 0; i < d->pipes.size(); i++) per_entry[i] = d->outstanding[i].load();
  return ZKHIP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------- partitioned key
struct zkhip_multi_prover {
  std::vector<int> devices;
  std::vector<zkhip_crs*> crs;
  std::vector<zkhip_prover*> prov;
  uint64_t alpha_g1[24], beta_g1[24], beta_g2[24], delta_g1[24], delta_g2[24];
  size_t n_vars = 0;
  std::mutex mu;                 // one proof at a time per instance (the slices' provers are not re-entrant)
  double last_ms[3] = {0, 0, 0};   // slowest slice, host additions, host tail
};

extern "C" {

void zkhip_multi_prover_free(zkhip_multi_prover* mp) {
  if (!mp) return;
  for (zkhip_prover* p : mp->prov) zkhip_prover_free(p);
  for (zkhip_crs* c : mp->crs) zkhip_crs_free(c);
  delete mp;
}

int zkhip_multi_prover_new(const zkhip_crs_desc* key, const zkhip_r1cs_desc* cs, const zkhip_key_opts* opts, const int* devices, int n_devices,
                           zkhip_multi_prover** out) {
  if (!key || !cs || !out || !key->alpha_g1 || !key->beta_g1 || !key->beta_g2 || !key->delta_g1 || !key->delta_g2) return fail(ZKHIP_ERR_ARG, "null pointer");
  int rc = check_devices(devices, n_devices);
  if (rc != ZKHIP_OK) return rc;
  if (key->n_vars != cs->n_vars || key->n_primary != cs->n_primary || key->n_vars < key->n_primary + 1 || key->domain_size < 1)
    return fail(ZKHIP_ERR_ARG, "proving key and constraint system do not match");
  // the key names the evaluation domain (zkhip.h): it must be one, and hold the system's n + l + 1 interpolation points
  if (!zkhip_domain_is_valid(key->domain_size) || key->domain_size < cs->n_constraints + cs->n_primary + 1)
    return fail(ZKHIP_ERR_ARG, "proving key: its evaluation domain is not a power of two or 2^k + 2^r, or too small for the constraint system");
  DeviceGuard guard;
  zkhip_multi_prover* mp = new zkhip_multi_prover();
  mp->n_vars = key->n_vars;
  memcpy(mp->alpha_g1, key->alpha_g1, 192); memcpy(mp->beta_g1, key->beta_g1, 192); memcpy(mp->beta_g2, key->beta_g2, 192);
  memcpy(mp->delta_g1, key->delta_g1, 192); memcpy(mp->delta_g2, key->delta_g2, 192);
  // slices of equal FINITE terms per query group (zkhip_key_partition)
  std::vector<size_t> ac(n_devices + 1), hc(n_devices + 1), lc(n_devices + 1);
  rc = zkhip_key_partition(key, n_devices, ac.data(), hc.data(), lc.data());
  for (int i = 0; i < n_devices && rc == ZKHIP_OK; i++) {
    const size_t a0 = ac[i], a1 = ac[i + 1], h0 = hc[i], h1 = hc[i + 1], l0 = lc[i], l1 = lc[i + 1];
    zkhip_crs* c = nullptr;
    zkhip_prover* p = nullptr;
    if ((rc = zkhip_init(devices[i])) == ZKHIP_OK && (rc = zkhip_set_device(devices[i])) == ZKHIP_OK &&
        (rc = zkhip_crs_upload_slice_ex(key, a0, a1 - a0, h0, h1 - h0, l0, l1 - l0, opts, &c)) == ZKHIP_OK) {
      mp->crs.push_back(c);
      if ((rc = zkhip_prover_new_slice(c, cs, a0, h0, l0, &p)) == ZKHIP_OK) mp->prov.push_back(p);
    }
    mp->devices.push_back(devices[i]);
  }
  if (rc != ZKHIP_OK) {
    const std::string why = zkhip_last_error();
    zkhip_multi_prover_free(mp);
    return fail(rc, why);
  }
  *out = mp;
  return ZKHIP_OK;
}

int zkhip_multi_prover_size(const zkhip_multi_prover* mp) { return mp ? (int)mp->prov.size() : 0; }

int zkhip_multi_prover_prove(zkhip_multi_prover* mp, const uint64_t* z, const uint64_t r[6], const uint64_t s[6], uint64_t proof_affine[72]) {
  if (!mp || !z || !r || !s || !proof_affine) return fail(ZKHIP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> lk(mp->mu);
  const size_t n = mp->prov.size();
  struct Part { uint64_t sums[180]; int rc = ZKHIP_OK; std::string err; double ms = 0; };
  std::vector<Part> parts(n);
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(now() - t0).count(); };
  auto run = [&](size_t i) {                 // (a thread that never called zkhip_init: the prover handle carries its device)
    const auto t0 = now();
    parts[i].rc = zkhip_prover_prove_partial(mp->prov[i], z, parts[i].sums);
    if (parts[i].rc != ZKHIP_OK) parts[i].err = zkhip_last_error();
    parts[i].ms = ms_since(t0);
  };
  std::vector<std::thread> th;
  for (size_t i = 1; i < n; i++) th.emplace_back(run, i);
  run(0);
  for (auto& t : th) t.join();
  double slowest = 0;
  for (size_t i = 0; i < n; i++) {
    if (parts[i].rc != ZKHIP_OK) return fail(parts[i].rc, "slice " + std::to_string(i) + " (device " + std::to_string(mp->devices[i]) + "): " + parts[i].err);
    if (parts[i].ms > slowest) slowest = parts[i].ms;
  }
  // the "all-reduce" of SURVEY 8e inside one process: 5 x 288 bytes per slice, added in list order (exact arithmetic)
  auto t0 = now();
  uint64_t total[180];
  memcpy(total, parts[0].sums, sizeof total);
  for (size_t i = 1; i < n; i++)
    for (int k = 0; k < 5; k++) {
      uint64_t acc[36];
      int rc = zkhip_jac_add(total + 36 * k, parts[i].sums + 36 * k, acc);
      if (rc != ZKHIP_OK) return rc;
      memcpy(total + 36 * k, acc, sizeof acc);
    }
  const double add_ms = ms_since(t0);
  t0 = now();
  const int rc = zkhip_groth16_finish(mp->alpha_g1, mp->beta_g1, mp->beta_g2, mp->delta_g1, mp->delta_g2, total, r, s, proof_affine);
  mp->last_ms[0] = slowest; mp->last_ms[1] = add_ms; mp->last_ms[2] = ms_since(t0);
  return rc;
}

// milliseconds of the last proof: [0] the slowest slice (upload z + QAP map + five MSMs), [1] host additions, [2] host tail
int zkhip_multi_prover_timings(zkhip_multi_prover* mp, double out_ms[3]) {
  if (!mp || !out_ms) return fail(ZKHIP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> lk(mp->mu);
  memcpy(out_ms, mp->last_ms, sizeof mp->last_ms);
  return ZKHIP_OK;
}

}  // extern "C"
