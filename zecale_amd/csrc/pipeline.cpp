// Streaming form of the wrapping prover: the shape of the reference's server loop (aggregator_server.cpp:300-420
// GenerateAggregatedTransaction -> aggregator_circuit::prove, one batch at a time, CPU) re-cut for one MI355X.
//
// A batch goes through three stages with very different resource needs:
//   witness   host only: ~1500 serial field inversions inside the in-circuit pairing, ~12 ms on 3 cores;
//   prove     GPU: upload z, QAP map, five MSMs.  At this circuit's size (50k constraints) every MSM phase after the
//             accumulation is a chain of short latency-bound launches, so ONE proof leaves most of the chip idle;
//   tail      host: three 377-bit scalar multiplications + affine normalisation, ~2 ms.
// The pipeline keeps `witness_workers` witnesses and `gpu_slots` proofs in flight: each GPU slot is a zkhip_prover
// (own streams, own MSM work space, own QAP buffers) driven by its own host thread, so the accumulation kernels of one
// proof fill the gaps in the reduction chains of another and the host stages overlap the device ones.
// Built only on the C ABI (zkhip_aggregator_witness, zkhip_prover_*): a host-side scheduler, no device code.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <set>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "../../include/zkhip.h"

namespace {
// One registered application of the pipeline (zkhip_aggregator_app: the constants of its nested key on this pipeline's proving key).
// Built by the first worker that meets the key (or by zkhip_aggregator_pipeline_register_app); batches that arrive while it is being
// built take the plain path.
struct AppEntry {
  zkhip_aggregator_app* app = nullptr;
  int state = 0;                  // 0: being built, 1: ready, 2: could not be built (its batches take the plain path and report their own errors)
  uint64_t last_used = 0;         // the pipeline's application clock at the last batch (the table evicts its least recently used entry)
  // An entry leaves the table when it is evicted or when the pipeline is freed; batches in flight keep it alive through their
  // shared_ptr, so the handle is freed by whoever drops the last reference - on whatever thread, whose library device is restored.
  ~AppEntry() {
    if (!app) return;
    const int saved = zkhip_get_device();
    zkhip_aggregator_app_free(app);
    if (saved >= 0) (void)zkhip_set_device(saved);
  }
};
struct Job {
  std::shared_ptr<AppEntry> app;  // set: the assignment is MASKED and the proof goes through zkhip_prover_prove_app[_dev]
  uint64_t id = 0;
  std::vector<uint64_t> vk, proofs, inputs, z;
  void* d_z = nullptr;            // GPU witness: the assignment in device memory (z then holds the primary inputs only)
  int slab = -1;                  // ... inside this slab of the pipeline
  uint64_t r[6], s[6], proof[72];
  int rc = ZKHIP_OK;
  bool done = false;
};
}  // namespace

struct zkhip_pipeline {
  std::atomic<uint64_t> st_wit_ns{0}, st_wit_n{0}, st_slot_wait_ns{0}, st_prove_ns{0}, st_prove_n{0};   // ZKHIP_PIPELINE_STATS: printed when the pipeline is freed
  zkhip_aggregator* agg = nullptr;
  const zkhip_crs* crs = nullptr;
  bool app_cache = true;                         // per-application constants (off: ZKHIP_PIPELINE_NO_APP_CACHE)
  std::mutex mu_apps;
  std::map<std::vector<uint64_t>, std::shared_ptr<AppEntry>> apps;      // by nested key: being built or ready
  std::set<std::vector<uint64_t>> bad_apps;      // keys that could not be given a handle (off-curve, degenerate): remembered, NOT counted against the table
  uint64_t app_clock = 0;
  bool app_evict_logged = false;
  std::atomic<uint64_t> st_app_hits{0};
  // assignment buffers (n_vars x 48 bytes: 2-4 MB) are reused from job to job: allocated per job they are mmap'ed and munmap'ed by the
  // allocator, and ten generator threads doing that held the process in the kernel (aggregator.cpp: SectionPool)
  std::mutex mu_z;
  std::vector<std::vector<uint64_t>> z_free;
  void take_z(std::vector<uint64_t>& z) {
    {
      std::lock_guard<std::mutex> lk(mu_z);
      if (!z_free.empty()) { z.swap(z_free.back()); z_free.pop_back(); }
    }
    z.resize(n_vars * 6);                          // (the generators write every entry)
  }
  void give_z(std::vector<uint64_t>& z) {
    std::lock_guard<std::mutex> lk(mu_z);
    if (z_free.size() < 64) { z_free.emplace_back(); z_free.back().swap(z); }
  }
  size_t n_vars = 0, n_primary = 0, vk_words = 0, proofs_words = 0, inputs_words = 0;
  std::vector<zkhip_prover*> provers;
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_wit, cv_gpu, cv_done, cv_room, cv_host;
  std::deque<std::shared_ptr<Job>> q_wit, q_gpu;
  std::deque<std::shared_ptr<Job>> q_host;         // GPU-witness mode: batches the device could not witness (degenerate nested points, a failed
                                                   // launch) wait here for the host generator thread - the batcher does not stop for them
  std::map<uint64_t, std::shared_ptr<Job>> jobs;   // submitted and not yet collected
  size_t max_unfinished = 0, unfinished = 0;     // back-pressure counts batches not yet proved (finished ones wait for their collector)
  uint64_t next_id = 1;
  bool stop = false;
  // witness generation on the GPU (ZKHIP_PIPELINE_GPU_WITNESS): device buffers for the assignments in flight
  bool gpu_witness = false;
  bool hybrid = false;                            // host generators wait on cv_wit beside the batchers: submit wakes all of them
  int device = 0;
  size_t wit_batch = 16;                          // batches per witness launch (one workgroup each: 16 take as long as one)
  struct Slab { void* base = nullptr; int outstanding = 0; };
  std::vector<Slab> slabs;                        // wit_batch assignments each
  std::vector<int> slab_free;
  std::condition_variable cv_buf;
};

namespace {

void finish_failed(zkhip_pipeline* p, const std::shared_ptr<Job>& j, int rc) {       // p->mu held
  j->rc = rc; j->done = true;
  p->unfinished--;
  p->cv_done.notify_all();
  p->cv_room.notify_one();
}

// The witness generators are the pipeline's CPU-heavy threads (7 ms on three threads per batch); the prover threads only enqueue
// launches and sleep on events, but the GPU waits for them when they do not get a core at once.  The generators therefore run at a
// lower scheduling priority (a per-thread nice value; the threads a generator spawns inherit it): ZKHIP_WITNESS_NICE, default 10.
void lower_priority() {
  static const int nice_by = [] { const char* e = getenv("ZKHIP_WITNESS_NICE"); int v = e ? atoi(e) : 10; return v < 0 || v > 19 ? 10 : v; }();
  if (nice_by) (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), nice_by);
}

constexpr size_t MAX_APPS = 32;          // per pipeline: each holds ~15 MB on the host and ~13 MB on the device
constexpr size_t MAX_BAD_APPS = 256;     // negative cache (keys only: a few hundred bytes each)

// The application of a batch's nested key, built on first sight.  Null: no cache, the handle is still being built by another
// thread, every entry of a full table is still being built, or the key cannot have a handle - the batch then takes the plain path.
// A full table evicts its least recently used READY entry (round 6, ADVICE r5: until then the 33rd key of a pipeline's life - valid
// or not - and every key after it silently took the plain path for good, and failed builds held their places for ever).
std::shared_ptr<AppEntry> find_app(zkhip_pipeline* p, const std::vector<uint64_t>& vk, bool wait_for_build = false) {
  if (!p->app_cache) return nullptr;
  std::shared_ptr<AppEntry> e, evicted;
  {
    std::lock_guard<std::mutex> lk(p->mu_apps);
    auto it = p->apps.find(vk);
    if (it != p->apps.end()) {
      if (it->second->state == 1) { it->second->last_used = ++p->app_clock; return it->second; }
      if (!wait_for_build) return nullptr;
      e = it->second;
    } else {
      if (p->bad_apps.count(vk)) return nullptr;
      if (p->apps.size() >= MAX_APPS) {
        auto lru = p->apps.end();
        for (auto jt = p->apps.begin(); jt != p->apps.end(); ++jt)
          if (jt->second->state == 1 && (lru == p->apps.end() || jt->second->last_used < lru->second->last_used)) lru = jt;
        if (lru == p->apps.end()) return nullptr;             // (all being built)
        if (!p->app_evict_logged) {
          p->app_evict_logged = true;
          fprintf(stderr, "zkhip pipeline: more than %zu applications in use - evicting the least recently used handle (logged once)\n", MAX_APPS);
        }
        evicted = lru->second;                                  // (freed below, outside the lock, unless batches in flight still hold it)
        p->apps.erase(lru);
      }
      e = std::make_shared<AppEntry>();
      p->apps[vk] = e;
      wait_for_build = false;
    }
  }
  evicted.reset();
  if (wait_for_build) {                         // (registration of a key a worker is already building: poll, it takes ~0.2 s)
    for (int i = 0; i < 2000; i++) {
      { std::lock_guard<std::mutex> lk(p->mu_apps); if (e->state) break; }
      usleep(1000);
    }
    std::lock_guard<std::mutex> lk(p->mu_apps);
    return e->state == 1 ? e : nullptr;
  }
  // (a registration comes from the caller's thread - a gRPC handler: its library device is put back)
  const int saved = zkhip_get_device();
  zkhip_aggregator_app* app = nullptr;
  const int rc = (zkhip_set_device(p->device) == ZKHIP_OK) ? zkhip_aggregator_app_new(p->agg, p->crs, vk.data(), &app) : ZKHIP_ERR_STATE;
  if (saved >= 0 && saved != p->device) (void)zkhip_set_device(saved);
  std::lock_guard<std::mutex> lk(p->mu_apps);
  if (rc == ZKHIP_OK) {
    e->app = app;
    e->state = 1;
    e->last_used = ++p->app_clock;
    return e;
  }
  e->state = 2;                                 // (a registration polling this entry sees the failure)
  auto it = p->apps.find(vk);
  if (it != p->apps.end() && it->second == e) p->apps.erase(it);
  if (p->bad_apps.size() >= MAX_BAD_APPS) p->bad_apps.clear();
  p->bad_apps.insert(vk);
  return nullptr;
}

// overflow_only (a HYBRID pipeline's host generators): take a batch only while MORE than one full witness launch is queued - what
// the batchers cannot start on anyway.  A host generator that takes every batch it sees leaves the batchers launching a few batches
// at a time, and a launch of 3 takes as long as a launch of 16 (nine inputs per nested proof, round 5: 149 proofs/s against 298
// from the batchers alone).
void witness_loop(zkhip_pipeline* p, bool overflow_only = false) {
  pthread_setname_np(pthread_self(), "zk-witness");
  lower_priority();
  for (;;) {
    std::shared_ptr<Job> j;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      if (overflow_only) {
        p->cv_wit.wait(lk, [&] { return p->stop || p->q_wit.size() > p->wit_batch; });
      } else
        p->cv_wit.wait(lk, [&] { return p->stop || !p->q_wit.empty(); });
      if (p->stop) return;
      j = p->q_wit.front();
      p->q_wit.pop_front();
    }
    int wf = 0;
    int rc = zkhip_aggregator_check_inputs(p->agg, j->vk.data(), j->proofs.data(), &wf);      // off-curve points: no proof exists
    if (rc == ZKHIP_OK && !wf) rc = ZKHIP_ERR_ARG;
    if (rc == ZKHIP_OK) {
      p->take_z(j->z);
      j->app = find_app(p, j->vk);
      const auto t0 = std::chrono::steady_clock::now();
      // a registered application: the proof sections only, MASKED assignment (the key's hash, lines and doubling chains are its constants)
      rc = j->app ? zkhip_aggregator_witness_app(j->app->app, j->proofs.data(), j->inputs.data(), j->z.data())
                  : zkhip_aggregator_witness(p->agg, j->vk.data(), j->proofs.data(), j->inputs.data(), j->z.data());
      if (j->app) p->st_app_hits++;
      p->st_wit_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      p->st_wit_n++;
    }
    std::lock_guard<std::mutex> lk(p->mu);
    if (rc != ZKHIP_OK) finish_failed(p, j, rc);
    else {
      p->q_gpu.push_back(j);
      p->cv_gpu.notify_one();
    }
  }
}

// ZKHIP_PIPELINE_GPU_WITNESS: a batcher thread takes up to wit_batch queued batches, generates their assignments with ONE launch
// sequence of the witness kernel (a workgroup per batch) into a slab of device buffers and hands them to the provers; the
// assignments never leave the device.  A degenerate batch (the device met an inversion of zero) goes through the host generator.
void gpu_witness_loop(zkhip_pipeline* p) {
  pthread_setname_np(pthread_self(), "zk-gpu-witness");
  zkhip_gpu_witness* gw = nullptr;
  if (zkhip_set_device(p->device) != ZKHIP_OK || zkhip_gpu_witness_new_batched(p->agg, p->wit_batch, &gw) != ZKHIP_OK) { witness_loop(p, false); return; }
  std::vector<std::shared_ptr<Job>> jobs, good;
  std::vector<const uint64_t*> vks, prs, ins;
  std::vector<uint64_t> prim(p->wit_batch * p->n_primary * 6);
  std::vector<int> deg(p->wit_batch);
  for (;;) {
    jobs.clear(); good.clear();
    int slab = -1;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_wit.wait(lk, [&] { return p->stop || !p->q_wit.empty(); });
      if (p->stop) break;
      p->cv_buf.wait(lk, [&] { return p->stop || !p->slab_free.empty(); });
      if (p->stop) break;
      // one launch = batches of ONE nested key (an application's program has its key folded in): the oldest batch decides, the
      // others of its key follow from anywhere in the queue, the rest wait for the next launch
      if (!p->q_wit.empty()) {
        const std::vector<uint64_t> key = p->q_wit.front()->vk;
        for (auto it = p->q_wit.begin(); it != p->q_wit.end() && jobs.size() < p->wit_batch;) {
          if (!p->app_cache || (*it)->vk == key) { jobs.push_back(*it); it = p->q_wit.erase(it); }
          else ++it;
        }
      }
      if (jobs.empty()) continue;                     // another batcher took them while this one waited for a slab
      slab = p->slab_free.back(); p->slab_free.pop_back();
    }
    for (auto& j : jobs) {
      int wf = 0;
      int rc = zkhip_aggregator_check_inputs(p->agg, j->vk.data(), j->proofs.data(), &wf);
      if (rc == ZKHIP_OK && !wf) rc = ZKHIP_ERR_ARG;
      if (rc == ZKHIP_OK) good.push_back(j);
      else { std::lock_guard<std::mutex> lk(p->mu); finish_failed(p, j, rc); }
    }
    // (after the input check: a key with a point off its curve never reaches the handle builder from here)
    std::shared_ptr<AppEntry> app = good.empty() ? nullptr : find_app(p, good[0]->vk);
    int rc = ZKHIP_OK;
    if (!good.empty()) {
      vks.clear(); prs.clear(); ins.clear();
      for (auto& j : good) { vks.push_back(j->vk.data()); prs.push_back(j->proofs.data()); ins.push_back(j->inputs.data()); }
      rc = app ? zkhip_gpu_witness_run_batched_app(gw, app->app, good.size(), prs.data(), ins.data(), p->slabs[slab].base, prim.data(), deg.data())
               : zkhip_gpu_witness_run_batched(gw, good.size(), vks.data(), prs.data(), ins.data(), p->slabs[slab].base, prim.data(), deg.data());
      if (app) p->st_app_hits += good.size();
    }
    int on_device = 0;
    for (size_t i = 0; i < good.size(); i++) {
      auto& j = good[i];
      if (rc == ZKHIP_OK && !deg[i]) {
        j->z.assign(prim.begin() + i * p->n_primary * 6, prim.begin() + (i + 1) * p->n_primary * 6);
        j->d_z = (char*)p->slabs[slab].base + i * p->n_vars * 48;
        j->slab = slab;
        j->app = app;
        on_device++;
      }
    }
    // the on-device jobs go to the provers and the slab is accounted for at once; what the device could not witness goes to the
    // host generator THREAD (crafted inputs - ABC_1 == ABC_0 - must not be able to stall the launches of everybody else's batches)
    std::lock_guard<std::mutex> lk(p->mu);
    p->slabs[slab].outstanding = on_device;
    if (on_device == 0) { p->slab_free.push_back(slab); p->cv_buf.notify_one(); }
    for (auto& j : good) {
      if (j->d_z) { p->q_gpu.push_back(j); p->cv_gpu.notify_one(); }
      else { p->q_host.push_back(j); p->cv_host.notify_one(); }
    }
  }
  zkhip_gpu_witness_free(gw);
}

// GPU-witness mode: the host generator for the batches the device handed back.
void host_fallback_loop(zkhip_pipeline* p) {
  pthread_setname_np(pthread_self(), "zk-wit-host");
  for (;;) {
    std::shared_ptr<Job> j;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_host.wait(lk, [&] { return p->stop || !p->q_host.empty(); });
      if (p->stop) return;
      j = p->q_host.front();
      p->q_host.pop_front();
    }
    p->take_z(j->z);
    j->app = nullptr;                                  // (the full assignment, the plain proof)
    const int rc = zkhip_aggregator_witness(p->agg, j->vk.data(), j->proofs.data(), j->inputs.data(), j->z.data());
    std::lock_guard<std::mutex> lk(p->mu);
    if (rc != ZKHIP_OK) finish_failed(p, j, rc);
    else { p->q_gpu.push_back(j); p->cv_gpu.notify_one(); }
  }
}

void gpu_loop(zkhip_pipeline* p, zkhip_prover* pr) {
  pthread_setname_np(pthread_self(), "zk-prover");
  for (;;) {
    std::shared_ptr<Job> j;
    const auto t0 = std::chrono::steady_clock::now();
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_gpu.wait(lk, [&] { return p->stop || !p->q_gpu.empty(); });
      if (p->stop) return;
      j = p->q_gpu.front();
      p->q_gpu.pop_front();
    }
    const auto t1 = std::chrono::steady_clock::now();
    int rc;
    if (j->app) rc = j->d_z ? zkhip_prover_prove_app_dev(pr, j->app->app, j->d_z, j->r, j->s, j->proof) : zkhip_prover_prove_app(pr, j->app->app, j->z.data(), j->r, j->s, j->proof);
    else rc = j->d_z ? zkhip_prover_prove_dev(pr, j->d_z, j->r, j->s, j->proof) : zkhip_prover_prove(pr, j->z.data(), j->r, j->s, j->proof);
    p->st_slot_wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
    p->st_prove_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t1).count();
    p->st_prove_n++;
    std::lock_guard<std::mutex> lk(p->mu);
    j->rc = rc; j->done = true;
    if (j->d_z) {
      j->d_z = nullptr;
      if (--p->slabs[j->slab].outstanding == 0) { p->slab_free.push_back(j->slab); p->cv_buf.notify_all(); }
    } else {
      std::vector<uint64_t> prim(j->z.begin() + 6, j->z.begin() + 6 + p->n_primary * 6);   // keep the primary inputs, drop the 2.4 MB witness
      j->z.swap(prim);
      p->give_z(prim);
    }
    p->unfinished--;
    p->cv_done.notify_all();
    p->cv_room.notify_one();
  }
}

}  // namespace

extern "C" {

int zkhip_aggregator_pipeline_new(zkhip_aggregator* a, const zkhip_crs* crs, int gpu_slots, int witness_workers, zkhip_pipeline** out) {
  return zkhip_aggregator_pipeline_new_ex(a, crs, gpu_slots, witness_workers, 0, out);
}

int zkhip_aggregator_pipeline_new_ex(zkhip_aggregator* a, const zkhip_crs* crs, int gpu_slots, int witness_workers, unsigned flags, zkhip_pipeline** out) {
  if (!a || !crs || !out || gpu_slots < 1 || gpu_slots > 64 || witness_workers < 1 || witness_workers > 64) return ZKHIP_ERR_ARG;
  zkhip_r1cs_desc desc;
  int rc = zkhip_aggregator_get_r1cs(a, &desc);
  if (rc != ZKHIP_OK) return rc;
  zkhip_pipeline* p = new zkhip_pipeline();
  p->agg = a; p->crs = crs;
  p->app_cache = (flags & ZKHIP_PIPELINE_NO_APP_CACHE) == 0 && !getenv("ZKHIP_NO_APP_CACHE");
  p->n_vars = desc.n_vars; p->n_primary = desc.n_primary;
  const size_t np = zkhip_aggregator_num_proofs(a), k = zkhip_aggregator_inputs_per_proof(a);
  p->vk_words = 60 + 12 * (k + 1); p->proofs_words = 48 * np; p->inputs_words = 6 * k * np;
  p->max_unfinished = (size_t)4 * (size_t)(gpu_slots + witness_workers);
  p->gpu_witness = (flags & ZKHIP_PIPELINE_GPU_WITNESS) != 0;
  if (const char* e = getenv("ZKHIP_WIT_BATCH")) {            // witnesses per launch of the GPU generator (tuning knob)
    const int v = atoi(e);
    if (v >= 1 && v <= 64) p->wit_batch = (size_t)v;
  }
  p->device = zkhip_crs_device(crs);
  if (p->gpu_witness) {
    if (witness_workers > 8) witness_workers = 8;     // batcher threads: each keeps one launch of wit_batch witnesses in flight
    if (zkhip_set_device(p->device) != ZKHIP_OK) { delete p; return ZKHIP_ERR_STATE; }
    const size_t n_slabs = (size_t)witness_workers + (size_t)gpu_slots / 2 + 2;
    for (size_t i = 0; i < n_slabs; i++) {
      void* b = nullptr;
      if (zkhip_device_alloc(p->wit_batch * p->n_vars * 48, &b) != ZKHIP_OK) {
        for (auto& q : p->slabs) zkhip_device_free(q.base);
        delete p;
        return ZKHIP_ERR_HIP;
      }
      zkhip_pipeline::Slab sl; sl.base = b;
      p->slabs.push_back(sl);
      p->slab_free.push_back((int)i);
    }
    p->max_unfinished = p->wit_batch * n_slabs + (size_t)4 * gpu_slots;      // deep enough to fill the witness launches
  }
  for (int i = 0; i < gpu_slots; i++) {
    zkhip_prover* pr = nullptr;
    rc = zkhip_prover_new(crs, &desc, &pr);
    if (rc == ZKHIP_OK) (void)zkhip_prover_set_streaming(pr, 1);
    if (rc != ZKHIP_OK) {
      for (zkhip_prover* q : p->provers) zkhip_prover_free(q);
      for (auto& q : p->slabs) zkhip_device_free(q.base);
      delete p;
      return rc;
    }
    p->provers.push_back(pr);
  }
  // the provers' busy streams first, one after the other, then their idle ones: evenly over the runtime's hardware queues
  // (zkhip_prover_create_streams; created by the provers' own threads they land as the race decides)
  for (int which = 0; which < 2; which++)
    for (zkhip_prover* pr : p->provers) (void)zkhip_prover_create_streams(pr, which);
  for (int i = 0; i < witness_workers; i++) { if (p->gpu_witness) p->threads.emplace_back(gpu_witness_loop, p); else p->threads.emplace_back(witness_loop, p, false); }
  if (p->gpu_witness) p->threads.emplace_back(host_fallback_loop, p);
  // HYBRID (ZKHIP_PIPELINE_HYBRID_WITNESS with ZKHIP_PIPELINE_GPU_WITNESS): host generators beside the GPU batchers, on the same queue - a
  // host worker takes one batch at a time - and only from what exceeds one full witness launch in the queue (witness_loop) - a batcher up
  // to wit_batch of one key: the host generators work when the batchers are behind, and idle otherwise.
  if (p->gpu_witness && (flags & ZKHIP_PIPELINE_HYBRID_WITNESS)) {
    int hw = 2;
    if (const char* e = getenv("ZKHIP_HYBRID_HOST_WORKERS")) { const int v = atoi(e); if (v >= 1 && v <= 16) hw = v; }
    p->hybrid = true;
    for (int i = 0; i < hw; i++) p->threads.emplace_back(witness_loop, p, true);
  }
  for (zkhip_prover* pr : p->provers) p->threads.emplace_back(gpu_loop, p, pr);
  *out = p;
  return ZKHIP_OK;
}

void zkhip_aggregator_pipeline_free(zkhip_pipeline* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->stop = true;
  }
  p->cv_wit.notify_all(); p->cv_gpu.notify_all(); p->cv_done.notify_all(); p->cv_room.notify_all(); p->cv_buf.notify_all(); p->cv_host.notify_all();
  for (auto& t : p->threads) t.join();
  if (getenv("ZKHIP_PIPELINE_STATS") && p->st_prove_n)
    fprintf(stderr, "zkhip pipeline: %llu proofs, %.2f ms in a prover each, provers waited %.2f ms per proof for an assignment; %llu host witnesses of %.2f ms; "
                    "%llu batches used an application's constants (%zu applications)\n",
            (unsigned long long)p->st_prove_n.load(), p->st_prove_ns / 1e6 / p->st_prove_n, p->st_slot_wait_ns / 1e6 / p->st_prove_n,
            (unsigned long long)p->st_wit_n.load(), p->st_wit_n ? p->st_wit_ns / 1e6 / p->st_wit_n : 0.0, (unsigned long long)p->st_app_hits.load(), p->apps.size());
  for (zkhip_prover* q : p->provers) zkhip_prover_free(q);
  p->apps.clear();                               // (~AppEntry frees the handles)
  for (auto& q : p->slabs) zkhip_device_free(q.base);
  delete p;
}

// RegisterApplication (aggregator_server.cpp:170-235): the application's constants are computed NOW instead of by the first batch
int zkhip_aggregator_pipeline_register_app(zkhip_pipeline* p, const uint64_t* nested_vk) {
  if (!p || !nested_vk) return ZKHIP_ERR_ARG;
  if (!p->app_cache) return ZKHIP_OK;
  std::vector<uint64_t> vk(nested_vk, nested_vk + p->vk_words);
  if (find_app(p, vk, true)) return ZKHIP_OK;
  // no handle: the key could not be given one (a point off its curve, a degenerate key: ZKHIP_ERR_ARG - its batches will be proved by
  // the plain path and report their own errors), or the pipeline's table of applications is full (the cache is best effort: OK)
  std::lock_guard<std::mutex> lk(p->mu_apps);
  return p->bad_apps.count(vk) ? ZKHIP_ERR_ARG : ZKHIP_OK;
}
size_t zkhip_aggregator_pipeline_app_hits(const zkhip_pipeline* p) { return p ? (size_t)p->st_app_hits.load() : 0; }

int zkhip_aggregator_pipeline_submit(zkhip_pipeline* p, const uint64_t* nested_vk, const uint64_t* nested_proofs, const uint64_t* nested_inputs,
                                     const uint64_t r[6], const uint64_t s[6], uint64_t* ticket) {
  if (!p || !nested_vk || !nested_proofs || !nested_inputs || !r || !s || !ticket) return ZKHIP_ERR_ARG;
  auto j = std::make_shared<Job>();
  j->vk.assign(nested_vk, nested_vk + p->vk_words);
  j->proofs.assign(nested_proofs, nested_proofs + p->proofs_words);
  j->inputs.assign(nested_inputs, nested_inputs + p->inputs_words);
  memcpy(j->r, r, 48); memcpy(j->s, s, 48);
  std::unique_lock<std::mutex> lk(p->mu);
  p->cv_room.wait(lk, [&] { return p->stop || p->unfinished < p->max_unfinished; });   // back-pressure on the caller
  p->unfinished++;
  if (p->stop) { p->unfinished--; return ZKHIP_ERR_STATE; }
  j->id = p->next_id++;
  p->jobs[j->id] = j;
  p->q_wit.push_back(j);
  *ticket = j->id;
  if (p->hybrid) p->cv_wit.notify_all(); else p->cv_wit.notify_one();   // (a host generator that wants more than wit_batch queued must not swallow a batcher's wake-up)
  return ZKHIP_OK;
}

int zkhip_aggregator_pipeline_wait(zkhip_pipeline* p, uint64_t ticket, uint64_t* primary_inputs, uint64_t proof_affine[72]) {
  if (!p || !proof_affine) return ZKHIP_ERR_ARG;
  std::unique_lock<std::mutex> lk(p->mu);
  auto it = p->jobs.find(ticket);
  if (it == p->jobs.end()) return ZKHIP_ERR_NO_TICKET;     // (not ZKHIP_ERR_ARG: that is what a batch's own failure returns)
  std::shared_ptr<Job> j = it->second;
  p->cv_done.wait(lk, [&] { return p->stop || j->done; });
  if (!j->done) return ZKHIP_ERR_STATE;
  p->jobs.erase(ticket);
  lk.unlock();
  if (j->rc != ZKHIP_OK) return j->rc;
  if (primary_inputs) memcpy(primary_inputs, j->z.data(), p->n_primary * 48);   // (the prover thread kept z[1 .. n_primary])
  memcpy(proof_affine, j->proof, sizeof j->proof);
  return ZKHIP_OK;
}

}  // extern "C"
