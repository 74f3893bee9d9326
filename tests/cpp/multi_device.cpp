// One process, several prover contexts (SURVEY 8e from the C ABI; the reference server is one process: aggregator_server.cpp:390-416):
// the proving key of the wrapping circuit is cut into two slices, each uploaded to its own device handle (two GPUs when the box
// has them, else both on device 0), two host threads run zkhip_groth16_prove_partial side by side, the 5 x 288-byte partial sums
// are added on the host, zkhip_groth16_finish completes the proof: it must equal the whole-key proof limb for limb.
// Input: the same binary file as boundary_gpu.cpp.  Test infrastructure only.
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "groth16_snark_hip.hpp"

using namespace zecale_amd;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::vector<uint64_t> in(84 + 96 + 12);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(in.data(), 8, in.size(), f) != in.size()) return 2;
  std::fclose(f);
  try {
    const int ndev = zkhip_device_count() >= 2 ? 2 : 1;
    for (int d = ndev - 1; d >= 0; d--) zk_check(zkhip_init(d), "zkhip_init");          // device 0 last: the thread's library device
    zkhip_aggregator* agg = nullptr;
    zk_check(zkhip_aggregator_new(2, 1, &agg), "zkhip_aggregator_new");
    zkhip_r1cs_desc cs;
    zk_check(zkhip_aggregator_get_r1cs(agg, &cs), "get_r1cs");
    uint64_t t[4][6];
    for (auto& s : t) zk_check(zkhip_fr_random(s), "zkhip_fr_random");
    zkhip_keypair* kp = nullptr;
    zk_check(zkhip_groth16_setup(&cs, t[0], t[1], t[2], t[3], &kp), "zkhip_groth16_setup");
    zkhip_crs_desc kd;
    zk_check(zkhip_keypair_crs_desc(kp, &kd), "crs_desc");
    std::vector<uint64_t> z(cs.n_vars * 6);
    zk_check(zkhip_aggregator_witness(agg, &in[0], &in[84], &in[84 + 96], z.data()), "witness");
    uint64_t r[6], s[6];
    zk_check(zkhip_fr_random(r), "r"); zk_check(zkhip_fr_random(s), "s");
    // whole key on device 0
    zkhip_crs* whole = nullptr; zkhip_r1cs* r0 = nullptr;
    zk_check(zkhip_crs_upload(&kd, &whole), "crs_upload");
    zk_check(zkhip_r1cs_upload(&cs, &r0), "r1cs_upload");
    uint64_t expect[72];
    zk_check(zkhip_groth16_prove(whole, r0, z.data(), r, s, expect), "prove");
    // two slices, one context each
    const size_t m = kd.n_vars, l = kd.n_primary, d = kd.domain_size;
    const size_t a_cut = m / 3, h_cut = (d - 1) / 2 + 5, l_cut = (m - l - 1) * 2 / 3;     // uneven on purpose
    struct Ctx { int dev; size_t a0, a1, h0, h1, l0, l1; zkhip_crs* crs = nullptr; zkhip_r1cs* rc = nullptr; uint64_t sums[180]; int rc_code = 0; std::string err; };
    Ctx cx[2] = {{0, 0, a_cut, 0, h_cut, 0, l_cut}, {ndev - 1, a_cut, m, h_cut, d - 1, l_cut, m - l - 1}};
    for (auto& c : cx) {
      zk_check(zkhip_set_device(c.dev), "set_device");
      zk_check(zkhip_crs_upload_slice(&kd, c.a0, c.a1 - c.a0, c.h0, c.h1 - c.h0, c.l0, c.l1 - c.l0, &c.crs), "upload_slice");
      zk_check(zkhip_r1cs_upload(&cs, &c.rc), "r1cs_upload");
    }
    zk_check(zkhip_set_device(0), "set_device");
    std::thread th[2];
    for (int i = 0; i < 2; i++)
      th[i] = std::thread([&, i] {            // these threads never called zkhip_init / zkhip_set_device: the handles carry their device
        Ctx& c = cx[i];
        c.rc_code = zkhip_groth16_prove_partial(c.crs, c.rc, z.data(), c.a0, c.h0, c.l0, c.sums);
        if (c.rc_code != ZKHIP_OK) c.err = zkhip_last_error();
      });
    for (auto& x : th) x.join();
    for (auto& c : cx) if (c.rc_code != ZKHIP_OK) throw std::runtime_error("prove_partial: " + c.err);
    uint64_t total[180];
    for (int k = 0; k < 5; k++) zk_check(zkhip_jac_add(cx[0].sums + 36 * k, cx[1].sums + 36 * k, total + 36 * k), "jac_add");
    uint64_t got[72];
    zk_check(zkhip_groth16_finish(kd.alpha_g1, kd.beta_g1, kd.beta_g2, kd.delta_g1, kd.delta_g2, total, r, s, got), "finish");
    uint64_t a[24], b[24], dl[24]; const uint64_t* abc;
    const size_t nabc = zkhip_keypair_vk(kp, a, b, dl, &abc);
    int ok = 0;
    zk_check(zkhip_groth16_verify(a, b, dl, abc, &z[6], nabc - 1, got, &ok), "verify");
    std::printf("MULTI devices=%d same_as_whole_key=%d verifies=%d\n", ndev, std::memcmp(got, expect, sizeof got) == 0, ok);
    for (auto& c : cx) { zkhip_crs_free(c.crs); zkhip_r1cs_free(c.rc); }
    zkhip_crs_free(whole); zkhip_r1cs_free(r0); zkhip_keypair_free(kp); zkhip_aggregator_free(agg);
    zkhip_shutdown();
  } catch (const std::exception& e) {
    std::printf("EXCEPTION %s\n", e.what());
    return 1;
  }
  return 0;
}
