// Drives the C++ mirror of libzecale::aggregator_circuit (include/aggregator_circuit_hip.hpp) on a real device, the way
// aggregator_server.cpp:480-514 and :300-348 use the reference class: circuit -> generate_trusted_setup -> prove -> verify, then
// the streaming form.  Input: a binary file of u64 limbs: nested vk (60 + 12 * 2) | two nested proofs (2 x 48) | two inputs (2 x 6)
// - the reference's dummy_app fixtures, written by the test.  Output: lines the test parses.  Test infrastructure only.
#include <cstdio>
#include <cstring>
#include <vector>

#include "aggregator_circuit_hip.hpp"

using namespace zecale_amd;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::vector<uint64_t> in(84 + 96 + 12);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(in.data(), 8, in.size(), f) != in.size()) { std::printf("cannot read %s\n", argv[1]); return 2; }
  std::fclose(f);
  nested_verification_key nvk;
  std::memcpy(nvk.alpha_g1.data(), &in[0], 96); std::memcpy(nvk.beta_g2.data(), &in[12], 192); std::memcpy(nvk.delta_g2.data(), &in[36], 192);
  nvk.abc_g1.resize(2);
  std::memcpy(nvk.abc_g1[0].data(), &in[60], 96); std::memcpy(nvk.abc_g1[1].data(), &in[72], 96);
  nested_extended_proof np[2];
  for (int p = 0; p < 2; p++) {
    const uint64_t* q = &in[84 + 48 * p];
    std::memcpy(np[p].proof.a.data(), q, 96); std::memcpy(np[p].proof.b.data(), q + 12, 192); std::memcpy(np[p].proof.c.data(), q + 36, 96);
    np[p].primary_inputs.resize(1);
    std::memcpy(np[p].primary_inputs[0].data(), &in[84 + 96 + 6 * p], 48);
  }
  try {
    zk_check(zkhip_init(0), "zkhip_init");
    aggregator_circuit<2> agg(1);
    auto kp = agg.generate_trusted_setup();
    std::printf("ABC %zu PRIMARY %zu\n", kp->vk_abc_size(), agg.num_primary_inputs());      // aggregator_server.cpp:490
    std::printf("DOMAIN %zu\n", kp->domain_size());                                          // the reference's forced power of two
    std::printf("VK %s\n", kp->verification_key_to_json().c_str());
    std::array<const nested_extended_proof*, 2> batch = {&np[0], &np[1]};
    extended_proof ep = agg.prove(nvk, batch, *kp);
    std::printf("VERIFY %d\n", kp->verify(ep) ? 1 : 0);
    std::printf("PROOF %s\n", ep.to_json().c_str());
    extended_proof ep2 = agg.prove(nvk, batch, *kp);                                         // fresh (r, s): another proof of the same statement
    std::printf("REPROVE_DIFFERS %d VERIFY %d\n", std::memcmp(ep.proof.a.data(), ep2.proof.a.data(), 192) != 0, kp->verify(ep2) ? 1 : 0);
    extended_proof bad = ep;
    bad.primary_inputs[1][0] ^= 1;
    std::printf("TAMPERED_VERIFY %d\n", kp->verify(bad) ? 1 : 0);
    // the reference's error path (tcc:138-141) and the well-formedness check
    nested_extended_proof wrong = np[0];
    wrong.primary_inputs.clear();
    std::array<const nested_extended_proof*, 2> bad_batch = {&np[0], &wrong};
    try { agg.prove(nvk, bad_batch, *kp); std::printf("THROW none\n"); } catch (const std::runtime_error& e) { std::printf("THROW %s\n", e.what()); }
    nested_extended_proof off = np[1];
    off.proof.a[6] ^= 1;
    std::array<const nested_extended_proof*, 2> off_batch = {&np[0], &off};
    try { agg.prove(nvk, off_batch, *kp); std::printf("OFFCURVE none\n"); } catch (const std::runtime_error& e) { std::printf("OFFCURVE %s\n", e.what()); }
    // streaming form
    auto st = agg.open_stream(*kp, 2, 2);
    uint64_t t[3];
    std::array<const nested_extended_proof*, 2> swapped = {&np[1], &np[0]};
    t[0] = st->submit(nvk, batch); t[1] = st->submit(nvk, swapped); t[2] = st->submit(nvk, batch);
    for (int i = 2; i >= 0; i--) {
      extended_proof e = st->wait(t[i]);
      std::printf("STREAM %d VERIFY %d %s\n", i, kp->verify(e) ? 1 : 0, e.to_json().c_str());
    }
    // the same stream with the witness generated on the GPU
    auto st2 = agg.open_stream(*kp, 2, 2, true);
    uint64_t tg = st2->submit(nvk, swapped);
    extended_proof eg = st2->wait(tg);
    std::printf("GPUWITNESS VERIFY %d INPUTS_EQUAL %d\n", kp->verify(eg) ? 1 : 0,
                (int)(eg.primary_inputs[2] == np[1].primary_inputs[0] && eg.primary_inputs[3] == np[0].primary_inputs[0]));
    // THE KEY IS AUTHORITATIVE FOR THE DOMAIN.  (a) A key handed over as raw arrays - what the bridge of INTEGRATION.md section 2 builds
    // from a reference-generated r1cs_gg_ppzksnark_proving_key, domain_size 65,536 - proves through hip_proving_key; (b) the same
    // circuit object proves with a key generated on libfqfft's unforced step domain (49,152 points) and then again with the first
    // key: its constraint-system handle follows the key; (c) a key whose domain cannot hold the system is refused.
    {
      zkhip_crs_desc d;
      zk_check(zkhip_keypair_crs_desc(kp->host(), &d), "zkhip_keypair_crs_desc");
      std::vector<uint64_t> vkf = nvk.flat(), prf, inf, z(agg.get_constraint_system().n_vars * 6);
      for (int p = 0; p < 2; p++) {
        prf.insert(prf.end(), np[p].proof.a.begin(), np[p].proof.a.end()); prf.insert(prf.end(), np[p].proof.b.begin(), np[p].proof.b.end());
        prf.insert(prf.end(), np[p].proof.c.begin(), np[p].proof.c.end());
        inf.insert(inf.end(), np[p].primary_inputs[0].begin(), np[p].primary_inputs[0].end());
      }
      zkhip_aggregator* raw = nullptr;
      zk_check(zkhip_aggregator_new(2, 1, &raw), "zkhip_aggregator_new");
      zk_check(zkhip_aggregator_witness(raw, vkf.data(), prf.data(), inf.data(), z.data()), "zkhip_aggregator_witness");
      zkhip_aggregator_free(raw);
      uint64_t r[6], s[6];
      zk_check(zkhip_fr_random(r), "zkhip_fr_random"); zk_check(zkhip_fr_random(s), "zkhip_fr_random");
      hip_proving_key imported(d, agg.get_constraint_system());
      groth16_proof gp = imported.generate_proof(z.data(), r, s);
      extended_proof ei;
      ei.proof = gp;
      for (size_t i = 0; i < agg.num_primary_inputs(); i++) { std::array<uint64_t, 6> x; std::memcpy(x.data(), &z[(i + 1) * 6], 48); ei.primary_inputs.push_back(x); }
      std::printf("IMPORTED_KEY domain=%zu key_domain=%zu VERIFY %d\n", imported.domain_size(), (size_t)d.domain_size, kp->verify(ei) ? 1 : 0);
      zkhip_crs_desc small = d;
      small.domain_size = 32768;                                       // a power of two, but below the circuit's 44,188 points
      try { hip_proving_key bad_key(small, agg.get_constraint_system()); std::printf("SMALL_DOMAIN accepted\n"); }
      catch (const std::runtime_error& e) { std::printf("SMALL_DOMAIN refused\n"); }
    }
    auto kps = agg.generate_trusted_setup(ZKHIP_DOMAIN_STEP);
    extended_proof es = agg.prove(nvk, batch, *kps);
    extended_proof eb = agg.prove(nvk, batch, *kp);                    // back on the 65,536-point key
    std::printf("STEP_KEY domain=%zu VERIFY %d CROSS %d BACK %d\n", kps->domain_size(), kps->verify(es) ? 1 : 0, kp->verify(es) ? 1 : 0, kp->verify(eb) ? 1 : 0);
    std::printf("DONE\n");
  } catch (const std::exception& e) {
    std::printf("EXCEPTION %s\n", e.what());
    return 1;
  }
  return 0;
}
