// One process, several GPUs (or several contexts of one) THROUGH THE C++ ADAPTER (the reference server is one process that owns
// its prover: aggregator_server.cpp:106-118, 390-416):
//   (1) hip_proving_key over a device list = the wrapping key partitioned into slices (zkhip_multi_prover: a prover instance and a
//       host thread per slice, partial sums added on the host, one tail): its proof must equal the whole-key proof limb for limb;
//   (2) aggregator_circuit::open_node_stream over a device list = replicas behind a dispatcher (zkhip_dispatcher): every batch is proved,
//       every proof verifies, and both entries of the list get work.
// The list is {0, 1} when the box has two GPUs, else {0, 0}: two contexts on device 0.
// Input: the same binary file as boundary_gpu.cpp.  Test infrastructure only.
#include <cstdio>
#include <cstring>
#include <vector>

#include "aggregator_circuit_hip.hpp"

using namespace zecale_amd;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::vector<uint64_t> in(84 + 96 + 12);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(in.data(), 8, in.size(), f) != in.size()) return 2;
  std::fclose(f);
  try {
    const std::vector<int> devices = zkhip_device_count() >= 2 ? std::vector<int>{0, 1} : std::vector<int>{0, 0};
    zk_check(zkhip_init(0), "zkhip_init");
    aggregator_circuit<2> circuit(1);
    const zkhip_r1cs_desc& cs = circuit.get_constraint_system();
    std::unique_ptr<keypair> kp = circuit.generate_trusted_setup();          // whole key on device 0
    zkhip_crs_desc kd;
    zk_check(zkhip_keypair_crs_desc(kp->host(), &kd), "crs_desc");

    // nested objects from the flat file
    nested_verification_key nvk;
    std::memcpy(nvk.alpha_g1.data(), &in[0], 96); std::memcpy(nvk.beta_g2.data(), &in[12], 192); std::memcpy(nvk.delta_g2.data(), &in[36], 192);
    nvk.abc_g1.resize(2);
    std::memcpy(nvk.abc_g1[0].data(), &in[60], 96); std::memcpy(nvk.abc_g1[1].data(), &in[72], 96);
    nested_extended_proof np[2];
    for (int i = 0; i < 2; i++) {
      const uint64_t* p = &in[84 + 48 * i];
      std::memcpy(np[i].proof.a.data(), p, 96); std::memcpy(np[i].proof.b.data(), p + 12, 192); std::memcpy(np[i].proof.c.data(), p + 36, 96);
      std::array<uint64_t, 6> x;
      std::memcpy(x.data(), &in[84 + 96 + 6 * i], 48);
      np[i].primary_inputs.push_back(x);
    }

    // (1) the partitioned key against the whole key, same assignment, same (r, s)
    std::vector<uint64_t> z(cs.n_vars * 6);
    {
      std::vector<uint64_t> proofs;
      for (int i = 0; i < 2; i++) { proofs.insert(proofs.end(), &in[84 + 48 * i], &in[84 + 48 * (i + 1)]); }
      zkhip_aggregator* agg = nullptr;
      zk_check(zkhip_aggregator_new(2, 1, &agg), "zkhip_aggregator_new");
      zk_check(zkhip_aggregator_witness(agg, &in[0], proofs.data(), &in[84 + 96], z.data()), "witness");
      zkhip_aggregator_free(agg);
    }
    uint64_t r[6], s[6];
    zk_check(zkhip_fr_random(r), "r"); zk_check(zkhip_fr_random(s), "s");
    groth16_proof whole, split;
    {
      hip_proving_key one(kd, cs);
      if (!one.is_satisfied(z.data())) throw std::runtime_error("assignment does not satisfy the circuit");
      whole = one.generate_proof(z.data(), r, s);
    }
    {
      hip_proving_key many(kd, cs, devices);
      split = many.generate_proof(z.data(), r, s);
      split = many.generate_proof(z.data(), r, s);                           // (a second proof on warm contexts)
      std::printf("PARTITIONED devices=%zu list=%d,%d same_as_whole_key=%d\n", many.num_devices(), devices[0], devices[1],
                  whole.a == split.a && whole.b == split.b && whole.c == split.c);
    }
    extended_proof ep;
    ep.proof = split;
    for (size_t i = 0; i < circuit.num_primary_inputs(); i++) {
      std::array<uint64_t, 6> x;
      std::memcpy(x.data(), &z[(i + 1) * 6], 48);
      ep.primary_inputs.push_back(x);
    }
    std::printf("PARTITIONED verifies=%d\n", (int)kp->verify(ep));

    // (2) replicas behind the dispatcher: eight batches over the list
    {
      auto stream = circuit.open_node_stream(*kp, devices, /*gpu_slots=*/3, /*witness_workers=*/2);
      std::vector<uint64_t> tickets;
      for (int b = 0; b < 8; b++) {
        const int flip = b & 1;                                              // the two nested proofs in either order
        tickets.push_back(stream->submit(nvk, {&np[flip], &np[1 - flip]}));
      }
      int verified = 0;
      for (uint64_t t : tickets) verified += kp->verify(stream->wait(t)) ? 1 : 0;
      const std::vector<size_t> per = stream->batches_per_device();
      std::printf("REPLICAS entries=%zu verified=%d of 8 batches_per_entry=%zu,%zu\n", per.size(), verified, per[0], per[1]);
    }
    kp.reset();
    zkhip_shutdown();
    std::printf("DONE\n");
  } catch (const std::exception& e) {
    std::printf("EXCEPTION %s\n", e.what());
    return 1;
  }
  return 0;
}
