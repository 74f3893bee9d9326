// Sanitizer driver for the host-side wrapping circuit (aggregator.cpp): build the circuit, generate the witness of the reference
// fixtures, check every constraint <A_i,z><B_i,z> = <C_i,z> with the host field arithmetic.  CPU only.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "zkhip.h"
#include "host_field.hpp"
using zkhip::host::HFr;
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  std::vector<uint64_t> in(84 + 96 + 12);
  if (!f || fread(in.data(), 8, in.size(), f) != in.size()) { puts("bad input"); return 2; }
  fclose(f);
  zkhip_aggregator* a = nullptr;
  if (zkhip_aggregator_new(2, 1, &a) != 0) return 3;
  zkhip_r1cs_desc d;
  zkhip_aggregator_get_r1cs(a, &d);
  std::vector<uint64_t> z(d.n_vars * 6);
  for (int rep = 0; rep < 2; rep++)
    if (zkhip_aggregator_witness(a, in.data(), in.data() + 84, in.data() + 180, z.data()) != 0) return 4;
  auto dot = [&](const void* rpv, const void* colv, const void* valv, size_t i) {
    const uint32_t* rp = (const uint32_t*)rpv; const uint32_t* col = (const uint32_t*)colv; const uint64_t* val = (const uint64_t*)valv;
    HFr acc = HFr::zero();
    for (uint32_t k = rp[i]; k < rp[i + 1]; k++) acc = acc + HFr::from_limbs(val + (size_t)k * 6) * HFr::from_limbs(&z[(size_t)col[k] * 6]);
    return acc;
  };
  size_t bad = 0;
  for (size_t i = 0; i < d.n_constraints; i++)
    if (dot(d.a_row_ptr, d.a_col, d.a_val, i) * dot(d.b_row_ptr, d.b_col, d.b_val, i) != dot(d.c_row_ptr, d.c_col, d.c_val, i)) bad++;
  uint64_t h[6];
  zkhip_aggregator_vk_hash(in.data(), 1, h);
  printf("constraints=%zu vars=%zu unsatisfied=%zu hash_matches_input0=%d\n", d.n_constraints, d.n_vars, bad, memcmp(h, &z[6], 48) == 0);
  zkhip_aggregator_free(a);
  return bad ? 1 : 0;
}
