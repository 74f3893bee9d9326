// CPU check of the GPU witness generator's PROGRAM (witness_tape.cpp), no GPU involved: the tape of the batch-2 circuit is built,
// its structure is verified - every operand defined in an earlier level (levelled part) or earlier in the chain (key-hash part),
// the static bounds the device relies on recomputed independently (a - b + 2^k r only with b below 2^k r, nothing above 2^12 r,
// inversions fed below 4 r) - and it is interpreted with the host field arithmetic; the assignment it produces must equal the host
// generator's (aggregator.cpp) limb for limb.  Also the sanitizer driver of the tape builder (ASan + UBSan build in the test).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "zkhip.h"
#include "host_field.hpp"
#include "witness_tape.h"
using zkhip::host::HFr;
using namespace zkhip;

#define FAIL(...) do { printf("FAIL: " __VA_ARGS__); printf("\n"); return 1; } while (0)

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  std::vector<uint64_t> in(84 + 96 + 12);
  if (!f || fread(in.data(), 8, in.size(), f) != in.size()) { puts("bad input"); return 2; }
  fclose(f);
  zkhip_aggregator* a = nullptr;
  if (zkhip_aggregator_new(2, 1, &a) != 0) return 3;
  zkhip_r1cs_desc d;
  zkhip_aggregator_get_r1cs(a, &d);
  std::vector<uint64_t> z(d.n_vars * 6);
  if (zkhip_aggregator_witness(a, in.data(), in.data() + 84, in.data() + 180, z.data()) != 0) return 4;

  WitnessTape T;
  std::string err;
  if (witness_tape_build(2, 1, &T, &err) != 0) FAIL("tape build: %s", err.c_str());
  if (T.n_vars != d.n_vars || T.n_inputs * 6 != in.size()) FAIL("tape shape: %zu variables, %zu inputs", T.n_vars, T.n_inputs);
  const size_t n = T.code.size(), cs = T.chain_start;
  if (cs % 64 || T.level_start.empty() || T.level_start.back() != cs) FAIL("layout: chain_start %zu, last level end %u", cs, T.level_start.back());
  std::vector<uint32_t> level_of(cs, 0);
  for (size_t l = 0; l + 1 < T.level_start.size(); l++) {
    if (T.level_start[l] % 64 || T.level_start[l] > T.level_start[l + 1]) FAIL("level %zu is not a whole number of chunks", l);
    for (uint32_t p = T.level_start[l]; p < T.level_start[l + 1]; p++) level_of[p] = (uint32_t)l;
  }
  std::vector<HFr> consts(T.consts.size() / 6), val(n);
  for (size_t i = 0; i < consts.size(); i++) consts[i] = HFr::from_limbs(&T.consts[i * 6]);
  std::vector<uint32_t> bound(n, 0);
  std::vector<uint8_t> defined(n, 0);
  size_t n_exec = 0, n_sub = 0, n_red = 0, max_bound = 0, kinds_mixed = 0;
  auto klass = [](uint8_t c) { return (c == WT_INV || c == WT_INV0) ? 2 : (c == WT_MUL || c == WT_BIT) ? 1 : 0; };
  for (size_t p = 0; p < n; p++) {
    const uint8_t c = T.code[p];
    if (c == WT_NOP) continue;
    const bool chain = p >= cs;
    auto check_ref = [&](int32_t r) -> bool {
      if (r < 0) return (size_t)(-1 - r) < consts.size();
      if ((size_t)r >= p || !defined[r]) return false;
      if (!chain) return (size_t)r < T.level_start[level_of[p]];            // an earlier LEVEL, not merely an earlier position
      return (size_t)r >= cs;                                              // the chain reads only itself (its own copies of the inputs)
    };
    auto get = [&](int32_t r) { return r < 0 ? consts[-1 - r] : val[r]; };
    auto bnd = [&](int32_t r) { return r < 0 ? 1u : bound[r]; };
    HFr v;
    uint32_t b = 0;
    if (c == WT_INPUT) {
      if (T.a[p] < 0 || (size_t)T.a[p] >= T.n_inputs) FAIL("position %zu: input index %d", p, T.a[p]);
      v = HFr::from_limbs(&in[(size_t)T.a[p] * 6]); b = 1;
    } else {
      if (!check_ref(T.a[p])) FAIL("position %zu (code %u): operand a = %d is not defined before it", p, c, T.a[p]);
      if (wt_binary(c) && !check_ref(T.b[p])) FAIL("position %zu (code %u): operand b = %d is not defined before it", p, c, T.b[p]);
      const HFr x = get(T.a[p]);
      if (c == WT_ADD) { v = x + get(T.b[p]); b = bnd(T.a[p]) + bnd(T.b[p]); }
      else if (c >= WT_SUBK) {
        const uint32_t k = c - WT_SUBK;
        if (k < 1 || k > 11) FAIL("position %zu: a - b + 2^%u r", p, k);
        if (bnd(T.b[p]) > (1u << k)) FAIL("position %zu: subtrahend bound %u above 2^%u", p, bnd(T.b[p]), k);
        v = x - get(T.b[p]); b = bnd(T.a[p]) + (1u << k); n_sub++;
      }
      else if (c == WT_RED) { v = x; b = 3; n_red++; }
      else if (c == WT_MUL) { v = x * get(T.b[p]); b = 2; }
      else if (c == WT_INV || c == WT_INV0) {
        if (bnd(T.a[p]) > 4) FAIL("position %zu: inversion of a value bounded by %u r", p, bnd(T.a[p]));
        if (c == WT_INV && x.is_zero()) FAIL("position %zu: inversion of zero on well-formed inputs", p);
        v = x.is_zero() ? HFr::zero() : x.inv(); b = 2;
      }
      else if (c == WT_BIT) {
        uint64_t w[6];
        x.to_canonical(w);
        if (T.b[p] < 0 || T.b[p] >= 384) FAIL("position %zu: bit %d", p, T.b[p]);
        v = ((w[T.b[p] / 64] >> (T.b[p] % 64)) & 1) ? HFr::one() : HFr::zero(); b = 1;
      }
      else FAIL("position %zu: unknown code %u (plain WT_SUB must not survive the bounds pass)", p, c);
    }
    if (b > 4096) FAIL("position %zu: bound %u r", p, b);
    if (!chain && p % 64 && T.code[p - 1] != WT_NOP && level_of[p - 1] == level_of[p] && klass(T.code[p - 1]) != klass(c)) kinds_mixed++;
    val[p] = v; bound[p] = b; defined[p] = 1;
    if (b > max_bound) max_bound = b;
    n_exec++;
  }
  if (kinds_mixed) FAIL("%zu levels mix instruction kinds", kinds_mixed);
  size_t diff = 0;
  for (size_t i = 0; i < T.n_vars; i++) {
    const int32_t r = T.out_ref[i];
    if (r >= 0 && ((size_t)r >= n || !defined[r])) FAIL("assignment entry %zu reads position %d", i, r);
    const HFr v = r < 0 ? consts[-1 - r] : val[r];
    uint64_t l[6];
    v.to_limbs(l);
    if (memcmp(l, &z[i * 6], 48) != 0 && !diff++) printf("first difference at variable %zu\n", i);
  }
  printf("tape: positions=%zu executed=%zu levels=%zu chain=%zu sub=%zu red=%zu max_bound=%zu mul=%zu inv=%zu differences=%zu\n", n, n_exec,
         T.level_start.size() - 1, n - cs, n_sub, n_red, max_bound, T.n_mul, T.n_inv, diff);
  zkhip_aggregator_free(a);
  return diff ? 1 : 0;
}
