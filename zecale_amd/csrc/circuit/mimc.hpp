// MiMC-based hash of the nested verification key: primary input 0 of every wrapping proof
// (libzecale/circuits/verification_key_hash_gadget.{hpp,tcc}: mimc_input_hasher over
// verification_key.get_all_vars(); compression_function_selector.hpp:23-31: MiMC-e17, 93 rounds,
// Miyaguchi-Preneel).  The reference's round constants and IV are derived inside libzeth, which is not in
// the reference tree, so they cannot be reproduced: this file keeps the STRUCTURE
// (exponent 17, 93 rounds, Miyaguchi-Preneel chaining, length block) with constants derived from SHA-256 of
// fixed strings.  Hash values therefore differ from a reference deployment's; everything else that depends on
// the hash (public-input layout, equality with compute_hash) is preserved.
#pragma once
#include <string>

#include "dsl.hpp"

namespace zkhip {
namespace ZK_CIRCUIT_NS {

// ---- SHA-256 (constants only) ----------------------------------------------------------------------
inline void sha256(const uint8_t* msg, size_t len, uint8_t out[32]) {
  static const uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
      0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
      0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
      0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
      0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
      0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  std::vector<uint8_t> m(msg, msg + len);
  m.push_back(0x80);
  while (m.size() % 64 != 56) m.push_back(0);
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 7; i >= 0; i--) m.push_back((uint8_t)(bits >> (8 * i)));
  auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
  for (size_t off = 0; off < m.size(); off += 64) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)m[off + 4 * i] << 24 | (uint32_t)m[off + 4 * i + 1] << 16 | (uint32_t)m[off + 4 * i + 2] << 8 | m[off + 4 * i + 3];
    for (int i = 16; i < 64; i++) {
      uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
      uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)(h[i] >> (24 - 8 * j));
}

inline HFr field_from_digest(const uint8_t d[32]) {     // big-endian 256-bit integer reduced mod r
  HFr acc = HFr::zero(), b = HFr::from_u64(256);
  for (int i = 0; i < 32; i++) acc = acc * b + HFr::from_u64(d[i]);
  return acc;
}

struct MimcConsts {
  static constexpr int ROUNDS = 93;
  HFr c[ROUNDS];
  HFr iv;
  MimcConsts() {
    const std::string seed = "zecale-amd/mimc-e17-r93/round-constants";
    uint8_t d[32];
    sha256((const uint8_t*)seed.data(), seed.size(), d);
    c[0] = HFr::zero();                                  // MiMC convention: first round constant 0
    for (int i = 1; i < ROUNDS; i++) { sha256(d, 32, d); c[i] = field_from_digest(d); }
    const std::string ivs = "zecale-amd/mimc-e17-r93/iv";
    sha256((const uint8_t*)ivs.data(), ivs.size(), d);
    iv = field_from_digest(d);
  }
};
inline const MimcConsts& mimc_consts() { static MimcConsts m; return m; }

template <class F> inline F pow17(const F& x) {
  F x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
  return x16 * x;
}
// E_k(m): 93 rounds of x <- (x + k + c_i)^17, then + k
template <class F> inline F mimc_permutation(const F& m, const F& k) {
  const MimcConsts& K = mimc_consts();
  F x = m;
  for (int i = 0; i < MimcConsts::ROUNDS; i++) x = pow17(x + k + F::constant(K.c[i]));
  return x + k;
}
// Miyaguchi-Preneel: h' = E_h(m) + h + m
template <class F> inline F mimc_mp(const F& m, const F& h) { return mimc_permutation(m, h) + h + m; }

// hash of a list of field elements: h_0 = IV; absorb every element, then the length
template <class F> inline F mimc_hash(const std::vector<F>& in) {
  F h = F::constant(mimc_consts().iv);
  for (const F& m : in) h = mimc_mp(m, h);
  return mimc_mp(F::constant(HFr::from_u64((uint64_t)in.size())), h);
}

}  // namespace ZK_CIRCUIT_NS
}  // namespace zkhip
