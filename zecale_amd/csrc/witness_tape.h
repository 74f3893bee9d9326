// The straight-line program of the GPU witness generator (built by witness_tape.cpp, interpreted by witness.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace zkhip {

// instruction codes; operands a, b: >= 0 the position (= result slot) of an earlier instruction, < 0 constant -1 - index
enum : uint8_t {
  WT_NOP = 0,
  WT_INPUT = 1,   // a: index of the input element (nested key | nested proofs | nested inputs, 6 limbs each)
  WT_ADD = 2,     // a + b, NOT reduced: the builder tracks an upper bound (a multiple of r) for every value
  WT_SUB = 3,     // a - b as recorded; the laid-out program holds WT_SUBK + log2 K instead
  WT_MUL = 4,     // Montgomery product: below 2r whatever the operands' bounds (they stay below 2^10 r)
  WT_INV = 5,     // inversion of a value that is never zero for well-formed inputs (the device raises a flag if it is)
  WT_INV0 = 6,    // inversion that maps 0 to 0 by design (the is-zero gadget's hint)
  WT_BIT = 7,     // bit b of the canonical integer of a, as a field element
  WT_RED = 8,     // a brought below 4r by subtracting an estimated multiple of r (no multiplication: the cheap kind of reduction)
  WT_SUBK = 16,   // WT_SUBK + k: a - b + 2^k r, for b below 2^k r (k = 1 .. 11)
};
inline bool wt_binary(uint8_t c) { return c == WT_ADD || c == WT_SUB || c == WT_MUL || c >= WT_SUBK; }

struct WitnessTape {
  std::vector<uint8_t> code;
  std::vector<int32_t> a, b;
  std::vector<uint32_t> level_start;     // positions; level l = [level_start[l], level_start[l + 1]), multiples of 64
  uint32_t chain_start = 0;              // positions [chain_start, code.size()): the key-hash chain, in execution order
  std::vector<int32_t> out_ref;          // assignment entry i = value at this reference
  std::vector<uint64_t> consts;          // 6 limbs each, ABI form (Montgomery 2^384)
  size_t n_vars = 0, n_inputs = 0, vk_words = 0, proofs_words = 0, inputs_words = 0;
  size_t n_recorded = 0, n_mul = 0, n_inv = 0, n_reductions = 0;      // n_reductions: multiplications by one inserted to keep the bounds
};

// 0 on success
// fixed_vk: null = the circuit's generic program; else the nested key (60 + 12 (k + 1) limbs) of one application, folded into the
// program as constants (witness_tape.cpp)
int witness_tape_build(size_t num_proofs, size_t inputs_per_proof, WitnessTape* out, std::string* err, const uint64_t* fixed_vk = nullptr);

}  // namespace zkhip
