// The wrapping circuit's witness generator as a straight-line program ("tape") for the GPU (witness.hip).
//
// The reference computes the assignment by walking its gadgets (aggregator_gadget::generate_r1cs_witness,
// libzecale/circuits/aggregator_gadget.tcc:87-112, called from aggregator_circuit::prove, aggregator_circuit.tcc:144-157).  Here
// the circuit is ONE body of template code (circuit/*.hpp) and this translation unit compiles it a second time over a
// RECORDING scalar: every field operation on a value that depends on the inputs appends an instruction to a tape instead of
// computing; operations on constants are folded.  The tape is then
//   re-associated  chains of additions nobody else reads become balanced trees,
//   pruned         instructions no assignment entry depends on are dropped,
//   bounded        every value gets a static upper bound (a multiple of r): the device does not reduce after additions; subtractions
//                  pick their K r, cheap reductions are inserted where a bound would pass 2^12 r,
//   split          the key hash (one sequential MiMC chain) becomes a program of its own,
//   levelled       by KIND: a level of additions while any is ready, else all ready multiplications, else all ready inversions,
//   laid out       level by level, each level padded to whole 64-lane chunks; an instruction's position is also the slot its result
//                  is stored in.
// The device interprets it with one wave per batch (witness.hip); the host generator (aggregator.cpp) stays the reference for
// parity: tests compare the two assignments limb for limb, on the GPU (tests/test_witness_gpu.py) and, interpreting the tape with
// the host field arithmetic, on the CPU (tools/sanitize/tape_check.cpp).
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "host_field.hpp"
#include "witness_tape.h"

namespace zkhip {
namespace rec {

using HV = host::HFr;

struct Recorder {
  struct Op { uint8_t code; int32_t a, b; };
  std::vector<Op> ops;                           // instruction i defines value i
  std::vector<HV> consts;
  std::map<std::string, int32_t> const_index;
  const uint64_t *arena_base = nullptr, *arena_lo = nullptr, *arena_hi = nullptr;   // inputs live at arena_base; [arena_lo, arena_hi) of them are VARIABLES
  int32_t intern(const HV& c) {
    std::string k((const char*)c.v, sizeof c.v);
    auto it = const_index.find(k);
    if (it != const_index.end()) return it->second;
    int32_t idx = (int32_t)consts.size();
    consts.push_back(c);
    const_index[k] = idx;
    return idx;
  }
};
static thread_local Recorder* g_rec = nullptr;
static thread_local size_t g_mark[3] = {0, 0, 0};    // the key-hash section: instructions [g_mark[0], g_mark[1]); [2]: end of the key's lines
static void on_section(int which) { if (which >= 0 && which < 3) g_mark[which] = g_rec->ops.size(); }

// The recording scalar.  id < 0: a constant (its value in `val`); id >= 0: the value defined by instruction id.
struct RecFr {
  HV val;
  int32_t id = -1;
  RecFr() : val(HV::zero()) {}
  explicit RecFr(const HV& c) : val(c) {}
  bool is_const() const { return id < 0; }
  int32_t ref() const { return id >= 0 ? id : -1 - g_rec->intern(val); }
  static RecFr emit(uint8_t code, int32_t a, int32_t b) {
    if (g_rec->ops.size() >= (size_t)0x7fffff00) throw std::runtime_error("witness tape too long");
    g_rec->ops.push_back({code, a, b});
    RecFr r;
    r.id = (int32_t)g_rec->ops.size() - 1;
    return r;
  }
  static RecFr zero() { return RecFr(HV::zero()); }
  static RecFr one() { return RecFr(HV::one()); }
  static RecFr from_u64(uint64_t x) { return RecFr(HV::from_u64(x)); }
  static RecFr from_limbs(const uint64_t* p) {
    if (g_rec && p >= g_rec->arena_lo && p < g_rec->arena_hi) return emit(WT_INPUT, (int32_t)((p - g_rec->arena_base) / 6), 0);
    return RecFr(HV::from_limbs(p));
  }
  void to_limbs(uint64_t* p) const { need_const("to_limbs"); val.to_limbs(p); }
  void to_canonical(uint64_t* p) const { need_const("to_canonical"); val.to_canonical(p); }
  void need_const(const char* what) const { if (!is_const()) throw std::runtime_error(std::string("recording build: ") + what + " of a variable"); }
  // value-dependent predicates: decided for constants, "generic" (non-zero, distinct) for variables - the recorded program must not
  // depend on the data (the device flags an inversion of zero where the host generator would have branched)
  bool is_zero() const { return is_const() && val.is_zero(); }
  bool operator==(const RecFr& o) const { return (is_const() && o.is_const()) ? val == o.val : id == o.id; }
  bool operator!=(const RecFr& o) const { return !(*this == o); }
  RecFr operator+(const RecFr& o) const {
    if (is_const() && o.is_const()) return RecFr(val + o.val);
    if (is_const() && val.is_zero()) return o;
    if (o.is_const() && o.val.is_zero()) return *this;
    return emit(WT_ADD, ref(), o.ref());
  }
  RecFr operator-(const RecFr& o) const {
    if (is_const() && o.is_const()) return RecFr(val - o.val);
    if (o.is_const() && o.val.is_zero()) return *this;
    return emit(WT_SUB, ref(), o.ref());
  }
  RecFr operator*(const RecFr& o) const {
    if (is_const() && o.is_const()) return RecFr(val * o.val);
    if ((is_const() && val.is_zero()) || (o.is_const() && o.val.is_zero())) return zero();
    if (is_const() && val == HV::one()) return o;
    if (o.is_const() && o.val == HV::one()) return *this;
    return emit(WT_MUL, ref(), o.ref());
  }
  RecFr neg() const { return zero() - *this; }
  RecFr dbl() const { return *this + *this; }
  RecFr sqr() const { return *this * *this; }
  RecFr inv() const {
    // (a constant that must not be zero IS zero: only possible with an application's key folded in - a degenerate key, where the
    //  host generator branches; the application then gets no program of its own and its batches take the generic one)
    if (is_const() && val.is_zero()) throw std::runtime_error("inversion of a constant zero (degenerate nested key)");
    return is_const() ? RecFr(val.inv()) : emit(WT_INV, ref(), 0);
  }
  RecFr pow_limbs(const uint64_t* e, int nlimbs) const {
    RecFr acc = one();
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
      acc = acc.sqr();
      if ((e[i / 64] >> (i % 64)) & 1) acc = acc * (*this);
    }
    return acc;
  }
};
inline RecFr fr_bit(const RecFr& v, int t) {
  if (v.is_const()) { uint64_t c[6]; v.val.to_canonical(c); return ((c[t / 64] >> (t % 64)) & 1) ? RecFr::one() : RecFr::zero(); }
  return RecFr::emit(WT_BIT, v.ref(), t);
}
inline RecFr fr_inv0(const RecFr& v) { return v.is_const() ? RecFr(v.val.is_zero() ? HV::zero() : v.val.inv()) : RecFr::emit(WT_INV0, v.ref(), 0); }

}  // namespace rec
}  // namespace zkhip

#define ZK_CIRCUIT_NS circuit_rec
#define ZK_CIRCUIT_FR ::zkhip::rec::RecFr
#include "circuit/sections.hpp"

namespace zkhip {

// fixed_vk != null: the program of ONE APPLICATION (a registered nested verification key: aggregator_server.cpp:170-235).  The key's
// 60 + 12 (k + 1) words are CONSTANTS of the recording: every operation that depends on the key only - the key's variables, its
// MiMC hash chain, the lines of -beta and -delta, the doubling chains 2^j ABC_i inside the proof sections - is folded by the
// recorder, the assignment entries it defines come out as constants (out_ref < 0) and the key-hash chain is empty.  The input block
// keeps its layout (key | proofs | inputs), the key's part is simply never read.
int witness_tape_build(size_t num_proofs, size_t inputs_per_proof, WitnessTape* out, std::string* err, const uint64_t* fixed_vk) {
  using namespace circuit_rec;
  using rec::Recorder;
  using rec::RecFr;
  try {
    const size_t k = inputs_per_proof;
    const size_t vk_w = 60 + 12 * (k + 1), pr_w = 48 * num_proofs, in_w = 6 * k * num_proofs;
    std::vector<uint64_t> arena(vk_w + pr_w + in_w, 0);
    Recorder R;
    if (fixed_vk) memcpy(arena.data(), fixed_vk, vk_w * 8);
    R.arena_base = arena.data(); R.arena_lo = arena.data() + (fixed_vk ? vk_w : 0); R.arena_hi = arena.data() + arena.size();
    rec::g_rec = &R;
    NestedData d{arena.data(), arena.data() + vk_w, arena.data() + vk_w + pr_w};
    Builder b;
    b.record = false;
    b.on_section = rec::on_section;
    synthesize<WV>(b, num_proofs, k, &d);
    size_t h0 = rec::g_mark[0], h1 = rec::g_mark[1];
    rec::g_rec = &R;                                  // (synthesize leaves the builder pointer cleared, not the recorder)
    const size_t n_vars = b.z.size();
    // assignment entry -> reference
    std::vector<int32_t> out_ref(n_vars);
    for (size_t i = 0; i < n_vars; i++) out_ref[i] = b.z[i].ref();
    rec::g_rec = nullptr;
    // Re-association of sums.  The host code accumulates left to right (a coefficient of an Fq12 product is a chain of twelve
    // additions), which is what it should do on a CPU and twelve dependent levels here.  Field addition is exact, so a chain of
    // additions / subtractions whose intermediate results nobody else reads is rewritten as a balanced tree (positive terms, negative
    // terms, one subtraction): the same value, log2 of the depth.
    {
      const size_t n0 = R.ops.size();
      auto is_lin = [&](size_t i) { return R.ops[i].code == WT_ADD || R.ops[i].code == WT_SUB; };
      std::vector<uint32_t> uses(n0, 0);
      std::vector<int32_t> user(n0, -1);
      for (size_t i = 0; i < n0; i++) {
        const auto& op = R.ops[i];
        if (op.code == WT_INPUT) continue;
        if (op.a >= 0) { uses[op.a]++; user[op.a] = (int32_t)i; }
        if ((op.code == WT_ADD || op.code == WT_SUB || op.code == WT_MUL) && op.b >= 0) { uses[op.b]++; user[op.b] = (int32_t)i; }
      }
      for (int32_t r : out_ref) if (r >= 0) uses[r] += 2;                       // an assignment entry is a reader
      std::vector<uint8_t> absorbed(n0, 0);
      for (size_t i = 0; i < n0; i++)
        absorbed[i] = is_lin(i) && uses[i] == 1 && user[i] >= 0 && is_lin((size_t)user[i]) && ((i >= h0 && i < h1) == ((size_t)user[i] >= h0 && (size_t)user[i] < h1));
      std::vector<Recorder::Op> ops2;
      std::vector<int32_t> remap(n0, -1);
      size_t nh0 = 0, nh1 = 0;
      bool seen_h0 = false;
      std::vector<std::pair<int32_t, int>> terms, stack;
      for (size_t i = 0; i < n0; i++) {
        if (i == h0) { nh0 = ops2.size(); seen_h0 = true; }
        if (i == h1) nh1 = ops2.size();
        if (absorbed[i]) continue;
        const auto& op = R.ops[i];
        auto rm = [&](int32_t ref) { return ref >= 0 ? remap[ref] : ref; };
        if (!is_lin(i)) {
          Recorder::Op o = op;
          if (op.code != WT_INPUT) {
            o.a = rm(op.a);
            if (op.code == WT_MUL) o.b = rm(op.b);
          }
          ops2.push_back(o);
          remap[i] = (int32_t)ops2.size() - 1;
          continue;
        }
        // gather the terms of this sum through the absorbed operands
        terms.clear(); stack.clear();
        stack.push_back({(int32_t)i, +1});
        while (!stack.empty()) {
          auto [ref, sign] = stack.back();
          stack.pop_back();
          if (ref >= 0 && (absorbed[ref] || ref == (int32_t)i)) {
            const auto& o = R.ops[ref];
            stack.push_back({o.a, sign});
            stack.push_back({o.b, o.code == WT_SUB ? -sign : sign});
          } else {
            terms.push_back({ref, sign});
          }
        }
        std::vector<int32_t> pos, neg;
        for (auto& t : terms) (t.second > 0 ? pos : neg).push_back(rm(t.first));
        auto tree = [&](std::vector<int32_t>& v) -> int32_t {
          while (v.size() > 1) {
            std::vector<int32_t> nx;
            for (size_t k = 0; k + 1 < v.size(); k += 2) { ops2.push_back({WT_ADD, v[k], v[k + 1]}); nx.push_back((int32_t)ops2.size() - 1); }
            if (v.size() & 1) nx.push_back(v.back());
            v.swap(nx);
          }
          return v[0];
        };
        int32_t res;
        if (neg.empty()) res = tree(pos);
        else {
          int32_t n_ = tree(neg);
          int32_t p_ = pos.empty() ? (-1 - R.intern(rec::HV::zero())) : tree(pos);
          ops2.push_back({WT_SUB, p_, n_});
          res = (int32_t)ops2.size() - 1;
        }
        if (res < 0 || (size_t)res + 1 != ops2.size()) {          // a single (remapped) term: keep a defining instruction for this value
          ops2.push_back({WT_ADD, res, -1 - R.intern(rec::HV::zero())});
          res = (int32_t)ops2.size() - 1;
        }
        remap[i] = res;
      }
      if (!seen_h0) nh0 = ops2.size();
      if (h1 >= n0) nh1 = ops2.size();
      for (auto& r : out_ref) if (r >= 0) r = remap[r];
      R.ops.swap(ops2);
      h0 = nh0; h1 = nh1;
    }
    size_t n_ops = R.ops.size();
    // prune: keep what the assignment depends on
    std::vector<uint8_t> live(n_ops, 0);
    for (int32_t r : out_ref) if (r >= 0) live[r] = 1;
    for (size_t i = n_ops; i-- > 0;) {
      if (!live[i]) continue;
      const auto& op = R.ops[i];
      if (op.code == WT_INPUT) continue;
      if (op.a >= 0) live[op.a] = 1;
      if ((op.code == WT_ADD || op.code == WT_SUB || op.code == WT_MUL) && op.b >= 0) live[op.b] = 1;
    }
    // Lazy reduction.  The device keeps a value as its 14 limbs and does NOT reduce after an addition or a subtraction (a
    // conditional subtraction of r costs more than the addition it follows): every value carries an upper bound here, a multiple
    // of r.  A product is below 2r whatever it multiplies (operands below 2^12 r), a sum is bounded by the sum, a - b becomes
    // a - b + K r with K the power of two that covers b's bound (WT_SUBK + log2 K), and where a bound would pass 2^12 r - or an
    // inversion's operand 4 r - a WT_RED brings the operand back below 4 r (an estimated multiple of r subtracted: an addition's
    // cost, not a multiplication's; 9 k of them in the batch-2 program, most in doubling chains).  (Dead instructions go here too.)
    {
      const uint32_t CAP = 4096, KMAX = 2048;      // (operands of a product below 2^12 r each: a b < 2^24 r^2 < R r)
      std::vector<Recorder::Op> ops3;
      std::vector<uint32_t> bound3;
      std::vector<int32_t> remap(n_ops, -1);
      ops3.reserve(n_ops); bound3.reserve(n_ops);
      auto bnd = [&](int32_t r3) -> uint32_t { return r3 < 0 ? 1u : bound3[r3]; };
      auto push = [&](uint8_t code, int32_t a, int32_t b, uint32_t bound) { ops3.push_back({code, a, b}); bound3.push_back(bound); return (int32_t)ops3.size() - 1; };
      auto reduce = [&](int32_t r3) { return push(WT_RED, r3, 0, 3); };       // (witness.hip, w_reduce: below 3r)
      auto rm = [&](int32_t ref) { return ref >= 0 ? remap[ref] : ref; };
      size_t nh0 = 0, nh1 = 0, n_red0 = 0;
      for (size_t i = 0; i < n_ops; i++) {
        if (i == h0) nh0 = ops3.size();
        if (i == h1) nh1 = ops3.size();
        if (!live[i]) continue;
        const auto& op = R.ops[i];
        const size_t before = ops3.size();
        switch (op.code) {
          case WT_INPUT: remap[i] = push(WT_INPUT, op.a, 0, 1); break;
          case WT_ADD: {
            int32_t a = rm(op.a), b = rm(op.b);
            while (bnd(a) + bnd(b) > CAP) { if (bnd(a) >= bnd(b)) a = reduce(a); else b = reduce(b); }
            remap[i] = push(WT_ADD, a, b, bnd(a) + bnd(b));
            break;
          }
          case WT_SUB: {
            int32_t a = rm(op.a), b = rm(op.b);
            if (bnd(b) > KMAX) b = reduce(b);
            uint32_t k = 2;
            while (k < bnd(b)) k <<= 1;
            if (bnd(a) + k > CAP) a = reduce(a);
            uint8_t lg = 0;
            while ((1u << lg) < k) lg++;
            const uint8_t code = (uint8_t)(WT_SUBK + lg);                  // WT_SUBK + log2 K
            remap[i] = push(code, a, b, bnd(a) + k);
            break;
          }
          case WT_MUL: remap[i] = push(WT_MUL, rm(op.a), rm(op.b), 2); break;
          case WT_INV:
          case WT_INV0: {
            int32_t a = rm(op.a);
            if (bnd(a) > 4) a = reduce(a);
            remap[i] = push(op.code, a, 0, 2);
            break;
          }
          case WT_BIT: remap[i] = push(WT_BIT, rm(op.a), op.b, 1); break;
          default: throw std::runtime_error("witness tape: unknown instruction");
        }
        n_red0 += ops3.size() - before - 1;
      }
      if (h0 >= n_ops) nh0 = ops3.size();
      if (h1 >= n_ops) nh1 = ops3.size();
      for (auto& r : out_ref) if (r >= 0) r = remap[r];
      R.ops.swap(ops3);
      n_ops = R.ops.size();
      live.assign(n_ops, 1);
      h0 = nh0; h1 = nh1;
      out->n_reductions = n_red0;
    }
    // The key hash (MiMC, Miyaguchi-Preneel chaining: 13 absorptions x 93 rounds x 7 dependent operations) is one long chain with
    // nothing to run beside it: as levels of the common program it would be 8,500 of its 9,800 levels, each paying a barrier and a
    // round trip through memory for ONE instruction.  It depends on the inputs only and nothing but the assignment reads it, so it
    // becomes a program of its own, run in recorded order by a single wave next to the levelled rest (k_witness_chain).
    auto in_chain = [&](size_t i) { return i >= h0 && i < h1; };
    // Levels.  ONE wave runs the levelled program, a chunk of 64 instructions at a time (witness.hip), so a level costs what its
    // dearest instruction costs: ~0.2 us if it only adds, ~2 us if it multiplies, ~45 us if it inverts - and as-soon-as-possible
    // levels mix the three in almost every level (8,100 of 9,700 levels held a multiplication, the longest path has 3,200).  The
    // levels are therefore built by KIND: as long as an addition / subtraction / input is ready, a level of those; else all the
    // multiplications that are ready; else all the inversions.  Every multiplication then sits in the level of its multiplication
    // depth and every inversion in that of its inversion depth - the fewest dear levels any order of execution can have.
    auto klass = [](uint8_t c) { return (c == WT_INV || c == WT_INV0) ? 2 : (c == WT_MUL || c == WT_BIT) ? 1 : 0; };
    std::vector<int32_t> level(n_ops, 0);
    int32_t max_level = 0;
    {
      std::vector<uint32_t> pending(n_ops, 0), succ_off(n_ops + 1, 0), succ;
      auto deps = [&](size_t i, int32_t (&d)[2]) {
        const auto& op = R.ops[i];
        d[0] = d[1] = -1;
        if (op.code == WT_INPUT) return;
        if (op.a >= 0) d[0] = op.a;
        if (wt_binary(op.code) && op.b >= 0 && op.b != op.a) d[1] = op.b;
      };
      for (size_t i = 0; i < n_ops; i++) {
        if (!live[i] || in_chain(i)) continue;
        int32_t d[2];
        deps(i, d);
        for (int32_t x : d) {
          if (x < 0) continue;
          // the levelled program and the key-hash chain run CONCURRENTLY on two streams: a levelled instruction that read a chain
          // value would never become ready here (its operand is never levelled) and would be laid out before it
          if (in_chain((size_t)x)) throw std::runtime_error("witness tape: a levelled instruction reads a value of the key-hash chain");
          pending[i]++; succ_off[(size_t)x + 1]++;
        }
      }
      for (size_t i = 0; i < n_ops; i++) succ_off[i + 1] += succ_off[i];
      succ.resize(succ_off[n_ops]);
      {
        std::vector<uint32_t> fill(succ_off.begin(), succ_off.end() - 1);
        for (size_t i = 0; i < n_ops; i++) {
          if (!live[i] || in_chain(i)) continue;
          int32_t d[2];
          deps(i, d);
          for (int32_t x : d) if (x >= 0) succ[fill[x]++] = (uint32_t)i;
        }
      }
      std::vector<uint32_t> ready[3], firing;
      for (size_t i = 0; i < n_ops; i++) if (live[i] && !in_chain(i) && pending[i] == 0) ready[klass(R.ops[i].code)].push_back((uint32_t)i);
      int32_t l = 0;
      for (;;) {
        const int k = !ready[0].empty() ? 0 : !ready[1].empty() ? 1 : !ready[2].empty() ? 2 : -1;
        if (k < 0) break;
        firing.swap(ready[k]);
        ready[k].clear();
        for (uint32_t i : firing) level[i] = l;
        for (uint32_t i : firing)
          for (uint32_t e = succ_off[i]; e < succ_off[i + 1]; e++) {
            const uint32_t j = succ[e];
            if (--pending[j] == 0) ready[klass(R.ops[j].code)].push_back(j);
          }
        max_level = l++;
      }
      // every live instruction outside the chain must have fired: one that is left waits for an operand that never will (a dead
      // or out-of-program reference) and would silently keep level 0
      for (size_t i = 0; i < n_ops; i++)
        if (live[i] && !in_chain(i) && pending[i] != 0) throw std::runtime_error("witness tape: an instruction's operand is never computed (levelling left it pending)");
    }
    // layout: by level, inside a level by kind (inversions first, then multiplications, then the cheap ones)
    auto kind_rank = [](uint8_t c) { return (c == WT_INV || c == WT_INV0) ? 0 : c == WT_MUL ? 1 : c == WT_INPUT ? 2 : c == WT_BIT ? 3 : 4; };
    std::vector<uint32_t> order;
    order.reserve(n_ops);
    for (size_t i = 0; i < n_ops; i++) if (live[i] && !in_chain(i)) order.push_back((uint32_t)i);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
      if (level[x] != level[y]) return level[x] < level[y];
      return kind_rank(R.ops[x].code) < kind_rank(R.ops[y].code);
    });
    std::vector<int32_t> pos_of(n_ops, -1);
    WitnessTape& T = *out;
    { const size_t nr = out->n_reductions; T = WitnessTape(); T.n_reductions = nr; }
    T.level_start.push_back(0);
    int32_t cur = 0;
    for (uint32_t i : order) {
      if (level[i] != cur) {
        while (T.code.size() % 64) { T.code.push_back(WT_NOP); T.a.push_back(0); T.b.push_back(0); }
        for (; cur < level[i]; cur++) T.level_start.push_back((uint32_t)T.code.size());
      }
      pos_of[i] = (int32_t)T.code.size();
      T.code.push_back(R.ops[i].code); T.a.push_back(R.ops[i].a); T.b.push_back(R.ops[i].b);
    }
    while (T.code.size() % 64) { T.code.push_back(WT_NOP); T.a.push_back(0); T.b.push_back(0); }
    T.level_start.push_back((uint32_t)T.code.size());
    // the chain program, appended: its own copies of the inputs it reads, then its instructions in recorded order
    T.chain_start = (uint32_t)T.code.size();
    {
      std::map<uint32_t, int32_t> input_copy;
      auto need_input = [&](int32_t ref) {
        if (ref < 0 || in_chain((size_t)ref)) return;
        if (R.ops[ref].code != WT_INPUT) throw std::runtime_error("the key-hash section reads a value of another section");
        if (!input_copy.count((uint32_t)ref)) input_copy[(uint32_t)ref] = 0;
      };
      for (size_t i = h0; i < h1; i++) {
        if (!live[i]) continue;
        const auto& op = R.ops[i];
        if (op.code == WT_INPUT) continue;
        need_input(op.a);
        if (wt_binary(op.code)) need_input(op.b);
      }
      for (auto& kv : input_copy) {
        kv.second = (int32_t)T.code.size();
        T.code.push_back(WT_INPUT); T.a.push_back(R.ops[kv.first].a); T.b.push_back(0);
      }
      // (inside the chain, references to an input outside it go to the chain's copy: remembered through chain_alias)
      for (size_t i = h0; i < h1; i++) {
        if (!live[i]) continue;
        pos_of[i] = (int32_t)T.code.size();
        T.code.push_back(R.ops[i].code); T.a.push_back(R.ops[i].a); T.b.push_back(R.ops[i].b);
      }
      for (size_t p = T.chain_start; p < T.code.size(); p++) {
        const uint8_t c = T.code[p];
        if (c == WT_INPUT) continue;
        auto fix = [&](int32_t ref) -> int32_t {
          if (ref < 0) return ref;
          if (in_chain((size_t)ref)) return pos_of[ref];
          return input_copy.at((uint32_t)ref);
        };
        T.a[p] = fix(T.a[p]);
        if (wt_binary(c)) T.b[p] = fix(T.b[p]);
      }
    }
    // operands: instruction ids -> positions
    for (size_t p = 0; p < T.chain_start; p++) {
      const uint8_t c = T.code[p];
      if (c == WT_NOP || c == WT_INPUT) continue;
      if (T.a[p] >= 0) T.a[p] = pos_of[T.a[p]];
      if (wt_binary(c) && T.b[p] >= 0) T.b[p] = pos_of[T.b[p]];
    }
    T.out_ref.resize(n_vars);
    for (size_t i = 0; i < n_vars; i++) T.out_ref[i] = out_ref[i] >= 0 ? pos_of[out_ref[i]] : out_ref[i];
    T.consts.resize(R.consts.size() * 6);
    for (size_t i = 0; i < R.consts.size(); i++) R.consts[i].to_limbs(&T.consts[i * 6]);
    T.n_vars = n_vars; T.n_inputs = arena.size() / 6;
    T.vk_words = vk_w; T.proofs_words = pr_w; T.inputs_words = in_w;
    T.n_recorded = n_ops;
    size_t n_mul = 0, n_inv = 0;
    for (uint8_t c : T.code) { n_mul += c == WT_MUL; n_inv += (c == WT_INV || c == WT_INV0); }
    T.n_mul = n_mul; T.n_inv = n_inv;
    return 0;
  } catch (const std::exception& e) {
    rec::g_rec = nullptr;
    if (err) *err = e.what();
    return -1;
  }
}

}  // namespace zkhip
