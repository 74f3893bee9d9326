// A small arithmetic-circuit DSL for the wrapping circuit (host C++).
//
// The reference builds its circuit out of libsnark gadgets: each gadget has a constructor that allocates
// variables and a pair generate_r1cs_constraints / generate_r1cs_witness (libzecale/circuits/
// aggregator_gadget.tcc:13-112, aggregator_circuit.tcc:17-98, 120-170).  Those gadget sources
// (libsnark gadgetlib1 pairing gadgets, libzeth MiMC) are not in the reference tree, so the circuit here is
// restated from the mathematics with one mechanism instead of two: every algorithm of the circuit (tower
// arithmetic, group law, Miller loop, final exponentiation, MiMC) is written ONCE as a template over a field
// type F and instantiated with
//     NF  native field element                      -> plain evaluation (used to cross-check and in tests)
//     CV  circuit value = linear combination + value  -> the same code emits the R1CS constraints and
//                                                       computes the witness in one pass
// Products of two CVs allocate a variable and emit  <a,z> * <b,z> = <c,z>;  sums and constant multiples stay
// linear combinations and cost nothing.  A witness-only pass (Builder::record = false) replays the same code
// and only fills the assignment; the structure of the code never depends on values.
//
// The field is Fr of BW6-761 = Fq of BLS12-377 (aggregator_gadget.hpp:20-30: nested coordinates are native
// wrapper scalars).
#pragma once
#include <algorithm>
#include <stdexcept>
#include <vector>

#include "../host_field.hpp"

// The circuit headers compile a second time, in another namespace and over another scalar type, for the GPU witness
// generator (witness_tape.cpp): there HFr is a RECORDING type whose operators append to a tape instead of (only) computing,
// so the very same code that defines the constraints and the host witness also defines the straight-line program the
// device interprets.  Everything value-dependent below is therefore written through four hooks the recording type overloads:
//   inv()  a field inversion that must not see zero        fr_inv0()  an inversion that maps 0 to 0 by design
//   fr_bit()  a bit of the canonical integer               batch_inv()  independent inversions
#ifndef ZK_CIRCUIT_NS
#define ZK_CIRCUIT_NS circuit
#endif

namespace zkhip {
namespace ZK_CIRCUIT_NS {

#ifdef ZK_CIRCUIT_FR
using HFr = ZK_CIRCUIT_FR;
#else
using host::HFr;
// bit t of the canonical integer, as a field element
inline HFr fr_bit(const HFr& v, int t) {
  uint64_t c[6];
  v.to_canonical(c);
  return ((c[t / 64] >> (t % 64)) & 1) ? HFr::one() : HFr::zero();
}
inline HFr fr_inv0(const HFr& v) { return v.is_zero() ? HFr::zero() : v.inv(); }
#endif

struct Term {
  uint32_t var;
  HFr coeff;
};
typedef std::vector<Term> LC;

inline void lc_normalise(LC& l) {
  std::sort(l.begin(), l.end(), [](const Term& a, const Term& b) { return a.var < b.var; });
  size_t o = 0;
  for (size_t i = 0; i < l.size();) {
    HFr c = l[i].coeff;
    size_t j = i + 1;
    while (j < l.size() && l[j].var == l[i].var) { c = c + l[j].coeff; j++; }
    if (!c.is_zero()) { l[o].var = l[i].var; l[o].coeff = c; o++; }
    i = j;
  }
  l.resize(o);
}

struct Builder {
  std::vector<HFr> z;       // full assignment, z[0] = 1
  std::vector<LC> A, B, C;  // constraints (recorded when `record`)
  bool record = true;
  void (*on_section)(int) = nullptr;      // told where the key-hash section begins (0) and ends (1) - the recording build splits its program there - and where the key's lines end (2)
  Builder() { z.push_back(HFr::one()); }
  uint32_t alloc(const HFr& value) { z.push_back(value); return (uint32_t)(z.size() - 1); }
  void enforce(LC a, LC b, LC c) {
    if (!record) return;
    lc_normalise(a); lc_normalise(b); lc_normalise(c);
    A.push_back(std::move(a)); B.push_back(std::move(b)); C.push_back(std::move(c));
  }
  size_t num_constraints() const { return A.size(); }
};

inline Builder*& current_builder() {
  static thread_local Builder* b = nullptr;
  return b;
}

// ---- native instantiation --------------------------------------------------------------------
struct NF {
  HFr v;
  NF() : v(HFr::zero()) {}
  explicit NF(const HFr& x) : v(x) {}
  static NF constant(const HFr& c) { return NF(c); }
  static NF witness(const HFr& value) { return NF(value); }
  static NF witness_bit(bool b) { return NF(b ? HFr::one() : HFr::zero()); }
  static NF witness_bitv(const HFr& bit) { return NF(bit); }
  const HFr& value() const { return v; }
  NF operator+(const NF& o) const { return NF(v + o.v); }
  NF operator-(const NF& o) const { return NF(v - o.v); }
  NF operator*(const NF& o) const { return NF(v * o.v); }
  NF mulc(const HFr& c) const { return NF(v * c); }
  NF neg() const { return NF(v.neg()); }
  static void assert_eq(const NF& a, const NF& b) {
    if (a.v != b.v) throw std::runtime_error("native assert_eq failed");
  }
  static void assert_product(const NF& a, const NF& b, const NF& c) {
    if (a.v * b.v != c.v) throw std::runtime_error("native assert_product failed");
  }
};

// ---- circuit instantiation -------------------------------------------------------------------
// "Constant" is a STRUCTURAL property (built from constants only), tracked by an explicit flag with the same
// propagation rules in CV and WV, so that both allocate variables in exactly the same order.
struct CV {
  LC lc;
  HFr val;
  bool cst = true;
  CV() : val(HFr::zero()) {}
  static CV constant(const HFr& c) {
    CV r;
    if (!c.is_zero()) r.lc.push_back({0, c});
    r.val = c;
    return r;
  }
  static CV witness(const HFr& value) {
    CV r;
    r.lc.push_back({current_builder()->alloc(value), HFr::one()});
    r.val = value;
    r.cst = false;
    return r;
  }
  static CV witness_bit(bool b) {
    CV r = witness(b ? HFr::one() : HFr::zero());
    CV m = r - constant(HFr::one());
    current_builder()->enforce(r.lc, m.lc, LC());      // b (b - 1) = 0
    return r;
  }
  static CV witness_bitv(const HFr& bit) {               // the bit as a field element (fr_bit)
    CV r = witness(bit);
    CV m = r - constant(HFr::one());
    current_builder()->enforce(r.lc, m.lc, LC());      // b (b - 1) = 0
    return r;
  }
  const HFr& value() const { return val; }
  bool is_const() const { return cst; }
  void compact() { if (lc.size() > 48) lc_normalise(lc); }
  CV operator+(const CV& o) const {
    CV r = *this;
    r.lc.insert(r.lc.end(), o.lc.begin(), o.lc.end());
    r.val = val + o.val;
    r.cst = cst && o.cst;
    r.compact();
    return r;
  }
  CV operator-(const CV& o) const { return *this + o.neg(); }
  CV neg() const {
    CV r = *this;
    for (auto& t : r.lc) t.coeff = t.coeff.neg();
    r.val = val.neg();
    return r;
  }
  CV mulc(const HFr& c) const {
    CV r;
    r.cst = cst;
    if (c.is_zero()) return r;
    r = *this;
    for (auto& t : r.lc) t.coeff = t.coeff * c;
    r.val = val * c;
    return r;
  }
  CV operator*(const CV& o) const {
    if (is_const()) return o.mulc(val);          // products with constants stay linear: no variable, no constraint
    if (o.is_const()) return mulc(o.val);
    CV r = witness(val * o.val);
    current_builder()->enforce(lc, o.lc, r.lc);
    return r;
  }
  static void assert_eq(const CV& a, const CV& b) {
    CV d = a - b;
    current_builder()->enforce(d.lc, constant(HFr::one()).lc, LC());   // (a - b) * 1 = 0
  }
  static void assert_product(const CV& a, const CV& b, const CV& c) { current_builder()->enforce(a.lc, b.lc, c.lc); }
};

// ---- witness-only instantiation: values and the const flag, no linear combinations ---------------------------
// Replays exactly the allocations of CV (same code, same flag rules) at native speed: this is the
// generate_r1cs_witness half of the reference's gadgets.
struct WV {
  HFr val;
  bool cst = true;
  WV() : val(HFr::zero()) {}
  static WV constant(const HFr& c) { WV r; r.val = c; return r; }
  static WV witness(const HFr& value) { WV r; r.val = value; r.cst = false; current_builder()->alloc(value); return r; }
  static WV witness_bit(bool b) { return witness(b ? HFr::one() : HFr::zero()); }
  static WV witness_bitv(const HFr& bit) { return witness(bit); }
  const HFr& value() const { return val; }
  bool is_const() const { return cst; }
  WV operator+(const WV& o) const { WV r; r.val = val + o.val; r.cst = cst && o.cst; return r; }
  WV operator-(const WV& o) const { WV r; r.val = val - o.val; r.cst = cst && o.cst; return r; }
  WV neg() const { WV r; r.val = val.neg(); r.cst = cst; return r; }
  WV mulc(const HFr& c) const { WV r; r.val = val * c; r.cst = cst; return r; }
  WV operator*(const WV& o) const {
    if (cst) return o.mulc(val);
    if (o.cst) return mulc(o.val);
    return witness(val * o.val);
  }
  static void assert_eq(const WV&, const WV&) {}
  static void assert_product(const WV&, const WV&, const WV&) {}
};
inline bool f_is_const(const WV& x) { return x.cst; }
// F only replays values (WV): a gadget's assertions are no-ops there, and so is everything that merely prepares their arguments -
// linear combinations and multiplications by constants allocate nothing, so skipping them leaves the allocation order alone
// (round 5: in the host generator a third of a cyclotomic squaring's products fed assertions that do nothing)
template <class F> struct witness_only { static constexpr bool value = false; };
template <> struct witness_only<WV> { static constexpr bool value = true; };

// helpers shared by the instantiations
template <class F> inline F f_const_u64(uint64_t x) { return F::constant(HFr::from_u64(x)); }
template <class F> inline F f_zero() { return F::constant(HFr::zero()); }
template <class F> inline F f_one() { return F::constant(HFr::one()); }
// a / b with b != 0 (value 0 if b == 0: the structure pass runs on dummy values): witness q, enforce q b = a
inline bool f_is_const(const NF&) { return false; }
inline bool f_is_const(const CV& x) { return x.is_const(); }
// Witness generation is a chain of field inversions (the slopes of the in-circuit point additions); the ones that do not depend on
// each other are inverted together: n inverses for one inversion and 3(n-1) multiplications (Montgomery's trick).  Zeros stay zero.
#ifdef ZK_CIRCUIT_FR
inline void batch_inv(HFr* v, int n) { for (int i = 0; i < n; i++) v[i] = v[i].inv(); }     // recorded as independent inversions
#else
inline void batch_inv(HFr* v, int n) {
  HFr pre[8];
  HFr acc = HFr::one();
  for (int i = 0; i < n; i++) { pre[i] = acc; if (!v[i].is_zero()) acc = acc * v[i]; }
  HFr inv = acc.inv();
  for (int i = n - 1; i >= 0; i--) {
    if (v[i].is_zero()) continue;
    HFr vi = inv * pre[i];
    inv = inv * v[i];
    v[i] = vi;
  }
}
#endif
// binv: the inverse of b's value if the caller already has it (batch_inv), else null
template <class F> inline F f_div(const F& a, const F& b, const HFr* binv = nullptr) {
  HFr bi = binv ? *binv : fr_inv0(b.value());
  if (f_is_const(a) && f_is_const(b)) return F::constant(a.value() * bi);
  F q = F::witness(a.value() * bi);
  F::assert_product(q, b, a);
  return q;
}
// [x == 0] as a field element (1 or 0): m = 1/x (or 0), z = 1 - x m, enforce x z = 0      (2 constraints)
template <class F> inline F f_is_zero(const F& x) {
  HFr m = fr_inv0(x.value());
  F mw = F::witness(m);
  F z = f_one<F>() - x * mw;
  F::assert_product(x, z, f_zero<F>());
  return z;
}
// Results of chained operations (point additions along a scalar, Miller-loop steps) are FRESH variables: the product
// constraint a b = r + off defines r directly instead of defining a product variable p and returning the linear combination
// p - off.  Same number of variables and constraints, but the linear combinations no longer grow along the chain (they used to
// reach 500 terms at the end of a 253-bit accumulator and made every row that touched them that long).
// r = a b - off
template <class F> inline F f_mul_minus(const F& a, const F& b, const F& off) {
  if (f_is_const(a) || f_is_const(b)) return a * b - off;          // stays linear: no variable, no constraint
  F r = F::witness(a.value() * b.value() - off.value());
  F::assert_product(a, b, r + off);
  return r;
}
// sel ? a : b   (sel boolean): one variable, one constraint  sel (a - b) = out - b
template <class F> inline F f_select(const F& sel, const F& a, const F& b) {
  F d = a - b;
  if (f_is_const(sel) || f_is_const(d)) return b + sel * d;
  F out = F::witness(b.value() + sel.value() * d.value());
  F::assert_product(sel, d, out - b);
  return out;
}

}  // namespace ZK_CIRCUIT_NS
}  // namespace zkhip
