// Host-side test shim: exposes the product's fp29.cuh arithmetic (compiled for the CPU by g++)
// through a C ABI so tests/test_fp29_host.py can compare it with Python big integers.
// Test infrastructure only.
#include "../zecale_amd/csrc/fp29.cuh"
#include "../zecale_amd/csrc/fp_inv.cuh"
using namespace zkhip;
extern "C" {
#define SHIM(F, PR, N64)                                                                       \
  void F##_mul(const uint64_t* a, const uint64_t* b, uint64_t* r) {                            \
    fp_to_abi<PR>(fp_mul(fp_from_abi<PR>(a), fp_from_abi<PR>(b)), r); }                        \
  /* (a b + c d) with one reduction, on lazily bounded operands: (a + b)(a - b + 16p) + (4p - a)(2b) */ \
  void F##_mul2(const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* d, uint64_t* r) { \
    fp_to_abi<PR>(fp_mul2(fp_from_abi<PR>(a), fp_from_abi<PR>(b), fp_from_abi<PR>(c), fp_from_abi<PR>(d)), r); } \
  void F##_mul2_lazy(const uint64_t* a, const uint64_t* b, uint64_t* r) {                      \
    Fp<PR> x = fp_from_abi<PR>(a), y = fp_from_abi<PR>(b);                                     \
    Fp<PR> s = fp_add(fp_add(x, y), fp_add(x, x));            /* 3x + y < 8p */                \
    Fp<PR> t = fp_sub<PR, 16>(s, fp_dbl(y));                  /* 3x - y + 16p < 24p */         \
    Fp<PR> n = fp_sub<PR, 4>(fp_zero<PR>(), x);               /* -x + 4p <= 4p */             \
    fp_to_abi<PR>(fp_mul2(s, t, n, fp_dbl(y)), r); }                                           \
  /* raw device limbs in and out: lets the test put every limb at its maximum (the column bound of the dual product) */ \
  void F##_mul2_raw(const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* r) { \
    Fp<PR> x, y, z, w;                                                                         \
    for (int i = 0; i < PR::NL; i++) { x.l[i] = a[i]; y.l[i] = b[i]; z.l[i] = c[i]; w.l[i] = d[i]; } \
    Fp<PR> o = fp_mul2(x, y, z, w);                                                            \
    for (int i = 0; i < PR::NL; i++) r[i] = o.l[i]; }                                          \
  void F##_sub_sub2(const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* r) {     \
    fp_to_abi<PR>(fp_sub_sub2<PR, 8>(fp_from_abi<PR>(a), fp_from_abi<PR>(b), fp_from_abi<PR>(c)), r); } \
  void F##_sqr(const uint64_t* a, uint64_t* r) { fp_to_abi<PR>(fp_sqr(fp_from_abi<PR>(a)), r); } \
  void F##_add(const uint64_t* a, const uint64_t* b, uint64_t* r) {                            \
    fp_to_abi<PR>(fp_add(fp_from_abi<PR>(a), fp_from_abi<PR>(b)), r); }                        \
  void F##_sub(const uint64_t* a, const uint64_t* b, uint64_t* r) {                            \
    fp_to_abi<PR>(fp_sub<PR, 2>(fp_from_abi<PR>(a), fp_from_abi<PR>(b)), r); }                 \
  void F##_inv(const uint64_t* a, uint64_t* r) { fp_to_abi<PR>(fp_inv<PR>(fp_from_abi<PR>(a)), r); }               \
  void F##_roundtrip(const uint64_t* a, uint64_t* r) { fp_to_abi<PR>(fp_from_abi<PR>(a), r); } \
  void F##_canon_words(const uint64_t* a, uint32_t* w) { fp_abi_to_canonical_words<PR>(a, w); } \
  /* a long lazy chain: ((a+b)*(a-b+8p) + 16p - b)^2 ... exercising the documented bounds */  \
  void F##_lazy_chain(const uint64_t* a, const uint64_t* b, uint64_t* r) {                     \
    Fp<PR> x = fp_from_abi<PR>(a), y = fp_from_abi<PR>(b);                                     \
    Fp<PR> s = fp_add(fp_add(x, y), fp_add(x, y));          /* < 8p */                         \
    Fp<PR> d = fp_sub<PR, 8>(s, fp_dbl(fp_dbl(y)));         /* < 16p, subtrahend < 8p */       \
    Fp<PR> m = fp_mul(s, d);                                /* < 2p */                         \
    Fp<PR> t = fp_sub<PR, 16>(m, d);                        /* < 18p */                        \
    fp_to_abi<PR>(fp_sqr(t), r); }
SHIM(fq, FqParams, 12)
SHIM(fr, FrParams, 6)
}
