// Point additions on MEMORY-resident XYZZ accumulators, written as small "micro-programs":
// one rolled loop over the 10 (mixed add) / 14 (full add) field multiplications of the formula,
// with a switch that selects the operands and files the result.  Why:
//   * A fully unrolled 761-bit Montgomery multiplication is ~12 KB of straight-line code.  Ten of
//     them inlined into one loop body (120 KB) overflow the CU's instruction cache and the kernel
//     becomes instruction-fetch bound (measured: 10x slower than the arithmetic allows).  Calling
//     an out-of-line multiply passes its 2 x 27 limbs through scratch memory (2x slower).  The
//     rolled micro-program keeps exactly ONE copy of the multiplier in the loop: ~20 KB per kernel.
//   * Only four temporaries (4 x 27 VGPRs) plus the multiplier's own operands are live, so the
//     kernel fits 256 VGPRs and runs two waves per SIMD; the accumulator itself stays in memory in
//     limb-major layout (word (k, idx) at base[k*stride + idx]): lane t <-> accumulator t, every
//     access is a coalesced 256-byte row per wave, served from L2.
// Formulas: madd-2008-s / add-2008-s (see ec.cuh for the register-resident versions and bounds).
#pragma once
#include "ec.cuh"

namespace zkhip {

// Addressing: buffer instructions with a wave-uniform descriptor (SGPRs), the row offset
// (k * stride) in the scalar offset and only the 32-bit lane offset in a VGPR - with flat 64-bit
// addresses hipcc hoists 108 address pairs per accumulator out of the loop and spills them.
typedef __amdgpu_buffer_rsrc_t zk_rsrc_t;
struct XyzzRef {
  zk_rsrc_t rs;        // wave-uniform: base pointer, 108 * stride words
  uint32_t stride_b;   // wave-uniform: distance between two limbs of a coordinate, in bytes (limb-major: the row stride; slot array: 4)
  uint32_t coord_b;    // wave-uniform: distance between two coordinates, in bytes (limb-major: 27 rows; slot array: 112)
  uint32_t voff;       // per lane: byte offset of the item (limb-major: 4 * index; slot array: 448 * index)
};
// base / stride must be wave-uniform (kernel arguments); 108 * stride * 4 must stay below 4 GiB.
__device__ __forceinline__ XyzzRef make_ref(uint32_t* base, uint32_t stride, uint32_t idx) {
  XyzzRef r;
  r.rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(108u * stride * 4u), 0x00020000);
  r.stride_b = stride * 4u;
  r.coord_b = 27u * stride * 4u;
  r.voff = idx * 4u;
  return r;
}
enum { CX = 0, CY = 1, CZZ = 2, CZZZ = 3 };

// THE SLOT ARRAY of a launch sequence - where k_accumulate leaves its runs: one slot per bucket, two per slice for the pieces of the
// buckets a slice boundary cuts - is an array of STRUCTURES since round 6: slot s = X | Y | ZZ | ZZZ, each coordinate 27 limbs padded
// to 28 words (112 bytes, 16-byte aligned), 448 bytes a slot.  A run is closed by ONE lane at a moment of its own (the lanes of a wave
// close their runs at different iterations), so in the limb-major layout every one of its 105 stores was a 4-byte write into a line of
// its own: 2.04 GB of WRITE_SIZE per 2^20-term launch for 0.33 GB of slots, partial lines the L2 also has to fetch.  In the
// structure a lane's stores fall into four lines back to back and a slot is read and written with 16-byte accesses (seven per
// coordinate: mem_ld / mem_st below) - WRITE_SIZE 0.59 GB, FETCH_SIZE 3.98 -> 3.68 GB, and, interleaved on one box
// (profiles/r06_slots_layout_ab.txt): k_accumulate<1> alone 11.10 -> 10.66 ms, MSM stream 83.1 -> 86.9 Mscalar/s, 2^20 prover
// 21.0 -> 22.2 proofs/s, wrapping stream 430.6 -> 450.0 proofs/s (4-byte accesses to 432-byte slots gave a quarter of that).
// Every other XYZZ array (the reduction's levels: one lane per item, read and written by whole waves) stays limb-major.
// -DZK_SLOTS_AOS=0 restores the limb-major slot array.
#ifndef ZK_SLOTS_AOS
#define ZK_SLOTS_AOS 1
#endif
#if ZK_SLOTS_AOS
#define ZK_SLOT_WORDS 112u
#define ZK_SLOT_PITCH 448u
__device__ __forceinline__ XyzzRef make_slot_ref(uint32_t* base, uint32_t n_slots, uint32_t idx) {
  XyzzRef r;
  r.rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(ZK_SLOT_WORDS * n_slots * 4u), 0x00020000);
  r.stride_b = 4u;
  r.coord_b = 112u;
  r.voff = idx * ZK_SLOT_PITCH;
  return r;
}
#else
#define ZK_SLOT_WORDS 108u
#define ZK_SLOT_PITCH 4u
__device__ __forceinline__ XyzzRef make_slot_ref(uint32_t* base, uint32_t n_slots, uint32_t idx) { return make_ref(base, n_slots, idx); }
#endif
// `in` of the first level of the bucket reduction is the slot array, of the later levels a limb-major array
__device__ __forceinline__ XyzzRef make_in_ref(uint32_t* base, uint32_t stride, uint32_t idx, int is_slots) {
  return is_slots ? make_slot_ref(base, stride, idx) : make_ref(base, stride, idx);
}

// a == 0 (mod p) for a in [0, 2p), for a value that is almost never zero (PP of an addition: zero only when the two points share
// their x coordinate).  The full test reads all 27 limbs twice (~100 instructions in every addition); zero and p both show in the
// LOWEST limb (0 or p's), so the full test runs only when some lane of the wave passes that necessary condition.
__device__ __forceinline__ bool fp_is_zero_2p_rare(const Fq& a) {
  const bool maybe = (a.l[0] == 0u) | (a.l[0] == FqParams::P[0]);
  if (!__any(maybe)) return false;
  return maybe && fp_is_zero_2p(a);
}

// a coordinate of a SLOT (27 consecutive words, 16-byte aligned): six 16-byte accesses and one of 12
typedef uint32_t zk_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t zk_u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ Fq slot_ld_wide(const zk_rsrc_t& rs, uint32_t vo, uint32_t so) {
  Fq v;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const zk_u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so + 16u * (uint32_t)j, 0);
    v.l[4 * j] = w.x; v.l[4 * j + 1] = w.y; v.l[4 * j + 2] = w.z; v.l[4 * j + 3] = w.w;
  }
  const zk_u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(rs, vo, so + 96u, 0);
  v.l[24] = t.x; v.l[25] = t.y; v.l[26] = t.z;
  return v;
}
__device__ __forceinline__ void slot_st_wide(const zk_rsrc_t& rs, uint32_t vo, uint32_t so, const Fq& v) {
#pragma unroll
  for (int j = 0; j < 6; j++) {
    zk_u32x4 w;
    w.x = v.l[4 * j]; w.y = v.l[4 * j + 1]; w.z = v.l[4 * j + 2]; w.w = v.l[4 * j + 3];
    __builtin_amdgcn_raw_buffer_store_b128(w, rs, vo, so + 16u * (uint32_t)j, 0);
  }
  zk_u32x3 t;
  t.x = v.l[24]; t.y = v.l[25]; t.z = v.l[26];
  __builtin_amdgcn_raw_buffer_store_b96(t, rs, vo, so + 96u, 0);
}
__device__ __forceinline__ Fq mem_ld(const XyzzRef& r, int c) {
#if ZK_SLOTS_AOS
  if (r.coord_b == 112u) return slot_ld_wide(r.rs, r.voff, (uint32_t)c * 112u);      // (wave-uniform: the reference of a SLOT - a limb-major array has coord_b = 108 x its stride, never 112)
#endif
  Fq v;
#pragma unroll
  for (int i = 0; i < 27; i++)
    v.l[i] = __builtin_amdgcn_raw_buffer_load_b32(r.rs, r.voff, (uint32_t)c * r.coord_b + (uint32_t)i * r.stride_b, 0);
  return v;
}
__device__ __forceinline__ void mem_st(const XyzzRef& r, int c, const Fq& v) {
#if ZK_SLOTS_AOS
  if (r.coord_b == 112u) { slot_st_wide(r.rs, r.voff, (uint32_t)c * 112u, v); return; }
#endif
#pragma unroll
  for (int i = 0; i < 27; i++)
    __builtin_amdgcn_raw_buffer_store_b32(v.l[i], r.rs, r.voff, (uint32_t)c * r.coord_b + (uint32_t)i * r.stride_b, 0);
}
__device__ __forceinline__ void mem_set_inf(const XyzzRef& r) {
  Fq z = fp_zero<FqParams>();
  mem_st(r, CX, z); mem_st(r, CY, z); mem_st(r, CZZ, z); mem_st(r, CZZZ, z);
}
__device__ __forceinline__ bool mem_is_inf(const XyzzRef& r) { return fp_is_zero_2p(mem_ld(r, CZZ)); }
__device__ __forceinline__ void mem_copy(const XyzzRef& dst, const XyzzRef& src) {
#pragma unroll
  for (int c = 0; c < 4; c++) mem_st(dst, c, mem_ld(src, c));
}

// packed affine point (device form) coordinate loads; neg: return p - y (as a [2] value)
__device__ __forceinline__ Fq aff_ld_x(const AffPacked* p) {
  uint32_t w[24];
#pragma unroll
  for (int i = 0; i < 24; i++) w[i] = p->x[i];
  return fp_unpack32<FqParams>(w);
}
__device__ __forceinline__ Fq aff_ld_y(const AffPacked* p, bool neg) {
  uint32_t w[24];
#pragma unroll
  for (int i = 0; i < 24; i++) w[i] = p->y[i];
  Fq y = fp_unpack32<FqParams>(w);
  if (neg) y = fp_sub<FqParams, 2>(fp_zero<FqParams>(), y);
  return y;
}
__device__ __forceinline__ bool aff_is_inf(const AffPacked* p) {
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < 24; i++) nz |= p->x[i] | p->y[i];
  return nz == 0;
}

// rare paths (same x coordinate), out of line: their cost does not matter, their code size does
__device__ __noinline__ void madd_same_x(XyzzRef acc, const AffPacked* p, bool neg) {
  Fq x2 = aff_ld_x(p), y2 = aff_ld_y(p, neg);
  Fq S2 = fp_mul(y2, mem_ld(acc, CZZZ));
  Fq R = fp_sub<FqParams, 4>(S2, mem_ld(acc, CY));
  if (fp_is_zero_2p(fp_sqr(R))) {
    XYZZ d = xyzz_dbl_affine(x2, y2);
    mem_st(acc, CX, d.X); mem_st(acc, CY, d.Y); mem_st(acc, CZZ, d.ZZ); mem_st(acc, CZZZ, d.ZZZ);
  } else {
    mem_set_inf(acc);
  }
}
__device__ __noinline__ void add_same_x(XyzzRef a, XyzzRef b) {
  Fq S1 = fp_mul(mem_ld(a, CY), mem_ld(b, CZZZ));
  Fq S2 = fp_mul(mem_ld(b, CY), mem_ld(a, CZZZ));
  Fq R = fp_sub<FqParams, 2>(S2, S1);
  if (fp_is_zero_2p(fp_sqr(R))) {
    XYZZ p;
    p.X = mem_ld(a, CX); p.Y = mem_ld(a, CY); p.ZZ = mem_ld(a, CZZ); p.ZZZ = mem_ld(a, CZZZ);
    XYZZ d = xyzz_dbl(p);
    mem_st(a, CX, d.X); mem_st(a, CY, d.Y); mem_st(a, CZZ, d.ZZ); mem_st(a, CZZZ, d.ZZZ);
  } else {
    mem_set_inf(a);
  }
}

// acc (memory, finite) += p (packed affine, finite; y negated when neg).
__device__ __forceinline__ bool madd_mem(const XyzzRef& acc, const AffPacked* p, bool neg) {
  Fq T0 = fp_zero<FqParams>(), T1 = T0, T2 = T0, T3 = T0;
  bool same_x = false;
#pragma unroll 1
  for (int step = 0; step < 10; step++) {
    Fq a, b;
    switch (step) {
      case 0: a = aff_ld_x(p); b = mem_ld(acc, CZZ); break;          // U2 = x2 ZZ1
      case 1: a = aff_ld_y(p, neg); b = mem_ld(acc, CZZZ); break;    // S2 = y2 ZZZ1
      case 2: a = T0; b = T0; break;                                 // PP = P^2
      case 3: a = T0; b = T2; break;                                 // PPP = P PP
      case 4: a = mem_ld(acc, CZZ); b = T2; break;                   // ZZ3 = ZZ1 PP
      case 5: a = mem_ld(acc, CZZZ); b = T3; break;                  // ZZZ3 = ZZZ1 PPP
      case 6: a = mem_ld(acc, CX); b = T2; break;                    // Q = X1 PP
      case 7: a = T1; b = T1; break;                                 // RR = R^2
      case 8: a = T1; b = fp_sub<FqParams, 16>(T0, T2); break;       // Y3a = R (Q - X3)
      default: a = mem_ld(acc, CY); b = T3; break;                   // Y3b = Y1 PPP
    }
    Fq r = fp_mul(a, b);
    switch (step) {
      case 0: T0 = fp_sub<FqParams, 16>(r, mem_ld(acc, CX)); break;  // P  [18]
      case 1: T1 = fp_sub<FqParams, 4>(r, mem_ld(acc, CY)); break;   // R  [6]
      case 2: T2 = r; same_x = fp_is_zero_2p_rare(r); break;              // PP
      case 3: T3 = r; break;                                         // PPP
      case 4: mem_st(acc, CZZ, r); break;
      case 5: mem_st(acc, CZZZ, r); break;
      case 6: T0 = r; break;                                         // Q
      case 7: T2 = fp_sub<FqParams, 4>(fp_sub<FqParams, 4>(r, T3), fp_dbl(T0)); mem_st(acc, CX, T2); break;   // X3 [10]
      case 8: T0 = r; break;                                         // Y3a
      default: mem_st(acc, CY, fp_sub<FqParams, 2>(T0, r)); break;   // Y3 [4]
    }
    if (same_x) break;
  }
  if (same_x) madd_same_x(acc, p, neg);
  return same_x;
}

// ---- LDS-staged accumulators -------------------------------------------------------------------
// X (packed), ZZ and ZZZ of a lane's running accumulator live in LDS for a whole run of additions ([k][256] images, word k of lane
// t at base[k*256 + t]: conflict-free), Y in registers (madd_lds_regy, add_lds_regy below).
#define ZK_LDS_STRIDE 256
__device__ __forceinline__ Fq lds_ld(const uint32_t* base) {
  Fq v;
#pragma unroll
  for (int i = 0; i < 27; i++) v.l[i] = base[i * ZK_LDS_STRIDE];
  return v;
}
__device__ __forceinline__ void lds_st(uint32_t* base, const Fq& v) {
#pragma unroll
  for (int i = 0; i < 27; i++) base[i * ZK_LDS_STRIDE] = v.l[i];
}

// X is LDS-staged too, as 24 packed words (the CU's 160 KiB hold 27 + 27 + 24 words for each of 512 lanes)
__device__ __forceinline__ Fq lds_ld_packed(const uint32_t* base) {
  uint32_t w[24];
#pragma unroll
  for (int i = 0; i < 24; i++) w[i] = base[i * ZK_LDS_STRIDE];
  return fp_unpack32<FqParams>(w);
}
__device__ __forceinline__ void lds_st_packed(uint32_t* base, const Fq& v) {   // v < 2^768, limbs normalised
  uint32_t w[24];
  fp_pack32<FqParams>(v, w);
#pragma unroll
  for (int i = 0; i < 24; i++) base[i * ZK_LDS_STRIDE] = w[i];
}

// acc += p (packed affine, finite; y negated when neg), the accumulator's Y carried in REGISTERS across the additions of a run
// (ty, bound [4]), X (packed), ZZ, ZZZ in LDS.  No global accumulator traffic in the steady state; `acc` is only touched on the
// rare same-x path.  Eight single products in the rolled loop (one multiplier body, one squarer body), then
// Y3 = R (Q - X3) + (-Y1) PPP as ONE dual product with one Montgomery reduction (fp_mul2): 13,149 v_mad_u64_u32 per addition
// (6 x 1,458 + 2 x 1,107 + 2,187) instead of 13,878, and one lazy subtraction less.  Five field elements live at most.
// (Prefetching the NEXT addition's point under this one was measured twice and lost twice: carried in registers in round 3 (+30 spilled
// dwords: 11.5 -> 12.1 ms at 2^20), and in round 4 as three LDS-DMA loads per point - global_load_lds_dword into a scratch row, no
// destination register - issued after this addition's last gather: 334 -> 325 wrapping proofs/s, 82.6 -> 80.8 Mscalar/s.  The
// gathers are not what the kernel waits for; the extra requests cost more than the latency they hide.)
__device__ __forceinline__ bool madd_lds_regy(const XyzzRef& acc, uint32_t* xs, uint32_t* zz, uint32_t* zzz, Fq& ty, const AffPacked* p, bool neg) {
  Fq T0, T1, T2, T3;      // (deliberately not initialised: every one is written by the step before the first that reads it, and zeroing
                          //  them costs 108 moves per addition)
  bool same_x = false;
#pragma unroll 1
  for (int step = 0; step < 8; step++) {
    Fq a, b;
    switch (step) {
      case 0: a = aff_ld_x(p); b = lds_ld(zz); break;                // U2 = x2 ZZ1
      case 1: a = aff_ld_y(p, neg); b = lds_ld(zzz); break;          // S2 = y2 ZZZ1
      case 2: a = T0; b = T0; break;                                 // PP = P^2
      case 3: a = T0; b = T2; break;                                 // PPP = P PP
      case 4: a = lds_ld(zz); b = T2; break;                         // ZZ3 = ZZ1 PP
      case 5: a = lds_ld(zzz); b = T3; break;                        // ZZZ3 = ZZZ1 PPP
      case 6: a = lds_ld_packed(xs); b = T2; break;                  // Q = X1 PP
      default: a = T1; b = T1; break;                                // RR = R^2
    }
    Fq r = (step == 2 || step == 7) ? fp_sqr(a) : fp_mul(a, b);
    switch (step) {
      case 0: T0 = fp_sub<FqParams, 16>(r, lds_ld_packed(xs)); break; // P  [18]
      case 1: T1 = fp_sub<FqParams, 4>(r, ty); break;                // R  [6]
      case 2: T2 = r; same_x = fp_is_zero_2p_rare(r); break;              // PP
      case 3: T3 = r; break;                                         // PPP
      case 4: lds_st(zz, r); break;
      case 5: lds_st(zzz, r); break;
      case 6: T0 = r; break;                                         // Q
      default: T2 = fp_sub_sub2<FqParams, 8>(r, T3, T0); lds_st_packed(xs, T2); break;   // X3 = RR - PPP - 2Q [10], one carry pass
    }
    if (same_x) break;
  }
  if (!same_x) {
    // Y3 = R (Q - X3) + (4p - Y1) PPP: [6] x [18] + [4] x [2] -> [2]
    ty = fp_mul2(T1, fp_sub<FqParams, 16>(T0, T2), fp_sub<FqParams, 4>(fp_zero<FqParams>(), ty), T3);
  } else {            // PP = 0 was detected at step 2: ty still holds Y1, nothing has been overwritten
    mem_st(acc, CX, lds_ld_packed(xs));
    mem_st(acc, CY, ty);
    mem_st(acc, CZZ, lds_ld(zz));
    mem_st(acc, CZZZ, lds_ld(zzz));
    madd_same_x(acc, p, neg);
    lds_st_packed(xs, fp_cond_sub_p(fp_mul(mem_ld(acc, CX), fp_one<FqParams>())));   // any bound -> canonical
    ty = mem_ld(acc, CY);
    lds_st(zz, mem_ld(acc, CZZ));
    lds_st(zzz, mem_ld(acc, CZZZ));
  }
  return same_x;
}

// Full addition into an accumulator that lives ON THE CU for a whole chain of additions: X (packed), ZZ, ZZZ in LDS, Y in registers
// (the layout of the bucket accumulation); B is an XYZZ point in memory (finite).  A chain acc = B0 + B1 + ... then costs six
// coordinate loads per addition (of B only: ZZ2 and ZZZ2 twice) and one accumulator store at the very end, where add_mem_s moves
// ten loads and four stores per addition - the plain sums of the bucket reduction are bound by those round trips, not by the
// multiplier.  `spill` is where the accumulator goes on the rare same-x path.  Returns true if that path ran.
// (Round 4, measured and left out: loading every coordinate of B once - parking ZZ2 / ZZZ2 in the dead temporary, carrying them in `b`
// across the loop's back edge, or a second call site of the multiplier for the product that shares the factor - takes the gathers
// from six to four per addition, but the rolled loop already holds T0 .. T3, Y and both operands live at the 256-register limit:
// the three forms compiled to 51 - 185 spilled registers where this one has 1.  Also without effect on the kernel's 368 us per
// launch: testing B for infinity on the first row of its ZZ only, and wave priorities by progress as in k_accumulate.)
__device__ __forceinline__ bool add_lds_regy(const XyzzRef& spill, uint32_t* xs, uint32_t* zz, uint32_t* zzz, Fq& ty, const XyzzRef& B) {
  Fq T0 = fp_zero<FqParams>(), T1 = T0, T2 = T0, T3 = T0;
  bool same_x = false;
#pragma unroll 1
  for (int step = 0; step < 12; step++) {
    Fq a, b;
    switch (step) {
      case 0: a = lds_ld_packed(xs); b = mem_ld(B, CZZ); break;      // U1 = X1 ZZ2
      case 1: a = mem_ld(B, CX); b = lds_ld(zz); break;              // U2 = X2 ZZ1
      case 2: a = T1; b = T1; break;                                 // PP
      case 3: a = T1; b = T2; break;                                 // PPP
      case 4: a = T0; b = T2; break;                                 // Q = U1 PP
      case 5: a = lds_ld(zz); b = T2; break;                         // ZZ1 PP
      case 6: a = lds_ld(zz); b = mem_ld(B, CZZ); break;             // (ZZ1 PP) ZZ2
      case 7: a = ty; b = mem_ld(B, CZZZ); break;                    // S1 = Y1 ZZZ2
      case 8: a = mem_ld(B, CY); b = lds_ld(zzz); break;             // S2 = Y2 ZZZ1
      case 9: a = lds_ld(zzz); b = T1; break;                        // ZZZ1 PPP
      case 10: a = lds_ld(zzz); b = mem_ld(B, CZZZ); break;          // (ZZZ1 PPP) ZZZ2
      default: a = T3; b = T3; break;                                // RR
    }
    Fq r = (step == 2 || step == 11) ? fp_sqr(a) : fp_mul(a, b);
    switch (step) {
      case 0: T0 = r; break;                                         // U1
      case 1: T1 = fp_sub<FqParams, 2>(r, T0); break;                // P [4]
      case 2: T2 = r; same_x = fp_is_zero_2p_rare(r); break;              // PP
      case 3: T1 = r; break;                                         // PPP
      case 4: T0 = r; break;                                         // Q
      case 5: lds_st(zz, r); break;
      case 6: lds_st(zz, r); break;                                  // ZZ3
      case 7: ty = r; break;                                         // S1 (Y1 is dead)
      case 8: T3 = fp_sub<FqParams, 2>(r, ty); break;                // R [4]
      case 9: lds_st(zzz, r); break;
      case 10: lds_st(zzz, r); break;                                // ZZZ3
      default: {
        Fq X3 = fp_sub_sub2<FqParams, 8>(r, T1, T0);                 // X3 = RR - PPP - 2Q [10], one carry pass
        lds_st_packed(xs, X3);
        T0 = fp_sub<FqParams, 16>(T0, X3);                           // Q - X3 [18]
        break;
      }
    }
    if (same_x) break;
  }
  if (!same_x) {
    ty = fp_mul2(T3, T0, fp_sub<FqParams, 2>(fp_zero<FqParams>(), ty), T1);     // Y3 = R (Q - X3) + (2p - S1) PPP  [2]
  } else {            // PP = 0 at step 2: the accumulator is untouched so far
    mem_st(spill, CX, lds_ld_packed(xs));
    mem_st(spill, CY, ty);
    mem_st(spill, CZZ, lds_ld(zz));
    mem_st(spill, CZZZ, lds_ld(zzz));
    add_same_x(spill, B);
    lds_st_packed(xs, fp_cond_sub_p(fp_mul(mem_ld(spill, CX), fp_one<FqParams>())));   // any bound -> canonical
    ty = mem_ld(spill, CY);
    lds_st(zz, mem_ld(spill, CZZ));
    lds_st(zzz, mem_ld(spill, CZZZ));
  }
  return same_x;
}

// a (memory) += b (memory); both may be infinite; a is updated in place, b is not written.  Per-lane LDS scratch of 2 x 27 words
// (limb-major images, ZK_LDS_STRIDE lanes per block): A's ZZ and ZZZ are fetched once instead of three times each, and the two-step products ZZ1 PP ZZ2 / ZZZ1 PPP ZZZ2 keep their intermediate in
// LDS instead of a global round trip: 15 coordinate transfers per addition instead of 22 (the throughput-bound reduction
// launches run at the L2's bandwidth, not at the multiplier's).
__device__ __forceinline__ void add_mem_s(const XyzzRef& A, const XyzzRef& B, uint32_t* zz, uint32_t* zzz) {
  if (mem_is_inf(B)) return;
  {
    Fq z1 = mem_ld(A, CZZ);
    if (fp_is_zero_2p(z1)) { mem_copy(A, B); return; }
    lds_st(zz, z1);
    lds_st(zzz, mem_ld(A, CZZZ));
  }
  Fq T0 = fp_zero<FqParams>(), T1 = T0, T2 = T0, T3 = T0;
  bool same_x = false;
  // twelve single products, then Y3 = R (Q - X3) + (-S1) PPP as one dual product (one reduction: fp_mul2)
#pragma unroll 1
  for (int step = 0; step < 12; step++) {
    Fq a, b;
    switch (step) {
      case 0: a = mem_ld(A, CX); b = mem_ld(B, CZZ); break;          // U1
      case 1: a = mem_ld(B, CX); b = lds_ld(zz); break;              // U2
      case 2: a = T1; b = T1; break;                                 // PP
      case 3: a = T1; b = T2; break;                                 // PPP
      case 4: a = T0; b = T2; break;                                 // Q = U1 PP
      case 5: a = lds_ld(zz); b = T2; break;                         // ZZ1 PP
      case 6: a = lds_ld(zz); b = mem_ld(B, CZZ); break;             // (ZZ1 PP) ZZ2
      case 7: a = mem_ld(A, CY); b = mem_ld(B, CZZZ); break;         // S1
      case 8: a = mem_ld(B, CY); b = lds_ld(zzz); break;             // S2
      case 9: a = lds_ld(zzz); b = T1; break;                        // ZZZ1 PPP
      case 10: a = lds_ld(zzz); b = mem_ld(B, CZZZ); break;          // (ZZZ1 PPP) ZZZ2
      default: a = T3; b = T3; break;                                // RR
    }
    Fq r = fp_mul(a, b);
    switch (step) {
      case 0: T0 = r; break;                                         // U1
      case 1: T1 = fp_sub<FqParams, 2>(r, T0); break;                // P [4]
      case 2: T2 = r; same_x = fp_is_zero_2p_rare(r); break;              // PP
      case 3: T1 = r; break;                                         // PPP
      case 4: T0 = r; break;                                         // Q
      case 5: lds_st(zz, r); break;
      case 6: mem_st(A, CZZ, r); break;                              // ZZ3
      case 7: T2 = r; break;                                         // S1
      case 8: T3 = fp_sub<FqParams, 2>(r, T2); break;                // R [4]
      case 9: lds_st(zzz, r); break;
      case 10: mem_st(A, CZZZ, r); break;                            // ZZZ3
      default: {
        Fq X3 = fp_sub_sub2<FqParams, 8>(r, T1, T0);                 // X3 = RR - PPP - 2Q [10], one carry pass
        mem_st(A, CX, X3);
        T0 = fp_sub<FqParams, 16>(T0, X3);                           // Q - X3 [18]
        break;
      }
    }
    if (same_x) break;
  }
  if (same_x) { add_same_x(A, B); return; }   // memory still holds A's original ZZ, ZZZ at this point (first stores are steps 6 and 10)
  mem_st(A, CY, fp_mul2(T3, T0, fp_sub<FqParams, 2>(fp_zero<FqParams>(), T2), T1));   // Y3 = R (Q - X3) + (2p - S1) PPP  [2]
}

// a (memory) = 2 a.   dbl-2008-s-1 as a micro-program: 9 multiplications
__device__ __forceinline__ void dbl_mem(const XyzzRef& A) {
  if (mem_is_inf(A)) return;
  Fq T0 = fp_zero<FqParams>(), T1 = T0, T2 = T0, T3 = T0;
  // seven single products, then Y3 = M (S - X3) + W (-Y1) as one dual product (fp_mul2)
#pragma unroll 1
  for (int step = 0; step < 7; step++) {
    Fq a, b;
    switch (step) {
      case 0: a = fp_dbl(mem_ld(A, CY)); b = a; break;               // V = U^2, U = 2 Y1 [8]
      case 1: a = fp_dbl(mem_ld(A, CY)); b = T0; break;              // W = U V
      case 2: a = mem_ld(A, CX); b = T0; break;                      // S = X1 V
      case 3: a = mem_ld(A, CZZ); b = T0; break;                     // ZZ3 = V ZZ1
      case 4: a = mem_ld(A, CZZZ); b = T1; break;                    // ZZZ3 = W ZZZ1
      case 5: a = mem_ld(A, CX); b = a; break;                       // X1^2
      default: a = T3; b = T3; break;                                // M^2
    }
    Fq r = fp_mul(a, b);
    switch (step) {
      case 0: T0 = r; break;                                         // V
      case 1: T1 = r; break;                                         // W
      case 2: T2 = r; break;                                         // S
      case 3: mem_st(A, CZZ, r); break;                              // (U = 0 gives ZZ3 = 0: infinity)
      case 4: mem_st(A, CZZZ, r); break;
      case 5: T3 = fp_add(fp_dbl(r), r); break;                      // M = 3 X1^2 [6]
      default: T0 = fp_sub<FqParams, 4>(r, fp_dbl(T2)); break;       // X3 = M^2 - 2S [6]
    }
  }
  // Y3 = M (S - X3) + W (4p - Y1): [6] x [10] + [2] x [4] -> [2]   (Y1 is read before X3 overwrites nothing it needs: CX only)
  Fq Y3 = fp_mul2(T3, fp_sub<FqParams, 8>(T2, T0), T1, fp_sub<FqParams, 4>(fp_zero<FqParams>(), mem_ld(A, CY)));
  mem_st(A, CX, T0);
  mem_st(A, CY, Y3);
}


// ================================================================================================
// Quad-cooperative point operations for the latency-bound kernels (bucket reduction, stitching).
// One lane needs ~6 us per Fq multiplication whatever the occupancy, and a full addition is a chain
// of 14 of them; the reduction levels have far fewer additions than the chip has lanes.  So FOUR
// lanes (a DPP quad) execute one addition: the formula's independent products run side by side
// (4 + 4 + 3 + 3 products in 4 rounds instead of 14 in sequence), operands travel between the lanes
// of a quad with DPP quad_perm moves.  Every lane of a quad is given the SAME operands; lane q of the
// quad performs the q-th product of each round.  Results are written by the lane that computed them.
// ================================================================================================
template <int Q>
__device__ __forceinline__ Fq quad_bcast(const Fq& v) {
  Fq r;
  constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6);   // quad_perm: [Q, Q, Q, Q]
#pragma unroll
  for (int i = 0; i < 27; i++) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.l[i], ctrl, 0xf, 0xf, true);
  return r;
}
__device__ __forceinline__ Fq fq_sel(bool c, const Fq& a, const Fq& b) {
  Fq r;
#pragma unroll
  for (int i = 0; i < 27; i++) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}
// coordinate chosen per lane (the row base goes into the lane offset instead of the scalar offset)
__device__ __forceinline__ Fq mem_ld_lane(const XyzzRef& r, uint32_t c) {
  Fq v;
  uint32_t vo = r.voff + c * r.coord_b;
#if ZK_SLOTS_AOS
  if (r.coord_b == 112u) return slot_ld_wide(r.rs, vo, 0u);
#endif
#pragma unroll
  for (int i = 0; i < 27; i++) v.l[i] = __builtin_amdgcn_raw_buffer_load_b32(r.rs, vo, (uint32_t)i * r.stride_b, 0);
  return v;
}

__device__ __forceinline__ void mem_st_lane(const XyzzRef& r, uint32_t c, const Fq& v) {
  uint32_t vo = r.voff + c * r.coord_b;
#if ZK_SLOTS_AOS
  if (r.coord_b == 112u) { slot_st_wide(r.rs, vo, 0u, v); return; }
#endif
#pragma unroll
  for (int i = 0; i < 27; i++) __builtin_amdgcn_raw_buffer_store_b32(v.l[i], r.rs, vo, (uint32_t)i * r.stride_b, 0);
}

// A += B, executed by the four lanes of a quad (all four pass the same A and B).
__device__ __forceinline__ void add_mem_quad(const XyzzRef& A, const XyzzRef& B, uint32_t q) {
  if (mem_is_inf(B)) return;
  if (mem_is_inf(A)) { mem_st_lane(A, q, mem_ld_lane(B, q)); return; }   // lane q copies coordinate q
  // round 1: q0 U1 = X1 ZZ2, q1 U2 = X2 ZZ1, q2 S1 = Y1 ZZZ2, q3 S2 = Y2 ZZZ1
  // round 2: q0 PP = P^2,    q1 RR = R^2,     q2 zz = ZZ1 ZZ2,  q3 zzz = ZZZ1 ZZZ2
  // round 3: q0 PPP = P PP,  q1 Q = U1 PP,    q2 ZZ3 = zz PP,   q3 -
  // round 4: q0 Y3b = S1 PPP, q1 Y3a = R (Q - X3), q2 -,        q3 ZZZ3 = zzz PPP
  Fq U1 = fp_zero<FqParams>(), P = U1, R = U1, S1 = U1, PP = U1, PPP = U1, keep = U1;   // keep: RR (q1), zz (q2), zzz (q3)
  Fq Qv = U1;
  bool same_x = false;
#pragma unroll 1
  for (int round = 0; round < 4; round++) {
    Fq a, b;
    if (round == 0) {
      const bool odd = (q & 1u) != 0;                       // q1, q3 take the first factor from B
      a = odd ? mem_ld_lane(B, q == 1 ? CX : CY) : mem_ld_lane(A, q == 0 ? CX : CY);
      b = odd ? mem_ld_lane(A, q == 1 ? CZZ : CZZZ) : mem_ld_lane(B, q == 0 ? CZZ : CZZZ);
    } else if (round == 1) {
      a = (q == 0) ? P : (q == 1) ? R : mem_ld_lane(A, q == 2 ? CZZ : CZZZ);
      b = (q == 0) ? P : (q == 1) ? R : mem_ld_lane(B, q == 2 ? CZZ : CZZZ);
    } else if (round == 2) {
      a = (q == 0) ? P : (q == 1) ? U1 : keep;              // q3: idle product (result unused)
      b = PP;
    } else {
      a = (q == 0) ? S1 : (q == 1) ? R : keep;
      b = (q == 1) ? fp_sub<FqParams, 16>(Qv, P /* holds X3 on q1 */) : PPP;
    }
    Fq r = fp_mul(a, b);
    if (round == 0) {
      Fq u1 = quad_bcast<0>(r), u2 = quad_bcast<1>(r), s1 = quad_bcast<2>(r), s2 = quad_bcast<3>(r);
      U1 = u1; S1 = s1;
      P = fp_sub<FqParams, 2>(u2, u1);                      // [4]
      R = fp_sub<FqParams, 2>(s2, s1);                      // [4]
    } else if (round == 1) {
      keep = r;                                             // q1: RR, q2: zz, q3: zzz
      PP = quad_bcast<0>(r);
      same_x = fp_is_zero_2p(PP);
    } else if (round == 2) {
      PPP = quad_bcast<0>(r);
      Qv = r;                                               // meaningful on q1 (Q)
      if (q == 2) mem_st(A, CZZ, r);                        // ZZ3
      // X3 = RR - PPP - 2Q on q1 (RR is its `keep`); parked in P, which q1 no longer needs
      if (q == 1) { P = fp_sub<FqParams, 4>(fp_sub<FqParams, 4>(keep, PPP), fp_dbl(Qv)); mem_st(A, CX, P); }
    } else {
      Fq y3b = quad_bcast<0>(r);
      if (q == 1) mem_st(A, CY, fp_sub<FqParams, 2>(r, y3b));   // Y3 [4]
      if (q == 3) mem_st(A, CZZZ, r);                       // ZZZ3
    }
    if (same_x) break;
  }
  if (same_x && q == 0) add_same_x(A, B);
}

// A = 2 A by a quad.
__device__ __forceinline__ void dbl_mem_quad(const XyzzRef& A, uint32_t q) {
  if (mem_is_inf(A)) return;
  // round 1: q0 V = U^2 (U = 2 Y1), q1 xx = X1^2
  // round 2: q0 W = U V, q1 S = X1 V, q2 ZZ3 = V ZZ1, q3 MM = M^2 (M = 3 xx)
  // round 3: q0 ZZZ3 = W ZZZ1, q1 wy = W Y1, q3 Y3a = M (S - X3), X3 = MM - 2 S
  Fq V = fp_zero<FqParams>(), M = V, keep = V, W = V, S = V;
#pragma unroll 1
  for (int round = 0; round < 3; round++) {
    Fq a, b;
    if (round == 0) {
      a = (q == 0) ? fp_dbl(mem_ld(A, CY)) : mem_ld(A, CX);          // q2, q3: idle
      b = a;
    } else if (round == 1) {
      a = (q == 0) ? fp_dbl(mem_ld(A, CY)) : (q == 1) ? mem_ld(A, CX) : (q == 2) ? mem_ld(A, CZZ) : M;
      b = (q == 3) ? M : V;
    } else {
      a = (q == 0) ? mem_ld(A, CZZZ) : (q == 1) ? mem_ld(A, CY) : M;
      b = (q == 3) ? fp_sub<FqParams, 8>(S, keep /* X3 on q3 */) : W;
    }
    Fq r = fp_mul(a, b);
    if (round == 0) {
      V = quad_bcast<0>(r);
      Fq xx = quad_bcast<1>(r);
      M = fp_add(fp_dbl(xx), xx);                                     // [6]
    } else if (round == 1) {
      W = quad_bcast<0>(r);
      S = quad_bcast<1>(r);
      if (q == 2) mem_st(A, CZZ, r);                                  // ZZ3
      if (q == 3) { keep = fp_sub<FqParams, 4>(r, fp_dbl(S)); }       // X3 = MM - 2S [6]
    } else {
      Fq wy = quad_bcast<1>(r);
      if (q == 0) mem_st(A, CZZZ, r);
      if (q == 3) { mem_st(A, CY, fp_sub<FqParams, 2>(r, wy)); mem_st(A, CX, keep); }
    }
  }
}

}  // namespace zkhip
