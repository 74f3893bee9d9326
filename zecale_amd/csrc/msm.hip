// Multi-scalar multiplication in G1 / G2 of BW6-761 on gfx950 (Pippenger / bucket method with
// signed digits).  Replaces the five libff::multi_exp calls the reference reaches through
// wsnarkT::generate_proof (libzecale/circuits/aggregator_circuit.tcc:168).
//
// Pipeline (all kernels hand-written for wave64; one lane owns one field element, see fp29.cuh):
//   k_bases_to_dev      ABI affine points -> packed device-form points (once per base set)
//   k_table_build       optional, once per resident base set: 2^(off_w) P_i for every window position (window table)
//   k_digit_pass<0/1>   Montgomery scalars -> canonical -> signed c-bit digits; counted, then placed, per PART of the bucket range (LDS)
//   k_scan_*            exclusive prefix sum of the (part, block) histogram
//   k_bucket_sort       a workgroup per part orders its entries by bucket: bucket-ordered list of (point index, sign), offsets, counts
//   k_accumulate        one lane per fixed-size SLICE of the sorted list: XYZZ mixed additions   <-- dominant
//   k_fixup_fold / k_fixup  stitch the buckets that slice boundaries cut
//   k_sum / k_seg       bucket reduction  sum_j (j+1) B_j : two-level split of the bucket index, then L-ary running sums
//   k_window_combine / k_hilo_combine   per bucket window: Horner over the levels, R * hi + lo
// Plain base sets: one bucket window per digit position, and the last step, sum_w 2^(off_w) W_w (about 380 serial
// doublings of ONE point), runs on the host.  Table-backed base sets: ONE bucket window for all digit positions (nothing
// left to combine), and up to five MSMs share one launch sequence with a bucket window each (msm_launch_multi).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>

#include "ec_affine.cuh"
#include "host_field.hpp"
#include "msm.h"
#include "wait.h"


namespace zkhip {

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------

// one thread per point: 24 u64 (x, y in ABI Montgomery form; all-zero = infinity) -> AffPacked
__global__ void __launch_bounds__(256) k_bases_to_dev(const uint64_t* __restrict__ in, AffPacked* __restrict__ out,
                                                       uint8_t* __restrict__ inf_flags /* may be null */, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x[12], y[12];
  uint64_t nz = 0;
#pragma unroll
  for (int k = 0; k < 12; k++) { x[k] = in[i * 24 + k]; y[k] = in[i * 24 + 12 + k]; nz |= x[k] | y[k]; }
  AffPacked p;
  if (nz == 0) {
#pragma unroll
    for (int k = 0; k < 24; k++) { p.x[k] = 0; p.y[k] = 0; }
  } else {
    Fq fx = fp_cond_sub_p(fp_from_abi<FqParams>(x));
    Fq fy = fp_cond_sub_p(fp_from_abi<FqParams>(y));
    fp_pack32<FqParams>(fx, p.x);
    fp_pack32<FqParams>(fy, p.y);
  }
  out[i] = p;
  if (inf_flags) inf_flags[i] = (nz == 0) ? 1 : 0;
}

// Window layout: W windows tile exactly 378 bits (scalars < r < 2^377, plus one bit for the
// signed-digit carry); window sizes differ by at most one bit (c or c-1) so that no window is a
// sparsely populated remainder - a short top window would funnel all n points of that window
// into a few hundred buckets, i.e. into a few hundred lanes.
struct WindowPlan {
  uint16_t off[96];   // first bit of window w
  uint8_t bits[96];   // size of window w
};

// ---- bucket sort of the digits: two passes with LDS histograms, no global atomics --------------------------------------
// A digit of a scalar is an ENTRY (which point, which sign) for a BUCKET (global index in [0, nb): window * B + |d| - 1, or
// job * B + |d| - 1 when all digit positions of a job share one bucket window).  k_accumulate wants the entries ordered by
// bucket.  bucket = part << LB | low:
//   k_digit_pass<0>   a block takes a tile of scalars, extracts their digits and counts them per PART in LDS; the counts go out
//                     as hist[part][block]
//   k_scan_*          exclusive scan of hist in (part, block) order: where each block's entries of each part start
//   k_digit_pass<1>   the same blocks extract the same digits again (two Fr multiplications per scalar: cheaper than storing
//                     and re-reading 4 bytes per digit), take their ranks from an LDS counter and write (entry, low bucket
//                     bits) pairs - one 8-byte store each -, grouped by part
//   k_bucket_sort     a workgroup per part: LDS histogram over the low bits, scan, placement - and the per-bucket
//                     offsets / counts the accumulation needs fall out on the way
// (Rounds 1-2 counted in global memory: one atomic per digit in the histogram and one RETURNING atomic per digit in the
// scatter, 1.6 ms of a 2^20-term MSM - 20 M device-scope atomics resolve in the memory-side cache, not in an XCD's L2.)
// The order of the entries inside a bucket is not defined (the sum is).
struct DigitJobs {
  const uint64_t* scalars[MSM_MAX_JOBS];
  const uint8_t* inf_flags[MSM_MAX_JOBS];
  size_t n[MSM_MAX_JOBS], tab_stride[MSM_MAX_JOBS];
  int mode[MSM_MAX_JOBS];
};
struct SortGeom {
  uint32_t LB;          // low bucket bits: a part is 2^LB consecutive buckets
  uint32_t NP;          // parts per bucket window (B >> LB)
  uint32_t bins;        // parts a block can hit: NP (merged plans: its job's window) or W * NP (plain plans: every window)
  uint32_t nbx;         // blocks per job = row length of hist
  uint32_t tile;        // scalars per block
  uint32_t nparts;      // nb >> LB
};

// canonical 32-bit words (13, the last one zero) of scalar i of a job
__device__ __forceinline__ void scalar_words(const uint64_t* __restrict__ scalars, size_t i, int mode, uint32_t* w32) {
  uint64_t s[6];
#pragma unroll
  for (int k = 0; k < 6; k++) s[k] = scalars[i * 6 + k];
  if (mode == 1) {
    fp_abi_to_canonical_words<FrParams>(s, w32);
  } else {
#pragma unroll
    for (int k = 0; k < 6; k++) { w32[2 * k] = (uint32_t)s[k]; w32[2 * k + 1] = (uint32_t)(s[k] >> 32); }
    if (mode == 2) {   // packed device form (value * 2^406, < 2^384), as the NTT kernels leave it
      Fp<FrParams> v = fp_unpack32<FrParams>(w32), one_raw = fp_zero<FrParams>();
      one_raw.l[0] = 1;
      fp_pack32<FrParams>(fp_cond_sub_p(fp_mul(v, one_raw)), w32);
    }
  }
  w32[12] = 0;
}

// PASS 0: count.  PASS 1: place.  NAF: scalars recoded in width-(c+1) non-adjacent form (merged == 2 plans).
// Dynamic LDS: cnt[bins] (+ base[bins] in pass 1).
template <int PASS, bool NAF>
__global__ void __launch_bounds__(256) k_digit_pass(DigitJobs jobs, int c, int W, WindowPlan plan, int merged, SortGeom ge,
                                                     uint32_t* __restrict__ hist, uint2* __restrict__ pairs) {
  extern __shared__ uint32_t sh_dyn[];
  __shared__ uint32_t s_w[NAF ? 13 : 1][256];
  uint32_t* s_cnt = sh_dyn;
  uint32_t* s_base = sh_dyn + ge.bins;
  const uint32_t job = blockIdx.y, bx = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
  const size_t n = jobs.n[job];
  const uint64_t* __restrict__ scalars = jobs.scalars[job];
  const uint8_t* __restrict__ inf_flags = jobs.inf_flags[job];
  const int mode = jobs.mode[job];
  const size_t tab_stride = jobs.tab_stride[job];
  const uint32_t g0 = merged ? job * ge.NP : 0u;                 // first part this block can hit
  const uint32_t lb_mask = (1u << ge.LB) - 1u;
  for (uint32_t j = tid; j < ge.bins; j += 256) {
    s_cnt[j] = 0;
    if (PASS == 1) s_base[j] = hist[(size_t)(g0 + j) * ge.nbx + bx];
  }
  if (PASS == 0 && bx == 0 && job == 0 && tid == 0) hist[(size_t)ge.nparts * ge.nbx] = 0;     // the scan leaves the total here
  __syncthreads();
  // one digit: bin = part inside this block's range, low = bucket bits below the part, e = entry word
  auto emit = [&](bool valid, bool hot, uint32_t bin, uint32_t low, uint32_t e) {
    // "scalar == 1" (boolean-heavy witnesses) puts a third of all points into ONE bucket: its lanes share one LDS operation
    const unsigned long long hot_mask = __ballot(valid && hot);
    if (hot_mask) {
      const int leader = __ffsll((long long)hot_mask) - 1;
      const uint32_t cntw = (uint32_t)__popcll(hot_mask);
      if (PASS == 0) {
        if ((int)lane == leader) atomicAdd(&s_cnt[bin], cntw);
      } else {
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&s_cnt[bin], cntw);
        base = __shfl(base, leader);
        if (valid && hot) {
          const uint32_t pos = s_base[bin] + base + (uint32_t)__popcll(hot_mask & ((1ull << lane) - 1ull));
          pairs[pos] = make_uint2(e, low);
        }
      }
    }
    if (valid && !hot) {
      if (PASS == 0) {
        atomicAdd(&s_cnt[bin], 1u);
      } else {
        const uint32_t pos = s_base[bin] + atomicAdd(&s_cnt[bin], 1u);
        pairs[pos] = make_uint2(e, low);
      }
    }
  };
  for (uint32_t it = 0; it < ge.tile; it += 256) {
    size_t i = (size_t)bx * ge.tile + it + tid;
    const bool live = i < n;
    if ((size_t)bx * ge.tile + it >= n) break;              // (block-uniform)
    if (!live) i = n - 1;                                   // whole waves stay in the loops: the hot-bucket path is wave-wide
    const bool skip0 = !live || (inf_flags && inf_flags[i]);   // a base at infinity contributes nothing: drop it here
    uint32_t w32[13];
    scalar_words(scalars, i, mode, w32);
    if constexpr (NAF) {
      // Width-(c+1) non-adjacent form: ODD signed digits |d| < 2^c at arbitrary bit positions, at least c+1 bits apart - 378 / (c + 2)
      // digits on average instead of 378 / c, over the same 2^(c-1) buckets (bucket = (|d| - 1) / 2, weight 2 * bucket + 1: the
      // reduction returns sum (bucket + 1) S_b and sum S_b, the finish makes 2 F - S of them).  The entry of a digit at bit position
      // j points at level j of the table (2^j P_i).  The LAST digit of a scalar covers only the bits that are left, so small
      // magnitudes are over-represented - LDS counters take that in their stride.
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 13; k++) s_w[k][tid] = w32[k];
      __syncthreads();
      const int wbits = c + 1;
      int pos = 0, slot = 0;
      uint32_t carry = 0;
      while (pos < 379 && slot < W) {
        const int j = pos >> 5, sh = pos & 31;
        uint64_t v = (uint64_t)s_w[j < 13 ? j : 12][tid] >> sh;
        if (j < 12) v |= (uint64_t)s_w[j + 1][tid] << (32 - sh);
        if (j >= 12) v = 0;
        const uint32_t v32 = (uint32_t)v;
        // bits + carry: skip the run that produces zeros (zeros without a carry, ones with it)
        const uint32_t run = carry ? ~v32 : v32;
        if ((run & 1u) == 0) { pos += run ? (__ffs((int)run) - 1) : 32; continue; }
        uint32_t win = (v32 & ((1u << wbits) - 1)) + carry;            // odd
        int32_t d;
        if (win > (1u << (wbits - 1))) { d = (int32_t)win - (int32_t)(1u << wbits); carry = 1; }
        else { d = (int32_t)win; carry = 0; }
        const uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
        const bool skip = skip0 || (inf_flags && pos > 0 && inf_flags[(size_t)pos * tab_stride + i]);      // 2^pos P_i = O
        if (!skip) {
          const uint32_t bl = (mag - 1) >> 1;
          const uint32_t e = (uint32_t)((size_t)pos * tab_stride + i) | (d < 0 ? 0x80000000u : 0u);
          // (divergent loop: no wave-wide aggregation here)
          if (PASS == 0) {
            atomicAdd(&s_cnt[bl >> ge.LB], 1u);
          } else {
            const uint32_t p_ = s_base[bl >> ge.LB] + atomicAdd(&s_cnt[bl >> ge.LB], 1u);
            pairs[p_] = make_uint2(e, bl & lb_mask);
          }
        }
        slot++;
        pos += wbits;
      }
    } else {
      uint32_t carry = 0;
      for (int w = 0; w < W; w++) {
        const int bit = plan.off[w], cw = plan.bits[w];
        const int j = bit >> 5, sh = bit & 31;
        uint64_t v = (uint64_t)w32[j] >> sh;
        if (j + 1 < 13) v |= (uint64_t)w32[j + 1] << (32 - sh);
        uint32_t d = ((uint32_t)v & ((1u << cw) - 1)) + carry;
        int32_t sd;
        if (d > (1u << (cw - 1))) { sd = (int32_t)d - (int32_t)(1u << cw); carry = 1; }
        else { sd = (int32_t)d; carry = 0; }
        const bool skip = skip0 || (merged && w > 0 && inf_flags && inf_flags[(size_t)w * tab_stride + i]);   // 2^(c w) P_i = O
        const bool valid = !skip && sd != 0;
        const uint32_t mag = sd < 0 ? (uint32_t)(-sd) : (uint32_t)sd;
        const uint32_t bw = valid ? mag - 1 : 0u;                                   // bucket inside its window
        const uint32_t bin = (merged ? 0u : (uint32_t)w * ge.NP) + (bw >> ge.LB);
        const uint32_t e = (uint32_t)(merged ? (size_t)w * tab_stride + i : i) | (sd < 0 ? 0x80000000u : 0u);
        emit(valid, mag == 1, bin, bw & lb_mask, e);
      }
    }
  }
  if (PASS == 0) {
    __syncthreads();
    for (uint32_t j = tid; j < ge.bins; j += 256) hist[(size_t)(g0 + j) * ge.nbx + bx] = s_cnt[j];
  }
}

// One workgroup per part g: its entries pairs[start, end) (grouped by part by k_digit_pass<1>) are ordered by their low bucket
// bits.  Writes offsets / counts of the part's 2^LB buckets (an empty bucket repeats the next offset) and the sorted entries.
__global__ void __launch_bounds__(512) k_bucket_sort(const uint32_t* __restrict__ hoff, SortGeom ge, const uint2* __restrict__ pairs,
                                                      uint32_t* __restrict__ entries,
                                                      uint32_t* __restrict__ offsets, uint32_t* __restrict__ counts) {
  __shared__ uint32_t s_cnt[1024], s_cur[1024], s_scan[512];
  const uint32_t g = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
  const uint32_t start = hoff[(size_t)g * ge.nbx], end = hoff[(size_t)(g + 1) * ge.nbx];     // (hoff has nparts * nbx + 1 elements)
  const uint32_t NB = 1u << ge.LB;
  for (uint32_t j = tid; j < NB; j += 512) s_cnt[j] = 0;
  __syncthreads();
  const uint32_t len = end - start, rounds = (len + 511) / 512;
  for (uint32_t r = 0; r < rounds; r++) {
    const uint32_t k = r * 512 + tid;
    const bool valid = k < len;
    const uint32_t lb = valid ? pairs[start + k].y : 0xffffu;
    const unsigned long long hot_mask = __ballot(lb == 0);       // the bucket of "digit 1" may hold a third of all entries
    if (hot_mask) {
      if ((int)lane == __ffsll((long long)hot_mask) - 1) atomicAdd(&s_cnt[0], (uint32_t)__popcll(hot_mask));
    }
    if (valid && lb != 0) atomicAdd(&s_cnt[lb], 1u);
  }
  __syncthreads();
  // exclusive scan of s_cnt[0 .. NB): two counters per thread
  const uint32_t c0 = (2 * tid < NB) ? s_cnt[2 * tid] : 0u, c1 = (2 * tid + 1 < NB) ? s_cnt[2 * tid + 1] : 0u;
  s_scan[tid] = c0 + c1;
  __syncthreads();
  for (int off = 1; off < 512; off <<= 1) {
    const uint32_t t_ = (tid >= (unsigned)off) ? s_scan[tid - off] : 0u;
    __syncthreads();
    s_scan[tid] += t_;
    __syncthreads();
  }
  const uint32_t ex = s_scan[tid] - (c0 + c1);
  if (2 * tid < NB) {
    s_cur[2 * tid] = ex;
    offsets[((size_t)g << ge.LB) + 2 * tid] = start + ex; counts[((size_t)g << ge.LB) + 2 * tid] = c0;
  }
  if (2 * tid + 1 < NB) {
    s_cur[2 * tid + 1] = ex + c0;
    offsets[((size_t)g << ge.LB) + 2 * tid + 1] = start + ex + c0; counts[((size_t)g << ge.LB) + 2 * tid + 1] = c1;
  }
  __syncthreads();
  for (uint32_t r = 0; r < rounds; r++) {
    const uint32_t k = r * 512 + tid;
    const bool valid = k < len;
    const uint2 pr = valid ? pairs[start + k] : make_uint2(0u, 0xffffu);
    const uint32_t lb = pr.y, e = pr.x;
    const unsigned long long hot_mask = __ballot(lb == 0);
    if (hot_mask) {
      const int leader = __ffsll((long long)hot_mask) - 1;
      uint32_t base = 0;
      if ((int)lane == leader) base = atomicAdd(&s_cur[0], (uint32_t)__popcll(hot_mask));
      base = __shfl(base, leader);
      if (lb == 0) entries[start + base + (uint32_t)__popcll(hot_mask & ((1ull << lane) - 1ull))] = e;
    }
    if (valid && lb != 0) entries[start + atomicAdd(&s_cur[lb], 1u)] = e;
  }
}

// ---- exclusive scan over `m` u32 counters, 1024 per block ----
// (in and out may be the same array - the histogram is scanned in place -: no __restrict__ on them)
__global__ void __launch_bounds__(256) k_scan_local(const uint32_t* in, uint32_t* out, uint32_t* __restrict__ block_tot, size_t m) {
  __shared__ uint32_t sh[256];
  size_t base = (size_t)blockIdx.x * 1024 + (size_t)threadIdx.x * 4;
  uint32_t v[4], sum = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { v[k] = (base + k < m) ? in[base + k] : 0; sum += v[k]; }
  sh[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    uint32_t t = (threadIdx.x >= (unsigned)off) ? sh[threadIdx.x - off] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t excl = sh[threadIdx.x] - sum;
#pragma unroll
  for (int k = 0; k < 4; k++) { if (base + k < m) out[base + k] = excl; excl += v[k]; }
  if (threadIdx.x == 255) block_tot[blockIdx.x] = sh[255];
}

__global__ void __launch_bounds__(1024) k_scan_tot(uint32_t* __restrict__ block_tot, size_t nb) {
  // single block, sequential over chunks of 1024
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (size_t base = 0; base < nb; base += 1024) {
    size_t i = base + threadIdx.x;
    uint32_t v = (i < nb) ? block_tot[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      uint32_t t = (threadIdx.x >= (unsigned)off) ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) block_tot[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) k_scan_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ block_tot, size_t m) {
  size_t base = (size_t)blockIdx.x * 1024 + (size_t)threadIdx.x * 4;
  uint32_t add = block_tot[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 4; k++) if (base + k < m) out[base + k] += add;
}

// Bucket accumulation, load-balanced: lane t owns a SLICE of the bucket-sorted entry list, whatever buckets it crosses.  A run of
// entries that covers a whole bucket is accumulated straight into that bucket's slot; a run cut by a slice boundary goes to a
// per-slice boundary slot (F = the slice starts inside the bucket, L = the slice ends inside it) and k_fixup stitches the pieces.
// All slots live in ONE array (one buffer descriptor; an array of structures, ec_mem.cuh): [0, nb) buckets, [nb, nb+T) F slots, [nb+T, nb+2T) L slots.
// Slices have equal WEIGHT, not equal length (round 4): the first entry of a bucket only OPENS a run - loads, no field arithmetic -
// so it weighs ZK_W_FIRST = 1 where an entry that is ADDED to a running sum weighs ZK_W_NEXT = 8.  Every lane then performs the same
// number of additions (to within one) and the openings ride along inside the iteration of the addition that follows them; with
// slices of equal length (rounds 1-3) every opening took a whole iteration of its wave: 7 % of the lane-iterations of a wrapping
// proof's launch, 5 % at 2^20 terms, did nothing.  (Thread-per-bucket loses ~45 % to the spread of bucket populations.)
// goff[b] = the weight in front of bucket b (exclusive scan of 8 count - 7 over the non-empty buckets, nb + 1 values); the entry at
// position p of bucket b starts at weight G(p) = goff[b] for p = offsets[b], goff[b] + 1 + 8 (p - offsets[b] - 1) otherwise; slice t
// holds the entries with t A <= G(p) < (t + 1) A.  The slot array's ZZ rows are zero-filled beforehand (all-zero ZZ = infinity), so
// empty buckets need no work.
#define ZK_W_NEXT 8u
#define ZK_W_FIRST 1u
__global__ void __launch_bounds__(256) k_slice_weights(const uint32_t* __restrict__ counts, uint32_t* __restrict__ w, size_t nb) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > nb) return;
  const uint32_t c = i < nb ? counts[i] : 0u;
  w[i] = c ? ZK_W_NEXT * c - (ZK_W_NEXT - ZK_W_FIRST) : 0u;
}
// first position p with G(p) >= x (M if there is none); *b_out: a bucket at or before the one that holds p
__device__ __forceinline__ uint32_t slice_pos(const uint32_t* __restrict__ goff, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                                              uint32_t nb, uint32_t M, uint64_t x, uint32_t* b_out) {
  if (x >= (uint64_t)goff[nb]) { *b_out = nb - 1; return M; }
  uint32_t lo = 0, hi = nb;   // last b with goff[b] <= x: a non-empty bucket (an empty one repeats the value of the next non-empty one)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((uint64_t)goff[mid] <= x) lo = mid; else hi = mid;
  }
  *b_out = lo;
  const uint32_t g = goff[lo], off = offsets[lo], cnt = counts[lo];
  if ((uint64_t)g >= x) return off;
  const uint32_t j = (uint32_t)((x - g - ZK_W_FIRST + (ZK_W_NEXT - 1)) / ZK_W_NEXT) + 1u;      // first j >= 1 with g + 1 + 8 (j - 1) >= x
  return off + (j < cnt ? j : cnt);
}
// slice of the entry at position p of bucket b
__device__ __forceinline__ uint32_t slice_of(const uint32_t* __restrict__ goff, const uint32_t* __restrict__ offsets, uint32_t b, uint32_t p, uint32_t A) {
  const uint32_t off = offsets[b];
  const uint64_t G = (uint64_t)goff[b] + (p == off ? 0u : ZK_W_FIRST + (uint64_t)ZK_W_NEXT * (p - off - 1));
  return (uint32_t)(G / A);
}
__device__ __forceinline__ uint32_t bucket_of(const uint32_t* __restrict__ offsets, uint32_t nb, uint32_t pos) {
  // last b with offsets[b] <= pos  (offsets is non-decreasing; empty buckets repeat the next offset)
  uint32_t lo = 0, hi = nb;   // invariant: offsets[lo] <= pos, (hi == nb or offsets[hi] > pos)
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (offsets[mid] <= pos) lo = mid; else hi = mid;
  }
  return lo;
}

// ZZ = 0 marks an empty slot (the point at infinity): the 27 ZZ words of every slot of the array of structures are cleared before a
// launch - a quarter of the array's bytes, as the limb-major layout's memset of its 27 ZZ rows was.  One word per lane, 27
// consecutive lanes per slot.
__global__ void __launch_bounds__(256) k_slots_clear_zz(uint32_t* __restrict__ slots, uint32_t n_slots) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_slots * 27) return;
  const size_t s_ = i / 27, w = i % 27;
  slots[s_ * ZK_SLOT_WORDS + (size_t)CZZ * (ZK_SLOT_WORDS / 4) + w] = 0u;
}

// Several MSMs may share one launch (merged plans: bucket window k belongs to job k): the base set of a run is chosen
// by its bucket, b >> bshift (plain plans: one base set, bshift = 31).
// NJ = 1: a plan for single MSMs (the selection folds away); NJ = MSM_MAX_JOBS: the five MSMs of a proof in one launch.  Two
// instantiations, two kernel names in a trace: k_accumulate<1> and k_accumulate<5>.
struct BasePtrs { const AffPacked* p[MSM_MAX_JOBS]; };
template <int NJ> __device__ __forceinline__ const AffPacked* base_of(const BasePtrs& bp, uint32_t k) {
  const AffPacked* r = bp.p[0];
#pragma unroll
  for (int j = 1; j < NJ; j++) r = (k == (uint32_t)j) ? bp.p[j] : r;
  return r;
}

// Slice weight A.  The host sizes the slices for the entries that CAN occur (digit positions x finite bases: S_host entries each); the
// list the sort produced may be shorter - zero digits produce no entry (a boolean-heavy witness: most of them), recoded scalars
// have fewer digits than positions.  A launch that has the chip to itself (tight != 0) then shrinks the slices so that all T lanes
// of the grid get work: A = ceil((G + 1) / T), at least 16 additions' worth, never above the host's (so the slot array still fits).
// A prover that shares the chip with others keeps the longer slices (fewer cut buckets to stitch; the chip is full anyway).
__device__ __forceinline__ uint32_t slice_weight(uint32_t G, uint32_t T, uint32_t S_host, int tight) {
  const uint32_t a_host = S_host * ZK_W_NEXT;
  if (!tight) return a_host;
  const uint32_t a = (uint32_t)(((uint64_t)G + T) / T), a_min = 16u * ZK_W_NEXT;
  if (a < a_min) return a_host < a_min ? a_host : a_min;
  return a < a_host ? a : a_host;
}

// Closing a run: the four coordinates go to the run's slot (ec_mem.cuh: the slot array is an array of structures)
#define ZK_CLOSE_RUN(acc, xs, zz, zzz, ty)                                                                          \
  do {                                                                                                               \
    mem_st(acc, CX, lds_ld_packed(xs)); mem_st(acc, CZZ, lds_ld(zz)); mem_st(acc, CZZZ, lds_ld(zzz));             \
    mem_st(acc, CY, ty);                                                                                             \
  } while (0)

// entries == nullptr: the sorted list IS the dense point array bp.p[0] (the output of the batched-affine levels, k_affine_level):
// entry k is point k, never negated; a point may be the level encoding of infinity (skipped).
template <int NJ>
__global__ void __launch_bounds__(256, 2) k_accumulate(BasePtrs bp, int bshift, const uint32_t* __restrict__ entries,
                                                        const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ goff,
                                                        uint32_t nb, uint32_t S_host, int tight, uint32_t T, uint32_t* __restrict__ slots,
                                                        uint32_t stride, uint32_t* __restrict__ fix_cnt /* [2] */, uint2* __restrict__ fix_short, uint2* __restrict__ fix_long,
                                                        uint64_t* __restrict__ dbg_times /* null, or [waves][4]: tools/acc_probe.py */, int prio_mode,
                                                        uint32_t* prio_board /* null, or one word per hardware wave slot */, uint32_t prio_tag) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t dbg_t0 = dbg_times ? wall_clock64() : 0;
  const uint32_t M = offsets[nb - 1] + counts[nb - 1];
  const uint32_t A = slice_weight(goff[nb], T, S_host, tight);
  if (t >= T) return;
  uint32_t b, b1_;
  const uint32_t pos0 = slice_pos(goff, offsets, counts, nb, M, (uint64_t)t * A, &b);
  if (pos0 >= M) return;
  const uint32_t pos1 = slice_pos(goff, offsets, counts, nb, M, (uint64_t)(t + 1) * A, &b1_);
  while (offsets[b] + counts[b] <= pos0) b++;          // (the slice may start right behind the last entry of the bucket the search found)
  {
    // A bucket cut by slice boundaries leaves an L piece (slice t0, where it starts) and F pieces in the slices t0+1 .. t1.  The
    // slice that holds the FIRST F piece of a bucket with several of them puts the bucket on a list: (first F slot, number of F
    // pieces) - a short list (2 .. 4 pieces: one lane or quad folds them) and a long one (a workgroup each): k_fixup_fold.  One atomic per
    // wave and list.
    uint32_t span = 0, tF0 = 0;
    if (offsets[b] < pos0) {
      const uint32_t t0 = slice_of(goff, offsets, b, offsets[b], A);
      tF0 = t0 + 1;
      span = slice_of(goff, offsets, b, offsets[b] + counts[b] - 1, A) - t0;       // number of F pieces
      if (t != tF0) span = 0;
    }
    const bool is_short = span >= 2 && span <= 4, is_long = span > 4;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long ms = __ballot(is_short), ml = __ballot(is_long);
    if (ms) {
      const int leader = __ffsll((long long)ms) - 1;
      uint32_t base = 0;
      if ((int)lane == leader) base = atomicAdd(&fix_cnt[0], (uint32_t)__popcll(ms));
      base = __shfl(base, leader);
      if (is_short) fix_short[base + (uint32_t)__popcll(ms & ((1ull << lane) - 1ull))] = make_uint2(tF0, span);
    }
    if (ml) {
      const int leader = __ffsll((long long)ml) - 1;
      uint32_t base = 0;
      if ((int)lane == leader) base = atomicAdd(&fix_cnt[1], (uint32_t)__popcll(ml));
      base = __shfl(base, leader);
      if (is_long) fix_long[base + (uint32_t)__popcll(ml & ((1ull << lane) - 1ull))] = make_uint2(tF0, tF0 + span - 1);
    }
  }
  uint32_t bend = pos0;            // forces the run set-up on the first iteration
  bool first = true;
  // 78 KiB per 256-lane block, two blocks per CU: ZZ, ZZZ as 27 limbs, X as 24 packed words
  __shared__ uint32_t lds_zz[27 * ZK_LDS_STRIDE], lds_zzz[27 * ZK_LDS_STRIDE], lds_x[24 * ZK_LDS_STRIDE];
  uint32_t* zz = lds_zz + threadIdx.x;
  uint32_t* zzz = lds_zzz + threadIdx.x;
  uint32_t* xs = lds_x + threadIdx.x;
  XyzzRef acc = make_slot_ref(slots, stride, 0);
  bool inf = true;
  Fq ty = fp_zero<FqParams>();     // Y of the running accumulator, carried in registers across the additions of a run
  const bool dense = entries == nullptr;
  uint32_t e_next = dense ? pos0 : entries[pos0];
  // The SIMD's arbiter serves its OLDEST wave first: of the two waves that share a SIMD one runs at full speed and the other on
  // what is left, so a launch of a single machine fill ends with a third of its time at one wave per SIMD (78 % of the
  // multiplier's rate; measured: waves 0 .. 1023 of 2,000 end at 1.6 ms, the others at 2.55 ms).  The wave that is behind asks
  // for the higher priority, by quarters of its slice: both advance together and end together.
  // ... by quarters of its slice when it cannot see its partner; when it can - both waves of a SIMD belong to this launch and post
  // their iteration count in prio_board, one word per hardware wave slot (XCC, SE, SH, CU, SIMD, wave from HW_ID / XCC_ID), tagged
  // with the launch - the wave that has done fewer iterations gets the SIMD: the two stay within one iteration of each other.
  const uint32_t len4 = (pos1 - pos0) >> 2, q1 = pos0 + len4, q2 = q1 + len4, q3 = q2 + len4;
  uint32_t *board_mine = nullptr, *board_other = nullptr;
  if (prio_mode && prio_board) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;
    const uint32_t simd = ((((xc * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u + ((hw >> 8) & 15u)) * 4u + ((hw >> 4) & 3u)) * 16u;
    board_mine = prio_board + simd + (hw & 15u);
    board_other = prio_board + simd + ((hw & 15u) ^ 1u);
  }
  uint32_t k = pos0, iter = 0;
  while (k < pos1) {
    if (prio_mode) {
      uint32_t other = 0;
      if (board_mine) {
        if ((threadIdx.x & 63u) == 0) __hip_atomic_store(board_mine, (prio_tag << 16) | (iter < 0xfffeu ? iter : 0xfffeu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        other = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(board_other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
      if (board_mine && (other >> 16) == prio_tag) {
        const uint32_t oi = other & 0xffffu;
        if (iter < oi) __builtin_amdgcn_s_setprio(3);
        else if (iter == oi) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      } else {
        const uint32_t kk = (uint32_t)__builtin_amdgcn_readfirstlane((int)k), a1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q1),
                       a2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q2), a3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q3);
        if (kk < a1) __builtin_amdgcn_s_setprio(3);
        else if (kk < a2) __builtin_amdgcn_s_setprio(2);
        else if (kk < a3) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
      iter++;
    }
    // ---- run boundaries and run openings: memory operations only.  The lanes that are in the middle of a run wait here for the few
    // that close one and open the next; whoever leaves this loop with k < pos1 has an addition to do.
    for (;;) {
      if (k == bend) {
        if (!first) {
          if (!inf) {   // close the finished run
            ZK_CLOSE_RUN(acc, xs, zz, zzz, ty);
          }
          b++;
          while (offsets[b] + counts[b] <= k) b++;     // next non-empty bucket
        }
        first = false;
        bend = offsets[b] + counts[b];
        const bool starts = (k == offsets[b]), ends = (bend <= pos1);
        const uint32_t slot = (starts && ends) ? b : (!starts ? nb + t : nb + T + t);
        acc.voff = slot * ZK_SLOT_PITCH;
        inf = true;
      }
      if (!inf) break;
      // open a run with entry k (also after a run cancelled to infinity in the middle of its bucket)
      const uint32_t e = e_next;
      if (k + 1 < pos1) e_next = dense ? k + 1 : entries[k + 1];
      const AffPacked* p = (dense ? bp.p[0] : base_of<NJ>(bp, b >> bshift)) + (e & 0x7fffffffu);
      k++;
      if (!(dense && p->x[23] == ZK_AFF_INF_WORD)) {               // (dense lists: a pair of the levels below may have cancelled)
#pragma unroll
        for (int i = 0; i < 24; i++) xs[i * ZK_LDS_STRIDE] = p->x[i];       // packed words straight into LDS
        ty = aff_ld_y(p, (e >> 31) != 0);
        lds_st(zz, fp_one<FqParams>());
        lds_st(zzz, fp_one<FqParams>());
        inf = false;
      }
      if (k >= pos1 || (k != bend && !inf)) break;
    }
    if (k >= pos1) break;
    // ---- one addition
    const uint32_t e = e_next;
    if (k + 1 < pos1) e_next = dense ? k + 1 : entries[k + 1];        // fetched a whole addition ahead of its use
    const AffPacked* p = (dense ? bp.p[0] : base_of<NJ>(bp, b >> bshift)) + (e & 0x7fffffffu);
    const bool neg = (e >> 31) != 0;        // (bases at infinity never reach the entry list: k_digit_pass drops them)
    k++;
    if (dense && p->x[23] == ZK_AFF_INF_WORD) continue;
    if (madd_lds_regy(acc, xs, zz, zzz, ty, p, neg)) inf = fp_is_zero_2p(lds_ld(zz));   // same-x path may have cancelled to infinity
  }
  if (board_mine && (threadIdx.x & 63u) == 0) __hip_atomic_store(board_mine, (prio_tag << 16) | 0xffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!inf) {
    ZK_CLOSE_RUN(acc, xs, zz, zzz, ty);
  }
  // a run that cancelled to infinity leaves ZZ = 0 in its slot: madd_same_x wrote the zeros, or the slot
  // was never written (zero-filled array)
  if (dbg_times && (threadIdx.x & 63u) == 0) {          // (wave-uniform: lane 0 of a wave that had work)
    const uint32_t wave = t >> 6;
    dbg_times[4 * wave] = dbg_t0; dbg_times[4 * wave + 1] = wall_clock64();
    dbg_times[4 * wave + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID: wave, SIMD, CU, SH, SE
    dbg_times[4 * wave + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
  }
}

// ---- batched-affine levels ------------------------------------------------------------------------------------------------
// The bucket-sorted entry list is summed PAIRWISE inside every bucket, level by level: a bucket of n points becomes ceil(n/2)
// points (floor(n/2) sums and, for odd n, its last point passed through), written densely in bucket order.  All additions of a
// level are independent, so a lane that produces m consecutive outputs shares one field inversion among them (ec_affine.cuh):
//   forward   d_j = x2 - x1 of pair j;  scratch[j] = d_0 ... d_(j-1);  acc = d_0 ... d_(m-1)
//   inv = 1 / acc                                       (fp_inv: division steps, all 64 lanes at once)
//   backward  1/d_j = inv * scratch[j];  inv *= d_j;  lambda, x3, y3;  store
// 5 M + 1 S per addition (+ ~30 M / m for the inversion) instead of the 8 M + 2 S of a mixed addition into an XYZZ accumulator.
// After a few levels (the buckets are down to a handful of points) k_accumulate sums what is left, in its dense mode.
__global__ void __launch_bounds__(256) k_half_counts(const uint32_t* __restrict__ in_cnt, uint32_t* __restrict__ out_cnt, size_t nb) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nb) out_cnt[i] = (in_cnt[i] + 1) >> 1;
}

struct LevelArgs {
  BasePtrs bp;                 // first level: the base tables, by job
  int bshift;
  const uint32_t* entries;     // first level: bucket-sorted (index | sign << 31)
  const AffPacked* src;        // later levels: the previous level's dense points
  const uint32_t* in_off;      // [nb] start of every bucket in the input level
  const uint32_t* in_cnt;      // [nb]
  const uint32_t* out_off;     // [nb] start of every bucket in the output level (counts: ceil(in_cnt / 2))
  uint32_t nb, m, lane0, lanes;  // outputs per lane; first lane of this launch; lanes of this launch (row length of `scratch`)
  AffPacked* dst;
  uint32_t* scratch;           // prefix products, limb-major: word (j * 27 + k) of lane t at scratch[(j * 27 + k) * lanes + t]
};

struct LevelWalk {             // the bucket that holds output j, and where its inputs start
  uint32_t b, ooff, oend, ioff, icnt;
};
__device__ __forceinline__ void walk_fwd(LevelWalk& w, uint32_t j, const uint32_t* __restrict__ in_off, const uint32_t* __restrict__ in_cnt) {
  while (j >= w.oend) {
    w.b++;
    const uint32_t c = in_cnt[w.b];
    if (!c) continue;
    w.ooff = w.oend; w.icnt = c; w.ioff = in_off[w.b]; w.oend = w.ooff + ((c + 1) >> 1);
  }
}
__device__ __forceinline__ void walk_bwd(LevelWalk& w, uint32_t j, const uint32_t* __restrict__ in_off, const uint32_t* __restrict__ in_cnt) {
  while (j < w.ooff) {
    w.b--;
    const uint32_t c = in_cnt[w.b];
    if (!c) continue;
    w.oend = w.ooff; w.icnt = c; w.ioff = in_off[w.b]; w.ooff = w.oend - ((c + 1) >> 1);
  }
}
template <int NJ, bool FIRST>
__device__ __forceinline__ PairRef level_pair(const LevelArgs& a, const LevelWalk& w, uint32_t j) {
  const uint32_t i = j - w.ooff, p0 = w.ioff + 2 * i;
  const bool has2 = 2 * i + 1 < w.icnt;
  PairRef pr;
  if constexpr (FIRST) {
    const AffPacked* base = base_of<NJ>(a.bp, w.b >> a.bshift);
    const uint32_t e0 = a.entries[p0], e1 = has2 ? a.entries[p0 + 1] : 0u;
    pr.p1 = base + (e0 & 0x7fffffffu); pr.neg1 = (e0 >> 31) != 0;
    pr.p2 = has2 ? base + (e1 & 0x7fffffffu) : nullptr; pr.neg2 = (e1 >> 31) != 0;
  } else {
    pr.p1 = a.src + p0; pr.neg1 = false;
    pr.p2 = has2 ? a.src + p0 + 1 : nullptr; pr.neg2 = false;
  }
  return pr;
}
__device__ __forceinline__ void pair_ld_x(const PairRef& pr, uint32_t* wx1, uint32_t* wx2) {
  aff_ld_words(pr.p1->x, wx1);
  if (pr.p2) aff_ld_words(pr.p2->x, wx2);
}

template <int NJ, bool FIRST>
__global__ void __launch_bounds__(256, 2) k_affine_level(LevelArgs a) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t M_out = a.out_off[a.nb - 1] + ((a.in_cnt[a.nb - 1] + 1) >> 1);
  const uint64_t j0_64 = (uint64_t)(a.lane0 + t) * a.m;
  if (t >= a.lanes || j0_64 >= M_out) return;
  const uint32_t j0 = (uint32_t)j0_64, cnt = min(a.m, M_out - j0);
  const zk_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.scratch, 0, (int)((size_t)a.m * 27u * a.lanes * 4u), 0x00020000);
  const uint32_t voff = t * 4u, row_b = a.lanes * 4u;
  LevelWalk w;
  w.b = bucket_of(a.out_off, a.nb, j0);
  w.icnt = a.in_cnt[w.b]; w.ioff = a.in_off[w.b]; w.ooff = a.out_off[w.b]; w.oend = w.ooff + ((w.icnt + 1) >> 1);
  // Memory: every operand is fetched ONE multiplication ahead of its use (a gather from a multi-gigabyte table costs about as
  // much as a sixth of a multiplication, and with two waves per SIMD a stalled wave is half the SIMD's throughput lost): the
  // words wait in 48 registers (`pw`) carried around the loops, and the pair of the next output is located while the current
  // one computes.
  // ---- forward: prefix products of the slope denominators
  Fq acc = fp_one<FqParams>();
  {
    uint32_t pw[48];
    walk_fwd(w, j0, a.in_off, a.in_cnt);
    PairRef pr = level_pair<NJ, FIRST>(a, w, j0);
    pair_ld_x(pr, pw, pw + 24);
#pragma unroll 1
    for (uint32_t jj = 0; jj < cnt; jj++) {
      const uint32_t kind = pair_kind(pr, pw, pw + 24);
      const Fq d = pair_denominator(pr, kind, pw, pw + 24);
      if (jj + 1 < cnt) {                                            // the next pair's x coordinates travel under this multiplication
        walk_fwd(w, j0 + jj + 1, a.in_off, a.in_cnt);
        pr = level_pair<NJ, FIRST>(a, w, j0 + jj + 1);
        pair_ld_x(pr, pw, pw + 24);
      }
#pragma unroll
      for (int k = 0; k < 27; k++) __builtin_amdgcn_raw_buffer_store_b32(acc.l[k], rs, voff, (jj * 27u + (uint32_t)k) * row_b, 0);
      acc = fp_mul(acc, d);
    }
  }
  Fq inv = fp_inv<FqParams>(acc);
  // ---- backward: one addition per output.  A rolled micro-program (one multiplier body and one squaring body in the loop, as
  // in ec_mem.cuh); three field elements are carried from step to step: inv, T0 (1/d, then x3), T1 (3 x1^2, then lambda).
  PairRef pr = level_pair<NJ, FIRST>(a, w, j0 + cnt - 1);           // (the forward pass left w on the last output)
#pragma unroll 1
  for (uint32_t jj = a.m; jj-- > 0;) {               // (a wave-uniform counter: it is the scalar row offset of the scratch loads)
    if (jj >= cnt) continue;
    const uint32_t j = j0 + jj;
    PairRef pr_next = pr;
    uint32_t kind = PK_FIRST;
    bool any_dbl = false;
    Fq T0 = fp_zero<FqParams>(), T1 = T0;
    uint32_t pw[48];
#pragma unroll
    for (int k = 0; k < 48; k++) pw[k] = 0;
#pragma unroll 1
    for (int step = 0; step < 6; step++) {
      if (step == 2 && !any_dbl) continue;
      Fq A, B;
      switch (step) {
        case 0:                                                     // 1/d = inv * (d_0 ... d_(j-1))
          A = inv;
#pragma unroll
          for (int k = 0; k < 27; k++) B.l[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (jj * 27u + (uint32_t)k) * row_b, 0);
          pair_ld_x(pr, pw, pw + 24);                               // for step 1
          break;
        case 1: {                                                   // inv <- inv * d
          kind = pair_kind(pr, pw, pw + 24);
          A = inv; B = pair_denominator(pr, kind, pw, pw + 24);
          aff_ld_words(pr.p1->y, pw);                               // for step 3
          if (pr.p2) aff_ld_words(pr.p2->y, pw + 24);
          break;
        }
        case 2: { uint32_t wx1[24]; aff_ld_words(pr.p1->x, wx1); A = fp_unpack32<FqParams>(wx1); B = A; break; }   // x1^2 (doublings)
        case 3: {                                                   // lambda = numerator / d
          if (kind == PK_ADD) A = fp_sub<FqParams, 2>(aff_y_eff(pw + 24, pr.neg2), aff_y_eff(pw, pr.neg1));      // [4]
          else A = T1;                                              // 3 x1^2 [6] (or unused)
          B = T0;
          pair_ld_x(pr, pw, pw + 24);                               // for the end of step 4
          break;
        }
        case 4: A = T1; B = T1; break;                              // lambda^2
        default: {                                                  // lambda (x1 - x3)
          A = T1; B = fp_sub<FqParams, 2>(fp_unpack32<FqParams>(pw), T0);                   // [3]
          if (jj > 0) {                                             // locate the pair of the next output under this multiplication
            walk_bwd(w, j - 1, a.in_off, a.in_cnt);
            pr_next = level_pair<NJ, FIRST>(a, w, j - 1);
          }
          break;
        }
      }
      const Fq r = (step == 2 || step == 4) ? fp_sqr(A) : fp_mul(A, B);
      switch (step) {
        case 0: T0 = r; break;
        case 1: inv = r; any_dbl = __any(kind == PK_DBL); break;
        case 2: T1 = fp_add(fp_dbl(r), r); break;                   // [6]
        case 3: T1 = r; break;
        case 4: {                                                   // x3 = lambda^2 - x1 - x2  (stored at once; kept in T0 for y3)
          Fq x1 = fp_unpack32<FqParams>(pw), x2 = x1;
          if (kind == PK_ADD) x2 = fp_unpack32<FqParams>(pw + 24);
          T0 = fp_canon_4p(fp_sub<FqParams, 2>(r, fp_add(x1, x2)));
          uint32_t ox[24];
          fp_pack32<FqParams>(T0, ox);
          aff_st_words(a.dst[j].x, ox);
          aff_ld_words(pr.p1->y, pw + 24);                          // y1 for the end of step 5 (x1 stays in pw[0 .. 24))
          break;
        }
        default: {                                                  // y3 = lambda (x1 - x3) - y1
          uint32_t oy[24];
          fp_pack32<FqParams>(fp_canon_4p(fp_sub<FqParams, 2>(r, aff_y_eff(pw + 24, pr.neg1))), oy);
          aff_st_words(a.dst[j].y, oy);
          break;
        }
      }
    }
    if (kind > PK_DBL) {                                            // no sum: overwrite what the program above stored for this output
      uint32_t ox[24], oy[24];
      if (kind == PK_INF) {
#pragma unroll
        for (int k = 0; k < 24; k++) { ox[k] = ZK_AFF_INF_WORD; oy[k] = 0; }
      } else {                                                      // the pair's finite point passes through (sign applied)
        const AffPacked* q = (kind == PK_FIRST) ? pr.p1 : pr.p2;
        const bool neg = (kind == PK_FIRST) ? pr.neg1 : pr.neg2;
        aff_ld_words(q->x, ox); aff_ld_words(q->y, oy);
        if (neg) fp_pack32<FqParams>(fp_canon_4p(aff_y_eff(oy, true)), oy);
      }
      aff_st_words(a.dst[j].x, ox); aff_st_words(a.dst[j].y, oy);
    }
    pr = pr_next;
  }
}

// Point operations on memory accumulators executed by one lane (QUAD = false: throughput-bound
// launches with at least as many additions as lanes) or by a DPP quad (QUAD = true: latency-bound
// launches; see ec_mem.cuh).  In quad mode all four lanes of a quad receive the same references.
// per-lane LDS scratch of the one-lane additions (add_mem_s): two 27-word limb-major images per 256-lane block
struct AddScratch { uint32_t* zz; uint32_t* zzz; };
#define ADD_SCRATCH_DECL(QUADFLAG)                                                                   \
  __shared__ uint32_t s_add_zz_[(QUADFLAG) ? 1 : 27 * ZK_LDS_STRIDE], s_add_zzz_[(QUADFLAG) ? 1 : 27 * ZK_LDS_STRIDE]; \
  const AddScratch sc{s_add_zz_ + ((QUADFLAG) ? 0 : threadIdx.x), s_add_zzz_ + ((QUADFLAG) ? 0 : threadIdx.x)}
template <bool QUAD> __device__ __forceinline__ void pt_add(const XyzzRef& a, const XyzzRef& b, uint32_t q, const AddScratch& sc) {
  if constexpr (QUAD) add_mem_quad(a, b, q); else add_mem_s(a, b, sc.zz, sc.zzz);
}
template <bool QUAD> __device__ __forceinline__ void pt_dbl(const XyzzRef& a, uint32_t q) {
  if constexpr (QUAD) dbl_mem_quad(a, q); else dbl_mem(a);
}
template <bool QUAD> __device__ __forceinline__ void pt_copy(const XyzzRef& dst, const XyzzRef& src, uint32_t q) {
  if constexpr (QUAD) mem_st_lane(dst, q, mem_ld_lane(src, q)); else mem_copy(dst, src);
}
template <bool QUAD> __device__ __forceinline__ void pt_set_inf(const XyzzRef& dst, uint32_t q) {
  if constexpr (QUAD) mem_st_lane(dst, q, fp_zero<FqParams>()); else mem_set_inf(dst);
}

// Stitch buckets that were cut by slice boundaries.  A bucket that starts in slice t0 and ends in slice t1 > t0 has pieces L[t0],
// F[t0+1], ..., F[t1].  First the F pieces are folded into F[t0+1] (k_fixup_fold, ONE launch, two kinds of workgroup):
//   * buckets of 2 .. 4 F pieces (the short list k_accumulate made): one lane - or one quad, in a latency-bound launch - folds them,
//     densely over the list.  With uniform scalars a few percent of the buckets have two F pieces, none more: one launch over
//     every slice, as rounds 1-2 had it, kept every wave busy for the sake of one lane in twelve.
//   * buckets of more pieces (the long list: "scalar == 1" in a boolean-heavy witness holds a third of all entries): a workgroup
//     each, its 64 quads fold the pieces pairwise (F[t] += F[t+d] for t - first divisible by 2d) in log2(pieces) rounds d = 1, 2,
//     4, ..., a barrier between rounds (the pieces live in global memory, coherent inside a CU).
// Both kinds touch different buckets, so they run side by side: a proof on its own waits for the longer of the two, not their sum.
template <bool QUAD>
__global__ void __launch_bounds__(256, 2) k_fixup_fold(const uint32_t* __restrict__ cnt /* [2]: short, long */, const uint2* __restrict__ list_short,
                                                        const uint2* __restrict__ list_long, uint32_t short_blocks, uint32_t nb,
                                                        uint32_t* __restrict__ slots, uint32_t stride) {
  ADD_SCRATCH_DECL(QUAD);
  const uint32_t q = threadIdx.x & 3u;
  if (blockIdx.x < short_blocks) {
    const uint32_t n = cnt[0], per = QUAD ? 64u : 256u;
    for (uint32_t i = blockIdx.x * per + (QUAD ? threadIdx.x >> 2 : threadIdx.x); i < n; i += short_blocks * per) {
      const uint2 w = list_short[i];                    // first F slot, number of F pieces
      const XyzzRef dst = make_slot_ref(slots, stride, nb + w.x);
      for (uint32_t j = 1; j < w.y; j++) pt_add<QUAD>(dst, make_slot_ref(slots, stride, nb + w.x + j), q, sc);
    }
    return;
  }
  const uint32_t n = cnt[1], quad = threadIdx.x >> 2;
  for (uint32_t i = blockIdx.x - short_blocks; i < n; i += gridDim.x - short_blocks) {
    const uint2 w = list_long[i];                       // first and last F piece of the bucket
#pragma unroll 1
    for (uint32_t d = 1; d <= w.y - w.x; d <<= 1) {
      for (uint64_t t = (uint64_t)w.x + (uint64_t)quad * 2 * d; t + d <= w.y; t += (uint64_t)64 * 2 * d)
        add_mem_quad(make_slot_ref(slots, stride, nb + (uint32_t)t), make_slot_ref(slots, stride, nb + (uint32_t)t + d), q);
      __threadfence_block();
      __syncthreads();
    }
  }
}

// final stitch: the slice in which a cut bucket STARTS owns it: bucket = L[t0] + F[t0+1] (folded).
__global__ void __launch_bounds__(256, 2) k_fixup(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ goff,
                                                   uint32_t nb, uint32_t S_host, int tight, uint32_t T, uint32_t* __restrict__ slots, uint32_t stride) {
  // the L piece goes to the CU (X, ZZ, ZZZ in LDS, Y in registers, as in k_accumulate / k_sum_lds), the F piece is added to it
  // there and the sum is stored once: ten coordinate loads and four stores per cut bucket (a copy into the bucket's slot followed by
  // an addition in memory - rounds 1-3 - moved fourteen and eight)
  __shared__ uint32_t lds_zz[27 * ZK_LDS_STRIDE], lds_zzz[27 * ZK_LDS_STRIDE], lds_x[24 * ZK_LDS_STRIDE];
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;   // throughput-bound (one addition per slice): one lane each
  const uint32_t M = offsets[nb - 1] + counts[nb - 1];
  const uint32_t A = slice_weight(goff[nb], T, S_host, tight);
  if (t >= T) return;
  uint32_t b0_, b1_;
  const uint32_t pos0 = slice_pos(goff, offsets, counts, nb, M, (uint64_t)t * A, &b0_);
  if (pos0 >= M) return;
  const uint32_t pos1 = slice_pos(goff, offsets, counts, nb, M, (uint64_t)(t + 1) * A, &b1_);
  uint32_t b = bucket_of(offsets, nb, pos1 - 1);
  uint32_t bend = offsets[b] + counts[b];
  if (bend <= pos1 || offsets[b] < pos0) return;      // not cut at this slice's end, or started earlier
  const XyzzRef dst = make_slot_ref(slots, stride, b), pl = make_slot_ref(slots, stride, nb + T + t), pf = make_slot_ref(slots, stride, nb + t + 1);
  const bool l_inf = mem_is_inf(pl), f_inf = mem_is_inf(pf);
  if (l_inf || f_inf) {                                 // (a piece whose points cancelled: rare)
    if (l_inf && f_inf) mem_set_inf(dst);
    else mem_copy(dst, l_inf ? pf : pl);
    return;
  }
  uint32_t* zz = lds_zz + threadIdx.x;
  uint32_t* zzz = lds_zzz + threadIdx.x;
  uint32_t* xs = lds_x + threadIdx.x;
  lds_st_packed(xs, mem_ld(pl, CX));                    // X of a stored point is an X3 [10]: below 2^768, packs into 24 words
  Fq ty = mem_ld(pl, CY);
  lds_st(zz, mem_ld(pl, CZZ));
  lds_st(zzz, mem_ld(pl, CZZZ));
  if (add_lds_regy(dst, xs, zz, zzz, ty, pf) && fp_is_zero_2p(lds_ld(zz))) { mem_set_inf(dst); return; }   // same-x path: L = -F
  mem_st(dst, CX, lds_ld_packed(xs)); mem_st(dst, CY, ty); mem_st(dst, CZZ, lds_ld(zz)); mem_st(dst, CZZZ, lds_ld(zzz));
}

// Segment pass of the bucket reduction.  in: n_in items (XYZZ limb-major, stride n_in), grouped in
// runs of L.  For segment t: S_t = sum_u item[tL+u],  R_t = sum_u (u + o) item[tL+u]   (o in {0,1}).
// The running sums live in the output arrays themselves (memory-resident accumulators).
template <bool QUAD>
__global__ void __launch_bounds__(256, 2) k_seg(uint32_t* __restrict__ in, size_t n_in, uint32_t in_stride, int L, int o,
                                                 uint32_t* __restrict__ outS, uint32_t* __restrict__ outR) {
  ADD_SCRATCH_DECL(QUAD);
  size_t n_out = n_in / L;
  size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x, t = QUAD ? gt >> 2 : gt;    // one lane / quad per segment
  const uint32_t q = (uint32_t)(gt & 3);
  if (t >= n_out) return;
  XyzzRef run = make_ref(outS, (uint32_t)n_out, (uint32_t)t), acc = make_ref(outR, (uint32_t)n_out, (uint32_t)t);
  pt_set_inf<QUAD>(run, q);
  pt_set_inf<QUAD>(acc, q);
  for (int u = L - 1; u >= 0; u--) {
    XyzzRef it = make_ref(in, in_stride, (uint32_t)(t * L + u));
    pt_add<QUAD>(run, it, q, sc);
    if (u + o > 0) pt_add<QUAD>(acc, run, q, sc);
  }
}

// out[t] = sum_u in[i(t) + u * row_len],  i(t) = (t / row_len) * (L * row_len) + (t % row_len).
// row_len = 1: sums of L consecutive items (row sums);  row_len = R: for items laid out [g][h][R] it sums L
// consecutive h for every (g, h / L, lo) (column sums).  in: n_in items with row stride in_stride words.
template <bool QUAD>
__global__ void __launch_bounds__(256, 2) k_sum(uint32_t* __restrict__ in, size_t n_in, uint32_t in_stride, int L, uint32_t row_len,
                                                 uint32_t* __restrict__ out, int in_slots /* `in` is the slot array (ec_mem.cuh) */) {
  ADD_SCRATCH_DECL(QUAD);
  size_t n_out = n_in / L;
  size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x, t = QUAD ? gt >> 2 : gt;    // one lane / quad per output
  const uint32_t q = (uint32_t)(gt & 3);
  if (t >= n_out) return;
  const size_t i0 = (t / row_len) * ((size_t)L * row_len) + (t % row_len);
  XyzzRef acc = make_ref(out, (uint32_t)n_out, (uint32_t)t);
  pt_copy<QUAD>(acc, make_in_ref(in, in_stride, (uint32_t)i0, in_slots), q);
  for (int u = 1; u < L; u++) pt_add<QUAD>(acc, make_in_ref(in, in_stride, (uint32_t)(i0 + (size_t)u * row_len), in_slots), q, sc);
}

// k_sum with one lane per output and the running sum held on the CU (add_lds_regy): the throughput-bound plain sums of the bucket
// reduction.  Same indexing as k_sum.
__global__ void __launch_bounds__(256, 2) k_sum_lds(uint32_t* __restrict__ in, size_t n_in, uint32_t in_stride, int L, uint32_t row_len,
                                                     uint32_t* __restrict__ out, int in_slots /* `in` is the slot array (ec_mem.cuh) */) {
  __shared__ uint32_t lds_zz[27 * ZK_LDS_STRIDE], lds_zzz[27 * ZK_LDS_STRIDE], lds_x[24 * ZK_LDS_STRIDE];
  const size_t n_out = n_in / L;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out) return;
  uint32_t* zz = lds_zz + threadIdx.x;
  uint32_t* zzz = lds_zzz + threadIdx.x;
  uint32_t* xs = lds_x + threadIdx.x;
  const size_t i0 = (t / row_len) * ((size_t)L * row_len) + (t % row_len);
  const XyzzRef dst = make_ref(out, (uint32_t)n_out, (uint32_t)t);
  bool inf = true;
  Fq ty = fp_zero<FqParams>();
  for (int u = 0; u < L; u++) {
    const XyzzRef B = make_in_ref(in, in_stride, (uint32_t)(i0 + (size_t)u * row_len), in_slots);
    if (mem_is_inf(B)) continue;
    if (inf) {
      lds_st_packed(xs, mem_ld(B, CX));             // X of a stored point is an X3 [10]: below 2^768, packs into 24 words
      ty = mem_ld(B, CY);
      lds_st(zz, mem_ld(B, CZZ));
      lds_st(zzz, mem_ld(B, CZZZ));
      inf = false;
      continue;
    }
    if (add_lds_regy(dst, xs, zz, zzz, ty, B)) inf = fp_is_zero_2p(lds_ld(zz));
  }
  if (inf) { mem_set_inf(dst); return; }
  mem_st(dst, CX, lds_ld_packed(xs)); mem_st(dst, CY, ty); mem_st(dst, CZZ, lds_ld(zz)); mem_st(dst, CZZZ, lds_ld(zzz));
}

// Two-level split of the bucket index j = hi * R + lo (R = 2^lo_bits, H = 2^hi_bits rows):
//   sum_j (j+1) B_j = R * sum_hi hi * Row[hi] + sum_lo (lo+1) * Col[lo].
// k_place_hilo lays both small weighted sums out as 2W groups of N = max(R, H) items with weights index+1:
//   group w      : item i = Row[w][i+1]  (i < H-1), infinity beyond      (weight i+1 = hi)
//   group W + w  : item i = Col[w][i]    (i < R),   infinity beyond      (weight i+1 = lo+1)
__global__ void __launch_bounds__(256, 2) k_place_hilo(uint32_t* __restrict__ rows /* W*H */, uint32_t* __restrict__ cols /* W*R */,
                                                        uint32_t W, uint32_t H, uint32_t R, uint32_t N, uint32_t* __restrict__ out /* 2W*N */) {
  uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x, t = gt >> 2, q = gt & 3u;
  if (t >= 2 * W * N) return;
  uint32_t g = t / N, i = t % N;
  XyzzRef dst = make_ref(out, 2 * W * N, t);
  if (g < W) {
    if (i + 1 < H) mem_st_lane(dst, q, mem_ld_lane(make_ref(rows, W * H, g * H + i + 1), q));
    else mem_st_lane(dst, q, fp_zero<FqParams>());
  } else {
    if (i < R) mem_st_lane(dst, q, mem_ld_lane(make_ref(cols, W * R, (g - W) * R + i), q));
    else mem_st_lane(dst, q, fp_zero<FqParams>());
  }
}

// per window w: out[w] = R[0][w] + L_0 (R[1][w] + L_1 (R[2][w] + ...)).
// R_all: `levels` arrays of W XYZZ points each, limb-major with stride W, consecutive (108*W words apart);
// `work`: scratch for W accumulators.
struct LevelShifts { uint8_t log_l[32]; };
__global__ void __launch_bounds__(64, 2) k_window_combine(uint32_t* __restrict__ R_all, int levels, int G, LevelShifts ls,
                                                           uint32_t* __restrict__ work /* G accumulators, result left here */) {
  int gt = blockIdx.x * blockDim.x + threadIdx.x, w = gt >> 2;   // one quad per group
  const uint32_t q = (uint32_t)(gt & 3);
  if (w >= G) return;
  XyzzRef acc = make_ref(work, (uint32_t)G, (uint32_t)w);
  mem_st_lane(acc, q, mem_ld_lane(make_ref(R_all + (size_t)(levels - 1) * 108 * G, (uint32_t)G, (uint32_t)w), q));
  for (int k = levels - 2; k >= 0; k--) {
    for (int d = 0; d < ls.log_l[k]; d++) dbl_mem_quad(acc, q);
    add_mem_quad(acc, make_ref(R_all + (size_t)k * 108 * G, (uint32_t)G, (uint32_t)w), q);
  }
}

// per window: R * (hi part) + (lo part), converted to ABI limbs
// total (optional): the plain sums of the 2W groups (the last S of the k_seg chain); the lo group's is sum_b S_b of the window and
// goes out behind the W weighted sums (the NAF finish needs it)
__global__ void __launch_bounds__(64, 2) k_hilo_combine(uint32_t* __restrict__ work /* 2W */, int W, int lo_bits,
                                                         uint64_t* __restrict__ out_abi /* W x 4 x 12 u64 (+ W more with total) */,
                                                         uint32_t* __restrict__ total /* 2W or null */) {
  int gt = blockIdx.x * blockDim.x + threadIdx.x, w = gt >> 2;
  const uint32_t q = (uint32_t)(gt & 3);
  if (w >= W) return;
  XyzzRef hi = make_ref(work, (uint32_t)(2 * W), (uint32_t)w), lo = make_ref(work, (uint32_t)(2 * W), (uint32_t)(W + w));
  for (int d = 0; d < lo_bits; d++) dbl_mem_quad(hi, q);
  add_mem_quad(hi, lo, q);
  uint64_t* o = out_abi + (size_t)w * 48;
  fp_to_abi<FqParams>(mem_ld_lane(hi, q), o + 12 * q);            // lane q converts coordinate q
  if (total) fp_to_abi<FqParams>(mem_ld_lane(make_ref(total, (uint32_t)(2 * W), (uint32_t)(W + w)), q), out_abi + (size_t)(W + w) * 48 + 12 * q);
}

// ---- batch fixed-base scalar multiplication: out[i] = k_i * G (the inner loop of Groth16 setup:
// reference libzecale/circuits/aggregator_circuit.tcc:100-109 -> wsnarkT::generate_setup) ----
// table[w*15 + (d-1)] = d * 2^(4w) * G, w < 95, d = 1..15 (packed device form).
__device__ const uint32_t FQ_P_LIMBS_DEV[27] = {
    FqParams::P[0], FqParams::P[1], FqParams::P[2], FqParams::P[3], FqParams::P[4], FqParams::P[5], FqParams::P[6],
    FqParams::P[7], FqParams::P[8], FqParams::P[9], FqParams::P[10], FqParams::P[11], FqParams::P[12], FqParams::P[13],
    FqParams::P[14], FqParams::P[15], FqParams::P[16], FqParams::P[17], FqParams::P[18], FqParams::P[19], FqParams::P[20],
    FqParams::P[21], FqParams::P[22], FqParams::P[23], FqParams::P[24], FqParams::P[25], FqParams::P[26]};

__device__ __forceinline__ Fq fq_inv_fermat(const Fq& a) {
  // a^(p-2); exponent bits taken from the modulus limbs
  Fq acc = fp_one<FqParams>();
  for (int i = FqParams::NBITS - 1; i >= 0; i--) {
    acc = fp_sqr(acc);
    // bit i of p - 2 (p odd, p0 = ...8b so p - 2 only changes limb 0: 0x8b - 2 = 0x89)
    uint32_t limb = FQ_P_LIMBS_DEV[i / 29];
    if (i / 29 == 0) limb -= 2;
    if ((limb >> (i % 29)) & 1) acc = fp_mul(acc, a);
  }
  return acc;
}

__global__ void __launch_bounds__(256, 2) k_fixed_base_mul(const AffPacked* __restrict__ table, const uint64_t* __restrict__ scalars,
                                                            size_t n, int montgomery, uint32_t* __restrict__ work /* 108 x n */,
                                                            uint64_t* __restrict__ out_abi /* n x 24 */) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t s[6];
#pragma unroll
  for (int k = 0; k < 6; k++) s[k] = scalars[i * 6 + k];
  uint32_t w32[12];
  if (montgomery) {
    fp_abi_to_canonical_words<FrParams>(s, w32);
  } else {
#pragma unroll
    for (int k = 0; k < 6; k++) { w32[2 * k] = (uint32_t)s[k]; w32[2 * k + 1] = (uint32_t)(s[k] >> 32); }
  }
  XyzzRef acc = make_ref(work, (uint32_t)n, (uint32_t)i);
  bool inf = true;
  for (int w = 0; w < 95; w++) {
    uint32_t d = (w32[w >> 3] >> ((w & 7) * 4)) & 15u;
    if (d == 0) continue;
    const AffPacked* p = &table[w * 15 + (d - 1)];
    if (aff_is_inf(p)) continue;
    if (inf) {
      mem_st(acc, CX, aff_ld_x(p)); mem_st(acc, CY, aff_ld_y(p, false));
      mem_st(acc, CZZ, fp_one<FqParams>()); mem_st(acc, CZZZ, fp_one<FqParams>());
      inf = false;
      continue;
    }
    if (madd_mem(acc, p, false)) inf = mem_is_inf(acc);
  }
  uint64_t* o = out_abi + i * 24;
  if (inf) {
#pragma unroll
    for (int k = 0; k < 24; k++) o[k] = 0;
    return;
  }
  // x = X / ZZ, y = Y / ZZZ with one inversion: zi = 1 / (ZZ ZZZ).  A rolled 6-step micro-program.
  Fq zi = fp_zero<FqParams>(), t = zi;
#pragma unroll 1
  for (int step = 0; step < 6; step++) {
    Fq a, b;
    switch (step) {
      case 0: a = mem_ld(acc, CZZ); b = mem_ld(acc, CZZZ); break;
      case 1: a = zi; b = mem_ld(acc, CZZZ); break;       // 1/ZZ
      case 2: a = t; b = mem_ld(acc, CX); break;          // x
      case 3: a = zi; b = mem_ld(acc, CZZ); break;        // 1/ZZZ
      case 4: a = t; b = mem_ld(acc, CY); break;          // y
      default: a = zi; b = zi; break;
    }
    Fq r = fp_mul(a, b);
    switch (step) {
      case 0: zi = fq_inv_fermat(r); break;
      case 1: t = r; break;
      case 2: mem_st(acc, CX, r); break;
      case 3: t = r; break;
      case 4: mem_st(acc, CY, r); break;
      default: break;
    }
  }
  fp_to_abi<FqParams>(mem_ld(acc, CX), o);
  fp_to_abi<FqParams>(mem_ld(acc, CY), o + 12);
}

// ---- window tables: table[w * n + i] = 2^(off_w) P_i, affine, for w = 1 .. levels-1 (level 0 is the base set itself; off_w is
// the first bit of window w in the balanced layout, c*w for most of them).
// With the table every digit position of a scalar addresses the same bucket window (the entry points at the
// pre-shifted base), so one MSM has 2^(c-1) buckets instead of W * 2^(c-1): the bucket reduction shrinks W-fold and c
// can grow until the accumulation (n * ceil(378/c) mixed additions) stops shrinking.  Built once per base set (a
// proving key lives in HBM for the life of the server; 288 GB hold a 2^22 key 19 times over).
// One lane per point: a chain of c doublings per level on a memory-resident XYZZ accumulator, every level kept;
// then ONE inversion per point normalises all levels (Montgomery's trick over the levels: prefix products in a fifth
// coordinate row of the work array).  Setup-time code: the multiplications outside dbl_mem are out-of-line calls.
__device__ __noinline__ Fq fq_mul_ool(const Fq& a, const Fq& b) { return fp_mul(a, b); }

// XyzzRef over 5 coordinate rows (X, Y, ZZ, ZZZ, scratch)
__device__ __forceinline__ XyzzRef make_ref5(uint32_t* base, uint32_t stride, uint32_t idx) {
  XyzzRef r;
  r.rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(135u * stride * 4u), 0x00020000);
  r.stride_b = stride * 4u;
  r.coord_b = 27u * stride * 4u;
  r.voff = idx * 4u;
  return r;
}

__global__ void __launch_bounds__(256, 2) k_table_build(AffPacked* __restrict__ table, uint8_t* __restrict__ tinf, size_t n, size_t i0,
                                                         uint32_t cn, WindowPlan plan, int levels, int every_bit, uint32_t* __restrict__ work, uint32_t stride) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cn) return;
  const size_t i = i0 + t;
  const AffPacked* p = &table[i];
  if (tinf[i]) {
    for (int w = 1; w < levels; w++) {
      AffPacked z;
#pragma unroll
      for (int k = 0; k < 24; k++) { z.x[k] = 0; z.y[k] = 0; }
      table[(size_t)w * n + i] = z;
      tinf[(size_t)w * n + i] = 1;
    }
    return;
  }
  const int CQ = 4;   // fifth row: prefix products
  // doubling chain
  for (int w = 1; w < levels; w++) {
    XyzzRef r = make_ref5(work, stride, (uint32_t)(w - 1) * cn + t);
    if (w == 1) {
      mem_st(r, CX, aff_ld_x(p)); mem_st(r, CY, aff_ld_y(p, false));
      mem_st(r, CZZ, fp_one<FqParams>()); mem_st(r, CZZZ, fp_one<FqParams>());
    } else {
      mem_copy(r, make_ref5(work, stride, (uint32_t)(w - 2) * cn + t));
    }
    const int nd = every_bit ? 1 : plan.bits[w - 1];       // level w = 2^(off_w) P, off_w - off_(w-1) = bits of window w-1 (every_bit: level w = 2^w P)
#pragma unroll 1
    for (int d = 0; d < nd; d++) dbl_mem(r);
  }
  // prefix products of ZZ * ZZZ over the finite levels
  Fq q = fp_one<FqParams>();
  for (int w = 1; w < levels; w++) {
    XyzzRef r = make_ref5(work, stride, (uint32_t)(w - 1) * cn + t);
    Fq zz = mem_ld(r, CZZ);
    if (!fp_is_zero_2p(zz)) q = fq_mul_ool(q, fq_mul_ool(zz, mem_ld(r, CZZZ)));
    mem_st(r, CQ, q);
  }
  Fq inv = fq_inv_fermat(q);      // q != 0: a product of non-zero field elements
  for (int w = levels - 1; w >= 1; w--) {
    XyzzRef r = make_ref5(work, stride, (uint32_t)(w - 1) * cn + t);
    Fq zz = mem_ld(r, CZZ);
    AffPacked o;
    uint8_t is_inf = 0;
    if (fp_is_zero_2p(zz)) {       // 2^(c w) P_i = O (a base outside the odd-order subgroup)
#pragma unroll
      for (int k = 0; k < 24; k++) { o.x[k] = 0; o.y[k] = 0; }
      is_inf = 1;
    } else {
      Fq zzz = mem_ld(r, CZZZ);
      Fq qprev = fp_one<FqParams>();
      if (w > 1) qprev = mem_ld(make_ref5(work, stride, (uint32_t)(w - 2) * cn + t), CQ);
      Fq tinv = fq_mul_ool(inv, qprev);                 // 1 / (ZZ ZZZ) of this level
      inv = fq_mul_ool(inv, fq_mul_ool(zz, zzz));       // inverse of the prefix below
      Fq x = fq_mul_ool(mem_ld(r, CX), fq_mul_ool(tinv, zzz));   // X / ZZ
      Fq y = fq_mul_ool(mem_ld(r, CY), fq_mul_ool(tinv, zz));    // Y / ZZZ
      fp_pack32<FqParams>(fp_cond_sub_p(x), o.x);
      fp_pack32<FqParams>(fp_cond_sub_p(y), o.y);
    }
    table[(size_t)w * n + i] = o;
    tinf[(size_t)w * n + i] = is_inf;
  }
}

// ---- the multiplier's own peak on THIS device, now (bench.py's fq_mul_frac divides by it) ----------------------------------------
// Two dependent chains of fp_mul per lane at the accumulation's occupancy (78 KiB of LDS per 256-lane block = two waves per SIMD):
// the kernel of tools/ubench/fqmul_occ_bench.hip inside the library.  The rate differs between boxes of the same model and between
// power states (420 .. 495 G wave-mads/s seen in round 4), so a fraction against a constant measured once means little.
__global__ void __launch_bounds__(256) k_fqmul_chain(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int iters) {
  extern __shared__ uint32_t lds_occ[];
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq x, y;
#pragma unroll
  for (int i = 0; i < 27; i++) { x.l[i] = (in[i] ^ (tid & 0xffu)) & M29; y.l[i] = (in[27 + i] ^ ((tid >> 8) & 0xffu)) & M29; }
  x.l[26] &= 0x3fu; y.l[26] &= 0x3fu;
  if (iters < 0) lds_occ[threadIdx.x] = x.l[0];       // (keeps the allocation)
#pragma unroll 1
  for (int it = 0; it < iters; it++) { x = fp_mul(x, y); y = fp_mul(y, x); }
  uint32_t s_ = 0;
#pragma unroll
  for (int i = 0; i < 27; i++) s_ ^= x.l[i] + y.l[i];
  out[tid] = s_;
}

int msm_measure_fqmul_rate(double* fq_mul_per_s, char* errbuf, size_t errlen) {
  const int blocks = 256 * 2 * 4, iters = 100;
  const size_t lds = 78 * 1024;
  uint32_t *in = nullptr, *out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  uint32_t h[54];
  for (int i = 0; i < 54; i++) h[i] = (0x9e3779b9u * (uint32_t)(i + 1)) & M29;
  hipError_t e = hipMalloc(&in, sizeof h);
  if (e == hipSuccess) e = hipMalloc(&out, (size_t)blocks * 256 * 4);
  if (e == hipSuccess) e = hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_fqmul_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  float ms = 0;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_fqmul_chain, dim3(blocks), dim3(256), lds, 0, in, out, 2);          // (code and clocks warm)
    hipLaunchKernelGGL(k_fqmul_chain, dim3(blocks), dim3(256), lds, 0, in, out, iters);
    for (int rep = 0; rep < 4 && e == hipSuccess; rep++) {                                   // a PEAK: the fastest of four launches
      float t = 0;
      e = hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_fqmul_chain, dim3(blocks), dim3(256), lds, 0, in, out, iters);
      if (e == hipSuccess) e = hipEventRecord(e1, 0);
      if (e == hipSuccess) e = hipEventSynchronize(e1);
      if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
      if (e == hipSuccess && (ms == 0 || t < ms)) ms = t;
    }
  }
  if (in) (void)hipFree(in);
  if (out) (void)hipFree(out);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (e != hipSuccess || ms <= 0) {
    if (errbuf) snprintf(errbuf, errlen, "msm_measure_fqmul_rate: %s", hipGetErrorString(e));
    return ZKHIP_ERR_HIP;
  }
  *fq_mul_per_s = (double)blocks * 256 * iters * 2 / (ms * 1e-3);
  return ZKHIP_OK;
}

// The three multiplier bodies of the DEVICE build on operands given limb by limb (tests/test_field_gpu.py: the worst cases of the
// column sums - every limb at 2^29 - 1, the top limb at the lazy bound - go through the same code the kernels run, and are checked
// against big integers on the host): out[i] = { a b / R, a^2 / R, (a b + c d) / R } for case i = (a, b, c, d), NL limbs each.
template <class PR>
__global__ void __launch_bounds__(64) k_field_selftest(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int N = PR::NL;
  Fp<PR> a, b, c, d;
  for (int k = 0; k < N; k++) { a.l[k] = in[(i * 4 + 0) * N + k]; b.l[k] = in[(i * 4 + 1) * N + k]; c.l[k] = in[(i * 4 + 2) * N + k]; d.l[k] = in[(i * 4 + 3) * N + k]; }
  const Fp<PR> m = fp_mul(a, b), q = fp_sqr(a), m2 = fp_mul2(a, b, c, d);
  for (int k = 0; k < N; k++) { out[(i * 3 + 0) * N + k] = m.l[k]; out[(i * 3 + 1) * N + k] = q.l[k]; out[(i * 3 + 2) * N + k] = m2.l[k]; }
}

int msm_field_selftest(int field, const uint32_t* in_host, size_t n, uint32_t* out_host, char* errbuf, size_t errlen) {
  const int N = field == 0 ? FqParams::NL : FrParams::NL;
  uint32_t *in = nullptr, *out = nullptr;
  hipError_t e = hipMalloc(&in, n * 4 * N * 4 + 4);
  if (e == hipSuccess) e = hipMalloc(&out, n * 3 * N * 4 + 4);
  if (e == hipSuccess) e = hipMemcpy(in, in_host, n * 4 * N * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess && n) {
    if (field == 0) hipLaunchKernelGGL(k_field_selftest<FqParams>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, in, out, n);
    else hipLaunchKernelGGL(k_field_selftest<FrParams>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, in, out, n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out_host, out, n * 3 * N * 4, hipMemcpyDeviceToHost);
  if (in) (void)hipFree(in);
  if (out) (void)hipFree(out);
  if (e != hipSuccess) {
    if (errbuf) snprintf(errbuf, errlen, "msm_field_selftest: %s", hipGetErrorString(e));
    return ZKHIP_ERR_HIP;
  }
  return ZKHIP_OK;
}

// ------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------
#define HIP_TRY(x)                                                                  \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      snprintf(ctx->errbuf, sizeof ctx->errbuf, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
      return ZKHIP_ERR_HIP;                                                         \
    }                                                                               \
  } while (0)

static inline unsigned nblk(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

// Window layout: W = ceil(378 / c) windows tile exactly 378 bits; the top W*c - 378 of them get c-1 bits.
static void window_layout(int c, uint16_t* off, uint8_t* bits) {
  const int W = (378 + c - 1) / c, n_small = W * c - 378;
  int bit = 0;
  for (int w = 0; w < W; w++) {
    int cw = (w >= W - n_small) ? c - 1 : c;
    off[w] = (uint16_t)bit; bits[w] = (uint8_t)cw;
    bit += cw;
  }
}

// target number of entries per lane and machine fill (tuning knob; ZKHIP_SLICE_TARGET overrides for experiments).  Round 4, with
// weighted slices and paced waves: 48 / 80 / 160 give 83.3-83.7 / 84.1 / 81.9-83.2 Mscalar/s in the 2^20 stream (slices of 38 / 76 /
// 152 entries: half the boundary pieces to stitch at 76; one fill per launch at 152 loses the overlap of fills), 20.5-20.6 / 20.8-20.9
// 2^20-proofs/s, the wrapping stream unchanged (one fill either way).
static size_t slice_target() {
  static size_t v = 0;
  if (!v) {
    const char* e = getenv("ZKHIP_SLICE_TARGET");
    v = e ? (size_t)atoi(e) : 80;
    if (v < 8 || v > 4096) v = 80;
  }
  return v;
}

static int env_int(const char* name, int dflt, int lo, int hi) {
  const char* e = getenv(name);
  if (!e) return dflt;
  int v = atoi(e);
  return (v < lo || v > hi) ? dflt : v;
}
// upper bound on the size of level l+1 given a bound on level l: ceil(n/2) summed over at most min(nb, m) non-empty buckets
static size_t level_bound(size_t m, size_t nb) { return (m + (m < nb ? m : nb)) / 2; }
static int g_aff_forced = -2;      // -2: not set (environment, then automatic); -1: automatic; >= 0: that many levels
void msm_force_aff_levels(int levels) { g_aff_forced = levels < -1 ? -1 : (levels > MSM_MAX_AFF_LEVELS ? MSM_MAX_AFF_LEVELS : levels); }
int msm_forced_aff_levels() { return g_aff_forced == -2 ? env_int("ZKHIP_AFF_LEVELS", -1, 0, MSM_MAX_AFF_LEVELS) : g_aff_forced; }
// How many batched-affine levels run before the XYZZ accumulation.  Automatic = NONE: measured on gfx950 (DESIGN.md section 5,
// profiles/r02_affine_*), one level over the 10.2 M pairs of a 2^20-term MSM takes 7.3 ms where the XYZZ kernel spends 6.1 ms on
// the same additions - 25 % fewer VALU instructions per addition, but 3.6 x the memory instructions (every operand is gathered
// in the forward AND the backward pass) at two waves per SIMD.  The levels stay available (zkhip_set_affine_levels,
// ZKHIP_AFF_LEVELS) and tested: they compute the same group element.
static int choose_aff_levels(size_t m_entries, size_t nb) {
  (void)m_entries; (void)nb;
  int forced = msm_forced_aff_levels();
  return forced >= 0 ? forced : 0;
}

// One event per device, recorded the first time a plan is made there - and again whenever the caller asks (msm_time_base_reset):
// the origin of the absolute launch times of k_accumulate.  hipEventElapsedTime returns FLOAT milliseconds: 1 us of resolution
// lasts ~8 s from the origin, 0.06 ms an hour - a caller that compares intervals re-bases at the start of its timed region.
static hipEvent_t g_time_base[64];
static bool g_time_base_made[64];
static uint32_t g_time_base_gen[64];        // bumped by every reset: an interval whose launch predates the current origin is dropped
static std::mutex g_time_base_mu;           // plans are made under per-device or per-prover locks: several threads may arrive here
static hipEvent_t time_base_locked(int dev, bool renew) {
  if (!g_time_base_made[dev] && hipEventCreate(&g_time_base[dev]) != hipSuccess) return nullptr;
  if (!g_time_base_made[dev] || renew) {
    if (hipEventRecord(g_time_base[dev], 0) != hipSuccess || hipEventSynchronize(g_time_base[dev]) != hipSuccess) {
      if (!g_time_base_made[dev]) (void)hipEventDestroy(g_time_base[dev]);
      return nullptr;
    }
    g_time_base_made[dev] = true;
  }
  return g_time_base[dev];
}
hipEvent_t msm_time_base() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> lk(g_time_base_mu);
  return time_base_locked(dev, false);
}
int msm_time_base_reset() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> lk(g_time_base_mu);
  g_time_base_gen[dev]++;
  return time_base_locked(dev, true) ? ZKHIP_OK : ZKHIP_ERR_HIP;
}
static uint32_t time_base_gen() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> lk(g_time_base_mu);
  return g_time_base_gen[dev];
}

// ZKHIP_DEBUG_DUMP=<dir>: the last launch's bucket populations and per-wave clocks of k_accumulate, for tools/acc_probe.py
// (synchronous copies: a measurement aid, never on in a timed run)
static void debug_dump(MsmCtx* ctx) {
  const char* dir = getenv("ZKHIP_DEBUG_DUMP");
  if (!dir || !ctx->dbg_times) return;
  const size_t nb = ctx->B * (size_t)ctx->W, nw = (size_t)ctx->T / 64 + 8;
  std::vector<uint32_t> cnt(nb);
  std::vector<uint64_t> tm(4 * nw);
  if (hipMemcpy(cnt.data(), ctx->counts, nb * 4, hipMemcpyDeviceToHost) != hipSuccess) return;
  if (hipMemcpy(tm.data(), ctx->dbg_times, nw * 32, hipMemcpyDeviceToHost) != hipSuccess) return;
  char path[512];
  snprintf(path, sizeof path, "%s/acc_K%d_c%d_m%d.bin", dir, ctx->K, ctx->c, ctx->merged);
  FILE* f = fopen(path, "wb");
  if (!f) return;
  const uint64_t hdr[8] = {nb, nw, ctx->last_S, ctx->last_T, (uint64_t)ctx->last_tight, (uint64_t)ctx->K, (uint64_t)ctx->c, (uint64_t)ctx->merged};
  fwrite(hdr, 8, 8, f);
  fwrite(cnt.data(), 4, nb, f);
  fwrite(tm.data(), 8, 4 * nw, f);
  fclose(f);
}

int msm_last_entries(MsmCtx* ctx, uint64_t* out) {
  *out = 0;
  if (!ctx->last_hist_m) return ZKHIP_OK;
  uint32_t v = 0;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(&v, ctx->hist + ctx->last_hist_m - 1, 4, hipMemcpyDeviceToHost));
  *out = v;
  return ZKHIP_OK;
}

static void read_accumulate_times(MsmCtx* ctx) {
  debug_dump(ctx);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev_acc0, ctx->ev_acc1);
  ctx->last_accumulate_ms = ms;
  hipEvent_t base = msm_time_base();
  float t0 = 0, t1 = 0;
  // (a launch recorded before the last msm_time_base_reset would be measured against an origin that lies AFTER it: no interval)
  if (ctx->acc_gen != time_base_gen()) { ctx->last_acc_begin_ms = ctx->last_acc_end_ms = -1.f; return; }
  if (base && hipEventElapsedTime(&t0, base, ctx->ev_acc0) == hipSuccess && hipEventElapsedTime(&t1, base, ctx->ev_acc1) == hipSuccess) {
    ctx->last_acc_begin_ms = t0; ctx->last_acc_end_ms = t1;
  }
}

int msm_plan_init(MsmCtx* ctx, size_t max_n, int c, int merged, int K, size_t total_terms, hipStream_t* adopt) {
  memset(ctx, 0, sizeof *ctx);
  // terms of ALL jobs of one launch sequence together: K * max_n unless the caller knows better (a proving key's five query vectors
  // differ in length and a third of the B query is the point at infinity: the wrapping key has 192,664 finite bases where
  // 5 * 65,535 = 327,675 would be planned - slices, boundary slots and the slot array's zero fill all scale with this bound)
  if (total_terms == 0 || total_terms > (size_t)K * max_n) total_terms = (size_t)K * max_n;
  if (total_terms < max_n && K == 1) total_terms = max_n;
  ctx->total_terms = total_terms;
  (void)msm_time_base();
  if (K < 1 || K > MSM_MAX_JOBS || (!merged && K != 1)) return ZKHIP_ERR_ARG;
  ctx->K = K;
  ctx->quad_below = (uint32_t)env_int("ZKHIP_QUAD_BELOW", 65536, 1, 1 << 30);
  ctx->one_stream = env_int("ZKHIP_MSM_ONE_STREAM", 0, 0, 1);
  // 108 * (W * 2^(c-1) + 2T) * 4 bytes must stay below 4 GiB (buffer descriptor); checked below
  if (c < 4 || c > (merged ? 22 : 18)) return ZKHIP_ERR_ARG;
  ctx->c = c;
  ctx->merged = merged;
  ctx->Wd = (378 + c - 1) / c;   // scalars < 2^377, +1 bit for the signed-digit carry
  if (merged == 2) ctx->Wd = 378 / (c + 1) + 2;      // width-(c+1) NAF: digits at least c+1 bits apart, +1 for the final carry
  ctx->W = merged ? K : ctx->Wd;
  // plain and merged plans share the balanced layout: a table's level w is 2^(off_w) P.  (A short top window would also
  // hurt a merged plan: its n digits of a few bits would all land in a handful of the shared buckets.)
  window_layout(c, ctx->win_off, ctx->win_bits);
  ctx->B = (size_t)1 << (c - 1);
  ctx->max_n = max_n;
  ctx->logL = env_int("ZKHIP_SUM_LOGL", 2, 2, 5); ctx->L = 1 << ctx->logL;      // fan-in of the reduction trees (tuning knob; the R arrays are sized for L >= 4)
  size_t nb = ctx->B * ctx->W;
  if (adopt && adopt[0] && adopt[1]) { ctx->stream = adopt[0]; ctx->stream2 = adopt[1]; adopt[0] = adopt[1] = nullptr; }
  else {
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));   // no implicit ordering against the null stream (the host application's, e.g. torch's)
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
  }
  HIP_TRY(hipEventCreateWithFlags(&ctx->ev, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&ctx->ev2, hipEventDisableTiming));
  HIP_TRY(hipEventCreate(&ctx->ev_acc0));
  HIP_TRY(hipEventCreate(&ctx->ev_acc1));
  // completion of a launch sequence: a BLOCKING event, so that the host thread that collects the result sleeps instead of
  // spinning on the stream (several prover instances wait side by side and share the cores with the witness generators)
  HIP_TRY(hipEventCreateWithFlags(&ctx->ev_done, hipEventBlockingSync | hipEventDisableTiming));
  if ((size_t)ctx->Wd * max_n >= ((size_t)1 << 31)) return ZKHIP_ERR_ARG;   // entry = 31-bit point index + sign
  if ((size_t)ctx->Wd * total_terms >= ((size_t)1 << 32)) return ZKHIP_ERR_ARG;  // positions in the entry list are 32-bit
  // geometry of the bucket sort: parts of 2^LB buckets (LB <= 10: k_bucket_sort keeps a part's counters in LDS); small bucket
  // windows get smaller parts so that k_bucket_sort still has about a thousand workgroups to spread over the chip
  {
    uint32_t LB = (uint32_t)(c - 1 < 10 ? c - 1 : 10);
    const size_t want_parts = (size_t)env_int("ZKHIP_SORT_PARTS", 1024, 64, 8192);      // (tuning knob)
    while (LB > 6 && (nb >> LB) < want_parts && (ctx->B >> (LB - 1)) * (merged ? 1 : (size_t)ctx->Wd) <= 4096) LB--;
    ctx->sort_LB = LB;
    ctx->sort_NP = (uint32_t)(ctx->B >> LB);
    ctx->sort_bins = merged ? ctx->sort_NP : ctx->sort_NP * (uint32_t)ctx->Wd;
    if ((size_t)ctx->sort_bins * 8 > 60 * 1024) return ZKHIP_ERR_ARG;             // LDS of k_digit_pass<1>: counters + bases
    uint32_t tile = (uint32_t)env_int("ZKHIP_SORT_TILE", 1024, 256, 16384) & ~255u;      // scalars per block of k_digit_pass (tuning knob)
    while ((max_n + tile - 1) / tile > 1024) tile *= 2;
    ctx->sort_tile = tile;
    const size_t nbx = (max_n + tile - 1) / tile;
    ctx->hist_len = (nb >> LB) * (nbx ? nbx : 1) + 1;
  }
  HIP_TRY(hipMalloc(&ctx->hist, ctx->hist_len * 4));
  HIP_TRY(hipMalloc(&ctx->pairs, ((size_t)ctx->Wd * total_terms + 1) * sizeof(uint2)));
  HIP_TRY(hipMalloc(&ctx->counts, nb * 4));
  HIP_TRY(hipMalloc(&ctx->offsets, nb * 4));
  HIP_TRY(hipMalloc(&ctx->goff, (nb + 1) * 4));
  if ((size_t)ctx->Wd * total_terms * ZK_W_NEXT >= ((size_t)1 << 32)) return ZKHIP_ERR_ARG;      // slice weights are 32-bit
  HIP_TRY(hipMalloc(&ctx->block_tot, ((nb > ctx->hist_len ? nb : ctx->hist_len) / 1024 + 2) * 4));
  HIP_TRY(hipMalloc(&ctx->entries, ((size_t)ctx->Wd * total_terms + 1) * 4));
  // batched-affine levels: bounds on the level sizes, the buffers of their outputs
  {
    size_t m = (size_t)ctx->Wd * total_terms;
    ctx->aff_levels = choose_aff_levels(m, nb);
    ctx->aff_forced = msm_forced_aff_levels();
    ctx->aff_m = (uint32_t)env_int("ZKHIP_AFF_M", 64, 4, 512);
    ctx->aff_lanes = 1u << 18;                                           // lanes per launch: 2^18 * aff_m * 108 B of scratch
    const size_t m_cap = (size_t)ctx->aff_m * 3 / 2;                      // a launch may use up to 1.5 aff_m outputs per lane (whole fills)
    while ((size_t)ctx->aff_lanes * m_cap * 108 >= ((size_t)1 << 32)) ctx->aff_lanes >>= 1;         // one buffer descriptor
    size_t bound[MSM_MAX_AFF_LEVELS + 1];
    bound[0] = m;
    for (int l = 0; l < ctx->aff_levels; l++) {
      bound[l + 1] = level_bound(bound[l], nb);
      HIP_TRY(hipMalloc(&ctx->lcnt[l], nb * 4));
      HIP_TRY(hipMalloc(&ctx->loff[l], nb * 4));
    }
    if (ctx->aff_levels > 0) {
      HIP_TRY(hipMalloc(&ctx->pbuf[0], (bound[1] + 1) * sizeof(AffPacked)));
      if (ctx->aff_levels > 1) HIP_TRY(hipMalloc(&ctx->pbuf[1], (bound[2] + 1) * sizeof(AffPacked)));
      size_t lanes = (bound[1] + ctx->aff_m - 1) / ctx->aff_m;
      if (lanes < ctx->aff_lanes) ctx->aff_lanes = (uint32_t)((lanes + 255) & ~(size_t)255);
      HIP_TRY(hipMalloc(&ctx->aff_scratch, (size_t)ctx->aff_lanes * m_cap * 108));
    }
    ctx->m_acc_max = bound[ctx->aff_levels];
  }
  // slice length: every lane gets the same number of point operations; aim at a whole number of
  // machine fills (256 CUs x 8 waves x 64 lanes at two waves per SIMD)
  {
    const size_t lanes = 131072, m_max = ctx->m_acc_max;
    size_t fills = (m_max + lanes * slice_target() - 1) / (lanes * slice_target());     // ~80 entries per lane and fill
    if (fills < 1) fills = 1;
    size_t S = (m_max + lanes * fills - 1) / (lanes * fills);
    if (S < 16) S = 16;
    // the slot array (nb buckets + two boundary slots per slice) is addressed through ONE buffer descriptor: 448 bytes (four
    // coordinates of 28 words; ec_mem.cuh) x slots < 4 GiB.  Large inputs (the five MSMs of a 2^22 key in one launch: 377 M entries) get longer slices instead of
    // more of them.
    const size_t max_slots = (((size_t)1 << 32) - 1) / (ZK_SLOT_WORDS * 4);
    if (nb + 64 >= max_slots) return ZKHIP_ERR_ARG;
    while (nb + 2 * ((m_max + S - 1) / S) >= max_slots) S += (S + 7) / 8;
    ctx->S = (uint32_t)S;
    ctx->T = (uint32_t)((m_max + S - 1) / S);
    ctx->slot_stride = (uint32_t)(nb + 2 * (size_t)ctx->T);
    if ((size_t)ctx->slot_stride * ZK_SLOT_WORDS * 4 >= ((size_t)1 << 32)) return ZKHIP_ERR_ARG;
  }
  HIP_TRY(hipMalloc(&ctx->buckets, (size_t)ctx->slot_stride * ZK_SLOT_WORDS * 4));
  if (getenv("ZKHIP_DEBUG_DUMP")) HIP_TRY(hipMalloc(&ctx->dbg_times, ((size_t)ctx->T / 64 + 8) * 32));
  HIP_TRY(hipMalloc(&ctx->prio_board, (size_t)ZK_PRIO_BOARD_WORDS * 4));          // 8 XCC x 8 SE x 2 SH x 16 CU x 4 SIMD x 16 wave slots
  HIP_TRY(hipMemset(ctx->prio_board, 0, (size_t)ZK_PRIO_BOARD_WORDS * 4));
  ctx->prio_seq = 0;
  HIP_TRY(hipMalloc(&ctx->fix_list, ((size_t)ctx->T / 5 + 2) * sizeof(uint2)));     // buckets of more than four F pieces: at most T / 5
  HIP_TRY(hipMalloc(&ctx->fix_short, ((size_t)ctx->T / 2 + 2) * sizeof(uint2)));    // buckets of two to four F pieces: at most T / 2
  // reduction scratch: S ping-pong (<= nb/L each) and R arrays (sum over levels <= nb/L * L/(L-1)), R sums
  HIP_TRY(hipMalloc(&ctx->segS[0], (nb / 2 + 1) * 108 * 4));
  HIP_TRY(hipMalloc(&ctx->segS[1], (nb / 2 + 1) * 108 * 4));
  HIP_TRY(hipMalloc(&ctx->segR, (nb / 2 + 64 * (size_t)ctx->W) * 108 * 4));   // one R array per level, back to back (sum < nb/3 at L = 4)
  HIP_TRY(hipMalloc(&ctx->sumR[0], (nb / ctx->L / ctx->L + 4 * ctx->W + 1) * 108 * 4));
  HIP_TRY(hipMalloc(&ctx->sumR[1], (nb / ctx->L / ctx->L + 4 * ctx->W + 1) * 108 * 4));
  HIP_TRY(hipMalloc(&ctx->Rlevels, (size_t)32 * 2 * ctx->W * 108 * 4));
  HIP_TRY(hipMalloc(&ctx->colS[0], (nb / 2 + 1) * 108 * 4));
  HIP_TRY(hipMalloc(&ctx->colS[1], (nb / 2 + 1) * 108 * 4));
  HIP_TRY(hipMalloc(&ctx->hilo, ((size_t)2 * ctx->W * ((size_t)1 << ((c - 1 + 1) / 2)) + 8) * 108 * 4));
  HIP_TRY(hipMalloc(&ctx->win_abi, (size_t)ctx->W * 48 * 8 * 2));       // (the second half: the plain totals of a NAF plan)
  HIP_TRY(hipHostMalloc(&ctx->win_host, (size_t)ctx->W * 48 * 8 * 2));
  return ZKHIP_OK;
}

void msm_plan_free(MsmCtx* ctx) {
  void* ptrs[] = {ctx->hist, ctx->pairs, ctx->counts, ctx->offsets, ctx->block_tot, ctx->entries, ctx->buckets,
                  ctx->segS[0], ctx->segS[1], ctx->segR, ctx->sumR[0], ctx->sumR[1], ctx->Rlevels, ctx->win_abi,
                  ctx->colS[0], ctx->colS[1], ctx->hilo, ctx->pbuf[0], ctx->pbuf[1], ctx->aff_scratch,
                  ctx->lcnt[0], ctx->lcnt[1], ctx->lcnt[2], ctx->lcnt[3], ctx->loff[0], ctx->loff[1], ctx->loff[2], ctx->loff[3],
                  ctx->fix_list, ctx->fix_short, ctx->dbg_times, ctx->goff, ctx->prio_board};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (ctx->win_host) (void)hipHostFree(ctx->win_host);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  if (ctx->ev) (void)hipEventDestroy(ctx->ev);
  if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
  if (ctx->ev_acc0) (void)hipEventDestroy(ctx->ev_acc0);
  if (ctx->ev_acc1) (void)hipEventDestroy(ctx->ev_acc1);
  if (ctx->ev_done) (void)hipEventDestroy(ctx->ev_done);
  memset(ctx, 0, sizeof *ctx);
}

int msm_bases_convert(const uint64_t* d_bases_abi, size_t n, AffPacked* d_out, uint8_t* d_inf_flags, char* errbuf, size_t errlen) {
  hipLaunchKernelGGL(k_bases_to_dev, dim3(nblk(n, 256)), dim3(256), 0, 0, d_bases_abi, d_out, d_inf_flags, n);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    if (errbuf) snprintf(errbuf, errlen, "msm_bases_convert: %s", hipGetErrorString(e));
    return ZKHIP_ERR_HIP;
  }
  return ZKHIP_OK;
}

// d_bases: packed device-form points; d_scalars: n x 6 u64 (device memory).  Result: Jacobian, ABI form (host).
int msm_run(MsmCtx* ctx, const AffPacked* d_bases, const uint8_t* d_inf_flags, const uint64_t* d_scalars, size_t n,
            int scalars_montgomery, size_t table_stride, uint64_t out_jac[36]) {
  int rc = msm_launch(ctx, d_bases, d_inf_flags, d_scalars, n, scalars_montgomery, table_stride);
  if (rc != ZKHIP_OK) return rc;
  return msm_finish(ctx, out_jac);
}

// Enqueue one MSM on the context's streams and return without waiting (the prover keeps two contexts
// in flight so that the latency-bound reduction of one MSM overlaps the accumulation of the next).
int msm_launch(MsmCtx* ctx, const AffPacked* d_bases, const uint8_t* d_inf_flags, const uint64_t* d_scalars, size_t n,
               int scalars_montgomery, size_t table_stride) {
  MsmJob job{d_bases, d_inf_flags, d_scalars, n, scalars_montgomery, table_stride, 0};
  return msm_launch_multi(ctx, 1, &job);
}

int msm_launch_multi(MsmCtx* ctx, int K, const MsmJob* jobs) {
  const int c = ctx->c, W = ctx->W, Wd = ctx->Wd, merged = ctx->merged;
  if (K < 1 || K > ctx->K) return ZKHIP_ERR_ARG;
  size_t n_tot = 0, n_eff = 0;
  for (int k = 0; k < K; k++) {
    if (jobs[k].n > ctx->max_n) return ZKHIP_ERR_ARG;
    n_eff += (jobs[k].n_finite && jobs[k].n_finite < jobs[k].n) ? jobs[k].n_finite : jobs[k].n;
    if (n_eff > ctx->total_terms) return ZKHIP_ERR_ARG;          // (the entry list and the slices are sized for total_terms)
    if (merged && jobs[k].n && (jobs[k].table_stride < jobs[k].n || (size_t)(merged == 2 ? 378 : Wd) * jobs[k].table_stride >= ((size_t)1 << 31))) return ZKHIP_ERR_ARG;
    n_tot += jobs[k].n;
  }
  const size_t B = ctx->B, nb = B * W;
  hipStream_t st = ctx->stream;
  ctx->pending_n = n_tot;
  if (n_tot == 0) { ctx->pending = true; return ZKHIP_OK; }
  WindowPlan plan;
  memset(&plan, 0, sizeof plan);
  for (int w = 0; w < Wd; w++) { plan.off[w] = ctx->win_off[w]; plan.bits[w] = ctx->win_bits[w]; }
  DigitJobs dj;
  memset(&dj, 0, sizeof dj);
  size_t n_max = 0;
  for (int k = 0; k < K; k++) {
    dj.scalars[k] = jobs[k].scalars; dj.inf_flags[k] = jobs[k].inf_flags; dj.n[k] = jobs[k].n; dj.tab_stride[k] = jobs[k].table_stride;
    dj.mode[k] = jobs[k].scalars_mode;
    if (jobs[k].n > n_max) n_max = jobs[k].n;
  }
  // bucket sort: count per (part, block) in LDS, scan, place by part, then order every part by its low bucket bits.  The bucket
  // windows of the jobs this call does not use (K < ctx->K) stay empty: their rows of hist are written as zeros by grid row k.
  SortGeom ge;
  ge.LB = ctx->sort_LB; ge.NP = ctx->sort_NP; ge.bins = ctx->sort_bins; ge.tile = ctx->sort_tile;
  ge.nbx = (uint32_t)((n_max + ge.tile - 1) / ge.tile);
  ge.nparts = (uint32_t)(nb >> ge.LB);
  const size_t hist_m = (size_t)ge.nparts * ge.nbx + 1;
  if (hist_m > ctx->hist_len) return ZKHIP_ERR_ARG;
  ctx->last_hist_m = hist_m;
  const int KK = merged ? ctx->K : 1;                            // every bucket window's rows of hist are (re)written
  for (int k = K; k < KK; k++) dj.n[k] = 0;
  const dim3 dgrid(ge.nbx, KK);
  const size_t lds0 = (size_t)ge.bins * 4, lds1 = (size_t)ge.bins * 8;
  if (merged == 2) hipLaunchKernelGGL((k_digit_pass<0, true>), dgrid, dim3(256), lds0, st, dj, c, Wd, plan, merged, ge, ctx->hist, ctx->pairs);
  else hipLaunchKernelGGL((k_digit_pass<0, false>), dgrid, dim3(256), lds0, st, dj, c, Wd, plan, merged, ge, ctx->hist, ctx->pairs);
  unsigned sb = nblk(hist_m, 1024);
  hipLaunchKernelGGL(k_scan_local, dim3(sb), dim3(256), 0, st, ctx->hist, ctx->hist, ctx->block_tot, hist_m);
  hipLaunchKernelGGL(k_scan_tot, dim3(1), dim3(1024), 0, st, ctx->block_tot, (size_t)sb);
  hipLaunchKernelGGL(k_scan_add, dim3(sb), dim3(256), 0, st, ctx->hist, ctx->block_tot, hist_m);
  if (merged == 2) hipLaunchKernelGGL((k_digit_pass<1, true>), dgrid, dim3(256), lds1, st, dj, c, Wd, plan, merged, ge, ctx->hist, ctx->pairs);
  else hipLaunchKernelGGL((k_digit_pass<1, false>), dgrid, dim3(256), lds1, st, dj, c, Wd, plan, merged, ge, ctx->hist, ctx->pairs);
  hipLaunchKernelGGL(k_bucket_sort, dim3(ge.nparts), dim3(512), 0, st, ctx->hist, ge, ctx->pairs, ctx->entries, ctx->offsets, ctx->counts);
  sb = nblk(nb, 1024);                                            // (the batched-affine levels below scan arrays of nb counters)
  BasePtrs bp;
  for (int k = 0; k < MSM_MAX_JOBS; k++) bp.p[k] = jobs[k < K ? k : 0].bases;
  const int bshift = merged ? c - 1 : 31;     // bucket -> job
  ctx->acc_gen = time_base_gen();
  if (ctx->aff_levels > 0) HIP_TRY(hipEventRecord(ctx->ev_acc0, st));     // the timed accumulation includes the affine levels
  // ---- batched-affine levels: the sorted list is summed pairwise inside every bucket, ctx->aff_levels times
  const uint32_t *cur_off = ctx->offsets, *cur_cnt = ctx->counts;
  size_t m_cur = (size_t)Wd * (n_eff ? n_eff : 1);       // entries that can occur: a base at infinity never produces one
  for (int l = 0; l < ctx->aff_levels; l++) {
    const size_t m_out = level_bound(m_cur, nb);
    hipLaunchKernelGGL(k_half_counts, dim3(nblk(nb, 256)), dim3(256), 0, st, cur_cnt, ctx->lcnt[l], nb);
    hipLaunchKernelGGL(k_scan_local, dim3(sb), dim3(256), 0, st, ctx->lcnt[l], ctx->loff[l], ctx->block_tot, nb);
    hipLaunchKernelGGL(k_scan_tot, dim3(1), dim3(1024), 0, st, ctx->block_tot, (size_t)sb);
    hipLaunchKernelGGL(k_scan_add, dim3(sb), dim3(256), 0, st, ctx->loff[l], ctx->block_tot, nb);
    LevelArgs la;
    la.bp = bp; la.bshift = bshift; la.entries = ctx->entries; la.src = l ? ctx->pbuf[(l - 1) & 1] : nullptr;
    la.in_off = cur_off; la.in_cnt = cur_cnt; la.out_off = ctx->loff[l];
    la.nb = (uint32_t)nb; la.m = ctx->aff_m; la.dst = ctx->pbuf[l & 1]; la.scratch = ctx->aff_scratch;
    // outputs per lane: as close to aff_m as a whole number of machine fills allows (131072 lanes are resident at two waves per
    // SIMD; a partial last fill runs at a fraction of the chip)
    {
      const size_t fill = 131072;
      size_t rounds = (m_out + fill * ctx->aff_m / 2) / (fill * ctx->aff_m);
      if (rounds >= 1) {
        size_t mm = (m_out + fill * rounds - 1) / (fill * rounds);
        la.m = (uint32_t)(mm > ctx->aff_m * 3 / 2 ? ctx->aff_m * 3 / 2 : mm);
      }
    }
    const size_t lanes_tot = (m_out + la.m - 1) / la.m;
    for (size_t lane0 = 0; lane0 < lanes_tot; lane0 += ctx->aff_lanes) {
      const size_t ln = lanes_tot - lane0 < ctx->aff_lanes ? lanes_tot - lane0 : ctx->aff_lanes;
      la.lane0 = (uint32_t)lane0; la.lanes = ctx->aff_lanes;       // row length of the scratch: fixed; lanes beyond ln do not exist
      const dim3 grid(nblk(ln, 256));
      if (l == 0) {
        if (ctx->K == 1) hipLaunchKernelGGL((k_affine_level<1, true>), grid, dim3(256), 0, st, la);
        else hipLaunchKernelGGL((k_affine_level<MSM_MAX_JOBS, true>), grid, dim3(256), 0, st, la);
      } else {
        hipLaunchKernelGGL((k_affine_level<1, false>), grid, dim3(256), 0, st, la);
      }
    }
    cur_off = ctx->loff[l]; cur_cnt = ctx->lcnt[l]; m_cur = m_out;
  }
  const bool dense = ctx->aff_levels > 0;
  if (dense) bp.p[0] = ctx->pbuf[(ctx->aff_levels - 1) & 1];
  // slice weights: goff = exclusive scan of (8 count - 7) over the non-empty buckets of the list k_accumulate sums (nb + 1 values)
  {
    const unsigned gb = nblk(nb + 1, 1024);
    hipLaunchKernelGGL(k_slice_weights, dim3(nblk(nb + 1, 256)), dim3(256), 0, st, cur_cnt, ctx->goff, nb);
    hipLaunchKernelGGL(k_scan_local, dim3(gb), dim3(256), 0, st, ctx->goff, ctx->goff, ctx->block_tot, nb + 1);
    hipLaunchKernelGGL(k_scan_tot, dim3(1), dim3(1024), 0, st, ctx->block_tot, (size_t)gb);
    hipLaunchKernelGGL(k_scan_add, dim3(gb), dim3(256), 0, st, ctx->goff, ctx->block_tot, nb + 1);
  }
  HIP_TRY(hipMemsetAsync(ctx->block_tot, 0, 8, st));   // block_tot[0], [1] are reused as the lengths of the two stitching lists (scans are done)
  // slice length for THIS n (the plan's slot array is sized for max_n)
  uint32_t S_run, T_run;
  {
    const size_t lanes = 131072, m = m_cur;
    size_t fills = (m + lanes * slice_target() - 1) / (lanes * slice_target());
    if (fills < 1) fills = 1;
    size_t S = (m + lanes * fills - 1) / (lanes * fills);
    if (S < 16) S = 16;
    // provers that share the chip (the streaming pipeline) take slices twice as long - half as many lanes a launch, half as many
    // buckets cut by a slice boundary to stitch afterwards; the other proofs in flight fill the chip.  Measured on the wrapping key,
    // 24 provers in flight, device side only: x1 349 proofs/s, x2 361, x3 358, x4 352 (tools/acc_probe.py --prove-stream).
    if (ctx->one_stream) S *= (size_t)env_int("ZKHIP_STREAM_SLICE_MULT", 2, 1, 16);
    // a bucket should not span more than ~3 slices (the stitching folds 2 .. 4 pieces by one lane; longer chains go through a
    // workgroup each, which is for the few heavy buckets of a witness, not for every bucket of a small MSM with a narrow window)
    {
      size_t n_jobs = 0;
      for (int k = 0; k < K; k++) n_jobs += jobs[k].n ? 1 : 0;
      const size_t live_buckets = merged ? B * (n_jobs ? n_jobs : 1) : nb, avg = m / (live_buckets ? live_buckets : 1);
      if (S < (avg + 1) / 2) S = (avg + 1) / 2;
    }
    while ((m + S - 1) / S > ctx->T) S++;
    S_run = (uint32_t)S; T_run = (uint32_t)((m + S - 1) / S);
  }
  const int tight = (ctx->one_stream || dense) ? 0 : env_int("ZKHIP_TIGHT_SLICES", 1, 0, 1);      // (a streaming prover shares the chip: see slice_len)
  // all-zero ZZ = infinity is what every reader of a slot tests first (mem_is_inf; X, Y, ZZZ of an infinite slot are copied along at
  // most, never used): only the 27 ZZ rows of the limb-major array need the zero fill - a quarter of the bytes
#if ZK_SLOTS_AOS
  {
    static const int clear_mode = env_int("ZKHIP_SLOTS_CLEAR", 0, 0, 1);      // 0: the ZZ words only (a kernel); 1: the whole array (one memset, four times the bytes, whole lines)
    if (clear_mode == 1) HIP_TRY(hipMemsetAsync(ctx->buckets, 0, (size_t)ctx->slot_stride * ZK_SLOT_WORDS * 4, st));
    else hipLaunchKernelGGL(k_slots_clear_zz, dim3(nblk((size_t)ctx->slot_stride * 27, 256)), dim3(256), 0, st, ctx->buckets, ctx->slot_stride);
  }
#else
  HIP_TRY(hipMemsetAsync(ctx->buckets + (size_t)CZZ * 27 * ctx->slot_stride, 0, (size_t)ctx->slot_stride * 27 * 4, st));
#endif
  if (ctx->dbg_times) HIP_TRY(hipMemsetAsync(ctx->dbg_times, 0, ((size_t)ctx->T / 64 + 8) * 32, st));
  if (ctx->acc_gate) HIP_TRY(hipStreamWaitEvent(st, ctx->acc_gate, 0));
  if (ctx->aff_levels == 0) HIP_TRY(hipEventRecord(ctx->ev_acc0, st));
  const uint32_t* acc_entries = dense ? nullptr : ctx->entries;
  static const int acc_prio = env_int("ZKHIP_ACC_PRIO", 1, 0, 2);      // 2: by quarters of the slice only (no board)
  const uint32_t prio_tag = (++ctx->prio_seq & 0x7fffu) + 1u;     // (see k_accumulate: the wave that is behind asks for priority; 0 = the arbiter's own order)
  if (ctx->K == 1 || dense)
    hipLaunchKernelGGL(k_accumulate<1>, dim3(nblk(T_run, 256)), dim3(256), 0, st, bp, bshift, acc_entries, cur_off, cur_cnt, ctx->goff,
                       (uint32_t)nb, S_run, tight, T_run, ctx->buckets, ctx->slot_stride, ctx->block_tot + 0, ctx->fix_short, ctx->fix_list, ctx->dbg_times, acc_prio, (acc_prio == 1 && !ctx->one_stream) ? ctx->prio_board : (uint32_t*)nullptr, prio_tag);
  else
    hipLaunchKernelGGL(k_accumulate<MSM_MAX_JOBS>, dim3(nblk(T_run, 256)), dim3(256), 0, st, bp, bshift, acc_entries, cur_off,
                       cur_cnt, ctx->goff, (uint32_t)nb, S_run, tight, T_run, ctx->buckets, ctx->slot_stride, ctx->block_tot + 0, ctx->fix_short, ctx->fix_list, ctx->dbg_times, acc_prio, (acc_prio == 1 && !ctx->one_stream) ? ctx->prio_board : (uint32_t*)nullptr, prio_tag);
  HIP_TRY(hipEventRecord(ctx->ev_acc1, st));
  ctx->last_S = S_run; ctx->last_T = T_run; ctx->last_tight = tight;
  // fold the F pieces of the buckets that have several (lists made by k_accumulate), then L + F for every cut bucket
  if (T_run > 2) {
    const bool fold_quads = T_run < (1u << 18) && ctx->quad_below > 1024;     // a small launch that has the chip to itself is latency-bound
    const uint32_t per = fold_quads ? 64u : 256u;
    uint32_t sblocks = nblk((size_t)T_run / 2 + 1, per);
    if (sblocks > 1024) sblocks = 1024;                                        // (the list is walked with a grid stride)
    // + 512 workgroups for the long list: a wrapping proof has ~290 buckets of 7-8 pieces (values that occur a couple of hundred
    // times in the assignment, in every window): with 128 workgroups they took three sweeps of three rounds each
    if (fold_quads) hipLaunchKernelGGL(k_fixup_fold<true>, dim3(sblocks + 512), dim3(256), 0, st, ctx->block_tot, ctx->fix_short, ctx->fix_list, sblocks,
                                       (uint32_t)nb, ctx->buckets, ctx->slot_stride);
    else hipLaunchKernelGGL(k_fixup_fold<false>, dim3(sblocks + 512), dim3(256), 0, st, ctx->block_tot, ctx->fix_short, ctx->fix_list, sblocks,
                            (uint32_t)nb, ctx->buckets, ctx->slot_stride);
  }
  hipLaunchKernelGGL(k_fixup, dim3(nblk(T_run, 256)), dim3(256), 0, st, cur_off, cur_cnt, ctx->goff, (uint32_t)nb, S_run, tight, T_run,
                     ctx->buckets, ctx->slot_stride);
  HIP_TRY(hipGetLastError());

  // ---- bucket reduction -------------------------------------------------------------------------------
  // (1) two-level split of the bucket index (row sums on stream 1, column sums on stream 2: plain trees),
  // (2) the two small weighted sums per window by recursive 4-ary running sums:
  //     F(items) = sum_t R_t + L * F0(S); the k_seg chain (stream 1) produces one R array per level, reducing
  //     each R array to one point per group (k_sum chain) is independent of the later levels: stream 2,
  // (3) per group Horner over the levels, then per window R * hi + lo.
  // fewer additions than this: the launch is latency-bound, a quad of lanes per addition finishes it sooner (at ~twice the lane-cycles:
  // a prover that shares the chip with others lowers the threshold, msm.h quad_below)
  const size_t QUAD_BELOW = ctx->quad_below;
  // the row and the column tree run side by side on two streams - unless this context shares the chip with others anyway (one_stream:
  // same throughput, no cross-stream events to wait on, ~0.8 host cores less for fourteen provers)
  hipStream_t st2 = ctx->one_stream ? ctx->stream : ctx->stream2;
  const int lo_bits = (c - 1 + 1) / 2, hi_bits = (c - 1) - lo_bits;
  const uint32_t Rr = 1u << lo_bits, Hh = 1u << hi_bits, Nn = Rr > Hh ? Rr : Hh;
  auto launch_sum = [&](hipStream_t s_, uint32_t* in, size_t n_in, uint32_t in_stride, int L, uint32_t row_len, uint32_t* out) {
    size_t n_out = n_in / L;
    const int in_slots = (in == ctx->buckets) ? 1 : 0;          // the first level reads the slot array (an array of structures: ec_mem.cuh)
    if (n_out >= QUAD_BELOW) hipLaunchKernelGGL(k_sum_lds, dim3(nblk(n_out, 256)), dim3(256), 0, s_, in, n_in, in_stride, L, row_len, out, in_slots);
    else hipLaunchKernelGGL(k_sum<true>, dim3(nblk(n_out * 4, 256)), dim3(256), 0, s_, in, n_in, in_stride, L, row_len, out, in_slots);
  };
  HIP_TRY(hipEventRecord(ctx->ev, st));
  HIP_TRY(hipStreamWaitEvent(st2, ctx->ev, 0));
  uint32_t *rows = nullptr, *cols = nullptr;
  {
    // row tree: [W][H][R] -> [W][H]  (sum over lo, contiguous)
    uint32_t* in = ctx->buckets; uint32_t in_stride = ctx->slot_stride; size_t n_in = nb; uint32_t left = Rr; int pp = 0;
    if (left == 1) rows = in;   // (c <= 2 never happens: c >= 4)
    // fan-in of a tree level: 4 while the level is throughput-bound (one lane per output, three additions each, fewer passes over
    // the data); 2 once it is latency-bound (a quad per output): ONE addition deep instead of three - the chain of dependent
    // additions of a row / column tree shrinks from 3 log4 to log2 of its length
    auto tree_fan_in = [&](size_t n_in_, uint32_t left_) {
      int L = ctx->L;
      if (n_in_ / (size_t)L < QUAD_BELOW) L = 2;
      while ((uint32_t)L > left_) L >>= 1;
      return L;
    };
    while (left > 1) {
      const int L = tree_fan_in(n_in, left);
      uint32_t* out = ctx->segS[pp];
      // halving-style grouping inside each row of `left` items (item lo' + u * left/L): lanes of a wave read adjacent
      // slots (summing L CONSECUTIVE items instead makes every lane stride L slots: a quarter of each sector used)
      launch_sum(st, in, n_in, in_stride, L, left / L, out);
      n_in /= L; left /= L; in = out; in_stride = (uint32_t)n_in; pp ^= 1; rows = out;
    }
    // column tree: [W][H][R] -> [W][R]  (sum over hi, stride R)
    in = ctx->buckets; in_stride = ctx->slot_stride; n_in = nb; left = Hh; pp = 0;
    if (left == 1) { cols = ctx->colS[0]; launch_sum(st2, in, n_in, in_stride, 1, Rr, cols); }
    while (left > 1) {
      const int L = tree_fan_in(n_in, left);
      uint32_t* out = ctx->colS[pp];
      launch_sum(st2, in, n_in, in_stride, L, Rr, out);
      n_in /= L; left /= L; in = out; in_stride = (uint32_t)n_in; pp ^= 1; cols = out;
    }
  }
  HIP_TRY(hipEventRecord(ctx->ev2, st2));
  HIP_TRY(hipStreamWaitEvent(st, ctx->ev2, 0));
  const int G = 2 * W;
  hipLaunchKernelGGL(k_place_hilo, dim3(nblk((size_t)G * Nn * 4, 256)), dim3(256), 0, st, rows, cols, (uint32_t)W, Hh, Rr, Nn, ctx->hilo);

  uint32_t* cur = ctx->hilo;
  size_t n_cur = (size_t)G * Nn;   // G groups of Nn items, weights index + 1
  int level = 0;
  LevelShifts ls;
  memset(&ls, 0, sizeof ls);
  size_t r_off = 0;                  // this level's R array starts here inside segR (in points)
  // (fan-in 2 in this recursion was measured and lost: 8 levels instead of 4 halve the chain of dependent point operations on paper,
  //  but every level brings its own k_seg launch, R-sum tree and cross-stream hand-over: one wrapping proof's MSM phase 5.7 -> 6.6 ms)
  while (n_cur > (size_t)G) {
    int L = ctx->L;
    while ((size_t)L > n_cur / G) L >>= 1;      // last level: fewer items per group than L
    int lg = 0; while ((1 << lg) < L) lg++;
    ls.log_l[level] = (uint8_t)lg;
    size_t n_out = n_cur / L;
    uint32_t* S = ctx->segS[level & 1];
    uint32_t* Rk = ctx->segR + r_off * 108;
    if (n_out >= QUAD_BELOW)
      hipLaunchKernelGGL(k_seg<false>, dim3(nblk(n_out, 256)), dim3(256), 0, st, cur, n_cur, (uint32_t)n_cur, L, level == 0 ? 1 : 0, S, Rk);
    else
      hipLaunchKernelGGL(k_seg<true>, dim3(nblk(n_out * 4, 256)), dim3(256), 0, st, cur, n_cur, (uint32_t)n_cur, L, level == 0 ? 1 : 0, S, Rk);
    HIP_TRY(hipEventRecord(ctx->ev, st));
    HIP_TRY(hipStreamWaitEvent(st2, ctx->ev, 0));
    // reduce R (n_out items, G groups) to G items: Rlevels[level]
    uint32_t* rc = Rk;
    size_t rn = n_out;
    int pp = 0;
    while (rn > (size_t)G) {
      int Ls = ctx->L;
      while ((size_t)Ls > rn / G) Ls >>= 1;
      size_t ro = rn / Ls;
      uint32_t* dst = (ro == (size_t)G) ? ctx->Rlevels + (size_t)level * 108 * G : ctx->sumR[pp];
      launch_sum(st2, rc, rn, (uint32_t)rn, Ls, (uint32_t)(rn / G / Ls), dst);      // same coalesced grouping inside each group
      rc = dst; rn = ro; pp ^= 1;
    }
    if (n_out == (size_t)G) {   // R already one per group
      HIP_TRY(hipMemcpyAsync(ctx->Rlevels + (size_t)level * 108 * G, Rk, (size_t)108 * G * 4, hipMemcpyDeviceToDevice, st2));
    }
    r_off += n_out;
    cur = S; n_cur = n_out; level++;
  }
  HIP_TRY(hipEventRecord(ctx->ev2, st2));
  HIP_TRY(hipStreamWaitEvent(st, ctx->ev2, 0));
  // the last S (one item per group) has weight 0 at its level (o = 0 for level >= 1) and is dropped.
  hipLaunchKernelGGL(k_window_combine, dim3(nblk((size_t)G * 4, 64)), dim3(64), 0, st, ctx->Rlevels, level, G, ls, ctx->sumR[0]);
  hipLaunchKernelGGL(k_hilo_combine, dim3(nblk((size_t)W * 4, 64)), dim3(64), 0, st, ctx->sumR[0], W, lo_bits, ctx->win_abi,
                     ctx->merged == 2 ? cur : (uint32_t*)nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(ctx->win_host, ctx->win_abi, (size_t)W * 48 * 8 * (ctx->merged == 2 ? 2 : 1), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipEventRecord(ctx->ev_done, st));
  ctx->pending = true;        // only a completely enqueued sequence is collectable; a failed launch leaves the context reusable
  return ZKHIP_OK;
}

// Wait for the MSM enqueued by msm_launch and finish it on the host.
static void xyzz_abi_to_jac(const uint64_t* p, host::HJac& q) {
  using namespace host;
  HFq X = HFq::from_limbs(p), Y = HFq::from_limbs(p + 12), ZZ = HFq::from_limbs(p + 24), ZZZ = HFq::from_limbs(p + 36);
  if (ZZ.is_zero()) { q = HJac::infinity(); return; }
  // XYZZ -> Jacobian with Z = ZZ*ZZZ:  X' = X ZZ ZZZ^2,  Y' = Y ZZ^3 ZZZ^2
  HFq z3s = ZZZ.sqr(), zz2 = ZZ.sqr();
  q.X = X * ZZ * z3s;
  q.Y = Y * zz2 * ZZ * z3s;
  q.Z = ZZ * ZZZ;
}

int msm_finish_multi(MsmCtx* ctx, int K, uint64_t* out_jac) {
  using namespace host;
  if (!ctx->pending || !ctx->merged || K < 1 || K > ctx->K) return ZKHIP_ERR_STATE;
  ctx->pending = false;
  if (ctx->pending_n) {
    HIP_TRY(zk_event_wait(ctx->ev_done));
    read_accumulate_times(ctx);
  }
  for (int k = 0; k < K; k++) {
    HJac q = HJac::infinity();
    if (ctx->pending_n) xyzz_abi_to_jac(ctx->win_host + (size_t)k * 48, q);    // one bucket window per job: nothing to combine
    if (ctx->pending_n && ctx->merged == 2) {                                  // odd digits: sum (2 b + 1) S_b = 2 F - sum S_b
      HJac tot;
      xyzz_abi_to_jac(ctx->win_host + (size_t)(ctx->K + k) * 48, tot);
      if (!tot.is_inf()) tot.Y = tot.Y.neg();
      q = q.dbl().add(tot);
    }
    q.X.to_limbs(out_jac + 36 * k); q.Y.to_limbs(out_jac + 36 * k + 12); q.Z.to_limbs(out_jac + 36 * k + 24);
  }
  return ZKHIP_OK;
}

int msm_finish(MsmCtx* ctx, uint64_t out_jac[36]) {
  using namespace host;
  if (!ctx->pending) return ZKHIP_ERR_STATE;
  if (ctx->merged) return msm_finish_multi(ctx, 1, out_jac);
  ctx->pending = false;
  const int W = ctx->W;
  if (ctx->pending_n == 0) {
    HJac inf = HJac::infinity();
    inf.X.to_limbs(out_jac); inf.Y.to_limbs(out_jac + 12); inf.Z.to_limbs(out_jac + 24);
    return ZKHIP_OK;
  }
  HIP_TRY(zk_event_wait(ctx->ev_done));
  read_accumulate_times(ctx);

  // host: sum_w 2^(c w) W_w  (Horner from the top window)
  HJac acc = HJac::infinity();
  for (int w = W - 1; w >= 0; w--) {
    for (int d = 0; d < ctx->win_bits[w]; d++) acc = acc.dbl();
    HJac q;
    xyzz_abi_to_jac(ctx->win_host + (size_t)w * 48, q);
    acc = acc.add(q);
  }
  acc.X.to_limbs(out_jac); acc.Y.to_limbs(out_jac + 12); acc.Z.to_limbs(out_jac + 24);
  return ZKHIP_OK;
}

int msm_table_build(AffPacked* d_table, uint8_t* d_tinf, size_t n, int c, int naf, char* errbuf, size_t errlen) {
  const int levels = msm_table_levels(c, naf);
  if (n == 0 || levels < 2) return ZKHIP_OK;
  WindowPlan plan;
  memset(&plan, 0, sizeof plan);
  window_layout(c, plan.off, plan.bits);
  // points per launch: (levels - 1) * chunk slots of 135 words, at most 4M slots (2.2 GB of work space)
  size_t chunk = ((size_t)1 << 22) / (size_t)(levels - 1);
  chunk &= ~(size_t)255;
  if (chunk > n) chunk = n;
  const uint32_t stride = (uint32_t)(chunk * (size_t)(levels - 1));
  uint32_t* work = nullptr;
  hipError_t e = hipMalloc(&work, (size_t)stride * 135 * 4);
  for (size_t i0 = 0; e == hipSuccess && i0 < n; i0 += chunk) {
    uint32_t cn = (uint32_t)((n - i0 < chunk) ? n - i0 : chunk);
    hipLaunchKernelGGL(k_table_build, dim3(nblk(cn, 256)), dim3(256), 0, 0, d_table, d_tinf, n, i0, cn, plan, levels, naf ? 1 : 0, work, stride);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  if (work) (void)hipFree(work);
  if (e != hipSuccess) {
    if (errbuf) snprintf(errbuf, errlen, "msm_table_build: %s", hipGetErrorString(e));
    return ZKHIP_ERR_HIP;
  }
  return ZKHIP_OK;
}

// out[i] = k_i * base for i < n.  base: ABI affine (host); d_scalars, d_out: device memory.
int fixed_base_mul(const uint64_t base_aff[24], const uint64_t* d_scalars, size_t n, int montgomery, uint64_t* d_out,
                   char* errbuf, size_t errlen) {
  using namespace host;
  struct { char errbuf[256]; } c_, *ctx = &c_;
  ctx->errbuf[0] = 0;
  // table on the host: 95 windows x 15 multiples, affine, ABI form -> device packed form
  std::vector<uint64_t> tab((size_t)95 * 15 * 24);
  HJac wbase = HJac::from_affine(HFq::from_limbs(base_aff), HFq::from_limbs(base_aff + 12));
  for (int w = 0; w < 95; w++) {
    HJac m = wbase;
    for (int d = 1; d <= 15; d++) {
      HFq x, y;
      m.to_affine(x, y);
      x.to_limbs(&tab[((size_t)w * 15 + d - 1) * 24]);
      y.to_limbs(&tab[((size_t)w * 15 + d - 1) * 24 + 12]);
      m = m.add(wbase);
    }
    for (int k = 0; k < 4; k++) wbase = wbase.dbl();
  }
  uint64_t* d_tab_abi = nullptr;
  AffPacked* d_tab = nullptr;
  uint32_t* d_work = nullptr;
  int rc = ZKHIP_OK;
  do {
    hipError_t e;
    if ((e = hipMalloc(&d_tab_abi, tab.size() * 8)) != hipSuccess) { rc = ZKHIP_ERR_HIP; break; }
    if ((e = hipMalloc(&d_tab, (size_t)95 * 15 * sizeof(AffPacked))) != hipSuccess) { rc = ZKHIP_ERR_HIP; break; }
    if ((e = hipMalloc(&d_work, n * 108 * 4)) != hipSuccess) { rc = ZKHIP_ERR_HIP; break; }
    if ((e = hipMemcpy(d_tab_abi, tab.data(), tab.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) { rc = ZKHIP_ERR_HIP; break; }
    hipLaunchKernelGGL(k_bases_to_dev, dim3(nblk(95 * 15, 256)), dim3(256), 0, 0, d_tab_abi, d_tab, (uint8_t*)nullptr, (size_t)95 * 15);
    hipLaunchKernelGGL(k_fixed_base_mul, dim3(nblk(n, 256)), dim3(256), 0, 0, d_tab, d_scalars, n, montgomery, d_work, d_out);
    if ((e = hipGetLastError()) != hipSuccess) { rc = ZKHIP_ERR_HIP; break; }
    if ((e = hipDeviceSynchronize()) != hipSuccess) { rc = ZKHIP_ERR_HIP; break; }
  } while (0);
  if (rc != ZKHIP_OK && errbuf) snprintf(errbuf, errlen, "fixed_base_mul: HIP failure (%s)", hipGetErrorString(hipGetLastError()));
  if (d_tab_abi) (void)hipFree(d_tab_abi);
  if (d_tab) (void)hipFree(d_tab);
  if (d_work) (void)hipFree(d_work);
  return rc;
}

}  // namespace zkhip
