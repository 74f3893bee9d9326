// Internal interface of the MSM engine (msm.hip).  The public C ABI is include/zkhip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/zkhip.h"

namespace zkhip {

struct AffPacked;

#define MSM_MAX_AFF_LEVELS 4
#define MSM_MAX_JOBS 5   // MSMs sharing one launch sequence (the five query vectors of a proof)
struct MsmJob {
  const AffPacked* bases;        // table-backed base set (level 0 at bases[0 .. n))
  const uint8_t* inf_flags;
  const uint64_t* scalars;       // device memory, n x 6 u64
  size_t n;
  int scalars_mode;              // 0 canonical, 1 Montgomery (ABI), 2 packed device form
  size_t table_stride;           // distance between table levels, in points
  size_t n_finite;               // upper bound on the bases of this job that are not the point at infinity (0: unknown, use n):
                                 // sizes the slices of the accumulation to the entries that can actually occur
};

#define ZK_PRIO_BOARD_WORDS (8 * 8 * 2 * 16 * 4 * 16)
struct MsmCtx {
  int c, W, L, logL;   // W: bucket windows (each owns 2^(c-1) buckets)
  int Wd;              // digits per scalar: W without a table; with a precomputed table all Wd digit positions share ONE bucket window
  int merged;          // 1: bases are a table  table[w * stride + i] = 2^(off_w) P_i  (msm_table_build), one digit per window;
                       // 2: a table with EVERY bit position, table[j * stride + i] = 2^j P_i: scalars are recoded in width-(c+1)
                       //    non-adjacent form (odd signed digits at arbitrary positions, 378 / (c + 2) of them on average)
  int K;               // merged plans: MSMs per launch sequence, one bucket window each (W == K); plain plans: 1
  uint16_t win_off[96];
  uint8_t win_bits[96];
  size_t B, max_n;
  size_t total_terms;   // bound on the terms of one launch sequence (all jobs): sizes the entry list, the slices and the slot array
  int one_stream;       // 1: the whole launch sequence on `stream` (a prover that shares the chip); 0: row / column trees side by side
  uint32_t quad_below;  // reduction launches with fewer outputs than this spread an addition over four lanes (latency) instead of one (total work)
  uint32_t S, T, slot_stride;   // slice length, slice count, words per row of the slot array
  hipStream_t stream, stream2;
  hipEvent_t ev, ev2, ev_acc0, ev_acc1, ev_done;
  hipEvent_t acc_gate;  // optional (not owned): the accumulation of this launch sequence starts after this event - the end of the
                        // accumulation of the MSM submitted before it on another context (zkhip_msm_submit): the sort of MSM i+1 and
                        // the reduction of MSM i-1 run under the accumulation of MSM i, two accumulations never share the chip
  // bucket sort (k_digit_pass / k_bucket_sort): hist[part][block] (+ 1: the total), (entry, low bucket bits) pairs grouped by part
  uint32_t sort_LB, sort_NP, sort_bins, sort_tile;
  uint32_t* hist;
  uint2* pairs;
  size_t hist_len;
  uint32_t *counts, *offsets, *block_tot, *entries;
  uint32_t* goff;       // slice weights in front of every bucket (nb + 1 values): see k_accumulate
  uint2* fix_list;      // (first, last) F slot of the buckets cut into more than four F pieces (k_accumulate -> k_fixup_fold)
  uint2* fix_short;     // (first F slot, number of F pieces) of the buckets with two to four (k_accumulate -> k_fixup_fold)
  uint32_t *buckets, *segS[2], *segR, *sumR[2], *Rlevels, *colS[2], *hilo;
  uint64_t *win_abi, *win_host;
  // batched-affine levels in front of the XYZZ accumulation (k_affine_level): per level the bucket counts / offsets of its output,
  // two point buffers used alternately, the prefix-product scratch of one launch
  int aff_levels;                    // 0: every entry goes straight to k_accumulate
  int aff_forced;                    // the msm_forced_aff_levels() this plan was made under
  uint32_t aff_m, aff_lanes;         // outputs per lane; lanes per launch (the scratch holds aff_m * 27 * aff_lanes words)
  uint32_t *lcnt[MSM_MAX_AFF_LEVELS], *loff[MSM_MAX_AFF_LEVELS];
  AffPacked* pbuf[2];
  uint32_t* aff_scratch;
  size_t m_acc_max;                  // upper bound on the entries that reach k_accumulate (sizes S, T and the slot array)
  float last_accumulate_ms;
  size_t last_hist_m;                // words of hist the last launch scanned: the last one holds the number of entries it sorted (msm_last_entries)
  uint32_t acc_gen;                           // generation of the device's time base when this launch's ev_acc0 was recorded (msm_time_base_reset bumps it)
  float last_acc_begin_ms, last_acc_end_ms;   // the same launch on the device's time base (msm_time_base): lets a caller that keeps
                                              // several MSMs in flight see how their accumulations overlap
  // measurement aid (ZKHIP_DEBUG_DUMP=<dir>, tools/acc_probe.py): per-wave begin / end clocks of k_accumulate and, at collection, a dump
  // of the bucket populations; null / unused otherwise
  uint64_t* dbg_times;
  uint32_t* prio_board;      // k_accumulate: iteration counts of the waves of a launch, one word per hardware wave slot (zeroed once)
  uint32_t prio_seq;
  uint32_t last_S, last_T;           // slice length / slices of the last launch (before the kernel's own shortening, see slice_len)
  int last_tight;
  bool pending;       // an MSM has been enqueued by msm_launch and not yet collected by msm_finish
  size_t pending_n;
  char errbuf[256];
};

// tuning / test knob: number of batched-affine levels of the plans made from now on (-1: automatic)
void msm_force_aff_levels(int levels);
int msm_forced_aff_levels();
// total_terms: upper bound on the terms (finite bases) of all K jobs of one launch together; 0 = K * max_n
// adopt: two streams created by the caller beforehand (or null): the plan's main and side stream - the plan owns them from here on
int msm_plan_init(MsmCtx* ctx, size_t max_n, int c, int merged, int K, size_t total_terms = 0, hipStream_t* adopt = nullptr);
// a per-device event recorded once: the origin of last_acc_begin_ms / last_acc_end_ms
hipEvent_t msm_time_base();
// record the origin again (now): intervals read afterwards are relative to this moment
int msm_time_base_reset();
void msm_plan_free(MsmCtx* ctx);
int msm_bases_convert(const uint64_t* d_bases_abi, size_t n, AffPacked* d_out, uint8_t* d_inf_flags, char* errbuf, size_t errlen);
// table_stride: distance (in points) between the levels of a precomputed table (merged plans only)
int msm_launch(MsmCtx* ctx, const AffPacked* d_bases, const uint8_t* d_inf_flags, const uint64_t* d_scalars, size_t n,
               int scalars_montgomery, size_t table_stride);
int msm_finish(MsmCtx* ctx, uint64_t out_jac[36]);
// The entries (non-zero digits = mixed additions of k_accumulate, no affine levels in front) of the last launch of this plan, read
// back from the sort's histogram: a measurement aid, synchronises the device.  0: nothing launched yet.
int msm_last_entries(MsmCtx* ctx, uint64_t* out);
// merged plans: up to ctx->K independent MSMs (same window, each at most max_n terms) through ONE launch sequence:
// one sort, one accumulation launch, one reduction chain with a bucket window per job.  out_jac: K x 36.
int msm_launch_multi(MsmCtx* ctx, int K, const MsmJob* jobs);
int msm_finish_multi(MsmCtx* ctx, int K, uint64_t* out_jac);
int msm_run(MsmCtx* ctx, const AffPacked* d_bases, const uint8_t* d_inf_flags, const uint64_t* d_scalars, size_t n,
            int scalars_montgomery, size_t table_stride, uint64_t out_jac[36]);
// levels of a window table for window size c
static inline int msm_table_levels(int c, int naf = 0) { return naf ? 378 : (378 + c - 1) / c; }
// d_table: levels x n points, level 0 (= the n base points) already in place; d_tinf: levels x n flags, level 0 in place.
// Fills levels 1 .. levels-1:  table[w * n + i] = 2^(c w) P_i  in affine packed form.
int msm_table_build(AffPacked* d_table, uint8_t* d_tinf, size_t n, int c, int naf, char* errbuf, size_t errlen);

// Fq multiplications per second of the whole device, measured now: dependent fp_mul chains at two waves per SIMD (~25 ms)
int msm_field_selftest(int field, const uint32_t* in_host, size_t n, uint32_t* out_host, char* errbuf, size_t errlen);
int msm_measure_fqmul_rate(double* fq_mul_per_s, char* errbuf, size_t errlen);

int fixed_base_mul(const uint64_t base_aff[24], const uint64_t* d_scalars, size_t n, int montgomery, uint64_t* d_out,
                   char* errbuf, size_t errlen);

}  // namespace zkhip
