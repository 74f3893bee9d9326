/* zkhip - C ABI of the MI355X (gfx950) wrapping-prover kernels.
 *
 * This is the drop-in boundary for the hot path of clearmatics/zecale:
 *   libzecale::aggregator_circuit::prove -> wsnarkT::generate_proof(pk, pb)
 *   (reference libzecale/circuits/aggregator_circuit.tcc:168; policy class wsnarkT =
 *    libzeth::groth16_snark<libff::bw6_761_pp>, aggregator_server/aggregator_server.cpp:61).
 * The reference has no FFI of its own (its seams are C++ template parameters); the
 * entry points below are what a `groth16_snark_hip<wppT>` policy class binds.
 *
 * Conventions
 *   - Field elements are little-endian arrays of 64-bit limbs in Montgomery form, radix 2^768
 *     (Fq, 12 limbs) / 2^384 (Fr, 6 limbs), fully reduced - libff's in-memory Fp_model layout.
 *   - Affine points: x then y (24 limbs).  The point at infinity is the all-zero pattern.
 *   - Jacobian points: X, Y, Z (36 limbs), infinity <=> Z = 0.  Both G1 and G2 of BW6-761 are
 *     over Fq, so every point entry point serves both groups.
 *   - All functions return ZKHIP_OK (0) or a negative error code; no exceptions cross the ABI.
 *   - Devices: zkhip_init(device) once per GPU the process uses.  Every handle (base set, proving key, constraint
 *     system, prover) remembers the GPU it was created on and every entry point binds the CALLING THREAD to that
 *     GPU first (HIP's current device is per host thread), so handles may be used from any thread.  Entry points
 *     without a handle run on the calling thread's library device: zkhip_set_device, default = the first GPU
 *     initialised.  One process can drive all GPUs of a node this way (the reference server is one process:
 *     aggregator_server/aggregator_server.cpp:390-416).
 *   - The handle-less entry points and the ones that share the library's own work space are serialised per GPU
 *     by the library; zkhip_prover / zkhip_pipeline instances run side by side.
 *   - `_dev` variants take DEVICE pointers (e.g. torch tensor data_ptr()); the others take host
 *     pointers and stage through HBM themselves.
 */
#ifndef ZKHIP_H
#define ZKHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKHIP_OK 0
#define ZKHIP_ERR_ARG (-1)        /* bad argument */
#define ZKHIP_ERR_NO_DEVICE (-2)  /* no gfx950 device / HIP runtime unavailable */
#define ZKHIP_ERR_HIP (-3)        /* a HIP call failed; see zkhip_last_error() */
#define ZKHIP_ERR_STATE (-4)      /* library not initialised / handle invalid */
#define ZKHIP_ERR_NO_TICKET (-5)  /* pipeline / dispatcher wait: this ticket was never issued or has been collected already
                                     (ZKHIP_ERR_ARG from a wait means the BATCH failed: malformed or degenerate nested proofs) */

typedef struct zkhip_bases zkhip_bases; /* opaque: a base-point set resident in HBM */

/* replaces: libff::bw6_761_pp::init_public_params() + device selection
 * (aggregator_server/aggregator_server.cpp:476-477) */
int zkhip_init(int device);
/* library device of the calling thread for the entry points that take no handle (the device must be initialised) */
int zkhip_set_device(int device);
int zkhip_get_device(void);              /* -1 before the first zkhip_init */
int zkhip_device_count(void);            /* GPUs visible to the process (0 without a HIP runtime) */
void zkhip_shutdown(void);
const char* zkhip_strerror(int code);
const char* zkhip_last_error(void);

/* Options of a base set / proving key.  They are resolved ONCE, when the key is uploaded, and travel with the handle: two threads
 * (or two GPUs) loading keys with different options do not see each other's choice.  A field left at its "default" value takes
 * the process-wide default, i.e. what the deprecated zkhip_set_* switches below last set. */
typedef struct {
  int precompute;   /* -1: default (on);  0: plain base sets;  1: window tables (2^(c w) P_i for every window position) */
  int table_naf;    /* -1: default (off, or ZKHIP_TABLE_NAF);  0: one table level per window;  1: EVERY bit position + scalars in
                       non-adjacent form (378 levels: within ZKHIP_NAF_TABLE_GB, default 48 GB per key, else one level per window) */
  int window;       /*  0: automatic from the length;  else the window c of the table, in [4, 22] */
  int batch_msms;   /* -1: default (on);  0: one launch sequence per MSM;  1: the five MSMs of a proof in one launch sequence */
} zkhip_key_opts;
#define ZKHIP_KEY_OPTS_DEFAULT {-1, -1, 0, -1}

/* DEPRECATED process-wide switches (kept for callers of rounds 1-2): they set the DEFAULTS that zkhip_key_opts falls back to.
 * Window size (bits) of the bucket method; 0 = automatic from len. */
int zkhip_set_msm_window(int c);
/* Bucket accumulation: number of batched-affine levels (pairwise sums inside the buckets, one shared inversion per lane) that
 * run before the XYZZ accumulation; -1 = automatic (currently NONE at every size: measured slower on gfx950).  A
 * measurement / test knob, process-wide, read when an MSM plan is made.  Results do not depend on it. */
int zkhip_set_affine_levels(int levels);

/* replaces: holding r1cs_gg_ppzksnark_proving_key query vectors in host memory
 * (aggregator_server/aggregator_server.cpp:483-514 loads the keypair once).
 * Copies `len` affine points (len x 24 limbs) into HBM in the kernels' packed layout. */
int zkhip_bases_upload(const uint64_t* bases_affine, size_t len, zkhip_bases** out);
int zkhip_bases_upload_dev(const void* d_bases_affine, size_t len, zkhip_bases** out);
size_t zkhip_bases_len(const zkhip_bases* b);
void zkhip_bases_free(zkhip_bases* b);
/* Window table for a resident base set (a proving key stays in HBM for the life of the server, the reference keeps it
 * in RAM: aggregator_server.cpp:483-514): stores 2^(c w) P_i for every window position w, so that all digit positions
 * of a scalar share one bucket window.  Costs ceil(378/c) x the memory of the base set, once; results of zkhip_msm are
 * unchanged (same group element).  c = 0 chooses the window from the length.  zkhip_crs_upload builds the tables of a
 * key by default; zkhip_set_crs_precompute(0) turns that off. */
int zkhip_bases_precompute(zkhip_bases* b, int c);
/* the same with the kind of table as an argument (zkhip_key_opts.table_naf: -1 default, 0 one level per window, 1 every bit position) */
int zkhip_bases_precompute_ex(zkhip_bases* b, int c, int table_naf);
int zkhip_bases_table_window(const zkhip_bases* b);      /* 0: no table */
/* plain base set (no table): the window c of the bucket method for the MSMs over THIS set, 0 = by the number of terms (default),
 * else 4 .. 18.  An option of the handle, like zkhip_key_opts of a key (the process-wide zkhip_set_msm_window is deprecated). */
int zkhip_bases_set_window(zkhip_bases* b, int c);
int zkhip_set_crs_precompute(int on);
/* on = 1: window tables built from now on hold EVERY bit position (378 levels, sixteen times the memory) and scalars are recoded in
   non-adjacent form: an eighth fewer point additions per scalar, but measured slower once the tables outgrow the TLB (DESIGN.md
   section 5) - off by default; 0: one level per window; -1: the environment's ZKHIP_TABLE_NAF decides.  Results are identical. */
int zkhip_set_table_naf(int on);
/* Table-backed keys of up to 2^20 points per query vector: the five MSMs of a proof go through one launch sequence
 * (default on; off = one launch sequence per MSM, 2 or 5 of them in flight).  A tuning / comparison switch. */
int zkhip_set_batch_msms(int on);

/* replaces: libff::multi_exp<G, Fr, multi_exp_method_BDLO12>(bases, scalars, chunks)
 * as called five times by r1cs_gg_ppzksnark_prover (reached from aggregator_circuit.tcc:168).
 * result = sum_{i<len} scalars[i] * bases[offset + i].
 * scalars: len x 6 limbs, Montgomery form (scalars_montgomery = 1) or canonical integers (0). */
int zkhip_msm(const zkhip_bases* bases, size_t offset, const uint64_t* scalars, size_t len, int scalars_montgomery,
              uint64_t out_jac[36]);
int zkhip_msm_dev(const zkhip_bases* bases, size_t offset, const void* d_scalars, size_t len, int scalars_montgomery,
                  uint64_t out_jac[36]);
/* Device pointers handed to *_dev / submit entry points must hold complete data: the library runs on its own non-blocking
 * streams, which are not ordered against the caller's streams (synchronise the producing stream first). */
/* Device memory for host code that has no HIP runtime of its own (the *_dev entry points take device pointers). */
int zkhip_device_alloc(size_t bytes, void** out);
int zkhip_device_free(void* p);
int zkhip_device_copy_in(void* dst, const void* src, size_t bytes);
int zkhip_device_copy_out(void* dst_host, const void* src_device, size_t bytes);

/* Asynchronous form of zkhip_msm_dev for a stream of MSMs on resident bases: submit enqueues the MSM on one of EIGHT
 * slots (0 .. 7) and returns, collect waits for it.  d_scalars must stay valid until collect.  With several slots in flight the
 * sort, the stitching and the latency-bound bucket reduction of one MSM run under the accumulation of another (measured at
 * 2^20 terms: 75 / 77 / 80 Mscalar/s with 2 / 3 / 4 in flight); each slot holds its own work space (1.3 GB at 2^20 terms). */
int zkhip_msm_submit(const zkhip_bases* bases, size_t offset, const void* d_scalars, size_t len, int scalars_montgomery, int slot);
int zkhip_msm_collect(int slot, uint64_t out_jac[36]);
/* The same stream of MSMs behind a HANDLE (the slot numbers above are process-wide: two threads streaming on one GPU collide).
 * A zkhip_msm_stream owns `depth` (1 .. 16) MSM contexts - streams and work space, like a zkhip_prover - on the GPU of its base
 * set; any number of streams run side by side.  submit enqueues on a free context and returns a ticket (ZKHIP_ERR_STATE when all
 * `depth` are in flight); collect waits for that ticket (0 = the oldest in flight).  replaces: one libff::multi_exp call
 * (reached from aggregator_circuit.tcc:168) per ticket.  d_scalars must stay valid until its ticket is collected.
 * submit_host takes HOST scalars (as libff holds them): one asynchronous copy in front of the MSM's kernels on the context's own
 * stream - truly asynchronous from pinned memory (zkhip_host_alloc), staged by the runtime from pageable memory - so the 48 MB
 * uploads of the MSMs in flight travel under their kernels. */
typedef struct zkhip_msm_stream zkhip_msm_stream;
int zkhip_msm_stream_new(const zkhip_bases* bases, int depth, zkhip_msm_stream** out);     /* bases must outlive the stream */
int zkhip_msm_stream_submit(zkhip_msm_stream* st, size_t offset, const void* d_scalars, size_t len, int scalars_montgomery, uint64_t* ticket);
int zkhip_msm_stream_submit_host(zkhip_msm_stream* st, size_t offset, const uint64_t* scalars, size_t len, int scalars_montgomery, uint64_t* ticket);
int zkhip_msm_stream_collect(zkhip_msm_stream* st, uint64_t ticket, uint64_t out_jac[36]);
float zkhip_msm_stream_last_accumulate_ms(zkhip_msm_stream* st);               /* of the MSM collected last */
int zkhip_msm_stream_last_accumulate_interval(zkhip_msm_stream* st, float out_ms[2]);
void zkhip_msm_stream_free(zkhip_msm_stream* st);                              /* waits for what is still in flight */
/* pinned host memory for callers without a HIP runtime of their own (the source of submit_host's asynchronous copies) */
int zkhip_host_alloc(size_t bytes, void** out);
int zkhip_host_free(void* p);
/* one-shot form (BASELINE config 2): host bases + host scalars */
int zkhip_msm_raw(const uint64_t* bases_affine, const uint64_t* scalars, size_t len, int scalars_montgomery,
                  uint64_t out_jac[36]);

/* replaces: the fixed-base batch exponentiations of r1cs_gg_ppzksnark_generator, reached from
 * aggregator_circuit::generate_trusted_setup (libzecale/circuits/aggregator_circuit.tcc:100-109).
 * out[i] = scalars[i] * base, affine (len x 24 limbs).  base: one affine point (host memory).
 * The _dev form reads scalars from and writes points to DEVICE memory. */
int zkhip_fixed_base_mul(const uint64_t base_affine[24], const uint64_t* scalars, size_t len, int scalars_montgomery,
                         uint64_t* out_affine);
int zkhip_fixed_base_mul_dev(const uint64_t base_affine[24], const void* d_scalars, size_t len, int scalars_montgomery,
                             void* d_out_affine);

/* replaces: libfqfft::basic_radix2_domain<Fr>::FFT / iFFT / cosetFFT / icosetFFT as used by
 * r1cs_to_qap_witness_map (reached from aggregator_circuit.tcc:168).  In place, natural order in
 * and out, 2^log_d elements of 6 limbs (Montgomery form); domain <omega>, omega = 15^((r-1)/2^log_d),
 * coset generator g = 15.   dir: 0 = forward, 1 = inverse.   coset: 0 / 1.   log_d <= 22.
 *   forward, coset:   a_i <- a_i g^i, then FFT          inverse, coset:   iFFT, then a_i <- a_i g^-i */
int zkhip_ntt(uint64_t* data, unsigned log_d, int dir, int coset);
int zkhip_ntt_dev(void* d_data, unsigned log_d, int dir, int coset);

/* ---- R1CS, QAP witness map and the Groth16 prover ---------------------------------------- */
/* A constraint system <A_i,z> * <B_i,z> = <C_i,z>, i < n_constraints, in CSR form; z = (1, primary,
 * auxiliary) has n_vars entries.  Replaces libsnark::r1cs_constraint_system<Fr<wppT>> as returned by
 * aggregator_circuit::get_constraint_system() (libzecale/circuits/aggregator_circuit.hpp:99-101). */
typedef struct {
  size_t n_constraints, n_vars, n_primary;
  const uint32_t *a_row_ptr, *a_col; const uint64_t* a_val;   /* row_ptr: n+1, col: nnz, val: nnz x 6 limbs */
  const uint32_t *b_row_ptr, *b_col; const uint64_t* b_val;
  const uint32_t *c_row_ptr, *c_col; const uint64_t* c_val;
} zkhip_r1cs_desc;
typedef struct zkhip_r1cs zkhip_r1cs;
/* THE EVALUATION DOMAIN of the QAP (a7).  A system of n constraints and l primary inputs interpolates n + l + 1 points.
 *   - The reference reaches libsnark through libzeth's groth16_snark, whose generate_setup / generate_proof
 *     (libzecale/circuits/aggregator_circuit.tcc:108, :168) pass force_pow_2_domain = true: the domain is libfqfft's
 *     basic_radix2_domain of 2^ceil(log2(n + l + 1)) points.  That is the DEFAULT of every entry
 *     point here: the wrapping circuit (44,183 constraints + 5) lives on 65,536 points and its key's H query has 65,535 entries.
 *   - ZKHIP_DOMAIN_STEP asks for what libfqfft's get_evaluation_domain picks when NOT forced: a power of two if n + l + 1 is one,
 *     else step_radix2_domain of 2^k + 2^r points (49,152 for the wrapping circuit: a quarter fewer H terms).  An explicit option -
 *     NOT what a reference deployment uses, and unpinnable here (libfqfft and the reference's keys are absent from its tree).
 *   - Any other value must be a size one of those two rules can return (zkhip_domain_is_valid) and at least n + l + 1.
 * THE PROVING KEY IS AUTHORITATIVE: zkhip_crs_desc.domain_size says which domain a key was generated on, and every prover
 * (zkhip_groth16_prove, zkhip_prover_new[_slice], zkhip_multi_prover_new, the pipelines) works on the key's domain - a 65,536-point
 * key from a reference-style setup and a 49,152-point key from zkhip_groth16_setup_ex(.., ZKHIP_DOMAIN_STEP, ..) both prove.  A
 * zkhip_r1cs handle handed to zkhip_groth16_prove with a key of another domain is moved to it (zkhip_r1cs_set_domain; the matrices
 * stay in HBM, the domain's work buffers are rebuilt once).  A key whose domain cannot hold the system is refused (ZKHIP_ERR_ARG). */
#define ZKHIP_DOMAIN_DEFAULT ((size_t)0)       /* the reference's forced power of two */
#define ZKHIP_DOMAIN_STEP (~(size_t)0)         /* libfqfft's unforced get_evaluation_domain: a power of two or 2^k + 2^r */
int zkhip_r1cs_upload(const zkhip_r1cs_desc* d, zkhip_r1cs** out);                        /* = _ex(d, ZKHIP_DOMAIN_DEFAULT, out) */
int zkhip_r1cs_upload_ex(const zkhip_r1cs_desc* d, size_t domain_size, zkhip_r1cs** out);
int zkhip_r1cs_set_domain(zkhip_r1cs* r, size_t domain_size);   /* no proof may be in flight on the handle */
void zkhip_r1cs_free(zkhip_r1cs* r);
size_t zkhip_r1cs_domain_size(const zkhip_r1cs* r);     /* d: the points of the handle's current domain (a key's H query: d - 1) */
/* ceil(log2 d): log2 d exactly for a radix-2 domain (the default); for a step domain 2^k + 2^r it is k + 1 - use
 * zkhip_r1cs_domain_size where the number of points matters */
unsigned zkhip_r1cs_log_domain(const zkhip_r1cs* r);
/* host code, no device: the domain size for min_size = n + l + 1 points under each rule, and whether a size is a domain at all */
size_t zkhip_domain_size(size_t min_size);              /* forced power of two (the reference, the default) */
size_t zkhip_step_domain_size(size_t min_size);         /* libfqfft unforced: 2^k, or 2^k + 2^r */
int zkhip_domain_is_valid(size_t domain_size);          /* 1 / 0 */

/* replaces: protoboard::is_satisfied() under DEBUG (aggregator_circuit.tcc:159-164); *ok = 1/0 */
int zkhip_r1cs_is_satisfied(zkhip_r1cs* r, const uint64_t* z, int* ok);

/* replaces: libsnark::r1cs_to_qap_witness_map(cs, primary, auxiliary, 0, 0, 0, force_pow_2 = true) -
 * coefficients_for_H, over the handle's domain (above).  h_out: d x 6 limbs (h_{d-1} = 0). */
int zkhip_qap_h(zkhip_r1cs* r, const uint64_t* z, uint64_t* h_out);

/* The proving key (replaces r1cs_gg_ppzksnark_proving_key<bw6_761_pp> held by the server,
 * aggregator_server/aggregator_server.cpp:483-514).  All points affine, 24 limbs each.
 * Lengths: a_query, b_g2_query, b_g1_query: n_vars;  h_query: d - 1;  l_query: n_vars - n_primary - 1. */
typedef struct {
  size_t n_vars, n_primary, domain_size;
  const uint64_t *alpha_g1, *beta_g1, *beta_g2, *delta_g1, *delta_g2;
  const uint64_t *a_query, *b_g2_query, *b_g1_query, *h_query, *l_query;
} zkhip_crs_desc;
typedef struct zkhip_crs zkhip_crs;
int zkhip_crs_upload(const zkhip_crs_desc* d, zkhip_crs** out);
/* the same with the key's options as an argument (opts == NULL: all defaults) */
int zkhip_crs_upload_ex(const zkhip_crs_desc* d, const zkhip_key_opts* opts, zkhip_crs** out);
void zkhip_crs_free(zkhip_crs* c);
int zkhip_crs_table_kind(const zkhip_crs* c);            /* 0: no tables, 1: one level per window, 2: every bit position (NAF scalars) */
/* bases of the five query vectors (A, B-G2, B-G1, H, L) that are not the point at infinity: the terms an MSM over the key can
 * actually have (a base at infinity produces no bucket entry) */
int zkhip_crs_finite_terms(const zkhip_crs* c, size_t out[5]);
int zkhip_crs_table_window(const zkhip_crs* c);          /* window of the key's tables, 0: none */
int zkhip_crs_device(const zkhip_crs* c);                /* the GPU that holds the key */

/* replaces: wsnarkT::generate_proof(pk, pb) = libsnark::r1cs_gg_ppzksnark_prover (called at
 * libzecale/circuits/aggregator_circuit.tcc:168), with the randomisers (r, s) injected so that
 * results can be compared bit for bit.  z: n_vars x 6 limbs.  proof_affine: A (G1), B (G2), C (G1). */
int zkhip_groth16_prove(const zkhip_crs* crs, zkhip_r1cs* r1cs, const uint64_t* z, const uint64_t r[6], const uint64_t s[6],
                        uint64_t proof_affine[72]);
/* Multi-GPU form (the proving key is partitioned across the GPUs of a node, one process per GPU):
 *   zkhip_crs_upload_slice   uploads this rank's slice of every query vector: [a_lo, a_lo+a_len) of the A / B queries,
 *                            [h_lo, ..) of H, [l_lo, ..) of L (full-key descriptor + ranges)
 *   zkhip_groth16_prove_partial  QAP map (replicated) + the five MSMs over the slice -> 5 Jacobian partial sums
 *                            (A, B-G2, B-G1, H, L: 5 x 36 limbs)
 *   ... the ranks exchange and add the partial sums (RCCL all-gather of 1440 bytes, zecale_amd/dist.py) ...
 *   zkhip_groth16_finish     host tail on the summed values.
 * zkhip_groth16_prove is exactly prove_partial over the whole key followed by finish. */
int zkhip_crs_upload_slice(const zkhip_crs_desc* full_key, size_t a_lo, size_t a_len, size_t h_lo, size_t h_len, size_t l_lo, size_t l_len,
                           zkhip_crs** out);
int zkhip_crs_upload_slice_ex(const zkhip_crs_desc* full_key, size_t a_lo, size_t a_len, size_t h_lo, size_t h_len, size_t l_lo, size_t l_len,
                              const zkhip_key_opts* opts, zkhip_crs** out);
int zkhip_groth16_prove_partial(const zkhip_crs* crs_slice, zkhip_r1cs* r1cs, const uint64_t* z, size_t a_lo, size_t h_lo, size_t l_lo,
                                uint64_t sums_jac[180]);
int zkhip_groth16_finish(const uint64_t alpha_g1[24], const uint64_t beta_g1[24], const uint64_t beta_g2[24], const uint64_t delta_g1[24],
                         const uint64_t delta_g2[24], const uint64_t sums_jac[180], const uint64_t r[6], const uint64_t s[6],
                         uint64_t proof_affine[72]);
/* phase timings of the last zkhip_groth16_prove, milliseconds: [0] upload z, [1] QAP (SpMV + 7 NTT),
 * [2..6] the five MSMs A, B2, B1, H, L, [7] host tail */
int zkhip_last_prove_timings(double out_ms[8]);

/* Prover instances.  The entry points above share one set of device work space and are serialised by the library;
 * a zkhip_prover owns its streams, MSM work space and QAP buffers, so several host threads (one instance each) keep
 * several proofs in flight on one GPU.  At the wrapping circuit's size (50k constraints) one proof cannot fill the
 * chip: the phases after the bucket accumulation are chains of short launches.  Same results as zkhip_groth16_prove.
 * replaces: one wsnarkT::generate_proof call (aggregator_circuit.tcc:168) per instance and call. */
typedef struct zkhip_prover zkhip_prover;
int zkhip_prover_new(const zkhip_crs* crs, const zkhip_r1cs_desc* cs, zkhip_prover** out);   /* crs must outlive the prover */
/* on != 0: this prover shares the GPU with others (a server's slots): the latency-bound launches of the bucket reduction then use
   one lane per point addition instead of four - less total work, a longer single proof -, slices of the bucket accumulation are twice
   as long, and from its second proof on the instance enqueues the upload of the assignment and the QAP map on the stream of its MSM
   launch sequence: the calling thread waits once, for the MSM results (zkhip_prover_timings then reports enqueueing times for the
   first two phases; ZKHIP_STREAM_CHAIN=0 in the environment keeps the waits).  The streaming pipeline sets it on its provers. */
int zkhip_prover_set_streaming(zkhip_prover* p, int on);
/* no counterpart: for a caller that owns SEVERAL instances which will prove at the same time (the streaming prover does this itself).
 * The runtime spreads streams over its hardware queues in creation order and kernels of streams that share a queue do not overlap; an
 * instance in streaming mode keeps one of its three streams busy.  Call with which = 0 for every instance (the busy streams: spread
 * evenly), then with which = 1 for every instance (the idle ones), before the first proof; without it the streams are created by the
 * first proofs, in whatever order the provers' threads arrive. */
int zkhip_prover_create_streams(zkhip_prover* p, int which);
int zkhip_prover_prove(zkhip_prover* p, const uint64_t* z, const uint64_t r[6], const uint64_t s[6], uint64_t proof_affine[72]);
/* the same with the assignment already in device memory (n_vars x 6 limbs, e.g. from zkhip_gpu_witness_run; must be complete) */
int zkhip_prover_prove_dev(zkhip_prover* p, const void* d_z, const uint64_t r[6], const uint64_t s[6], uint64_t proof_affine[72]);
int zkhip_prover_timings(zkhip_prover* p, double out_ms[8]);   /* as zkhip_last_prove_timings, for p's last proof */
/* 1 if p's last proof was chained (streaming mode, see zkhip_prover_set_streaming): slots [0] and [1] above are then the time it took
 * to ENQUEUE the upload and the QAP map, and the five MSM slots hold the whole device time of the proof */
int zkhip_prover_timings_chained(zkhip_prover* p);
/* (2: p's last proof was SPLIT - with ZKHIP_PROVE_SPLIT=1, an option that is OFF by default because it measured slower, a prover
 * that is not streaming runs the four MSMs over the assignment beside the QAP map and the H MSM behind it, two launch sequences:
 * slot [1] is then the time to enqueue the map and both sequences, the MSM slots the device time of all of it.)  The same question
 * for the plain entry points
 * (zkhip_groth16_prove[_partial|_app]) on the calling thread's device: */
int zkhip_last_prove_split(void);
/* the option itself, process-wide (provers that are NOT streaming): 0 one launch sequence (default), 1 two sequences with the H
 * accumulation gated behind the first, 2 not gated.  Proofs are bit-identical in every mode. */
int zkhip_set_prove_split(int mode);
float zkhip_prover_last_accumulate_ms(zkhip_prover* p);          /* as zkhip_last_accumulate_ms, for p's last proof */
int zkhip_prover_last_accumulate_entries(zkhip_prover* p, uint64_t* out);   /* as zkhip_last_accumulate_entries, for p's last proof */
void zkhip_prover_free(zkhip_prover* p);
/* A prover instance over a SLICE of the key (zkhip_crs_upload_slice[_ex] with the same three offsets): own streams and work space
 * like any zkhip_prover, so several slices - on several GPUs, or several contexts of one - run side by side.  It only produces
 * partial sums: zkhip_prover_prove_partial = zkhip_groth16_prove_partial on this instance (QAP map + five MSMs over the slice). */
int zkhip_prover_new_slice(const zkhip_crs* crs_slice, const zkhip_r1cs_desc* cs, size_t a_lo, size_t h_lo, size_t l_lo, zkhip_prover** out);
int zkhip_prover_prove_partial(zkhip_prover* p, const uint64_t* z, uint64_t sums_jac[180]);

/* replaces: wsnarkT::verify(primary_inputs, proof, vk) (libzecale/tests/aggregator/aggregator_dummy_test.cpp:61-62)
 * = libsnark r1cs_gg_ppzksnark_verifier_strong_IC for the Clearmatics Groth16 (no gamma in the key:
 * testdata/dummy_app/aggregator_vk.json), i.e. the equation of contracts/Groth16BW6_761.sol:166-176.
 * Host code (verification is not on the prover's hot path); needs no device and no zkhip_init.
 * vk_abc: (n_inputs + 1) x 24 limbs; inputs: n_inputs x 6 limbs; proof: A (G1), B (G2), C (G1) affine. */
int zkhip_groth16_verify(const uint64_t vk_alpha_g1[24], const uint64_t vk_beta_g2[24], const uint64_t vk_delta_g2[24],
                         const uint64_t* vk_abc, const uint64_t* inputs, size_t n_inputs, const uint64_t proof_affine[72],
                         int* ok);

/* ---- the wrapping (aggregator) circuit: host code, no device needed ------------------------------- */
/* Nested objects are over BLS12-377, whose base field is Fr of BW6-761: coordinates are 6-limb Montgomery
 * elements.  G1 affine = x | y (12 limbs); G2 affine = x.c0 | x.c1 | y.c0 | y.c1 (24 limbs), Fq2 = Fq[u]/(u^2+5).
 * nested_vk   = alpha (12) | beta (24) | delta (24) | ABC_0 .. ABC_k (12 each), k = inputs_per_proof
 * nested_proofs = num_proofs x [ a (12) | b (24) | c (12) ];  nested_inputs = num_proofs x k x 6 limbs. */

/* replaces: nsnark::verify for the nested curve (libzecale/tests/circuits/dummy_application_test.cpp:32-44):
 * Groth16 over BLS12-377, Clearmatics variant without gamma. */
int zkhip_bls12_377_groth16_verify(const uint64_t vk_alpha_g1[12], const uint64_t vk_beta_g2[24], const uint64_t vk_delta_g2[24],
                                   const uint64_t* vk_abc, const uint64_t* inputs, size_t n_inputs,
                                   const uint64_t proof_a[12], const uint64_t proof_b[24], const uint64_t proof_c[12], int* ok);

/* replaces: libzecale::aggregator_circuit<wppT, wsnarkT, nverifierT, NumProofs>(inputs_per_nested_proof)
 * (libzecale/circuits/aggregator_circuit.hpp:32-114; constructor .tcc:17-98): builds the constraint system. */
typedef struct zkhip_aggregator zkhip_aggregator;
int zkhip_aggregator_new(size_t num_proofs, size_t inputs_per_proof, zkhip_aggregator** out);
void zkhip_aggregator_free(zkhip_aggregator* a);
size_t zkhip_aggregator_num_proofs(const zkhip_aggregator* a);
size_t zkhip_aggregator_inputs_per_proof(const zkhip_aggregator* a);
size_t zkhip_aggregator_num_constraints(const zkhip_aggregator* a);
size_t zkhip_aggregator_num_variables(const zkhip_aggregator* a);      /* including the constant ONE */
size_t zkhip_aggregator_num_primary_inputs(const zkhip_aggregator* a); /* aggregator_circuit::num_primary_inputs, .tcc:172-180 */
/* replaces: get_constraint_system() (aggregator_circuit.hpp:99-101); pointers stay valid while `a` lives */
int zkhip_aggregator_get_r1cs(const zkhip_aggregator* a, zkhip_r1cs_desc* out);
/* replaces: the generate_r1cs_witness calls of aggregator_circuit::prove (.tcc:136-157): full assignment
 * z = (1, primary, auxiliary), n_vars x 6 limbs; primary = [vk hash, packed results, nested inputs ...] */
int zkhip_aggregator_witness(zkhip_aggregator* a, const uint64_t* nested_vk, const uint64_t* nested_proofs,
                             const uint64_t* nested_inputs, uint64_t* z_out);
/* replaces: the well-formedness checks libsnark applies to a proof before it is used (proof.is_well_formed(); in the circuit:
 * the G1 / G2 checker gadgets of r1cs_gg_ppzksnark_proof_variable, groth16_verifier_parameters.hpp:16-27): *ok = 1 iff every
 * point of the nested key and of the nested proofs lies on its curve (BLS12-377 G1: y^2 = x^3 + 1, G2: y^2 = x^3 + 1/u).
 * zkhip_aggregator_witness itself computes an assignment for ANY input; for an off-curve proof point that assignment violates
 * the circuit's curve constraints (no wrapping proof exists).  The streaming prover and the C++ mirror call this first. */
int zkhip_aggregator_check_inputs(const zkhip_aggregator* a, const uint64_t* nested_vk, const uint64_t* nested_proofs, int* ok);
/* replaces: verification_key_hash_gadget::compute_hash(vk, num_inputs) (verification_key_hash_gadget.tcc:42-59) */
int zkhip_aggregator_vk_hash(const uint64_t* nested_vk, size_t inputs_per_proof, uint64_t out[6]);

/* Witness generation ON THE GPU: the same assignment as zkhip_aggregator_witness, limb for limb, computed by a device kernel that
 * interprets the straight-line program recorded from the circuit (witness_tape.cpp, witness.hip).  One batch occupies four waves for
 * ~tens of milliseconds: the gain is not latency (the host generator takes 8 ms on three cores) but host cores - a server keeps
 * many batches in flight and the GPU generates their witnesses under the provers' kernels.
 *   zkhip_aggregator_witness_gpu   one batch, host in / host out (allocates its work space: a test / convenience entry point)
 *   zkhip_gpu_witness_*            work space for one batch in flight on the calling thread's device; run() leaves the assignment
 *                                  in DEVICE memory (d_z_out: n_vars x 6 limbs, ready for zkhip_prover_prove_dev) and returns the
 *                                  primary inputs.  ZKHIP_ERR_ARG when an inversion met zero (degenerate nested points, where the
 *                                  host generator branches): fall back to zkhip_aggregator_witness for that batch.
 * replaces: the generate_r1cs_witness calls of aggregator_circuit::prove (aggregator_circuit.tcc:136-157, aggregator_gadget.tcc:87-112) */
int zkhip_aggregator_witness_gpu(zkhip_aggregator* a, const uint64_t* nested_vk, const uint64_t* nested_proofs, const uint64_t* nested_inputs,
                                 uint64_t* z_out);
typedef struct zkhip_gpu_witness zkhip_gpu_witness;
int zkhip_gpu_witness_new(zkhip_aggregator* a, zkhip_gpu_witness** out);
int zkhip_gpu_witness_run(zkhip_gpu_witness* w, const uint64_t* nested_vk, const uint64_t* nested_proofs, const uint64_t* nested_inputs,
                          void* d_z_out, uint64_t* primary_inputs);
 /* several batches in ONE launch sequence (a workgroup each: n witnesses take as long as one).  d_z_out: n x n_vars x 6 limbs,
 * contiguous; primary_inputs: n x n_primary x 6 limbs; degenerate[i] != 0: batch i met an inversion of zero, its assignment is void */
int zkhip_gpu_witness_new_batched(zkhip_aggregator* a, size_t max_batches, zkhip_gpu_witness** out);
int zkhip_gpu_witness_run_batched(zkhip_gpu_witness* w, size_t n, const uint64_t* const* nested_vk, const uint64_t* const* nested_proofs,
                                  const uint64_t* const* nested_inputs, void* d_z_out, uint64_t* primary_inputs, int* degenerate);
void zkhip_gpu_witness_free(zkhip_gpu_witness* w);
/* the recorded program: [0] operations recorded, [1] positions after levelling (with padding), [2] dependent levels,
 * [3] multiplications, [4] inversions, [5] distinct constants */
int zkhip_gpu_witness_stats(zkhip_aggregator* a, size_t out[6]);

/* ---- PER-APPLICATION CONSTANTS -----------------------------------------------------------------------------------------------
 * The reference registers an application - a nested verification key - once (RegisterApplication, aggregator_server.cpp:170-235)
 * and then aggregates batch after batch of ITS proofs (GenerateAggregatedTransaction, :279-348).  Everything the wrapping circuit
 * derives from the nested key alone is the same in every one of those batches: the key's variables, the MiMC chain of its hash
 * (verification_key_hash_gadget.tcc:36-40), the lines of -beta and -delta (aggregator_gadget.tcc:93) and the doubling chains
 * 2^j ABC_i of the input accumulators - 9,038 of the 44,206 variables at batch 2 with one input per nested proof.  Their share of
 * the A, B-G2, B-G1 and L sums of a proof, sum z_i Base_i over those positions, is FOUR CONSTANT POINTS per (application, proving
 * key).  A zkhip_aggregator_app computes both once:
 *   - the constant positions (auxiliary variables only: the constant ONE and the primary inputs - the key's hash among them - stay
 *     ordinary entries) and their values, by recording the circuit with the key as constants (witness_tape.cpp): what the
 *     recorder folds does not depend on the proofs;
 *   - the four points, by one masked MSM per query over the resident key.
 * Per batch, the generators then produce a MASKED assignment - zero at the constant positions: a zero scalar produces no bucket
 * entry, so the five MSMs lose a sixth of their entries - the QAP map runs on (masked | constants), H is untouched, and the tail adds
 * the four points.  The proof is the same group elements as without the handle: bit-identical (tests/test_app_cache_gpu.py).
 * The handle belongs to ONE proving key (its GPU) and ONE circuit; crs and a must outlive it. */
typedef struct zkhip_aggregator_app zkhip_aggregator_app;
int zkhip_aggregator_app_new(zkhip_aggregator* a, const zkhip_crs* crs, const uint64_t* nested_vk, zkhip_aggregator_app** out);
void zkhip_aggregator_app_free(zkhip_aggregator_app* app);
size_t zkhip_aggregator_app_num_constants(const zkhip_aggregator_app* app);
/* positions (num_constants, ascending), their values (x 6 limbs), the key's hash (primary input 0) and the four cached points
 * (A, B-G2, B-G1, L: 4 x 36 limbs, Jacobian); any pointer may be null */
int zkhip_aggregator_app_constants(const zkhip_aggregator_app* app, uint32_t* positions, uint64_t* values, uint64_t vk_hash[6], uint64_t points_jac[144]);
/* a FULL assignment generated under the application's key -> masked in place (ZKHIP_ERR_ARG if a constant position holds another value) */
int zkhip_aggregator_app_mask(const zkhip_aggregator_app* app, uint64_t* z);
/* replaces: the generate_r1cs_witness calls of aggregator_circuit::prove (aggregator_circuit.tcc:136-157) for a batch of the
 * registered application: the proof sections only (the key's hash and lines are not recomputed), masked assignment out */
int zkhip_aggregator_witness_app(const zkhip_aggregator_app* app, const uint64_t* nested_proofs, const uint64_t* nested_inputs, uint64_t* z_out);
/* replaces: wsnarkT::generate_proof (aggregator_circuit.tcc:168) for a masked assignment: same proof as zkhip_groth16_prove /
 * zkhip_prover_prove[_dev] on the full one.  A host assignment that is not masked is refused (ZKHIP_ERR_ARG). */
int zkhip_groth16_prove_app(const zkhip_crs* crs, zkhip_r1cs* r1cs, const zkhip_aggregator_app* app, const uint64_t* z_masked, const uint64_t r[6],
                            const uint64_t s[6], uint64_t proof_affine[72]);
int zkhip_prover_prove_app(zkhip_prover* p, const zkhip_aggregator_app* app, const uint64_t* z_masked, const uint64_t r[6], const uint64_t s[6],
                           uint64_t proof_affine[72]);
/* PRECONDITION of the device form (not checked - a check would cost a pass over the assignment per proof): d_z_masked is an assignment
 * of THIS application with zero at every constant position (zkhip_aggregator_app_constants' `positions`) - what
 * zkhip_gpu_witness_run_batched_app writes.  The QAP map ORs the constants into it limb-wise, so a full or foreign assignment gives a
 * wrong H and a proof that does not verify, with ZKHIP_OK.  The host forms above refuse such an assignment. */
int zkhip_prover_prove_app_dev(zkhip_prover* p, const zkhip_aggregator_app* app, const void* d_z_masked, const uint64_t r[6], const uint64_t s[6],
                               uint64_t proof_affine[72]);
/* zkhip_gpu_witness_run_batched for n batches of ONE application, by the application's own device program (the key folded in: no
 * key-hash launch, an eighth fewer multiplications): masked assignments in device memory, ready for zkhip_prover_prove_app_dev */
int zkhip_gpu_witness_run_batched_app(zkhip_gpu_witness* w, zkhip_aggregator_app* app, size_t n, const uint64_t* const* nested_proofs,
                                      const uint64_t* const* nested_inputs, void* d_z_out, uint64_t* primary_inputs, int* degenerate);

/* Streaming form of aggregator_circuit::prove for a server that wraps batch after batch (the reference's
 * GenerateAggregatedTransaction loop, aggregator_server.cpp:300-420, handles one batch at a time on the CPU):
 * `witness_workers` host threads generate witnesses (zkhip_aggregator_witness) while `gpu_slots` prover instances
 * keep that many proofs in flight on the GPU; the host tail of one proof overlaps the device work of the next.
 * submit copies its inputs and returns a ticket (it blocks only while 4 x (gpu_slots + witness_workers) batches are
 * still unproved; finished batches wait for their collector and never block a submitter); wait blocks until that batch is done and returns the extended proof: primary inputs
 * (num_primary_inputs x 6 limbs: vk hash, packed results, nested inputs) and the proof (a | b | c, 72 limbs).
 * Every result is bit-identical to zkhip_aggregator_witness + zkhip_groth16_prove on the same inputs, r and s.
 * gpu_slots and witness_workers: 1 .. 64 each.  ZKHIP_PIPELINE_STATS=1 in the environment prints, when a pipeline is freed, the time
 * a proof spent in a prover, the time a prover waited for an assignment and the host witness time (stderr). */
typedef struct zkhip_pipeline zkhip_pipeline;
int zkhip_aggregator_pipeline_new(zkhip_aggregator* a, const zkhip_crs* crs, int gpu_slots, int witness_workers, zkhip_pipeline** out);
/* flags: ZKHIP_PIPELINE_GPU_WITNESS - the witness workers generate the assignment on the GPU (zkhip_gpu_witness_run: each worker is
 * a host thread that waits on its batch's kernel; the assignment never leaves the device) and fall back to the host generator for a
 * degenerate batch.  Results are the same proofs. */
#define ZKHIP_PIPELINE_GPU_WITNESS 1u
/* ZKHIP_PIPELINE_NO_APP_CACHE - a pipeline keeps a zkhip_aggregator_app per nested key it has seen (at most 32; built by the first
 * batch of a key or by zkhip_aggregator_pipeline_register_app) and proves that key's batches from masked assignments; this flag (or
 * ZKHIP_NO_APP_CACHE in the environment) turns that off.  The proofs are the same either way. */
#define ZKHIP_PIPELINE_NO_APP_CACHE 2u
/* ZKHIP_PIPELINE_HYBRID_WITNESS (with ZKHIP_PIPELINE_GPU_WITNESS) - two host generator threads (ZKHIP_HYBRID_HOST_WORKERS: 1 .. 16) work
 * the same queue as the GPU batchers, taking a batch only while MORE than one full witness launch (16 batches) is queued - what the
 * batchers could not start on anyway (a launch of three batches takes as long as one of sixteen).  Round 5, one MI355X, batch 2: between
 * the pure modes in host cores (2.9-3.4; host 4.1-4.7, GPU 2.1-2.4) and, run by run, between 6 % below and 3 % above the host
 * generator in proofs/s.  Results are the same proofs. */
#define ZKHIP_PIPELINE_HYBRID_WITNESS 4u
int zkhip_aggregator_pipeline_new_ex(zkhip_aggregator* a, const zkhip_crs* crs, int gpu_slots, int witness_workers, unsigned flags, zkhip_pipeline** out);
/* replaces: RegisterApplication's part in the prover (aggregator_server.cpp:170-235 stores the key; here its constants are computed,
 * ~0.2 s, so that the application's first batch does not pay for them).  ZKHIP_ERR_ARG: a key with a point off its curve. */
int zkhip_aggregator_pipeline_register_app(zkhip_pipeline* p, const uint64_t* nested_vk);
size_t zkhip_aggregator_pipeline_app_hits(const zkhip_pipeline* p);     /* batches proved from an application's constants so far */
int zkhip_aggregator_pipeline_submit(zkhip_pipeline* p, const uint64_t* nested_vk, const uint64_t* nested_proofs,
                                     const uint64_t* nested_inputs, const uint64_t r[6], const uint64_t s[6], uint64_t* ticket);
int zkhip_aggregator_pipeline_wait(zkhip_pipeline* p, uint64_t ticket, uint64_t* primary_inputs, uint64_t proof_affine[72]);
void zkhip_aggregator_pipeline_free(zkhip_pipeline* p);

/* replaces: wsnarkT::generate_setup(pb) = libsnark::r1cs_gg_ppzksnark_generator, reached from
 * aggregator_circuit::generate_trusted_setup (libzecale/circuits/aggregator_circuit.tcc:100-109).
 * QAP evaluation at tau on the host, the batch exponentiations on the GPU (zkhip_fixed_base_mul).
 * The toxic waste is an argument so that tests can reproduce keys; a deployment passes fresh randomness and
 * forgets it.  All four scalars: 6 limbs, Montgomery form, non-zero. */
typedef struct zkhip_keypair zkhip_keypair;
int zkhip_groth16_setup(const zkhip_r1cs_desc* cs, const uint64_t tau[6], const uint64_t alpha[6], const uint64_t beta[6],
                        const uint64_t delta[6], zkhip_keypair** out);
/* the same with the evaluation domain as an argument (see zkhip_r1cs_upload_ex): ZKHIP_DOMAIN_DEFAULT = the forced power of two that
 * libzeth's generate_setup uses (what zkhip_groth16_setup does), ZKHIP_DOMAIN_STEP = libfqfft's unforced choice, else a valid size */
int zkhip_groth16_setup_ex(const zkhip_r1cs_desc* cs, const uint64_t tau[6], const uint64_t alpha[6], const uint64_t beta[6],
                           const uint64_t delta[6], size_t domain_size, zkhip_keypair** out);
/* ONE RANK'S SHARE of the same setup, for a key partitioned over the GPUs of a node (BASELINE configs[3]): the exponents are
 * evaluated on the host as above, the three index ranges are cut into `parts` contiguous slices of equal FINITE terms - the cuts
 * zkhip_key_partition makes on a finished key: an exponent of zero is a base at infinity - and only slice `part` is multiplied out,
 * on the device, straight into the base sets and window tables of *slice_out (a handle like zkhip_crs_upload_slice_ex's; the points
 * never visit the host).  ranges: a_lo, a_hi, h_lo, h_hi, l_lo, l_hi of the slice.  *vk_out (optional): a keypair WITHOUT query
 * vectors - its verification half (zkhip_keypair_vk) and the five constants zkhip_groth16_finish needs (zkhip_keypair_crs_desc).
 * replaces: the same generate_setup call, run by N processes that each keep an N-th of pk. */
int zkhip_groth16_setup_slice(const zkhip_r1cs_desc* cs, const uint64_t tau[6], const uint64_t alpha[6], const uint64_t beta[6],
                              const uint64_t delta[6], size_t domain_size, int parts, int part, const zkhip_key_opts* opts,
                              zkhip_crs** slice_out, size_t ranges[6], zkhip_keypair** vk_out);
/* proving half: pointers into the keypair (valid while it lives) */
int zkhip_keypair_crs_desc(const zkhip_keypair* kp, zkhip_crs_desc* out);
/* verification half: alpha (G1), beta, delta (G2), abc = (n_primary + 1) x 24 limbs; returns n_primary + 1 */
size_t zkhip_keypair_vk(const zkhip_keypair* kp, uint64_t alpha_g1[24], uint64_t beta_g2[24], uint64_t delta_g2[24],
                        const uint64_t** abc);
/* replaces: wsnarkT::keypair_write_bytes / keypair_read_bytes (aggregator_server.cpp:77-94: the server writes the key after
 * the first setup and reads it on later starts).  The reference's byte layout lives in the absent libzeth; this is the library's
 * own container (header, the limb arrays as they cross this ABI, checksum): not interchangeable with a reference key file. */
int zkhip_keypair_write(const zkhip_keypair* kp, const char* path);
int zkhip_keypair_read(const char* path, zkhip_keypair** out);
void zkhip_keypair_free(zkhip_keypair* kp);

/* duration (ms, HIP events on the library's stream) of the dominant kernel of the last MSM */
float zkhip_last_accumulate_ms(void);
/* The entries that launch accumulated: the non-zero digits of its scalars = the mixed additions of k_accumulate (a zero scalar, a zero
 * digit and a base at infinity produce none - a witness of mostly small values has far fewer than terms x digits).  Read back from the
 * sort's histogram: a measurement aid that synchronises the device (bench.py's roofline numerator); 0 before the first launch. */
int zkhip_last_accumulate_entries(uint64_t* out);
/* begin and end of that launch (ms, HIP events) on a per-device time base: a caller that keeps several MSMs in flight
 * (zkhip_msm_submit / collect) can see how their accumulations overlap and take the union of the intervals */
int zkhip_last_accumulate_interval(float out_ms[2]);
/* The time base above is recorded when the library first plans an MSM on a GPU; the values are FLOAT milliseconds since then (1 us
 * of resolution for ~8 s, 0.06 ms after an hour).  A caller that compares intervals records the origin again at the start of its
 * timed region (calling thread's library device). */
int zkhip_reset_time_base(void);
/* The Fq multiplier's own peak on the calling thread's library device, measured NOW: dependent chains of 761-bit Montgomery products
 * at the accumulation kernel's occupancy, chip-wide products per second (~25 ms of device time).  The bound that binds this path is
 * the integer multiplier, and its rate differs between boxes and power states of the same model: a measurement states
 * its fraction against the peak of the run it was taken in. */
int zkhip_measure_fq_mul_rate(double* fq_mul_per_s);
/* The transform kernels of zkhip_ntt ALONE (no conversion from / to the ABI's limbs, no allocation): `batch` (1 .. 3: the QAP map
 * transforms its A, B, C vectors in the same launches) resident vectors of 2^log_d elements, `reps` timed transforms after one
 * untimed, HIP events on the stream the passes run on.  *ms_per_transform = elapsed / (reps x batch).  dir / coset as zkhip_ntt.
 * replaces: nothing - the measurement behind bench.py's ntt roofline (2 d 48 B per transform). */
int zkhip_measure_ntt(unsigned log_d, int dir, int coset, int batch, int reps, double* ms_per_transform);
/* Test hook (no counterpart): the device build's three multiplier bodies - the code every kernel runs (fp29.cuh / fp29_chain.cuh) - on
 * operands given limb by limb: field 0 = Fq (27 limbs of 29 bits), 1 = Fr (14); limbs_in holds n cases of (a, b, c, d), limbs_out
 * receives n x { a b / R, a^2 / R, (a b + c d) / R } in the device's own form (R = 2^783 / 2^406).  Operands must respect the
 * bodies' contract (limbs below 2^29 except the top one, products below 2^10 R p). */
int zkhip_internal_field_selftest(int field, const uint32_t* limbs_in, size_t n, uint32_t* limbs_out);
/* Test hook (no counterpart), HOST ONLY - works without a device: the prover's tail (row a9) on the paths a healthy proof never
 * takes: `rounds` proofs abandoned between the start of the tail's key-only scalar multiplications and their collection, the key's
 * tables freed at once (run under the CPU AddressSanitizer build); a delta of small order (`small_order_g1`, e.g. (1, 0)) whose
 * fixed-base table must be refused in favour of variable-base products.  g1 / g2: the curve generators, 24 limbs each. */
int zkhip_internal_tail_selftest(const uint64_t g1[24], const uint64_t g2[24], const uint64_t small_order_g1[24], int rounds);

/* replaces: libff::Fr<wppT>::random_element() as r1cs_gg_ppzksnark_prover draws the proof's randomisers r, s (reached from
 * aggregator_circuit.tcc:168) and the generator its toxic waste: one field element uniform in Fr, 6 Montgomery limbs, from the
 * operating system's entropy source.  Host code. */
int zkhip_fr_random(uint64_t out[6]);

/* Montgomery limbs -> canonical integer limbs (little-endian), for the JSON / EVM encodings of the reference
 * (SURVEY App. A.2, A.3): which = 0 for Fq (12 limbs), 1 for Fr (6 limbs).  Host code. */
int zkhip_to_canonical(int which, const uint64_t* in, uint64_t* out);

/* ---- the GPUs of one node behind one process (multi_device.cpp: host code on the entry points above, no collective) ----------
 * The reference server is ONE process that owns the prover (aggregator_server.cpp:106-118, 279-348, 390-416; OpenMP is its only
 * parallelism, CMakeLists.txt:80-84); a drop-in on an 8-GPU node must drive all eight from that process.  `devices` lists GPU
 * indices; an index may repeat ("0,0": two contexts on GPU 0 - how a one-GPU box rehearses N > 1).  Both constructors call
 * zkhip_init for the GPUs they use and leave the calling thread's library device as they found it.
 *
 * Replicas (BASELINE configs[4], SURVEY 8e "whole proofs"): one resident copy of the key and one streaming pipeline
 * (zkhip_aggregator_pipeline_*) per list entry; submit hands a batch to the entry with the fewest batches outstanding.
 * replaces: aggregator_circuit::prove as GenerateAggregatedTransaction calls it (aggregator_server.cpp:318-319), on every GPU. */
typedef struct zkhip_dispatcher zkhip_dispatcher;
int zkhip_dispatcher_new(zkhip_aggregator* a, const zkhip_crs_desc* key, const zkhip_key_opts* opts, const int* devices, int n_devices,
                         int gpu_slots, int witness_workers, unsigned flags /* ZKHIP_PIPELINE_* */, zkhip_dispatcher** out);
int zkhip_dispatcher_size(const zkhip_dispatcher* d);                          /* entries of the device list */
int zkhip_dispatcher_submit(zkhip_dispatcher* d, const uint64_t* nested_vk, const uint64_t* nested_proofs, const uint64_t* nested_inputs,
                            const uint64_t r[6], const uint64_t s[6], uint64_t* ticket);
int zkhip_dispatcher_wait(zkhip_dispatcher* d, uint64_t ticket, uint64_t* primary_inputs, uint64_t proof_affine[72]);
int zkhip_dispatcher_register_app(zkhip_dispatcher* d, const uint64_t* nested_vk);    /* zkhip_aggregator_pipeline_register_app on every entry */
int zkhip_dispatcher_stats(const zkhip_dispatcher* d, size_t* submitted_per_entry);   /* batches given to each entry so far */
int zkhip_dispatcher_outstanding(const zkhip_dispatcher* d, size_t* per_entry);        /* submitted and not yet collected, per entry */
void zkhip_dispatcher_free(zkhip_dispatcher* d);
/* One proof over a key PARTITIONED across the list (BASELINE configs[3]): entry k holds the k-th contiguous slice of
 * the A / B, H and L queries (equal finite terms per slice: zkhip_key_partition) and a prover instance on it; prove
 * runs the slices on a host thread each, adds the 5 x 288-byte partial sums in list order on the host (zkhip_jac_add) and finishes
 * once (zkhip_groth16_finish).  Same proof as zkhip_groth16_prove over the whole key.
 * replaces: wsnarkT::generate_proof (aggregator_circuit.tcc:168) for a key that one GPU should not hold or prove alone. */
/* How a key is cut for N devices / ranks: cuts[0 .. parts] (parts + 1 values each) of the A / B queries (one range for the three: the
 * weight of an index is the number of its finite bases among A, B-G2, B-G1), of H and of L - contiguous slices of equal FINITE terms
 * (a base at infinity - a third of a real key's B query - produces no bucket entry).  Host code; zkhip_multi_prover_new and
 * zecale_amd/dist.py (one process per GPU) cut by this rule. */
int zkhip_key_partition(const zkhip_crs_desc* key, int parts, size_t* a_cuts, size_t* h_cuts, size_t* l_cuts);
typedef struct zkhip_multi_prover zkhip_multi_prover;
int zkhip_multi_prover_new(const zkhip_crs_desc* key, const zkhip_r1cs_desc* cs, const zkhip_key_opts* opts, const int* devices, int n_devices,
                           zkhip_multi_prover** out);
int zkhip_multi_prover_size(const zkhip_multi_prover* mp);
int zkhip_multi_prover_prove(zkhip_multi_prover* mp, const uint64_t* z, const uint64_t r[6], const uint64_t s[6], uint64_t proof_affine[72]);
/* ms of the last proof: [0] the slowest slice (upload z + QAP map + five MSMs), [1] the host additions, [2] the host tail */
int zkhip_multi_prover_timings(zkhip_multi_prover* mp, double out_ms[3]);
void zkhip_multi_prover_free(zkhip_multi_prover* mp);

/* host helpers on results (tiny, serial): Jacobian -> affine (infinity -> all zero), a + b */
int zkhip_jac_to_affine(const uint64_t jac[36], uint64_t aff[24]);
int zkhip_jac_add(const uint64_t a[36], const uint64_t b[36], uint64_t out[36]);

#ifdef __cplusplus
}
#endif
#endif /* ZKHIP_H */
