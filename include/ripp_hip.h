/* ripp_hip.h -- C ABI of the MI355X-native inner-pairing-product engine (libripp_hip.so).
 *
 * Drop-in boundary for the data-parallel hot path of arkworks-rs/ripp on BLS12-381.  Each entry point names the
 * reference interface it replaces (paths relative to the reference repository).  A Rust maintainer binds these
 * with `extern "C"` and implements the `InnerProduct` / `DoublyHomomorphicCommitment` traits on top -- see
 * INTEGRATION.md.
 *
 * Data layout (all plain-old-data, no torch / HIP types in any signature):
 *   little-endian u64 limbs in MONTGOMERY form, R = 2^384 (Fp) / 2^256 (Fr) -- the limbs ark-ff 0.4 holds in
 *   `Fp384.0.0` / `Fp256.0.0`, so the shim copies limbs without arithmetic.
 *   affine infinity == (0, 0); Jacobian infinity == Z = 0.
 *
 * Group elements are elements of G1 / G2 PROPER (the prime-order subgroups), which is what ark-ec's `G1Affine` / `G2Affine` values
 * are after `deserialize_*` with validation or as outputs of group arithmetic.  The G2 folds, the MSMs and the table folds use the
 * curve endomorphisms (GLV on G1, psi on G2), which equal the corresponding scalar multiples only on those subgroups; a point that is
 * merely on the curve (`new_unchecked`) gives the reference's value only on the G1 side of the SIPP prover, which uses plain arithmetic throughout.
 * The deserialisers of section "wire format" reject such points like arkworks does.
 *
 * Status codes: 0 ok, 1 message length mismatch (InnerProductError::MessageLengthInvalid,
 * inner_products/src/lib.rs:65-70), 2 length not a power of two (the asserts at sipp/src/lib.rs:48-53),
 * 3 HIP runtime / no device, 4 bad argument.  Nothing throws or aborts.  There is NO CPU fallback: every compute
 * entry point returns 3 when no gfx950 device is usable.
 */
#ifndef RIPP_HIP_H
#define RIPP_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[6]; } ripp_fp;
typedef struct { uint64_t l[4]; } ripp_fr;
typedef struct { ripp_fp c0, c1; } ripp_fp2;
typedef struct { ripp_fp2 c[6]; } ripp_gt;            /* c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2 (Fp12 = Fp6[w], Fp6 = Fp2[v]) */
typedef struct { ripp_fp x, y; } ripp_g1a;
typedef struct { ripp_fp x, y, z; } ripp_g1j;
typedef struct { ripp_fp2 x, y; } ripp_g2a;
typedef struct { ripp_fp2 x, y, z; } ripp_g2j;

enum { RIPP_OK = 0, RIPP_ERR_LENGTH = 1, RIPP_ERR_POW2 = 2, RIPP_ERR_DEVICE = 3, RIPP_ERR_ARG = 4 };

/* per-call phase timings (milliseconds, HIP events on the engine's stream) -- counterpart of the reference's
 * `start_timer!/end_timer!` scopes (ip_proofs/src/gipa.rs:197-291) */
typedef struct {
    double total_ms, upload_ms, scale_ms, miller_lines_ms, miller_products_ms, fold_ms, normalize_ms, host_ms, hash_ms;
    double kernel_miller_lines_ms_sum, kernel_line_products_ms_sum;   /* summed launch durations of the two dominant kernels */
    uint64_t kernel_miller_lines_launches, kernel_line_products_launches;
    uint64_t pairs_lines, pairs_products;                             /* units processed by those launches */
    /* sharded proofs and the hash-window look-ahead (appended in build round 3) */
    double exchange_ms;                                               /* time spent in the per-round all-gathers (incl. waiting for the slowest rank) */
    double look_ms;                                                   /* device + host time of the look-ahead evaluation of rounds 1..k in the hash window */
    uint64_t look_items, look_pairs;                                  /* (round, side) values pre-evaluated; pairs that took */
    /* the statement hash of THIS call (appended in build round 4; 0 on ranks that did not hash): Blake2s itself / waiting for the serialisation workers */
    double statement_hash_ms, statement_hash_wait_ms;
    uint64_t chains_lines;                                            /* G2 chains the carry-free stage-1 kernel walked (<= pairs_lines: products over one Q vector share a chain) */
    /* memory-aware degradation (appended in build round 5): the deepest fall-back the call took -- 0 none, 1 half-vector round-0 tables instead of the
     * three-quarter / eight-multiple ones, 2 pre-doubled bases only, 3 no round-0 precomputation; +8 when the line buffer (pairs per launch) was cut,
     * +16 when an in-round G2 table fold ran in the 4-lane split form for lack of room --
     * and the device memory the library holds at the end of the call */
    uint64_t mem_tier, device_bytes;
} ripp_stats;
/* ABI guard.  The library WRITES sizeof(ripp_stats) bytes through every `ripp_stats*` it is given, and the struct has grown twice: a caller
 * compiled against an older header would be overrun.  Bindings must check at load time that RIPP_ABI_VERSION == ripp_abi_version() and
 * sizeof(their ripp_stats) == ripp_stats_size() (ripp_amd/_lib.py and rust/ripp-hip do). */
#define RIPP_ABI_VERSION 7
int32_t ripp_abi_version(void);
size_t  ripp_stats_size(void);

/* ---- lifecycle.  The traits are static (inner_products/src/lib.rs:40-49: no &self), so the engine is a
 * process-global, lazily created context bound to ONE device (one process per GPU). */
int32_t ripp_init(int32_t device_ordinal);
void    ripp_shutdown(void);
int32_t ripp_device_count(void);
/* ripp_init(other_ordinal) while job / SRS handles of the current device are alive returns RIPP_ERR_ARG (it would free their streams).
 * ripp_release_scratch frees the engine's grow-only scratch (line buffer, fold tables, MSM scratch: ~19 GB after an n = 2^20 proof);
 * the next call re-allocates what it needs. */
int32_t ripp_release_scratch(void);
/* device memory the library holds right now through its own buffers (scratch, tables, job / SRS / vector handles): the quantity
 * ripp_config.mem_cap_bytes bounds and ripp_stats.device_bytes reports at the end of a proof */
size_t  ripp_device_bytes(void);
const char* ripp_last_error(void);   /* message of the calling thread's last failed call (thread-local, errno-style) */

/* ---- configuration (no counterpart in the reference: its tuning knobs are cargo features and rayon's thread count) --------------------------
 * Every choice among implementations of the same function -- which kernels, from which launch size, the hash-window look-ahead plan -- is a
 * member of ripp_config.  Usage: ripp_config c; ripp_config_default(&c); c.tail_pipe_max = 0; ripp_configure(&c);   (process-wide, takes
 * effect from the next call; ripp_configure(NULL) returns to the defaults).  Every setting selects among forms that produce the same bytes.
 * The RIPP_* environment variables of DESIGN.md section 7b remain as a DEBUG override on top (read in one place, once per call). */
typedef struct {
    uint32_t struct_size;            /* sizeof(ripp_config), set by ripp_config_default; ripp_configure rejects any other value */
    /* implementation selectors: non-zero switches the named form OFF (DESIGN.md section 7b has the effect of each) */
    uint32_t no_vm, no_precompute, no_fold_tables, no_msm_glv, lp_one_lane, no_endo, no_fq, no_xscale, scale_no_fq, agg_sequential, look_static, quiet_waits, no_share, no_fuse, fuse_tables;
    int32_t  look_eighths;           /* hash-window look-ahead plan: -1 automatic (cost model / adaptive), 8 k + f = k (round, side) items and f/8 of the next, FORCED */
    int32_t  ranks_per_device;       /* ranks sharing one GPU (test rigs): the look-ahead plan prices the window per device */
    int32_t  msm_c; uint32_t msm_ch, msm_gmin;       /* MSM window width, slot length, grouping threshold; 0 = the plan's own choice */
    uint32_t no_prebuild;            /* (selector, in what was padding) in-round G2 fold tables only after the challenge, not in the host phase before it */
    /* crossover launch sizes between the latency and the throughput forms */
    uint64_t vm_lines_max, vm_fold_max, vm_tree_max, gls_split_max, msm_vm_merge_max, fold_tab_min, fq_min, lp_fq_min, vm_joint_max, vm_scale_max,
             tail_pipe_max, ml_fq_min, fq_min_g1,
             msm_lds_sort_min,       /* MSMs below this many terms use the lane-per-term digit sort instead of the LDS-tile sort (default 0: never) */
             msm_chunk_min;          /* host-slice MSMs from this many G1 bases (half as many G2 bases) run as two halves on two streams (default 2^20) */
    /* appended in build round 5 (ABI version 6) */
    uint64_t mem_cap_bytes;          /* a BUDGET for the device memory the library holds (scratch, tables, jobs, SRS / vector handles: ripp_device_bytes()), consulted where an
                                      * OPTIONAL structure is sized -- a caller's own uploads are never refused because of it; 0 = automatic: what hipMemGetInfo
                                      * reports free, less a margin.  Short of it the engine cuts the line buffer (pairs per launch), then steps the round-0 fold
                                      * tables down (ripp_stats.mem_tier); the proof bytes are the same in every tier */
    uint32_t hot_workers;            /* polling host workers after the digest: 0 automatic (CPUs of the process tree / ranks >= 8), 1 always, 2 never */
    uint32_t no_job_cache;           /* one-shot proofs free their job buffers (~1 KB per element + four pinned row buffers) instead of parking them for the next call */
    /* appended in build round 6 (ABI version 7) */
    uint32_t no_lp_karatsuba;        /* (selector) stage 2 of the pairing product with the six-product sums of k_line_products_q instead of the Karatsuba form k_line_products_k (BLS12-381) */
    uint32_t comm_timeout_ms;        /* deadline of one exchange of the RCCL transport; 0 = 60 000.  On expiry the call returns RIPP_ERR_DEVICE and the communicator is unusable */
    uint32_t plan_derate_pct;        /* the look-ahead planner prices the GPU this many per cent SLOWER than measured (tests of the plan on a slow device; 0 = as measured) */
    uint32_t n_devices;              /* stateless trait calls (pairing products, MSMs on host slices) split over this many devices IN THIS PROCESS; 0 / 1 = the bound device only */
} ripp_config;
int32_t ripp_config_default(ripp_config* cfg);      /* the built-in defaults of this build; needs no device */
int32_t ripp_configure(const ripp_config* cfg);     /* NULL: back to the defaults */
int32_t ripp_config_get(ripp_config* cfg);          /* what the next call will run with: defaults < ripp_configure < environment */

/* ---- L1 trait surface on host slices ------------------------------------------------------------------ */
/* PairingInnerProduct::inner_product(left: &[G1], right: &[G2])  -- inner_products/src/lib.rs:61-73 (cfg_multi_pairing :77-116) */
int32_t ripp_pairing_product_j(const ripp_g1j* left, size_t n_left, const ripp_g2j* right, size_t n_right, ripp_gt* out);
/* sipp::product_of_pairings(a: &[G1Affine], b: &[G2Affine])  -- sipp/src/lib.rs:219-224 */
int32_t ripp_pairing_product_a(const ripp_g1a* a, const ripp_g2a* b, size_t n, ripp_gt* out);
/* sipp::product_of_pairings_with_coeffs(a, b, r)  -- sipp/src/lib.rs:184-217 */
int32_t ripp_pairing_product_coeffs_a(const ripp_g1a* a, const ripp_g2a* b, const ripp_fr* r, size_t n, ripp_gt* out);
/* MultiexponentiationInnerProduct::<G1|G2>::inner_product(left: &[G], right: &[Fr])  -- inner_products/src/lib.rs:128-141 */
int32_t ripp_msm_g1_j(const ripp_g1j* bases, size_t n_left, const ripp_fr* scalars, size_t n_right, ripp_g1j* out);
int32_t ripp_msm_g2_j(const ripp_g2j* bases, size_t n_left, const ripp_fr* scalars, size_t n_right, ripp_g2j* out);
/* VariableBaseMSM::msm on affine bases (sipp/src/lib.rs:174-175) */
int32_t ripp_msm_g1_a(const ripp_g1a* bases, const ripp_fr* scalars, size_t n, ripp_g1j* out);
int32_t ripp_msm_g2_a(const ripp_g2a* bases, const ripp_fr* scalars, size_t n, ripp_g2j* out);
/* In-process multi-device dispatch (ripp_config.n_devices = D > 1, no communicator): the pairing products and MSMs on host slices above cut their index range
 * into D contiguous parts, one per device (bound device + d), each on its own engine and host thread; the host multiplies / adds the D partial results and runs
 * the one final exponentiation.  The value is the single-device call's (an MSM result may be another projective representative of the same point).  Parts below
 * 4 096 pairs / 32 768 terms are not split off.  ripp_device_slots_used: how many parts the LAST such call of this process used (1 = not split).
 * RIPP_VIRTUAL_DEVICES=D in the environment maps all D slots onto the bound device (a one-GPU test rig).  Proofs (SIPP, GIPA, TIPA) shard across PROCESSES. */
int32_t ripp_device_slots_used(void);
/* ScalarInnerProduct::inner_product (inner_products/src/lib.rs:144-166): sum_i l_i * r_i in Fr; RIPP_ERR_LENGTH as above */
int32_t ripp_scalar_inner_product(const ripp_fr* left, size_t nl, const ripp_fr* right, size_t nr, ripp_fr* out);
/* Sharded evaluation (vectors partitioned by index residue, one process per GPU): this rank's share of a
 * pairing product = the Miller value of its pairs BEFORE the final exponentiation.  All-gather the shares (576 B each), multiply
 * them (ripp_combine_partials) and apply ripp_final_exp ONCE: the result equals ripp_pairing_product_j of the whole vectors.  A
 * sharded MSM is ripp_msm_g{1,2}_j per shard + ripp_sum_g{1,2}_j of the gathered points. */
int32_t ripp_pairing_miller_j(const ripp_g1j* left, size_t nl, const ripp_g2j* right, size_t nr, ripp_gt* miller_value);
int32_t ripp_sum_g1_j(const ripp_g1j* pts, size_t n, ripp_g1j* out);
int32_t ripp_sum_g2_j(const ripp_g2j* pts, size_t n, ripp_g2j* out);

/* ---- halving-round fold  out[i] = s * hi[i] + lo[i],  i < half ------------------------------------------ */
/* SIPP form: affine in, batch-normalised affine out  -- sipp/src/lib.rs:87-92 (G1) and :95-100 (G2) */
int32_t ripp_fold_g1_a(const ripp_g1a* hi, const ripp_g1a* lo, size_t half, const ripp_fr* s, ripp_g1a* out);
int32_t ripp_fold_g2_a(const ripp_g2a* hi, const ripp_g2a* lo, size_t half, const ripp_fr* s, ripp_g2a* out);
/* GIPA form: projective in/out through `mul_helper`  -- ip_proofs/src/gipa.rs:262-290, ip_proofs/src/lib.rs:15-19 */
int32_t ripp_fold_g1_j(const ripp_g1j* hi, const ripp_g1j* lo, size_t half, const ripp_fr* s, ripp_g1j* out);
int32_t ripp_fold_g2_j(const ripp_g2j* hi, const ripp_g2j* lo, size_t half, const ripp_fr* s, ripp_g2j* out);
/* the same fold on a scalar vector (GIPA with Message = Fr, e.g. Pedersen messages: gipa.rs:270-274) */
int32_t ripp_fold_fr(const ripp_fr* hi, const ripp_fr* lo, size_t half, const ripp_fr* s, ripp_fr* out);
/* a_i <- r_i * a_i with per-element scalars, normalised  -- sipp/src/lib.rs:61-66 */
int32_t ripp_scale_g1_a(const ripp_g1a* a, const ripp_fr* r, size_t n, ripp_g1a* out);
/* CurveGroup::normalize_batch  -- inner_products/src/lib.rs:80-81,140; sipp/src/lib.rs:66,92,100 */
int32_t ripp_normalize_g1(const ripp_g1j* in, size_t n, ripp_g1a* out);
int32_t ripp_normalize_g2(const ripp_g2j* in, size_t n, ripp_g2a* out);

/* ---- device-resident vectors ---------------------------------------------------
 * The entry points above take host slices, as the reference's traits do (inner_products/src/lib.rs:45-48): each call uploads and
 * normalises its inputs.  A ripp_vec keeps a vector in HBM across calls -- group elements affine (normalised once), scalars in
 * Montgomery form -- so that a caller running the reference's generic GIPA loop (ip_proofs/src/gipa.rs:196-297) uploads each
 * vector once and then works on views: `ripp_vec_halves` is the split of a halving round, `ripp_vec_fold` its fold
 * (gipa.rs:262-291), and the three inner products read the views in place.  Views share the storage; every handle (vector or view)
 * is freed with ripp_vec_free, the storage goes with the last one.  Length mismatches return RIPP_ERR_LENGTH like the slice forms. */
typedef struct ripp_vec ripp_vec;
enum { RIPP_VEC_G1 = 1, RIPP_VEC_G2 = 2, RIPP_VEC_FR = 3 };
int32_t ripp_vec_upload_g1a(const ripp_g1a* p, size_t n, ripp_vec** out);
int32_t ripp_vec_upload_g2a(const ripp_g2a* p, size_t n, ripp_vec** out);
int32_t ripp_vec_upload_g1j(const ripp_g1j* p, size_t n, ripp_vec** out);     /* normalised on the device */
int32_t ripp_vec_upload_g2j(const ripp_g2j* p, size_t n, ripp_vec** out);
int32_t ripp_vec_upload_fr(const ripp_fr* p, size_t n, ripp_vec** out);
size_t  ripp_vec_len(const ripp_vec* v);
int32_t ripp_vec_kind(const ripp_vec* v);
int32_t ripp_vec_slice(const ripp_vec* v, size_t off, size_t len, ripp_vec** view);
int32_t ripp_vec_halves(const ripp_vec* v, ripp_vec** lo, ripp_vec** hi);      /* (v[..len/2], v[len/2..]) */
int32_t ripp_vec_download(const ripp_vec* v, void* out);                       /* ripp_g1a / ripp_g2a / ripp_fr elements by kind */
void    ripp_vec_free(ripp_vec* v);
/* PairingInnerProduct / MultiexponentiationInnerProduct / ScalarInnerProduct on resident vectors (inner_products/src/lib.rs:61-166) */
int32_t ripp_vec_pairing_product(const ripp_vec* left_g1, const ripp_vec* right_g2, ripp_gt* out);
int32_t ripp_vec_msm(const ripp_vec* bases, const ripp_vec* scalars, void* out /* ripp_g1j or ripp_g2j by the bases' kind */);
int32_t ripp_vec_scalar_inner_product(const ripp_vec* left, const ripp_vec* right, ripp_fr* out);
/* out[i] = s * hi[i] + lo[i], a new resident vector (hi and lo of one kind and length; typically the two halves of one vector) */
int32_t ripp_vec_fold(const ripp_vec* hi, const ripp_vec* lo, const ripp_fr* s, ripp_vec** out);

/* ---- SIPP prover  -- SIPP::<Bls12_381, Blake2s>::prove, sipp/src/lib.rs:42-106 ---------------------------- */
/* proof: 2*log2(n) GT elements, (z_l, z_r) per round in round order (Proof::gt_elems, sipp/src/lib.rs:32-34).
 * challenges (optional, may be NULL): log2(n) Fr values x of sipp/src/lib.rs:85.
 * RETENTION: a one-shot call parks its job buffers (~1 KB of device memory per element and four pinned row buffers: ~1 GB after an n = 2^20
 * proof) for the next one-shot call, beside the engine's grow-only scratch (line buffer, fold tables: ~19 GB at n = 2^20).  ripp_release_scratch
 * / ripp_shutdown free both; ripp_config.no_job_cache turns the parking off; ripp_config.mem_cap_bytes bounds the total. */
int32_t ripp_sipp_prove(const ripp_g1a* a, const ripp_g2a* b, const ripp_fr* r, size_t n, const ripp_gt* value,
                        ripp_gt* proof, ripp_fr* challenges, ripp_stats* stats);
/* SIPP::verify, sipp/src/lib.rs:109-180.  *accept = 1/0. */
int32_t ripp_sipp_verify(const ripp_g1a* a, const ripp_g2a* b, const ripp_fr* r, size_t n, const ripp_gt* claimed,
                         const ripp_gt* proof, size_t proof_rounds, int32_t* accept);

/* Device-resident statement for repeated / timed proving (inputs stay in HBM between calls).
 * Sharded proving (one process per GPU): rank `rank` of `world` holds the elements with index = rank (mod world);
 * a[], b[], r[] passed here are that shard in local order (local j <-> global j*world + rank).  Partner elements
 * of every halving round (i, i + len/2) then live on the same GPU for the whole proof. */
typedef struct ripp_sipp_job ripp_sipp_job;
int32_t ripp_sipp_job_create(const ripp_g1a* a, const ripp_g2a* b, const ripp_fr* r, size_t n_local,
                             int32_t rank, int32_t world, ripp_sipp_job** job);
void    ripp_sipp_job_destroy(ripp_sipp_job* job);
/* whole proof on one GPU (world == 1) from the resident statement; the job's working vectors are consumed (ripp_sipp_job_local_len() == 0
 * afterwards: the prover discards the last fold, sipp/src/lib.rs:87-104, so it is not computed); ripp_sipp_job_begin() restarts from the statement */
int32_t ripp_sipp_job_prove(ripp_sipp_job* job, const ripp_gt* value, ripp_gt* proof, ripp_fr* challenges, ripp_stats* stats);
/* staged interface for world > 1 (the caller all-gathers 2 x 576 B per round over RCCL):
 *   begin: re-arm the job from the resident statement (scaling a_i <- r_i a_i of this shard)
 *   round_partials: this shard's two MILLER VALUES (before the final exponentiation) of the current round's
 *                   z_l and z_r.  prod_ranks miller_combine(rows_rank) == miller_combine(prod_ranks rows_rank) because the
 *                   f <- f^2 * L recurrence is multiplicative, so every rank folds its own 68 step products first.
 *                   A Miller value is defined UP TO A FACTOR IN Fp* (the kernels scale line elements freely; the final
 *                   exponentiation removes such factors): only its image under ripp_final_exp is a reference value.
 *   round_finish: given the product over ranks of those values, final-exponentiate to z_l, z_r, derive x, fold the
 *                 local halves.  seed_digest: Blake2s digest of the statement (needed in round 0 only). */
int32_t ripp_sipp_job_begin(ripp_sipp_job* job);
size_t  ripp_sipp_job_rounds_left(const ripp_sipp_job* job);
int32_t ripp_sipp_job_round_partials(ripp_sipp_job* job, ripp_gt* partials /* [2] */);
int32_t ripp_sipp_job_round_finish(ripp_sipp_job* job, const ripp_gt* combined /* [2] */, const uint8_t seed_digest[32],
                                   ripp_gt* z_l, ripp_gt* z_r, ripp_fr* x);
int32_t ripp_sipp_job_stats(const ripp_sipp_job* job, ripp_stats* stats);
/* tail of a sharded proof: once every rank holds ONE element (global length == world) the ranks all-gather the
 * exported elements and each imports the whole remaining vector (global order = rank order); from then on the
 * job behaves as world == 1 (partials are complete products) while keeping its Fiat-Shamir state. */
size_t  ripp_sipp_job_local_len(const ripp_sipp_job* job);
int32_t ripp_sipp_job_export(ripp_sipp_job* job, ripp_g1a* a_out, ripp_g2a* b_out);
int32_t ripp_sipp_job_import(ripp_sipp_job* job, const ripp_g1a* a, const ripp_g2a* b, size_t len);
/* element-wise product over `world` ranks of their [count] partial step-products (what replaces a custom RCCL
 * reduction op: all-gather + local multiply) */
int32_t ripp_combine_partials(const ripp_gt* gathered /* [world][count] */, int32_t world, size_t count, ripp_gt* out /* [count] */);
/* Blake2s digest of (a, b, r, value).serialize_uncompressed for a FULL statement held on the host (sipp/src/lib.rs:56-59) */
int32_t ripp_sipp_seed_digest(const ripp_g1a* a, const ripp_g2a* b, const ripp_fr* r, size_t n, const ripp_gt* value, uint8_t digest[32]);
/* timing of the LAST statement hash of this process: milliseconds spent in Blake2s itself / waiting for the serialisation workers
 * (the sequential hash is the serial floor of a proof; bench.py reports it beside the time the prover was blocked on it) */
void    ripp_statement_hash_times(double* hash_ms, double* wait_ms);

/* ---- multi-GPU: one process per GPU, collectives INSIDE the library ----------------------------------
 * Transport "rccl": librccl.so is loaded on first use; rank 0 creates the 128-byte id, the HOST hands it to the other ranks (it owns
 * the rendezvous: torch.distributed store, MPI, a file), every rank calls ripp_comm_init after ripp_init(its device).
 * Transport "callback": the host supplies the all-gather (recv holds `world` blocks of `bytes` in rank order); returns 0 on success. */
typedef int32_t (*ripp_allgather_fn)(void* user, const void* send, void* recv, size_t bytes);
int32_t ripp_comm_unique_id(uint8_t id[128]);
int32_t ripp_comm_init(const uint8_t id[128], int32_t rank, int32_t world);
int32_t ripp_comm_init_callback(int32_t rank, int32_t world, ripp_allgather_fn allgather, void* user);
void    ripp_comm_destroy(void);
int32_t ripp_comm_rank(void);
int32_t ripp_comm_world(void);
int32_t ripp_comm_allgather(const void* send, void* recv, size_t bytes);     /* host buffers; recv: world x bytes */
/* Transport "replay" (MEASUREMENT rig, no counterpart in the reference): a proof is deterministic, so the blocks a rank receives are too.
 * ripp_comm_record(1) on a live communicator keeps every exchange since the start of the last sharded proof (gathered blocks + this rank's
 * gap = its own time between the previous exchange and its arrival at this one); ripp_comm_recording_save writes them to a file.
 * ripp_comm_init_replay makes THIS process rank `rank` of `world` with the peers' blocks served from such a file: alone on its GPU, at full
 * speed, each all-gather completing no earlier than the slowest peer whose gaps the file holds would have arrived, plus latency_us.  The
 * rank's own blocks and gaps replace the recorded ones (save again to hand them to the next pass).  tools/replay_ranks.py; DESIGN.md section 6. */
int32_t ripp_comm_record(int32_t on);
int32_t ripp_comm_recording_save(const char* path);
int32_t ripp_comm_init_replay(int32_t rank, int32_t world, const char* path, double latency_us);
int32_t ripp_comm_replay_info(uint64_t* served, uint64_t* own_differs, double* waited_ms);
/* the replay's verdict, called after the proofs: own blocks whose BYTES differed from the recording (clock readings of the plan message, Miller values before the
 * final exponentiation, projective representatives of partial MSM sums do, legitimately) are compared by what they MEAN; *mismatches > 0 (and RIPP_ERR_ARG,
 * the first one named in ripp_last_error) says the live run left the recorded protocol -- the measurement is void */
int32_t ripp_comm_replay_check(uint64_t* differing, uint64_t* mismatches);
/* InnerProduct implementations over vectors sharded by index residue (element i on rank i mod world); every rank passes ITS shard
 * and receives the full result.  Same status codes as the single-GPU forms (inner_products/src/lib.rs:61-73, 128-141). */
int32_t ripp_pairing_product_sharded_j(const ripp_g1j* left, size_t nl, const ripp_g2j* right, size_t nr, ripp_gt* out);
int32_t ripp_msm_g1_sharded_j(const ripp_g1j* bases, size_t nl, const ripp_fr* scalars, size_t nr, ripp_g1j* out);
int32_t ripp_msm_g2_sharded_j(const ripp_g2j* bases, size_t nl, const ripp_fr* scalars, size_t nr, ripp_g2j* out);
/* SIPP::prove (sipp/src/lib.rs:42-106) across the communicator: a, b, r = this rank's shard (local j <-> global j * world + rank).
 * Rank 0 passes the full statement (hashed on a host thread while round 0 runs) or its precomputed digest; other ranks pass NULL. */
/* Argument errors (NULL pointers, a shard size that is no power of two, rank 0 with neither statement nor digest) are returned before the first
 * exchange and must be symmetric across ranks; every later failure of one rank travels with its next message and makes ALL ranks return non-zero. */
int32_t ripp_sipp_prove_sharded(const ripp_g1a* a, const ripp_g2a* b, const ripp_fr* r, size_t n_local, const ripp_gt* value,
                                const ripp_g1a* full_a, const ripp_g2a* full_b, const ripp_fr* full_r, const uint8_t* seed_digest,
                                ripp_gt* proof, ripp_fr* challenges, ripp_stats* stats);
/* the same on a resident shard: job from ripp_sipp_job_create(shard, n_local, ripp_comm_rank(), ripp_comm_world()) */
int32_t ripp_sipp_job_prove_sharded(ripp_sipp_job* job, const ripp_gt* value, const ripp_g1a* full_a, const ripp_g2a* full_b, const ripp_fr* full_r,
                                    const uint8_t* seed_digest, ripp_gt* proof, ripp_fr* challenges, ripp_stats* stats);
/* TEST HOOK (no counterpart in the reference): the fold of round `round` on rank `rank` of this process's NEXT SIPP proof reports
 * RIPP_ERR_DEVICE instead of running -- a local failure in the middle of a sharded proof.  One shot; (-1, -1) disarms.  Every rank
 * of the proof must then return non-zero from the same exchange (tests/test_sharded_gloo.py::test_collective_error_exit_*).
 * round + 1000: the process KILLS ITSELF there (SIGKILL) -- a rank that dies in the middle of a proof; the survivors' exchange runs into its deadline. */
void    ripp_test_inject_failure(int32_t rank, int32_t round);

/* ---- GIPA prover, TIPP instantiation  -- GIPA::prove_with_aux / _prove, ip_proofs/src/gipa.rs:162-312 ------------
 * GIPA<PairingInnerProduct, AFGHOCommitmentG1, AFGHOCommitmentG2, IdentityCommitment<GT, Fr>, Blake2b> (the instantiation
 * of the reference's tests, gipa.rs:470-497, and of TIPA).  m_a, ck_b: G1 projective; m_b, ck_a: G2 projective.
 * Outputs in ROUND order (the reference reverses both vectors at gipa.rs:298-299):
 *   com_steps[round][6] = (com_1.0, com_1.1, com_1.2[0], com_2.0, com_2.1, com_2.2[0]),  transcript[round] = c,
 *   base = (m_a[0], m_b[0]) = proof.r_base,  ck_base = (ck_a[0], ck_b[0]) = aux.ck_base  (any projective representative). */
int32_t ripp_gipa_tipp_prove(const ripp_g1j* m_a, const ripp_g2j* m_b, const ripp_g2j* ck_a, const ripp_g1j* ck_b, size_t n,
                             ripp_gt* com_steps, ripp_fr* transcript, ripp_g1j* base_a, ripp_g2j* base_b,
                             ripp_g2j* ck_base_a, ripp_g1j* ck_base_b, ripp_stats* stats);

/* ---- TIPA: GIPA + KZG openings of the final commitment keys -- ip_proofs/src/tipa/mod.rs ---------------------------
 * The SRS (tipa/mod.rs:94-102: g_alpha_powers, h_beta_powers with 2n-1 entries each) is normalised ONCE and stays
 * resident in HBM behind a handle; provers take commitment keys from it (even powers, get_commitment_keys :114-118)
 * and run the two opening MSMs of size 2n-1 against it. */
typedef struct ripp_srs ripp_srs;
int32_t ripp_srs_create(const ripp_g1j* g_alpha_powers, const ripp_g2j* h_beta_powers, size_t num /* = 2n-1 */, ripp_srs** srs);
void    ripp_srs_destroy(ripp_srs* srs);
/* structured_generators_scalar_power (tipa/mod.rs:372-391): out[i] = s^i * generator, on the device */
int32_t ripp_srs_powers_g1(const ripp_fr* s, size_t num, ripp_g1j* out);
int32_t ripp_srs_powers_g2(const ripp_fr* s, size_t num, ripp_g2j* out);
/* SRS::get_commitment_keys (tipa/mod.rs:114-118): ck_1 = even h_beta powers, ck_2 = even g_alpha powers (n each, Z = 1) */
int32_t ripp_srs_commitment_keys(const ripp_srs* srs, ripp_g2j* ck_1, ripp_g1j* ck_2);

/* TIPA::prove_with_srs_shift (tipa/mod.rs:176-231), TIPP instantiation (PairingInnerProductAB, groth16_aggregation.rs:24-31).
 * Outputs: the GIPA proof as for ripp_gipa_tipp_prove (ROUND order), final_ck = aux.ck_base, final_ck_proof = the two
 * KZG openings (prove_commitment_key_kzg_opening, :304-337), and the KZG challenge point for inspection.  n >= 2. */
int32_t ripp_tipa_tipp_prove(const ripp_srs* srs, const ripp_g1j* m_a, const ripp_g2j* m_b, const ripp_g2j* ck_a, const ripp_g1j* ck_b, size_t n,
                             const ripp_fr* r_shift, ripp_gt* com_steps, ripp_fr* transcript, ripp_g1j* base_a, ripp_g2j* base_b,
                             ripp_g2j* final_ck_a, ripp_g1j* final_ck_b, ripp_g2j* opening_a, ripp_g1j* opening_b, ripp_fr* kzg_challenge,
                             ripp_stats* stats);
/* TIPAWithSSM::prove_with_structured_scalar_message (tipa/structured_scalar_message.rs:211-268), MultiExpInnerProductC
 * instantiation (groth16_aggregation.rs:42-48): m_a in G1, m_b in Fr, ck_a in G2.
 *   com_gt[round][2] = (com_1.0, com_2.0)   com_g1[round][2] = (com_1.2[0], com_2.2[0])   (com_x.1 is Fr::zero(), ssm.rs:44-46) */
int32_t ripp_tipa_ssm_prove(const ripp_srs* srs, const ripp_g1j* m_a, const ripp_fr* m_b, const ripp_g2j* ck_a, size_t n,
                            ripp_gt* com_gt, ripp_g1j* com_g1, ripp_fr* transcript, ripp_g1j* base_a, ripp_fr* base_b,
                            ripp_g2j* final_ck_a, ripp_g2j* opening_a, ripp_fr* kzg_challenge, ripp_stats* stats);

/* ---- Groth16 proof aggregation -- aggregate_proofs, ip_proofs/src/applications/groth16_aggregation.rs:77-160 --------
 * AggregateProof (:59-69).  Step arrays are caller-allocated ([rounds = log2 n] entries as sized below) and filled in
 * ROUND order; projective members are any representative of the group element. */
typedef struct {
    ripp_gt com_a, com_b, com_c, ip_ab;
    ripp_g1j agg_c;
    ripp_fr r;                                  /* the Fiat-Shamir combination scalar (:105-116) */
    /* tipa_proof_ab */
    ripp_gt* ab_com_steps;                      /* [rounds][6] */
    ripp_fr* ab_transcript;                     /* [rounds] */
    ripp_g1j ab_base_a; ripp_g2j ab_base_b;
    ripp_g2j ab_final_ck_a; ripp_g1j ab_final_ck_b;
    ripp_g2j ab_opening_a; ripp_g1j ab_opening_b;
    ripp_fr ab_kzg_c;
    /* tipa_proof_c */
    ripp_gt* c_com_gt;                          /* [rounds][2] */
    ripp_g1j* c_com_g1;                         /* [rounds][2] */
    ripp_fr* c_transcript;                      /* [rounds] */
    ripp_g1j c_base_a; ripp_fr c_base_b;
    ripp_g2j c_final_ck_a; ripp_g2j c_opening_a;
    ripp_fr c_kzg_c;
} ripp_aggregate_proof;
/* a, b, c: the (A, B, C) members of n Groth16 proofs (affine, as ark_groth16::Proof stores them); n a power of two >= 2 and
 * srs built for the same n.  The reference's assert_eq!(com_a, IP(a_r, ck_1_r)) (:133-136) holds by construction here: the engine
 * never materialises ck_1_r (bilinearity: IP(a_r, ck_1_r) IS IP(a, ck_1) = com_a), see DESIGN.md section 4b. */
int32_t ripp_aggregate_proofs(const ripp_srs* srs, const ripp_g1a* a, const ripp_g2a* b, const ripp_g1a* c, size_t n,
                              ripp_aggregate_proof* out, ripp_stats* stats);

/* the same across the communicator's ranks (ripp_comm_*): a, b, c = this rank's shard of the n_local * world proofs (local j <->
 * global j * world + rank), srs created for the GLOBAL n on every rank (the SRS is resident in full, the work is split); per round
 * the ranks all-gather their 6 (TIPP) resp. 2 + 2 (SSM) partial values, the KZG openings are per-rank MSMs over a residue class
 * of the powers.  Every rank receives the whole AggregateProof. */
int32_t ripp_aggregate_proofs_sharded(const ripp_srs* srs, const ripp_g1a* a, const ripp_g2a* b, const ripp_g1a* c, size_t n_local,
                                      ripp_aggregate_proof* out, ripp_stats* stats);
/* GIPA::prove_with_aux, TIPP instantiation, on sharded vectors (outputs as ripp_gipa_tipp_prove, replicated) */
int32_t ripp_gipa_tipp_prove_sharded(const ripp_g1j* m_a, const ripp_g2j* m_b, const ripp_g2j* ck_a, const ripp_g1j* ck_b, size_t n_local,
                                     ripp_gt* com_steps, ripp_fr* transcript, ripp_g1j* base_a, ripp_g2j* base_b,
                                     ripp_g2j* ck_base_a, ripp_g1j* ck_base_b, ripp_stats* stats);

/* ---- verifiers: O(log n) GT / group exponentiations on the host, pairings and MSMs on the device --
 * Every function writes *accept = 1 / 0 and returns RIPP_OK when the check itself ran. */
/* VerifierSRS (tipa/mod.rs:104-110, get_verifier_key :120-127) */
typedef struct { ripp_g1j g; ripp_g2j h; ripp_g1j g_beta; ripp_g2j h_alpha; } ripp_verifier_srs;
/* ark_groth16::VerifyingKey as read by verify_aggregate_proof (groth16_aggregation.rs:211-227) */
typedef struct { ripp_g1a alpha_g1; ripp_g2a beta_g2, gamma_g2, delta_g2; const ripp_g1a* gamma_abc_g1; size_t gamma_abc_len; } ripp_groth16_vk;
/* GIPA::verify (gipa.rs:135-160), TIPP instantiation: com = (com_a, com_b, com_t[0]); steps in ROUND order.  The final keys are two
 * n-term MSMs on the device (the reference folds them sequentially, gipa.rs:383-396 with a TODO to use an MSM). */
int32_t ripp_gipa_tipp_verify(const ripp_g2j* ck_a, const ripp_g1j* ck_b, size_t n, const ripp_gt com[3], const ripp_gt* com_steps, size_t rounds,
                              const ripp_g1j* base_a, const ripp_g2j* base_b, int32_t* accept);
/* TIPA::verify_with_srs_shift (tipa/mod.rs:242-301) */
int32_t ripp_tipa_tipp_verify(const ripp_verifier_srs* v_srs, const ripp_gt com[3], const ripp_gt* com_steps, size_t rounds,
                              const ripp_g1j* base_a, const ripp_g2j* base_b, const ripp_g2j* final_ck_a, const ripp_g1j* final_ck_b,
                              const ripp_g2j* opening_a, const ripp_g1j* opening_b, const ripp_fr* r_shift, int32_t* accept);
/* TIPAWithSSM::verify_with_structured_scalar_message (structured_scalar_message.rs:270-331); com = (com_a in GT, com_t in G1) */
int32_t ripp_tipa_ssm_verify(const ripp_verifier_srs* v_srs, const ripp_gt* com_a, const ripp_g1j* com_t, const ripp_fr* scalar_b,
                             const ripp_gt* com_gt, const ripp_g1j* com_g1, size_t rounds, const ripp_g1j* base_a,
                             const ripp_g2j* final_ck_a, const ripp_g2j* opening_a, int32_t* accept);
/* verify_aggregate_proof (groth16_aggregation.rs:162-231); public_inputs[n][m] row-major, vk->gamma_abc_len == m + 1 */
int32_t ripp_verify_aggregate_proof(const ripp_verifier_srs* v_srs, const ripp_groth16_vk* vk, const ripp_fr* public_inputs, size_t n, size_t m,
                                    const ripp_aggregate_proof* proof, int32_t* accept);

/* ---- wire format: ark-serialize 0.4 images of the proof structs, host only ---------------------
 * compress = 0: `serialize_uncompressed`, 1: `serialize_compressed`.  Steps are passed in ROUND order and written reversed, as
 * GIPAProof stores them (gipa.rs:298-299).  Serialisers return the image size and write it when cap is large enough (out may be
 * NULL to query); deserialisers validate like `deserialize_*` with Validate::Yes (range, curve equation, prime-order subgroup) and
 * return RIPP_ERR_ARG on a malformed image.
 *   GIPAProof (gipa.rs:24-51): pass final_ck_a == NULL.   TIPAProof (tipa/mod.rs:41-65): all members. */
size_t  ripp_ser_tipa_tipp_proof(const ripp_gt* com_steps, size_t rounds, const ripp_g1j* base_a, const ripp_g2j* base_b,
                                 const ripp_g2j* final_ck_a, const ripp_g1j* final_ck_b, const ripp_g2j* opening_a, const ripp_g1j* opening_b,
                                 int32_t compress, uint8_t* out, size_t cap);
int32_t ripp_de_tipa_tipp_proof(const uint8_t* in, size_t len, int32_t compress, int32_t with_tipa, size_t max_rounds, size_t* rounds,
                                ripp_gt* com_steps, ripp_g1j* base_a, ripp_g2j* base_b,
                                ripp_g2j* final_ck_a, ripp_g1j* final_ck_b, ripp_g2j* opening_a, ripp_g1j* opening_b);
/* TIPAWithSSMProof (tipa/structured_scalar_message.rs:138-156) */
size_t  ripp_ser_tipa_ssm_proof(const ripp_gt* com_gt, const ripp_g1j* com_g1, size_t rounds, const ripp_g1j* base_a, const ripp_fr* base_b,
                                const ripp_g2j* final_ck_a, const ripp_g2j* opening_a, int32_t compress, uint8_t* out, size_t cap);
int32_t ripp_de_tipa_ssm_proof(const uint8_t* in, size_t len, int32_t compress, size_t max_rounds, size_t* rounds,
                               ripp_gt* com_gt, ripp_g1j* com_g1, ripp_g1j* base_a, ripp_fr* base_b, ripp_g2j* final_ck_a, ripp_g2j* opening_a);
/* single group elements, compressed (48 / 96 bytes) */
size_t  ripp_ser_g1_compressed(const ripp_g1a* p, uint8_t out[48]);
size_t  ripp_ser_g2_compressed(const ripp_g2a* p, uint8_t out[96]);

/* ---- host-side helpers (no device needed): what the host code around the kernels computes ---------------- */
/* BLAKE2s-256 (RFC 7693; `blake2::Blake2s` of sipp/src/lib.rs:230) of a host buffer: the implementation the statement hash runs, exposed for tests */
int32_t ripp_blake2s(const uint8_t* in, size_t len, uint8_t out[32]);
int32_t ripp_final_exp(const ripp_gt* miller_value, ripp_gt* out);                 /* Pairing::final_exponentiation */
int32_t ripp_miller_combine(const ripp_gt* step_products /* [68] */, ripp_gt* out); /* stage (3) of pairing.hpp */
/* out[k] = ripp_final_exp(ripp_miller_combine(step_products + 68 k)), k < count, the way the provers compute it between two kernels: the 63 bits
 * of every product cut into `parts` ranges (0 = the library's default) that run on the library's host workers and are joined with cyclotomic
 * squarings -- the final exponentiation is a homomorphism, so the value is the same field element. */
int32_t ripp_pairing_values(const ripp_gt* step_products /* [count][68] */, int32_t count, int32_t parts, ripp_gt* out /* [count] */);
int32_t ripp_gt_mul(const ripp_gt* a, const ripp_gt* b, ripp_gt* out);
int32_t ripp_gt_pow(const ripp_gt* a, const ripp_fr* k, ripp_gt* out);
int32_t ripp_fr_inverse(const ripp_fr* a, ripp_fr* out);
/* ark-serialize 0.4 `serialize_uncompressed` byte images (what Fiat-Shamir hashes): */
size_t  ripp_ser_gt(const ripp_gt* f, uint8_t out[576]);
size_t  ripp_ser_g1(const ripp_g1a* p, uint8_t out[96]);
size_t  ripp_ser_g2(const ripp_g2a* p, uint8_t out[192]);
size_t  ripp_ser_fr(const ripp_fr* s, uint8_t out[32]);
/* FiatShamirRng<Blake2s> step of sipp/src/lib.rs:80-85: absorb (z_l, z_r) into `seed` (in/out), draw x */
int32_t ripp_sipp_challenge(uint8_t seed[32], const ripp_gt* z_l, const ripp_gt* z_r, ripp_fr* x);

/* ---- synthetic inputs for benchmarks: generated ON THE DEVICE ---------------------- */
/* out[i] = (start + first + i*stride) * G  (affine); scalars from SplitMix64(seed) skipping to element first + i*stride */
int32_t ripp_synth_g1(uint64_t start, size_t first, size_t stride, size_t n, ripp_g1a* out);
int32_t ripp_synth_g2(uint64_t start, size_t first, size_t stride, size_t n, ripp_g2a* out);
int32_t ripp_synth_fr(uint64_t seed, size_t first, size_t stride, size_t n, ripp_fr* out);

#ifdef __cplusplus
}
#endif
#endif
