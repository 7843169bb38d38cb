//! `extern "C"` declarations of `include/ripp_hip.h` (one to one; see that header for the reference interface each entry replaces).
//! The `extern` block is GENERATED from the header (tools/gen_rust_ffi.py) and checked against it by tests/test_rust_ffi_signatures_cpu.py.
//! All structs are plain `repr(C)` limb arrays: little-endian u64 limbs in MONTGOMERY form -- exactly `Fp.0 .0` of ark-ff 0.4.
#![allow(non_camel_case_types)]
use core::ffi::c_void;

#[repr(C)] #[derive(Copy, Clone, Default)] pub struct RippFp { pub l: [u64; 6] }
#[repr(C)] #[derive(Copy, Clone, Default)] pub struct RippFr { pub l: [u64; 4] }
#[repr(C)] #[derive(Copy, Clone, Default)] pub struct RippFp2 { pub c0: RippFp, pub c1: RippFp }
/// c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2  (Fq12 = Fq6[w], Fq6 = Fq2[v])
#[repr(C)] #[derive(Copy, Clone, Default)] pub struct RippGt { pub c: [RippFp2; 6] }
#[repr(C)] #[derive(Copy, Clone, Default)] pub struct RippG1A { pub x: RippFp, pub y: RippFp }
#[repr(C)] #[derive(Copy, Clone, Default)] pub struct RippG1J { pub x: RippFp, pub y: RippFp, pub z: RippFp }
#[repr(C)] #[derive(Copy, Clone, Default)] pub struct RippG2A { pub x: RippFp2, pub y: RippFp2 }
#[repr(C)] #[derive(Copy, Clone, Default)] pub struct RippG2J { pub x: RippFp2, pub y: RippFp2, pub z: RippFp2 }

/// opaque handles (`typedef struct ripp_vec ripp_vec;` etc.): only ever used behind pointers
#[repr(C)] pub struct RippVec { _private: [u8; 0] }
#[repr(C)] pub struct RippSippJob { _private: [u8; 0] }
#[repr(C)] pub struct RippSrs { _private: [u8; 0] }
/// `ripp_allgather_fn`: host-supplied all-gather of the "callback" transport (recv holds `world` blocks of `bytes` in rank order; 0 = ok)
pub type RippAllgatherFn = Option<unsafe extern "C" fn(user: *mut c_void, send: *const c_void, recv: *mut c_void, bytes: usize) -> i32>;
/// `ripp_aggregate_proof` (AggregateProof, groth16_aggregation.rs:59-69): step arrays are caller-allocated, filled in ROUND order
#[repr(C)] #[derive(Copy, Clone)]
pub struct RippAggregateProof {
    pub com_a: RippGt, pub com_b: RippGt, pub com_c: RippGt, pub ip_ab: RippGt,
    pub agg_c: RippG1J,
    pub r: RippFr,
    pub ab_com_steps: *mut RippGt,
    pub ab_transcript: *mut RippFr,
    pub ab_base_a: RippG1J, pub ab_base_b: RippG2J,
    pub ab_final_ck_a: RippG2J, pub ab_final_ck_b: RippG1J,
    pub ab_opening_a: RippG2J, pub ab_opening_b: RippG1J,
    pub ab_kzg_c: RippFr,
    pub c_com_gt: *mut RippGt,
    pub c_com_g1: *mut RippG1J,
    pub c_transcript: *mut RippFr,
    pub c_base_a: RippG1J, pub c_base_b: RippFr,
    pub c_final_ck_a: RippG2J, pub c_opening_a: RippG2J,
    pub c_kzg_c: RippFr,
}
/// `ripp_verifier_srs` (VerifierSRS, tipa/mod.rs:104-110)
#[repr(C)] #[derive(Copy, Clone, Default)]
pub struct RippVerifierSrs { pub g: RippG1J, pub h: RippG2J, pub g_beta: RippG1J, pub h_alpha: RippG2J }
/// `ripp_groth16_vk`: the members of ark_groth16::VerifyingKey that verify_aggregate_proof reads (groth16_aggregation.rs:162-231)
#[repr(C)] #[derive(Copy, Clone)]
pub struct RippGroth16Vk { pub alpha_g1: RippG1A, pub beta_g2: RippG2A, pub gamma_g2: RippG2A, pub delta_g2: RippG2A, pub gamma_abc_g1: *const RippG1A, pub gamma_abc_len: usize }

pub const RIPP_OK: i32 = 0;
pub const RIPP_ERR_LENGTH: i32 = 1;
pub const RIPP_ERR_POW2: i32 = 2;
pub const RIPP_ERR_DEVICE: i32 = 3;
pub const RIPP_ERR_ARG: i32 = 4;
/// `ripp_vec_kind`: what a device-resident vector holds (include/ripp_hip.h: RIPP_VEC_*); `ripp_vec_download` writes ripp_g1a / ripp_g2a / ripp_fr elements accordingly
pub const RIPP_VEC_G1: i32 = 1;
pub const RIPP_VEC_G2: i32 = 2;
pub const RIPP_VEC_FR: i32 = 3;

#[repr(C)] #[derive(Copy, Clone, Default, Debug)]
pub struct RippStats {
    pub total_ms: f64, pub upload_ms: f64, pub scale_ms: f64, pub miller_lines_ms: f64, pub miller_products_ms: f64,
    pub fold_ms: f64, pub normalize_ms: f64, pub host_ms: f64, pub hash_ms: f64,
    pub kernel_miller_lines_ms_sum: f64, pub kernel_line_products_ms_sum: f64,
    pub kernel_miller_lines_launches: u64, pub kernel_line_products_launches: u64,
    pub pairs_lines: u64, pub pairs_products: u64,
    pub exchange_ms: f64, pub look_ms: f64, pub look_items: u64, pub look_pairs: u64,
    pub statement_hash_ms: f64, pub statement_hash_wait_ms: f64,
    pub chains_lines: u64,
    pub mem_tier: u64, pub device_bytes: u64,
}
/// `ripp_config` of include/ripp_hip.h: every choice among implementations of the same function (fill with `ripp_config_default` first).
#[repr(C)] #[derive(Copy, Clone, Default, Debug)]
pub struct RippConfig {
    pub struct_size: u32,
    pub no_vm: u32, pub no_precompute: u32, pub no_fold_tables: u32, pub no_msm_glv: u32, pub lp_one_lane: u32, pub no_endo: u32, pub no_fq: u32,
    pub no_xscale: u32, pub scale_no_fq: u32, pub agg_sequential: u32, pub look_static: u32, pub quiet_waits: u32, pub no_share: u32, pub no_fuse: u32, pub fuse_tables: u32,
    pub look_eighths: i32, pub ranks_per_device: i32, pub msm_c: i32, pub msm_ch: u32, pub msm_gmin: u32, pub no_prebuild: u32,
    pub vm_lines_max: u64, pub vm_fold_max: u64, pub vm_tree_max: u64, pub gls_split_max: u64, pub msm_vm_merge_max: u64, pub fold_tab_min: u64,
    pub fq_min: u64, pub lp_fq_min: u64, pub vm_joint_max: u64, pub vm_scale_max: u64, pub tail_pipe_max: u64, pub ml_fq_min: u64, pub fq_min_g1: u64, pub msm_lds_sort_min: u64, pub msm_chunk_min: u64,
    pub mem_cap_bytes: u64, pub hot_workers: u32, pub no_job_cache: u32,
    pub no_lp_karatsuba: u32, pub comm_timeout_ms: u32, pub plan_derate_pct: u32, pub n_devices: u32,
}
/// `RIPP_ABI_VERSION` of include/ripp_hip.h this binding was written against; `abi_check()` compares it (and the size of `RippStats`, which
/// the library writes in full through every stats pointer) with the loaded library.
pub const RIPP_ABI_VERSION: i32 = 7;
#[cfg(feature = "ffi")]
pub fn abi_check() -> bool { unsafe { ripp_abi_version() == RIPP_ABI_VERSION && ripp_stats_size() == core::mem::size_of::<RippStats>() } }

#[cfg(feature = "ffi")]
extern "C" {
    // ---- GENERATED from include/ripp_hip.h by tools/gen_rust_ffi.py: do not edit by hand ----
    pub fn ripp_abi_version() -> i32;
    pub fn ripp_stats_size() -> usize;
    pub fn ripp_init(device_ordinal: i32) -> i32;
    pub fn ripp_shutdown();
    pub fn ripp_device_count() -> i32;
    pub fn ripp_release_scratch() -> i32;
    pub fn ripp_device_bytes() -> usize;
    pub fn ripp_last_error() -> *const core::ffi::c_char;
    pub fn ripp_config_default(cfg: *mut RippConfig) -> i32;
    pub fn ripp_configure(cfg: *const RippConfig) -> i32;
    pub fn ripp_config_get(cfg: *mut RippConfig) -> i32;
    pub fn ripp_pairing_product_j(left: *const RippG1J, n_left: usize, right: *const RippG2J, n_right: usize, out: *mut RippGt) -> i32;
    pub fn ripp_pairing_product_a(a: *const RippG1A, b: *const RippG2A, n: usize, out: *mut RippGt) -> i32;
    pub fn ripp_pairing_product_coeffs_a(a: *const RippG1A, b: *const RippG2A, r: *const RippFr, n: usize, out: *mut RippGt) -> i32;
    pub fn ripp_msm_g1_j(bases: *const RippG1J, n_left: usize, scalars: *const RippFr, n_right: usize, out: *mut RippG1J) -> i32;
    pub fn ripp_msm_g2_j(bases: *const RippG2J, n_left: usize, scalars: *const RippFr, n_right: usize, out: *mut RippG2J) -> i32;
    pub fn ripp_msm_g1_a(bases: *const RippG1A, scalars: *const RippFr, n: usize, out: *mut RippG1J) -> i32;
    pub fn ripp_msm_g2_a(bases: *const RippG2A, scalars: *const RippFr, n: usize, out: *mut RippG2J) -> i32;
    pub fn ripp_device_slots_used() -> i32;
    pub fn ripp_scalar_inner_product(left: *const RippFr, nl: usize, right: *const RippFr, nr: usize, out: *mut RippFr) -> i32;
    pub fn ripp_pairing_miller_j(left: *const RippG1J, nl: usize, right: *const RippG2J, nr: usize, miller_value: *mut RippGt) -> i32;
    pub fn ripp_sum_g1_j(pts: *const RippG1J, n: usize, out: *mut RippG1J) -> i32;
    pub fn ripp_sum_g2_j(pts: *const RippG2J, n: usize, out: *mut RippG2J) -> i32;
    pub fn ripp_fold_g1_a(hi: *const RippG1A, lo: *const RippG1A, half: usize, s: *const RippFr, out: *mut RippG1A) -> i32;
    pub fn ripp_fold_g2_a(hi: *const RippG2A, lo: *const RippG2A, half: usize, s: *const RippFr, out: *mut RippG2A) -> i32;
    pub fn ripp_fold_g1_j(hi: *const RippG1J, lo: *const RippG1J, half: usize, s: *const RippFr, out: *mut RippG1J) -> i32;
    pub fn ripp_fold_g2_j(hi: *const RippG2J, lo: *const RippG2J, half: usize, s: *const RippFr, out: *mut RippG2J) -> i32;
    pub fn ripp_fold_fr(hi: *const RippFr, lo: *const RippFr, half: usize, s: *const RippFr, out: *mut RippFr) -> i32;
    pub fn ripp_scale_g1_a(a: *const RippG1A, r: *const RippFr, n: usize, out: *mut RippG1A) -> i32;
    pub fn ripp_normalize_g1(in_: *const RippG1J, n: usize, out: *mut RippG1A) -> i32;
    pub fn ripp_normalize_g2(in_: *const RippG2J, n: usize, out: *mut RippG2A) -> i32;
    pub fn ripp_vec_upload_g1a(p: *const RippG1A, n: usize, out: *mut *mut RippVec) -> i32;
    pub fn ripp_vec_upload_g2a(p: *const RippG2A, n: usize, out: *mut *mut RippVec) -> i32;
    pub fn ripp_vec_upload_g1j(p: *const RippG1J, n: usize, out: *mut *mut RippVec) -> i32;
    pub fn ripp_vec_upload_g2j(p: *const RippG2J, n: usize, out: *mut *mut RippVec) -> i32;
    pub fn ripp_vec_upload_fr(p: *const RippFr, n: usize, out: *mut *mut RippVec) -> i32;
    pub fn ripp_vec_len(v: *const RippVec) -> usize;
    pub fn ripp_vec_kind(v: *const RippVec) -> i32;
    pub fn ripp_vec_slice(v: *const RippVec, off: usize, len: usize, view: *mut *mut RippVec) -> i32;
    pub fn ripp_vec_halves(v: *const RippVec, lo: *mut *mut RippVec, hi: *mut *mut RippVec) -> i32;
    pub fn ripp_vec_download(v: *const RippVec, out: *mut c_void) -> i32;
    pub fn ripp_vec_free(v: *mut RippVec);
    pub fn ripp_vec_pairing_product(left_g1: *const RippVec, right_g2: *const RippVec, out: *mut RippGt) -> i32;
    pub fn ripp_vec_msm(bases: *const RippVec, scalars: *const RippVec, out: *mut c_void) -> i32;
    pub fn ripp_vec_scalar_inner_product(left: *const RippVec, right: *const RippVec, out: *mut RippFr) -> i32;
    pub fn ripp_vec_fold(hi: *const RippVec, lo: *const RippVec, s: *const RippFr, out: *mut *mut RippVec) -> i32;
    pub fn ripp_sipp_prove(a: *const RippG1A, b: *const RippG2A, r: *const RippFr, n: usize, value: *const RippGt, proof: *mut RippGt, challenges: *mut RippFr, stats: *mut RippStats) -> i32;
    pub fn ripp_sipp_verify(a: *const RippG1A, b: *const RippG2A, r: *const RippFr, n: usize, claimed: *const RippGt, proof: *const RippGt, proof_rounds: usize, accept: *mut i32) -> i32;
    pub fn ripp_sipp_job_create(a: *const RippG1A, b: *const RippG2A, r: *const RippFr, n_local: usize, rank: i32, world: i32, job: *mut *mut RippSippJob) -> i32;
    pub fn ripp_sipp_job_destroy(job: *mut RippSippJob);
    pub fn ripp_sipp_job_prove(job: *mut RippSippJob, value: *const RippGt, proof: *mut RippGt, challenges: *mut RippFr, stats: *mut RippStats) -> i32;
    pub fn ripp_sipp_job_begin(job: *mut RippSippJob) -> i32;
    pub fn ripp_sipp_job_rounds_left(job: *const RippSippJob) -> usize;
    pub fn ripp_sipp_job_round_partials(job: *mut RippSippJob, partials: *mut RippGt) -> i32;
    pub fn ripp_sipp_job_round_finish(job: *mut RippSippJob, combined: *const RippGt, seed_digest: *const u8, z_l: *mut RippGt, z_r: *mut RippGt, x: *mut RippFr) -> i32;
    pub fn ripp_sipp_job_stats(job: *const RippSippJob, stats: *mut RippStats) -> i32;
    pub fn ripp_sipp_job_local_len(job: *const RippSippJob) -> usize;
    pub fn ripp_sipp_job_export(job: *mut RippSippJob, a_out: *mut RippG1A, b_out: *mut RippG2A) -> i32;
    pub fn ripp_sipp_job_import(job: *mut RippSippJob, a: *const RippG1A, b: *const RippG2A, len: usize) -> i32;
    pub fn ripp_combine_partials(gathered: *const RippGt, world: i32, count: usize, out: *mut RippGt) -> i32;
    pub fn ripp_sipp_seed_digest(a: *const RippG1A, b: *const RippG2A, r: *const RippFr, n: usize, value: *const RippGt, digest: *mut u8) -> i32;
    pub fn ripp_statement_hash_times(hash_ms: *mut f64, wait_ms: *mut f64);
    pub fn ripp_comm_unique_id(id: *mut u8) -> i32;
    pub fn ripp_comm_init(id: *const u8, rank: i32, world: i32) -> i32;
    pub fn ripp_comm_init_callback(rank: i32, world: i32, allgather: RippAllgatherFn, user: *mut c_void) -> i32;
    pub fn ripp_comm_destroy();
    pub fn ripp_comm_rank() -> i32;
    pub fn ripp_comm_world() -> i32;
    pub fn ripp_comm_allgather(send: *const c_void, recv: *mut c_void, bytes: usize) -> i32;
    pub fn ripp_comm_record(on: i32) -> i32;
    pub fn ripp_comm_recording_save(path: *const core::ffi::c_char) -> i32;
    pub fn ripp_comm_init_replay(rank: i32, world: i32, path: *const core::ffi::c_char, latency_us: f64) -> i32;
    pub fn ripp_comm_replay_info(served: *mut u64, own_differs: *mut u64, waited_ms: *mut f64) -> i32;
    pub fn ripp_comm_replay_check(differing: *mut u64, mismatches: *mut u64) -> i32;
    pub fn ripp_pairing_product_sharded_j(left: *const RippG1J, nl: usize, right: *const RippG2J, nr: usize, out: *mut RippGt) -> i32;
    pub fn ripp_msm_g1_sharded_j(bases: *const RippG1J, nl: usize, scalars: *const RippFr, nr: usize, out: *mut RippG1J) -> i32;
    pub fn ripp_msm_g2_sharded_j(bases: *const RippG2J, nl: usize, scalars: *const RippFr, nr: usize, out: *mut RippG2J) -> i32;
    pub fn ripp_sipp_prove_sharded(a: *const RippG1A, b: *const RippG2A, r: *const RippFr, n_local: usize, value: *const RippGt, full_a: *const RippG1A, full_b: *const RippG2A, full_r: *const RippFr, seed_digest: *const u8, proof: *mut RippGt, challenges: *mut RippFr, stats: *mut RippStats) -> i32;
    pub fn ripp_sipp_job_prove_sharded(job: *mut RippSippJob, value: *const RippGt, full_a: *const RippG1A, full_b: *const RippG2A, full_r: *const RippFr, seed_digest: *const u8, proof: *mut RippGt, challenges: *mut RippFr, stats: *mut RippStats) -> i32;
    pub fn ripp_test_inject_failure(rank: i32, round: i32);
    pub fn ripp_gipa_tipp_prove(m_a: *const RippG1J, m_b: *const RippG2J, ck_a: *const RippG2J, ck_b: *const RippG1J, n: usize, com_steps: *mut RippGt, transcript: *mut RippFr, base_a: *mut RippG1J, base_b: *mut RippG2J, ck_base_a: *mut RippG2J, ck_base_b: *mut RippG1J, stats: *mut RippStats) -> i32;
    pub fn ripp_srs_create(g_alpha_powers: *const RippG1J, h_beta_powers: *const RippG2J, num: usize, srs: *mut *mut RippSrs) -> i32;
    pub fn ripp_srs_destroy(srs: *mut RippSrs);
    pub fn ripp_srs_powers_g1(s: *const RippFr, num: usize, out: *mut RippG1J) -> i32;
    pub fn ripp_srs_powers_g2(s: *const RippFr, num: usize, out: *mut RippG2J) -> i32;
    pub fn ripp_srs_commitment_keys(srs: *const RippSrs, ck_1: *mut RippG2J, ck_2: *mut RippG1J) -> i32;
    pub fn ripp_tipa_tipp_prove(srs: *const RippSrs, m_a: *const RippG1J, m_b: *const RippG2J, ck_a: *const RippG2J, ck_b: *const RippG1J, n: usize, r_shift: *const RippFr, com_steps: *mut RippGt, transcript: *mut RippFr, base_a: *mut RippG1J, base_b: *mut RippG2J, final_ck_a: *mut RippG2J, final_ck_b: *mut RippG1J, opening_a: *mut RippG2J, opening_b: *mut RippG1J, kzg_challenge: *mut RippFr, stats: *mut RippStats) -> i32;
    pub fn ripp_tipa_ssm_prove(srs: *const RippSrs, m_a: *const RippG1J, m_b: *const RippFr, ck_a: *const RippG2J, n: usize, com_gt: *mut RippGt, com_g1: *mut RippG1J, transcript: *mut RippFr, base_a: *mut RippG1J, base_b: *mut RippFr, final_ck_a: *mut RippG2J, opening_a: *mut RippG2J, kzg_challenge: *mut RippFr, stats: *mut RippStats) -> i32;
    pub fn ripp_aggregate_proofs(srs: *const RippSrs, a: *const RippG1A, b: *const RippG2A, c: *const RippG1A, n: usize, out: *mut RippAggregateProof, stats: *mut RippStats) -> i32;
    pub fn ripp_aggregate_proofs_sharded(srs: *const RippSrs, a: *const RippG1A, b: *const RippG2A, c: *const RippG1A, n_local: usize, out: *mut RippAggregateProof, stats: *mut RippStats) -> i32;
    pub fn ripp_gipa_tipp_prove_sharded(m_a: *const RippG1J, m_b: *const RippG2J, ck_a: *const RippG2J, ck_b: *const RippG1J, n_local: usize, com_steps: *mut RippGt, transcript: *mut RippFr, base_a: *mut RippG1J, base_b: *mut RippG2J, ck_base_a: *mut RippG2J, ck_base_b: *mut RippG1J, stats: *mut RippStats) -> i32;
    pub fn ripp_gipa_tipp_verify(ck_a: *const RippG2J, ck_b: *const RippG1J, n: usize, com: *const RippGt, com_steps: *const RippGt, rounds: usize, base_a: *const RippG1J, base_b: *const RippG2J, accept: *mut i32) -> i32;
    pub fn ripp_tipa_tipp_verify(v_srs: *const RippVerifierSrs, com: *const RippGt, com_steps: *const RippGt, rounds: usize, base_a: *const RippG1J, base_b: *const RippG2J, final_ck_a: *const RippG2J, final_ck_b: *const RippG1J, opening_a: *const RippG2J, opening_b: *const RippG1J, r_shift: *const RippFr, accept: *mut i32) -> i32;
    pub fn ripp_tipa_ssm_verify(v_srs: *const RippVerifierSrs, com_a: *const RippGt, com_t: *const RippG1J, scalar_b: *const RippFr, com_gt: *const RippGt, com_g1: *const RippG1J, rounds: usize, base_a: *const RippG1J, final_ck_a: *const RippG2J, opening_a: *const RippG2J, accept: *mut i32) -> i32;
    pub fn ripp_verify_aggregate_proof(v_srs: *const RippVerifierSrs, vk: *const RippGroth16Vk, public_inputs: *const RippFr, n: usize, m: usize, proof: *const RippAggregateProof, accept: *mut i32) -> i32;
    pub fn ripp_ser_tipa_tipp_proof(com_steps: *const RippGt, rounds: usize, base_a: *const RippG1J, base_b: *const RippG2J, final_ck_a: *const RippG2J, final_ck_b: *const RippG1J, opening_a: *const RippG2J, opening_b: *const RippG1J, compress: i32, out: *mut u8, cap: usize) -> usize;
    pub fn ripp_de_tipa_tipp_proof(in_: *const u8, len: usize, compress: i32, with_tipa: i32, max_rounds: usize, rounds: *mut usize, com_steps: *mut RippGt, base_a: *mut RippG1J, base_b: *mut RippG2J, final_ck_a: *mut RippG2J, final_ck_b: *mut RippG1J, opening_a: *mut RippG2J, opening_b: *mut RippG1J) -> i32;
    pub fn ripp_ser_tipa_ssm_proof(com_gt: *const RippGt, com_g1: *const RippG1J, rounds: usize, base_a: *const RippG1J, base_b: *const RippFr, final_ck_a: *const RippG2J, opening_a: *const RippG2J, compress: i32, out: *mut u8, cap: usize) -> usize;
    pub fn ripp_de_tipa_ssm_proof(in_: *const u8, len: usize, compress: i32, max_rounds: usize, rounds: *mut usize, com_gt: *mut RippGt, com_g1: *mut RippG1J, base_a: *mut RippG1J, base_b: *mut RippFr, final_ck_a: *mut RippG2J, opening_a: *mut RippG2J) -> i32;
    pub fn ripp_ser_g1_compressed(p: *const RippG1A, out: *mut u8) -> usize;
    pub fn ripp_ser_g2_compressed(p: *const RippG2A, out: *mut u8) -> usize;
    pub fn ripp_blake2s(in_: *const u8, len: usize, out: *mut u8) -> i32;
    pub fn ripp_final_exp(miller_value: *const RippGt, out: *mut RippGt) -> i32;
    pub fn ripp_miller_combine(step_products: *const RippGt, out: *mut RippGt) -> i32;
    pub fn ripp_pairing_values(step_products: *const RippGt, count: i32, parts: i32, out: *mut RippGt) -> i32;
    pub fn ripp_gt_mul(a: *const RippGt, b: *const RippGt, out: *mut RippGt) -> i32;
    pub fn ripp_gt_pow(a: *const RippGt, k: *const RippFr, out: *mut RippGt) -> i32;
    pub fn ripp_fr_inverse(a: *const RippFr, out: *mut RippFr) -> i32;
    pub fn ripp_ser_gt(f: *const RippGt, out: *mut u8) -> usize;
    pub fn ripp_ser_g1(p: *const RippG1A, out: *mut u8) -> usize;
    pub fn ripp_ser_g2(p: *const RippG2A, out: *mut u8) -> usize;
    pub fn ripp_ser_fr(s: *const RippFr, out: *mut u8) -> usize;
    pub fn ripp_sipp_challenge(seed: *mut u8, z_l: *const RippGt, z_r: *const RippGt, x: *mut RippFr) -> i32;
    pub fn ripp_synth_g1(start: u64, first: usize, stride: usize, n: usize, out: *mut RippG1A) -> i32;
    pub fn ripp_synth_g2(start: u64, first: usize, stride: usize, n: usize, out: *mut RippG2A) -> i32;
    pub fn ripp_synth_fr(seed: u64, first: usize, stride: usize, n: usize, out: *mut RippFr) -> i32;
    // ---- end of the generated block ----
}
