//! `extern "C"` declarations of `include/ripp_hip.h` (one to one; see that header for the reference interface each entry replaces).
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

pub const RIPP_OK: i32 = 0;
pub const RIPP_ERR_LENGTH: i32 = 1;
pub const RIPP_ERR_POW2: i32 = 2;
pub const RIPP_ERR_DEVICE: i32 = 3;
pub const RIPP_ERR_ARG: i32 = 4;

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
}
/// `RIPP_ABI_VERSION` of include/ripp_hip.h this binding was written against; `abi_check()` compares it (and the size of `RippStats`, which
/// the library writes in full through every stats pointer) with the loaded library.
pub const RIPP_ABI_VERSION: i32 = 5;
#[cfg(feature = "ffi")]
pub fn abi_check() -> bool { unsafe { ripp_abi_version() == RIPP_ABI_VERSION && ripp_stats_size() == core::mem::size_of::<RippStats>() } }

#[cfg(feature = "ffi")]
extern "C" {
    pub fn ripp_init(device_ordinal: i32) -> i32;
    pub fn ripp_shutdown();
    pub fn ripp_device_count() -> i32;
    pub fn ripp_last_error() -> *const core::ffi::c_char;
    pub fn ripp_config_default(cfg: *mut RippConfig) -> i32;
    pub fn ripp_configure(cfg: *const RippConfig) -> i32;
    pub fn ripp_config_get(cfg: *mut RippConfig) -> i32;
    pub fn ripp_test_inject_failure(rank: i32, round: i32);
    pub fn ripp_abi_version() -> i32;
    pub fn ripp_stats_size() -> usize;
    // InnerProduct implementations on host slices
    pub fn ripp_pairing_product_j(l: *const RippG1J, nl: usize, r: *const RippG2J, nr: usize, out: *mut RippGt) -> i32;
    pub fn ripp_pairing_product_a(a: *const RippG1A, b: *const RippG2A, n: usize, out: *mut RippGt) -> i32;
    pub fn ripp_pairing_product_coeffs_a(a: *const RippG1A, b: *const RippG2A, r: *const RippFr, n: usize, out: *mut RippGt) -> i32;
    pub fn ripp_msm_g1_j(b: *const RippG1J, nl: usize, s: *const RippFr, nr: usize, out: *mut RippG1J) -> i32;
    pub fn ripp_msm_g2_j(b: *const RippG2J, nl: usize, s: *const RippFr, nr: usize, out: *mut RippG2J) -> i32;
    pub fn ripp_msm_g1_a(b: *const RippG1A, s: *const RippFr, n: usize, out: *mut RippG1J) -> i32;
    pub fn ripp_msm_g2_a(b: *const RippG2A, s: *const RippFr, n: usize, out: *mut RippG2J) -> i32;
    pub fn ripp_scalar_inner_product(l: *const RippFr, nl: usize, r: *const RippFr, nr: usize, out: *mut RippFr) -> i32;
    // folds / normalisation
    pub fn ripp_fold_g1_j(hi: *const RippG1J, lo: *const RippG1J, half: usize, s: *const RippFr, out: *mut RippG1J) -> i32;
    pub fn ripp_fold_g2_j(hi: *const RippG2J, lo: *const RippG2J, half: usize, s: *const RippFr, out: *mut RippG2J) -> i32;
    pub fn ripp_fold_g1_a(hi: *const RippG1A, lo: *const RippG1A, half: usize, s: *const RippFr, out: *mut RippG1A) -> i32;
    pub fn ripp_fold_g2_a(hi: *const RippG2A, lo: *const RippG2A, half: usize, s: *const RippFr, out: *mut RippG2A) -> i32;
    pub fn ripp_normalize_g1(p: *const RippG1J, n: usize, out: *mut RippG1A) -> i32;
    pub fn ripp_normalize_g2(p: *const RippG2J, n: usize, out: *mut RippG2A) -> i32;
    // device-resident vectors: upload once, split / fold / take inner products on views (include/ripp_hip.h, "device-resident vectors")
    pub fn ripp_vec_upload_g1a(p: *const RippG1A, n: usize, out: *mut *mut c_void) -> i32;
    pub fn ripp_vec_upload_g2a(p: *const RippG2A, n: usize, out: *mut *mut c_void) -> i32;
    pub fn ripp_vec_upload_g1j(p: *const RippG1J, n: usize, out: *mut *mut c_void) -> i32;
    pub fn ripp_vec_upload_g2j(p: *const RippG2J, n: usize, out: *mut *mut c_void) -> i32;
    pub fn ripp_vec_upload_fr(p: *const RippFr, n: usize, out: *mut *mut c_void) -> i32;
    pub fn ripp_vec_len(v: *const c_void) -> usize;
    pub fn ripp_vec_kind(v: *const c_void) -> i32;
    pub fn ripp_vec_slice(v: *const c_void, off: usize, len: usize, view: *mut *mut c_void) -> i32;
    pub fn ripp_vec_halves(v: *const c_void, lo: *mut *mut c_void, hi: *mut *mut c_void) -> i32;
    pub fn ripp_vec_download(v: *const c_void, out: *mut c_void) -> i32;
    pub fn ripp_vec_free(v: *mut c_void);
    pub fn ripp_vec_pairing_product(left_g1: *const c_void, right_g2: *const c_void, out: *mut RippGt) -> i32;
    pub fn ripp_vec_msm(bases: *const c_void, scalars: *const c_void, out: *mut c_void) -> i32;
    pub fn ripp_vec_scalar_inner_product(l: *const c_void, r: *const c_void, out: *mut RippFr) -> i32;
    pub fn ripp_vec_fold(hi: *const c_void, lo: *const c_void, s: *const RippFr, out: *mut *mut c_void) -> i32;
    // SIPP
    pub fn ripp_sipp_prove(a: *const RippG1A, b: *const RippG2A, r: *const RippFr, n: usize, value: *const RippGt,
                           proof: *mut RippGt, challenges: *mut RippFr, stats: *mut RippStats) -> i32;
    pub fn ripp_sipp_verify(a: *const RippG1A, b: *const RippG2A, r: *const RippFr, n: usize, claimed: *const RippGt,
                            proof: *const RippGt, rounds: usize, accept: *mut i32) -> i32;
    // GIPA / TIPA (TIPP instantiation) -- see the header for the output layout
    pub fn ripp_gipa_tipp_prove(m_a: *const RippG1J, m_b: *const RippG2J, ck_a: *const RippG2J, ck_b: *const RippG1J, n: usize,
                                com_steps: *mut RippGt, transcript: *mut RippFr, base_a: *mut RippG1J, base_b: *mut RippG2J,
                                ck_base_a: *mut RippG2J, ck_base_b: *mut RippG1J, stats: *mut RippStats) -> i32;
    pub fn ripp_srs_create(g_alpha_powers: *const RippG1J, h_beta_powers: *const RippG2J, num: usize, srs: *mut *mut c_void) -> i32;
    pub fn ripp_srs_destroy(srs: *mut c_void);
    // multi-GPU collectives (one process per GPU; RCCL inside the library)
    pub fn ripp_comm_unique_id(id: *mut u8 /* [128] */) -> i32;
    pub fn ripp_comm_init(id: *const u8, rank: i32, world: i32) -> i32;
    pub fn ripp_comm_destroy();
    pub fn ripp_comm_init_callback(rank: i32, world: i32, allgather: extern "C" fn(*mut c_void, *const c_void, *mut c_void, usize) -> i32, user: *mut c_void) -> i32;
    pub fn ripp_comm_rank() -> i32;
    pub fn ripp_comm_world() -> i32;
    pub fn ripp_comm_allgather(send: *const c_void, recv: *mut c_void, bytes: usize) -> i32;
    pub fn ripp_pairing_product_sharded_j(l: *const RippG1J, nl: usize, r: *const RippG2J, nr: usize, out: *mut RippGt) -> i32;
    pub fn ripp_msm_g1_sharded_j(b: *const RippG1J, nl: usize, s: *const RippFr, nr: usize, out: *mut RippG1J) -> i32;
    pub fn ripp_msm_g2_sharded_j(b: *const RippG2J, nl: usize, s: *const RippFr, nr: usize, out: *mut RippG2J) -> i32;
    /// rank 0 passes the full statement (or its 32-byte digest); the other ranks pass null for both
    pub fn ripp_sipp_prove_sharded(a: *const RippG1A, b: *const RippG2A, r: *const RippFr, n_local: usize, value: *const RippGt,
                                   full_a: *const RippG1A, full_b: *const RippG2A, full_r: *const RippFr, seed_digest: *const u8,
                                   proof: *mut RippGt, challenges: *mut RippFr, stats: *mut RippStats) -> i32;
}
