"""ctypes loader of libripp_hip.so -- the ONLY compute backend of this package.

There is deliberately no CPU fallback: if the HIP library is missing this raises, and every compute entry
point of the library itself returns RIPP_ERR_DEVICE when no gfx950 device is usable.
"""
import ctypes
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RIPP_HIP_LIB") or os.path.join(PKG_DIR, "lib", "libripp_hip.so")      # RIPP_HIP_LIB: another build of the same library (A/B runs)

RIPP_OK, RIPP_ERR_LENGTH, RIPP_ERR_POW2, RIPP_ERR_DEVICE, RIPP_ERR_ARG = 0, 1, 2, 3, 4
RIPP_ABI_VERSION = 7              # include/ripp_hip.h; checked against the library at load time together with sizeof(ripp_stats)


class RippStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in (
        "total_ms", "upload_ms", "scale_ms", "miller_lines_ms", "miller_products_ms", "fold_ms", "normalize_ms",
        "host_ms", "hash_ms", "kernel_miller_lines_ms_sum", "kernel_line_products_ms_sum")] + [
        (n, ctypes.c_uint64) for n in ("kernel_miller_lines_launches", "kernel_line_products_launches", "pairs_lines", "pairs_products")] + [
        ("exchange_ms", ctypes.c_double), ("look_ms", ctypes.c_double), ("look_items", ctypes.c_uint64), ("look_pairs", ctypes.c_uint64),
        ("statement_hash_ms", ctypes.c_double), ("statement_hash_wait_ms", ctypes.c_double), ("chains_lines", ctypes.c_uint64),
        ("mem_tier", ctypes.c_uint64), ("device_bytes", ctypes.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class RippConfig(ctypes.Structure):
    """`ripp_config` of include/ripp_hip.h: every choice among implementations of the same function (kernel forms, crossover sizes, look-ahead plan)."""
    _fields_ = [("struct_size", ctypes.c_uint32)] + [(n, ctypes.c_uint32) for n in (
        "no_vm", "no_precompute", "no_fold_tables", "no_msm_glv", "lp_one_lane", "no_endo", "no_fq", "no_xscale", "scale_no_fq", "agg_sequential", "look_static", "quiet_waits", "no_share", "no_fuse", "fuse_tables")] + [
        ("look_eighths", ctypes.c_int32), ("ranks_per_device", ctypes.c_int32), ("msm_c", ctypes.c_int32), ("msm_ch", ctypes.c_uint32), ("msm_gmin", ctypes.c_uint32), ("no_prebuild", ctypes.c_uint32)] + [
        (n, ctypes.c_uint64) for n in ("vm_lines_max", "vm_fold_max", "vm_tree_max", "gls_split_max", "msm_vm_merge_max", "fold_tab_min", "fq_min", "lp_fq_min", "vm_joint_max",
                                       "vm_scale_max", "tail_pipe_max", "ml_fq_min", "fq_min_g1", "msm_lds_sort_min", "msm_chunk_min", "mem_cap_bytes")] + [
        ("hot_workers", ctypes.c_uint32), ("no_job_cache", ctypes.c_uint32),
        ("no_lp_karatsuba", ctypes.c_uint32), ("comm_timeout_ms", ctypes.c_uint32), ("plan_derate_pct", ctypes.c_uint32), ("n_devices", ctypes.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def _u64(n):
    return ctypes.c_uint64 * n


class AggregateProofStruct(ctypes.Structure):
    """`ripp_aggregate_proof` of include/ripp_hip.h (AggregateProof, groth16_aggregation.rs:59-69).  Step arrays are
    caller-allocated and filled in ROUND order."""
    _fields_ = [
        ("com_a", _u64(72)), ("com_b", _u64(72)), ("com_c", _u64(72)), ("ip_ab", _u64(72)), ("agg_c", _u64(18)), ("r", _u64(4)),
        ("ab_com_steps", ctypes.c_void_p), ("ab_transcript", ctypes.c_void_p),
        ("ab_base_a", _u64(18)), ("ab_base_b", _u64(36)), ("ab_final_ck_a", _u64(36)), ("ab_final_ck_b", _u64(18)),
        ("ab_opening_a", _u64(36)), ("ab_opening_b", _u64(18)), ("ab_kzg_c", _u64(4)),
        ("c_com_gt", ctypes.c_void_p), ("c_com_g1", ctypes.c_void_p), ("c_transcript", ctypes.c_void_p),
        ("c_base_a", _u64(18)), ("c_base_b", _u64(4)), ("c_final_ck_a", _u64(36)), ("c_opening_a", _u64(36)), ("c_kzg_c", _u64(4))]


class VerifierSRSStruct(ctypes.Structure):
    """`ripp_verifier_srs` (VerifierSRS, tipa/mod.rs:104-110)."""
    _fields_ = [("g", _u64(18)), ("h", _u64(36)), ("g_beta", _u64(18)), ("h_alpha", _u64(36))]


class Groth16VKStruct(ctypes.Structure):
    """`ripp_groth16_vk` (the members of ark_groth16::VerifyingKey that verify_aggregate_proof reads)."""
    _fields_ = [("alpha_g1", _u64(12)), ("beta_g2", _u64(24)), ("gamma_g2", _u64(24)), ("delta_g2", _u64(24)),
                ("gamma_abc_g1", ctypes.c_void_p), ("gamma_abc_len", ctypes.c_size_t)]


class AggregateProof:
    """Owner of an AggregateProofStruct and of its step arrays (numpy); `field(name)` views a fixed-size member."""

    def __init__(self, n):
        import numpy as np
        self.n, self.rounds = n, max(n.bit_length() - 1, 1)
        r = self.rounds
        self.ab_com_steps = np.zeros((r * 6, 72), dtype=np.uint64); self.ab_transcript = np.zeros((r, 4), dtype=np.uint64)
        self.c_com_gt = np.zeros((r * 2, 72), dtype=np.uint64); self.c_com_g1 = np.zeros((r * 2, 18), dtype=np.uint64); self.c_transcript = np.zeros((r, 4), dtype=np.uint64)
        self.s = AggregateProofStruct()
        for name in ("ab_com_steps", "ab_transcript", "c_com_gt", "c_com_g1", "c_transcript"):
            setattr(self.s, name, getattr(self, name).ctypes.data)

    def field(self, name):
        import numpy as np
        return np.ctypeslib.as_array(getattr(self.s, name))

    def ref(self):
        return ctypes.byref(self.s)

    FIXED = ("com_a", "com_b", "com_c", "ip_ab", "agg_c", "r", "ab_base_a", "ab_base_b", "ab_final_ck_a", "ab_final_ck_b", "ab_opening_a",
             "ab_opening_b", "ab_kzg_c", "c_base_a", "c_base_b", "c_final_ck_a", "c_opening_a", "c_kzg_c")
    STEPS = ("ab_com_steps", "ab_transcript", "c_com_gt", "c_com_g1", "c_transcript")


_lib = None


def lib():
    """Load (once) and return the HIP engine library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  ripp_amd has no CPU fallback.")
        _lib = load_library(LIB_PATH)
    return _lib


def load_library(path):
    L = ctypes.CDLL(path)
    # the library writes sizeof(ripp_stats) bytes through every stats pointer: a binding built for another layout would be overrun
    ok = hasattr(L, "ripp_abi_version") and hasattr(L, "ripp_stats_size")
    if ok:
        L.ripp_stats_size.restype = ctypes.c_size_t
        ok = L.ripp_abi_version() == RIPP_ABI_VERSION and L.ripp_stats_size() == ctypes.sizeof(RippStats)
    if not ok:
        raise RuntimeError(f"{path}: ABI mismatch (this binding: version {RIPP_ABI_VERSION}, ripp_stats of {ctypes.sizeof(RippStats)} bytes); rebuild the library")
    L.ripp_last_error.restype = ctypes.c_char_p
    L.ripp_test_inject_failure.restype = None
    for name in ("ripp_ser_gt", "ripp_ser_g1", "ripp_ser_g2", "ripp_ser_fr", "ripp_sipp_job_rounds_left", "ripp_sipp_job_local_len", "ripp_vec_len", "ripp_device_bytes"):
        getattr(L, name).restype = ctypes.c_size_t
    L.ripp_vec_free.restype = None; L.ripp_vec_free.argtypes = [ctypes.c_void_p]
    L.ripp_statement_hash_times.restype = None
    return L


def last_error():
    return lib().ripp_last_error().decode()
