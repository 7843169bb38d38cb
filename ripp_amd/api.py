"""Host-side mirror of the reference interface for the hot path, over the C ABI (include/ripp_hip.h).

Names, argument order and error behaviour follow the reference so that parity tests read like its own:
  inner_products/src/lib.rs:40-142   InnerProduct, PairingInnerProduct, MultiexponentiationInnerProduct, InnerProductError
  dh_commitments/src/afgho16/mod.rs  AFGHOCommitmentG1 / AFGHOCommitmentG2
  dh_commitments/src/pedersen/mod.rs PedersenCommitment
  sipp/src/lib.rs:42-224             SIPP.prove / SIPP.verify, product_of_pairings(_with_coeffs)

Values are numpy uint64 arrays in the flat C-ABI layouts (little-endian Montgomery limbs):
  Fr (n,4)  G1Affine (n,12)  G1 projective/Jacobian (n,18)  G2Affine (n,24)  G2 projective (n,36)  GT (72,)
"""
import ctypes
import numpy as np
from ._lib import lib, last_error, RippStats, AggregateProof, VerifierSRSStruct, Groth16VKStruct, RIPP_OK, RIPP_ERR_LENGTH, RIPP_ERR_POW2, RIPP_ERR_DEVICE

__all__ = ["InnerProductError", "DeviceError", "Vec", "PairingInnerProduct", "MultiexponentiationInnerProductG1",
           "MultiexponentiationInnerProductG2", "ScalarInnerProduct", "AFGHOCommitmentG1", "AFGHOCommitmentG2", "PedersenCommitmentG1",
           "PedersenCommitmentG2", "SIPP", "SippJob", "GIPA_TIPP", "SRS", "TIPA_TIPP", "TIPAWithSSM", "aggregate_proofs", "aggregate_proofs_sharded", "gipa_tipp_prove_sharded", "verify_aggregate_proof", "AggregateProof", "ser_tipa_tipp_proof", "de_tipa_tipp_proof", "ser_tipa_ssm_proof", "de_tipa_ssm_proof", "ser_g1_compressed", "ser_g2_compressed", "product_of_pairings", "product_of_pairings_with_coeffs",
           "normalize_batch_g1", "normalize_batch_g2", "fold_g1_affine", "fold_g2_affine", "fold_g1", "fold_g2",
           "scale_g1_affine", "synth_g1", "synth_g2", "synth_fr", "init", "device_count", "final_exponentiation",
           "ser_gt", "ser_g1", "ser_g2", "ser_fr", "sipp_seed_digest", "gt_mul", "statement_hash_times", "configure", "config_default", "config_get", "release_scratch", "device_bytes"]


class InnerProductError(Exception):
    """InnerProductError::MessageLengthInvalid(left, right) -- inner_products/src/lib.rs:18-38."""

    def __init__(self, left, right):
        self.left, self.right = left, right
        super().__init__(f"left length, right length: {left}, {right}")


class DeviceError(RuntimeError):
    pass


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, cols):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim == 1 and cols and a.size == cols:
        a = a.reshape(1, cols)
    if cols and (a.ndim != 2 or a.shape[1] != cols):
        if a.size == 0:
            return a.reshape(0, cols)
        raise ValueError(f"expected shape (n, {cols}), got {a.shape}")
    return a


def _check(rc, left=None, right=None):
    if rc == RIPP_OK:
        return
    if rc == RIPP_ERR_LENGTH:
        raise InnerProductError(left, right)
    if rc == RIPP_ERR_POW2:
        raise AssertionError("vector length must be a power of two (sipp/src/lib.rs:48-53)")
    if rc == RIPP_ERR_DEVICE:
        raise DeviceError("HIP engine unavailable: " + last_error())
    raise ValueError(f"libripp_hip status {rc}: {last_error()}")


def config_default():
    """ripp_config_default: the built-in defaults (a RippConfig to modify and hand to `configure`)."""
    from ._lib import RippConfig
    c = RippConfig(); _check(lib().ripp_config_default(ctypes.byref(c))); return c


def configure(cfg=None, **changes):
    """ripp_configure: process-wide settings from the next call on.  configure() / configure(None) returns to the defaults;
    configure(tail_pipe_max=0, look_eighths=24) changes the named members of the defaults; configure(cfg) installs a RippConfig."""
    if cfg is None and not changes:
        _check(lib().ripp_configure(None)); return None
    c = cfg if cfg is not None else config_default()
    for k, v in changes.items():
        if not hasattr(c, k):
            raise AttributeError(f"ripp_config has no member {k!r}")
        setattr(c, k, v)
    _check(lib().ripp_configure(ctypes.byref(c))); return c


def config_get():
    """ripp_config_get: what the next call runs with (defaults < configure() < RIPP_* environment overrides)."""
    from ._lib import RippConfig
    c = RippConfig(); _check(lib().ripp_config_get(ctypes.byref(c))); return c


def init(device=0):
    _check(lib().ripp_init(ctypes.c_int32(device)))


def device_count():
    return int(lib().ripp_device_count())


def release_scratch():
    """ripp_release_scratch: free the engine's grow-only scratch (line buffer, fold tables, MSM scratch, parked one-shot buffers)."""
    _check(lib().ripp_release_scratch())


def device_bytes():
    """ripp_device_bytes: device memory the library holds right now (what ripp_config.mem_cap_bytes bounds)."""
    return int(lib().ripp_device_bytes())


# ------------------------------------------------------------------ InnerProduct implementations
# ------------------------------------------------------------------ device-resident vectors (ripp_vec_*)
_FP_ONE = None


def _fp_one():
    """Fp::one() in Montgomery form, read off the library's own generator table (Z of the first SRS power) -- curve-agnostic."""
    global _FP_ONE
    if _FP_ONE is None:
        g = np.zeros((1, 18), dtype=np.uint64)
        _check(lib().ripp_srs_powers_g1(_p(FR_ONE), ctypes.c_size_t(1), _p(g)))
        a = normalize_batch_g1(g)
        # (x, y, Z) with Z = 1: Z is recovered as the Jacobian Z of an already-affine point, i.e. the one the library wrote when Z == 1
        _FP_ONE = g[0, 12:18].copy() if np.array_equal(g[0, :12], a[0]) else None
        if _FP_ONE is None:
            raise RuntimeError("generator came back with Z != 1")
    return _FP_ONE


class Vec:
    """A vector kept in HBM across calls (include/ripp_hip.h: ripp_vec).  kind: "G1" | "G2" | "Fr".  Group elements are stored affine
    (normalised on upload / after a fold).  Slicing with a contiguous range gives a VIEW sharing the storage -- `v[:s]`, `v[s:]` are the
    halves of a GIPA round; `v[i]` downloads one element (projective layout for groups, like the host-slice API)."""
    KINDS = {"G1": 1, "G2": 2, "Fr": 3}
    COLS = {1: 12, 2: 24, 3: 4}

    def __init__(self, handle, kind, n):
        self._h, self.kind, self._n = handle, kind, n

    @staticmethod
    def upload(kind, array):
        """array: G1 (n,12) affine or (n,18) projective; G2 (n,24) or (n,36); Fr (n,4)."""
        k = Vec.KINDS[kind]
        a = np.ascontiguousarray(array, dtype=np.uint64)
        if a.ndim == 1: a = a.reshape(1, -1)
        fn = {(1, 12): "ripp_vec_upload_g1a", (1, 18): "ripp_vec_upload_g1j", (2, 24): "ripp_vec_upload_g2a", (2, 36): "ripp_vec_upload_g2j", (3, 4): "ripp_vec_upload_fr"}.get((k, a.shape[1]))
        if fn is None:
            raise ValueError(f"no {kind} layout with {a.shape[1]} limbs per element")
        h = ctypes.c_void_p()
        _check(getattr(lib(), fn)(_p(a), ctypes.c_size_t(len(a)), ctypes.byref(h)))
        return Vec(h, k, len(a))

    def __len__(self): return self._n

    def __getitem__(self, key):
        if isinstance(key, slice):
            start, stop, step = key.indices(self._n)
            if step != 1: raise IndexError("device vectors slice by contiguous ranges only")
            h = ctypes.c_void_p()
            _check(lib().ripp_vec_slice(self._h, ctypes.c_size_t(start), ctypes.c_size_t(max(stop - start, 0)), ctypes.byref(h)))
            return Vec(h, self.kind, max(stop - start, 0))
        i = key + self._n if key < 0 else key
        if not 0 <= i < self._n: raise IndexError(key)
        return self[i:i + 1].to_host()[0]

    def download(self):
        """the stored elements: G1 (n,12) / G2 (n,24) affine, Fr (n,4)"""
        out = np.zeros((self._n, Vec.COLS[self.kind]), dtype=np.uint64)
        _check(lib().ripp_vec_download(self._h, _p(out))); return out

    def to_host(self):
        """the host-slice layouts of the trait-level API: projective (Z = 1, or Z = 0 for the point at infinity) for groups"""
        a = self.download()
        if self.kind == 3: return a
        half = a.shape[1] // 2                                     # limbs of one coordinate
        out = np.zeros((self._n, 3 * half), dtype=np.uint64); out[:, :2 * half] = a
        inf = ~a.any(axis=1)
        out[~inf, 2 * half:2 * half + 6] = _fp_one()
        out[inf, 0:6] = _fp_one(); out[inf, half:half + 6] = _fp_one()     # (1, 1, 0), arkworks' zero()
        return out

    def fold(self, lo, s):
        """self * s + lo, element-wise (gipa.rs:262-291) -> a new resident vector"""
        h = ctypes.c_void_p()
        _check(lib().ripp_vec_fold(self._h, lo._h, _p(np.ascontiguousarray(s, dtype=np.uint64).reshape(4)), ctypes.byref(h)), len(self), len(lo))
        return Vec(h, self.kind, self._n)

    def close(self):
        if self._h:
            lib().ripp_vec_free(self._h); self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PairingInnerProduct:
    """inner_products/src/lib.rs:52-75.  left: G1 projective (n,18); right: G2 projective (n,36) -> GT (72,)."""

    @staticmethod
    def inner_product(left, right):
        if isinstance(left, Vec) or isinstance(right, Vec):
            out = np.zeros(72, dtype=np.uint64)
            _check(lib().ripp_vec_pairing_product(left._h, right._h, _p(out)), len(left), len(right)); return out
        l, r = _c(left, 18), _c(right, 36)
        out = np.zeros(72, dtype=np.uint64)
        _check(lib().ripp_pairing_product_j(_p(l), ctypes.c_size_t(len(l)), _p(r), ctypes.c_size_t(len(r)), _p(out)), len(l), len(r))
        return out


class MultiexponentiationInnerProductG1:
    """inner_products/src/lib.rs:119-142 with G = G1.  left: bases (n,18); right: scalars (n,4) -> G1 projective (18,)."""

    @staticmethod
    def inner_product(left, right):
        if isinstance(left, Vec) or isinstance(right, Vec):
            out = np.zeros(18, dtype=np.uint64)
            _check(lib().ripp_vec_msm(left._h, right._h, _p(out)), len(left), len(right)); return out
        l, r = _c(left, 18), _c(right, 4)
        out = np.zeros(18, dtype=np.uint64)
        _check(lib().ripp_msm_g1_j(_p(l), ctypes.c_size_t(len(l)), _p(r), ctypes.c_size_t(len(r)), _p(out)), len(l), len(r))
        return out


class MultiexponentiationInnerProductG2:
    @staticmethod
    def inner_product(left, right):
        if isinstance(left, Vec) or isinstance(right, Vec):
            out = np.zeros(36, dtype=np.uint64)
            _check(lib().ripp_vec_msm(left._h, right._h, _p(out)), len(left), len(right)); return out
        l, r = _c(left, 36), _c(right, 4)
        out = np.zeros(36, dtype=np.uint64)
        _check(lib().ripp_msm_g2_j(_p(l), ctypes.c_size_t(len(l)), _p(r), ctypes.c_size_t(len(r)), _p(out)), len(l), len(r))
        return out


class ScalarInnerProduct:
    """inner_products/src/lib.rs:144-166: left, right (n,4) Fr -> Fr (4,)."""

    @staticmethod
    def inner_product(left, right):
        if isinstance(left, Vec) or isinstance(right, Vec):
            out = np.zeros(4, dtype=np.uint64)
            _check(lib().ripp_vec_scalar_inner_product(left._h, right._h, _p(out)), len(left), len(right)); return out
        l, r = _c(left, 4), _c(right, 4)
        out = np.zeros(4, dtype=np.uint64)
        _check(lib().ripp_scalar_inner_product(_p(l), ctypes.c_size_t(len(l)), _p(r), ctypes.c_size_t(len(r)), _p(out)), len(l), len(r))
        return out


# ------------------------------------------------------------------ DoublyHomomorphicCommitment implementations
class _Commitment:
    @classmethod
    def verify(cls, k, m, com):
        """Default `verify` of the trait: commit(k, m)? == *com  (dh_commitments/src/lib.rs:52-54)."""
        return bool(np.array_equal(cls._canon(cls.commit(k, m)), cls._canon(com)))

    @staticmethod
    def _canon(x):
        return x


class AFGHOCommitmentG1(_Commitment):
    """Message G1, Key G2: commit(k, m) = PairingInnerProduct(m, k)  (afgho16/mod.rs:20-32)."""

    @staticmethod
    def commit(k, m):
        return PairingInnerProduct.inner_product(m, k)


class AFGHOCommitmentG2(_Commitment):
    """Message G2, Key G1: commit(k, m) = PairingInnerProduct(k, m)  (afgho16/mod.rs:35-47)."""

    @staticmethod
    def commit(k, m):
        return PairingInnerProduct.inner_product(k, m)


class PedersenCommitmentG1(_Commitment):
    """commit(k, m) = MultiexponentiationInnerProduct(k, m)  (pedersen/mod.rs:14-26); output compared as a group element."""

    @staticmethod
    def commit(k, m):
        return MultiexponentiationInnerProductG1.inner_product(k, m)

    @staticmethod
    def _canon(x):
        return normalize_batch_g1(np.asarray(x, dtype=np.uint64).reshape(1, 18))


class PedersenCommitmentG2(_Commitment):
    @staticmethod
    def commit(k, m):
        return MultiexponentiationInnerProductG2.inner_product(k, m)

    @staticmethod
    def _canon(x):
        return normalize_batch_g2(np.asarray(x, dtype=np.uint64).reshape(1, 36))


# ------------------------------------------------------------------ sipp crate free functions
def product_of_pairings(a, b):
    """sipp/src/lib.rs:219-224 (affine inputs)."""
    a, b = _c(a, 12), _c(b, 24)
    if len(a) != len(b):
        raise InnerProductError(len(a), len(b))
    out = np.zeros(72, dtype=np.uint64)
    _check(lib().ripp_pairing_product_a(_p(a), _p(b), ctypes.c_size_t(len(a)), _p(out)))
    return out


def product_of_pairings_with_coeffs(a, b, r):
    """sipp/src/lib.rs:184-217."""
    a, b, r = _c(a, 12), _c(b, 24), _c(r, 4)
    if not (len(a) == len(b) == len(r)):
        raise AssertionError(f"slice lengths differ: {len(a)}, {len(b)}, {len(r)}")
    out = np.zeros(72, dtype=np.uint64)
    _check(lib().ripp_pairing_product_coeffs_a(_p(a), _p(b), _p(r), ctypes.c_size_t(len(a)), _p(out)))
    return out


def normalize_batch_g1(pj):
    pj = _c(pj, 18); out = np.zeros((len(pj), 12), dtype=np.uint64)
    _check(lib().ripp_normalize_g1(_p(pj), ctypes.c_size_t(len(pj)), _p(out))); return out


def normalize_batch_g2(pj):
    pj = _c(pj, 36); out = np.zeros((len(pj), 24), dtype=np.uint64)
    _check(lib().ripp_normalize_g2(_p(pj), ctypes.c_size_t(len(pj)), _p(out))); return out


def _fold(fn, hi, lo, s, cin, cout):
    hi, lo, s = _c(hi, cin), _c(lo, cin), np.ascontiguousarray(s, dtype=np.uint64).reshape(4)
    assert len(hi) == len(lo)
    out = np.zeros((len(hi), cout), dtype=np.uint64)
    _check(fn(_p(hi), _p(lo), ctypes.c_size_t(len(hi)), _p(s), _p(out))); return out


def fold_g1_affine(hi, lo, s):
    """out[i] = s*hi[i] + lo[i], normalised (sipp/src/lib.rs:87-92)."""
    return _fold(lib().ripp_fold_g1_a, hi, lo, s, 12, 12)


def fold_g2_affine(hi, lo, s):
    return _fold(lib().ripp_fold_g2_a, hi, lo, s, 24, 24)


def fold_g1(hi, lo, s):
    """projective in / out (ip_proofs/src/gipa.rs:262-290)."""
    return _fold(lib().ripp_fold_g1_j, hi, lo, s, 18, 18)


def fold_g2(hi, lo, s):
    return _fold(lib().ripp_fold_g2_j, hi, lo, s, 36, 36)


def scale_g1_affine(a, r):
    a, r = _c(a, 12), _c(r, 4); out = np.zeros((len(a), 12), dtype=np.uint64)
    _check(lib().ripp_scale_g1_a(_p(a), _p(r), ctypes.c_size_t(len(a)), _p(out))); return out


# ------------------------------------------------------------------ SIPP
class SippJob:
    """Device-resident SIPP statement (shard).  world == 1: `prove`.  world > 1: staged rounds (ripp_amd/sharded.py)."""

    def __init__(self, a, b, r, rank=0, world=1):
        a, b, r = _c(a, 12), _c(b, 24), _c(r, 4)
        assert len(a) == len(b) == len(r)
        self.n_local, self.rank, self.world = len(a), rank, world
        self._h = ctypes.c_void_p()
        _check(lib().ripp_sipp_job_create(_p(a), _p(b), _p(r), ctypes.c_size_t(len(a)), ctypes.c_int32(rank), ctypes.c_int32(world), ctypes.byref(self._h)))

    def close(self):
        if self._h:
            lib().ripp_sipp_job_destroy(self._h); self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prove(self, value):
        n = self.n_local; lg = n.bit_length() - 1
        value = np.ascontiguousarray(value, dtype=np.uint64).reshape(72)
        proof = np.zeros((2 * lg, 72), dtype=np.uint64); ch = np.zeros((lg, 4), dtype=np.uint64); st = RippStats()
        _check(lib().ripp_sipp_job_prove(self._h, _p(value), _p(proof), _p(ch), ctypes.byref(st)))
        return proof, ch, st.as_dict()

    # staged interface
    def begin(self):
        _check(lib().ripp_sipp_job_begin(self._h))

    def rounds_left(self):
        return int(lib().ripp_sipp_job_rounds_left(self._h))

    def round_partials(self):
        out = np.zeros((2, 72), dtype=np.uint64)
        _check(lib().ripp_sipp_job_round_partials(self._h, _p(out))); return out

    def round_finish(self, combined, seed_digest):
        combined = np.ascontiguousarray(combined, dtype=np.uint64)
        zl = np.zeros(72, dtype=np.uint64); zr = np.zeros(72, dtype=np.uint64); x = np.zeros(4, dtype=np.uint64)
        d = (ctypes.c_uint8 * 32).from_buffer_copy(seed_digest) if seed_digest is not None else None
        _check(lib().ripp_sipp_job_round_finish(self._h, _p(combined), d, _p(zl), _p(zr), _p(x)))
        return zl, zr, x

    def local_len(self):
        return int(lib().ripp_sipp_job_local_len(self._h))

    def export(self):
        n = self.local_len(); a = np.zeros((n, 12), dtype=np.uint64); b = np.zeros((n, 24), dtype=np.uint64)
        _check(lib().ripp_sipp_job_export(self._h, _p(a), _p(b))); return a, b

    def import_(self, a, b):
        a, b = _c(a, 12), _c(b, 24)
        _check(lib().ripp_sipp_job_import(self._h, _p(a), _p(b), ctypes.c_size_t(len(a))))

    @staticmethod
    def combine(gathered):
        """element-wise GT product over ranks of their (count,72) partial arrays."""
        g = np.ascontiguousarray(np.stack(gathered), dtype=np.uint64); world, count = g.shape[0], g.shape[1]
        out = np.zeros((count, 72), dtype=np.uint64)
        _check(lib().ripp_combine_partials(_p(g), ctypes.c_int32(world), ctypes.c_size_t(count), _p(out))); return out

    def stats(self):
        st = RippStats(); _check(lib().ripp_sipp_job_stats(self._h, ctypes.byref(st))); return st.as_dict()


class SIPP:
    """SIPP::<Bls12_381, Blake2s> (sipp/src/lib.rs:25-181)."""

    @staticmethod
    def prove(a, b, r, value):
        return SIPP.prove_one_shot(a, b, r, value)[0]

    @staticmethod
    def prove_one_shot(a, b, r, value):
        """ripp_sipp_prove on HOST slices -- the call SURVEY.md section 8(d) defines the metric on: the statement hash starts on the caller's
        buffers, the upload of the statement happens inside.  Returns (proof, challenges, stats)."""
        a, b, r = _c(a, 12), _c(b, 24), _c(r, 4)
        if not (len(a) == len(b) == len(r)):      # the C side reads n elements of each: never let a short slice through
            raise AssertionError(f"assert_eq!(a.len(), b.len()) / r.len(): {len(a)}, {len(b)}, {len(r)}  (sipp/src/lib.rs:48-49)")
        n = len(a)
        if n == 0 or n & (n - 1):
            raise AssertionError("vector length must be a power of two (sipp/src/lib.rs:48-53)")
        lg = n.bit_length() - 1
        value = np.ascontiguousarray(value, dtype=np.uint64).reshape(72)
        proof = np.zeros((2 * lg, 72), dtype=np.uint64); ch = np.zeros((lg, 4), dtype=np.uint64); st = RippStats()
        _check(lib().ripp_sipp_prove(_p(a), _p(b), _p(r), ctypes.c_size_t(n), _p(value), _p(proof), _p(ch), ctypes.byref(st)))
        return proof, ch, st.as_dict()

    @staticmethod
    def prove_with_stats(a, b, r, value):
        job = SippJob(a, b, r)
        try:
            return job.prove(value)
        finally:
            job.close()

    @staticmethod
    def verify(a, b, r, claimed_value, proof):
        a, b, r = _c(a, 12), _c(b, 24), _c(r, 4)
        if not (len(a) == len(b) == len(r)):      # sipp/src/lib.rs:116-117; the C side reads len(a) elements of b and r
            raise AssertionError(f"assert_eq!(a.len(), b.len()) / r.len(): {len(a)}, {len(b)}, {len(r)}")
        proof = np.ascontiguousarray(proof, dtype=np.uint64).reshape(-1, 72)
        if len(proof) % 2:
            raise ValueError("a SIPP proof is a list of (z_l, z_r) pairs: odd number of GT elements")
        claimed_value = np.ascontiguousarray(claimed_value, dtype=np.uint64).reshape(72)
        acc = ctypes.c_int32(0)
        _check(lib().ripp_sipp_verify(_p(a), _p(b), _p(r), ctypes.c_size_t(len(a)), _p(claimed_value), _p(proof), ctypes.c_size_t(len(proof) // 2), ctypes.byref(acc)))
        return bool(acc.value)


class GIPA_TIPP:
    """GIPA<PairingInnerProduct, AFGHOCommitmentG1, AFGHOCommitmentG2, IdentityCommitment<GT,Fr>, Blake2b>
    (ip_proofs/src/gipa.rs:97-312).  prove_with_aux returns, in the reference's (reversed) order,
    proof = {r_commitment_steps: [(com_1, com_2)], r_base: (m_a[0], m_b[0])}, aux = {r_transcript, ck_base}."""

    @staticmethod
    def prove_with_aux(m_a, m_b, ck_a, ck_b):
        m_a, m_b, ck_a, ck_b = _c(m_a, 18), _c(m_b, 36), _c(ck_a, 36), _c(ck_b, 18)
        n = len(m_a)
        assert len(m_b) == len(ck_a) == len(ck_b) == n
        if n == 0 or n & (n - 1):
            raise AssertionError("assert!(m_a.len().is_power_of_two())  (ip_proofs/src/gipa.rs:195)")
        rounds = n.bit_length() - 1
        steps = np.zeros((max(rounds, 1) * 6, 72), dtype=np.uint64); tr = np.zeros((max(rounds, 1), 4), dtype=np.uint64)
        ba = np.zeros(18, dtype=np.uint64); bb = np.zeros(36, dtype=np.uint64); ka = np.zeros(36, dtype=np.uint64); kb = np.zeros(18, dtype=np.uint64)
        st = RippStats()
        _check(lib().ripp_gipa_tipp_prove(_p(m_a), _p(m_b), _p(ck_a), _p(ck_b), ctypes.c_size_t(n), _p(steps), _p(tr), _p(ba), _p(bb), _p(ka), _p(kb), ctypes.byref(st)))
        steps, tr = steps[: rounds * 6].reshape(rounds, 6, 72), tr[:rounds]
        proof = {"r_commitment_steps": [((s[0], s[1], [s[2]]), (s[3], s[4], [s[5]])) for s in steps[::-1]], "r_base": (ba, bb)}
        aux = {"r_transcript": tr[::-1].copy(), "ck_base": (ka, kb)}
        return proof, aux, {"round_order_steps": steps.reshape(rounds * 6, 72), "round_order_transcript": tr, "stats": st.as_dict()}


    @staticmethod
    def verify(ck_a, ck_b, com, round_order_steps, r_base):
        """GIPA::verify (gipa.rs:135-160): com = (com_a, com_b, com_t[0]); steps in ROUND order as returned by prove_with_aux."""
        ck_a, ck_b = _c(ck_a, 36), _c(ck_b, 18); n = len(ck_a)
        com = np.ascontiguousarray(np.stack([np.asarray(x, dtype=np.uint64).reshape(72) for x in com]))
        steps = np.ascontiguousarray(round_order_steps, dtype=np.uint64).reshape(-1, 72)
        ba = np.ascontiguousarray(r_base[0], dtype=np.uint64).reshape(18); bb = np.ascontiguousarray(r_base[1], dtype=np.uint64).reshape(36)
        acc = ctypes.c_int32(0)
        _check(lib().ripp_gipa_tipp_verify(_p(ck_a), _p(ck_b), ctypes.c_size_t(n), _p(com), _p(steps), ctypes.c_size_t(len(steps) // 6), _p(ba), _p(bb), ctypes.byref(acc)))
        return bool(acc.value)


# ------------------------------------------------------------------ TIPA, TIPAWithSSM, Groth16 aggregation
def _vsrs(v):
    s = VerifierSRSStruct()
    for k in ("g", "h", "g_beta", "h_alpha"):
        arr = np.ascontiguousarray(v[k], dtype=np.uint64).reshape(-1)
        ctypes.memmove(getattr(s, k), arr.ctypes.data, arr.nbytes)
    return s


def _a(x, n):
    return np.ascontiguousarray(x, dtype=np.uint64).reshape(n)


FR_ONE = np.array([0x00000001fffffffe, 0x5884b7fa00034802, 0x998c4fefecbc4ff5, 0x1824b159acc5056f], dtype=np.uint64)   # Fr::one(), Montgomery form


class SRS:
    """SRS<Bls12_381> (ip_proofs/src/tipa/mod.rs:94-128): g_alpha_powers (2n-1,18), h_beta_powers (2n-1,36), normalised once and
    kept resident in HBM; g_beta / h_alpha are carried for `get_verifier_key`."""

    def __init__(self, g_alpha_powers, h_beta_powers, g_beta=None, h_alpha=None):
        gap, hbp = _c(g_alpha_powers, 18), _c(h_beta_powers, 36)
        assert len(gap) == len(hbp)
        self.g_alpha_powers, self.h_beta_powers, self.g_beta, self.h_alpha = gap, hbp, g_beta, h_alpha
        self.size = (len(gap) + 1) // 2
        self._h = ctypes.c_void_p()
        _check(lib().ripp_srs_create(_p(gap), _p(hbp), ctypes.c_size_t(len(gap)), ctypes.byref(self._h)))

    @staticmethod
    def from_trapdoors(alpha, beta, size):
        """TIPA::setup with the two field elements given instead of drawn (tipa/mod.rs:150-165): powers computed on the device."""
        alpha = np.ascontiguousarray(alpha, dtype=np.uint64).reshape(4); beta = np.ascontiguousarray(beta, dtype=np.uint64).reshape(4)
        num = 2 * size - 1
        gap = np.zeros((num, 18), dtype=np.uint64); hbp = np.zeros((num, 36), dtype=np.uint64)
        _check(lib().ripp_srs_powers_g1(_p(alpha), ctypes.c_size_t(num), _p(gap)))
        _check(lib().ripp_srs_powers_g2(_p(beta), ctypes.c_size_t(num), _p(hbp)))
        tmp = np.zeros((2, 18), dtype=np.uint64); _check(lib().ripp_srs_powers_g1(_p(beta), ctypes.c_size_t(2), _p(tmp))); g_beta = tmp[1].copy()
        tmp = np.zeros((2, 36), dtype=np.uint64); _check(lib().ripp_srs_powers_g2(_p(alpha), ctypes.c_size_t(2), _p(tmp))); h_alpha = tmp[1].copy()
        return SRS(gap, hbp, g_beta, h_alpha)

    def get_commitment_keys(self):
        """(ck_1, ck_2) = even powers (tipa/mod.rs:114-118)."""
        ck_1 = np.zeros((self.size, 36), dtype=np.uint64); ck_2 = np.zeros((self.size, 18), dtype=np.uint64)
        _check(lib().ripp_srs_commitment_keys(self._h, _p(ck_1), _p(ck_2))); return ck_1, ck_2

    def get_verifier_key(self):
        return {"g": self.g_alpha_powers[0].copy(), "h": self.h_beta_powers[0].copy(), "g_beta": self.g_beta, "h_alpha": self.h_alpha}

    def close(self):
        if self._h:
            lib().ripp_srs_destroy(self._h); self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TIPA_TIPP:
    """TIPA<PairingInnerProduct, AFGHOCommitmentG1, AFGHOCommitmentG2, IdentityCommitment<GT,Fr>, Bls12_381, Blake2b>
    = PairingInnerProductAB (groth16_aggregation.rs:24-31)."""

    @staticmethod
    def prove_with_srs_shift(srs, values, ck, r_shift):
        """tipa/mod.rs:176-231.  values = (m_a, m_b), ck = (ck_a, ck_b).  Returns a dict with the GIPA proof in ROUND order."""
        m_a, m_b, ck_a, ck_b = _c(values[0], 18), _c(values[1], 36), _c(ck[0], 36), _c(ck[1], 18)
        n = len(m_a); assert len(m_b) == len(ck_a) == len(ck_b) == n
        r_shift = np.ascontiguousarray(r_shift, dtype=np.uint64).reshape(4)
        rounds = max(n.bit_length() - 1, 1)
        o = dict(steps=np.zeros((rounds * 6, 72), dtype=np.uint64), tr=np.zeros((rounds, 4), dtype=np.uint64), base_a=np.zeros(18, dtype=np.uint64),
                 base_b=np.zeros(36, dtype=np.uint64), final_ck_a=np.zeros(36, dtype=np.uint64), final_ck_b=np.zeros(18, dtype=np.uint64),
                 opening_a=np.zeros(36, dtype=np.uint64), opening_b=np.zeros(18, dtype=np.uint64), kzg_c=np.zeros(4, dtype=np.uint64))
        st = RippStats()
        _check(lib().ripp_tipa_tipp_prove(srs._h, _p(m_a), _p(m_b), _p(ck_a), _p(ck_b), ctypes.c_size_t(n), _p(r_shift), _p(o["steps"]), _p(o["tr"]),
                                          _p(o["base_a"]), _p(o["base_b"]), _p(o["final_ck_a"]), _p(o["final_ck_b"]), _p(o["opening_a"]), _p(o["opening_b"]),
                                          _p(o["kzg_c"]), ctypes.byref(st)))
        o["stats"] = st.as_dict(); return o

    @staticmethod
    def prove(srs, values, ck):
        """tipa/mod.rs:168-174: r_shift = 1."""
        return TIPA_TIPP.prove_with_srs_shift(srs, values, ck, FR_ONE)

    @staticmethod
    def verify_with_srs_shift(v_srs, com, proof, r_shift):
        """tipa/mod.rs:242-301.  v_srs: dict from SRS.get_verifier_key(); com = (com_a, com_b, com_t[0]); proof: dict of prove_with_srs_shift."""
        vs = _vsrs(v_srs)
        com = np.ascontiguousarray(np.stack([np.asarray(x, dtype=np.uint64).reshape(72) for x in com]))
        steps = np.ascontiguousarray(proof["steps"], dtype=np.uint64).reshape(-1, 72)
        acc = ctypes.c_int32(0)
        _check(lib().ripp_tipa_tipp_verify(ctypes.byref(vs), _p(com), _p(steps), ctypes.c_size_t(len(steps) // 6), _p(_a(proof["base_a"], 18)), _p(_a(proof["base_b"], 36)),
                                           _p(_a(proof["final_ck_a"], 36)), _p(_a(proof["final_ck_b"], 18)), _p(_a(proof["opening_a"], 36)), _p(_a(proof["opening_b"], 18)),
                                           _p(_a(r_shift, 4)), ctypes.byref(acc)))
        return bool(acc.value)

    @staticmethod
    def verify(v_srs, com, proof):
        return TIPA_TIPP.verify_with_srs_shift(v_srs, com, proof, FR_ONE)


class TIPAWithSSM:
    """TIPAWithSSM<MultiexponentiationInnerProduct<G1>, AFGHOCommitmentG1, IdentityCommitment<G1,Fr>, Bls12_381, Blake2b>
    = MultiExpInnerProductC (groth16_aggregation.rs:42-48)."""

    @staticmethod
    def prove_with_structured_scalar_message(srs, values, ck):
        """structured_scalar_message.rs:211-268.  values = (m_a G1 (n,18), m_b Fr (n,4)), ck = (ck_a G2 (n,36),)."""
        m_a, m_b, ck_a = _c(values[0], 18), _c(values[1], 4), _c(ck[0], 36)
        n = len(m_a); assert len(m_b) == len(ck_a) == n
        rounds = max(n.bit_length() - 1, 1)
        o = dict(com_gt=np.zeros((rounds * 2, 72), dtype=np.uint64), com_g1=np.zeros((rounds * 2, 18), dtype=np.uint64), tr=np.zeros((rounds, 4), dtype=np.uint64),
                 base_a=np.zeros(18, dtype=np.uint64), base_b=np.zeros(4, dtype=np.uint64), final_ck_a=np.zeros(36, dtype=np.uint64),
                 opening_a=np.zeros(36, dtype=np.uint64), kzg_c=np.zeros(4, dtype=np.uint64))
        st = RippStats()
        _check(lib().ripp_tipa_ssm_prove(srs._h, _p(m_a), _p(m_b), _p(ck_a), ctypes.c_size_t(n), _p(o["com_gt"]), _p(o["com_g1"]), _p(o["tr"]),
                                         _p(o["base_a"]), _p(o["base_b"]), _p(o["final_ck_a"]), _p(o["opening_a"]), _p(o["kzg_c"]), ctypes.byref(st)))
        o["stats"] = st.as_dict(); return o


    @staticmethod
    def verify_with_structured_scalar_message(v_srs, com, scalar_b, proof):
        """structured_scalar_message.rs:270-331.  com = (com_a GT, com_t G1 projective)."""
        vs = _vsrs(v_srs)
        com_gt = np.ascontiguousarray(proof["com_gt"], dtype=np.uint64).reshape(-1, 72); com_g1 = np.ascontiguousarray(proof["com_g1"], dtype=np.uint64).reshape(-1, 18)
        acc = ctypes.c_int32(0)
        _check(lib().ripp_tipa_ssm_verify(ctypes.byref(vs), _p(_a(com[0], 72)), _p(_a(com[1], 18)), _p(_a(scalar_b, 4)), _p(com_gt), _p(com_g1),
                                          ctypes.c_size_t(len(com_gt) // 2), _p(_a(proof["base_a"], 18)), _p(_a(proof["final_ck_a"], 36)), _p(_a(proof["opening_a"], 36)),
                                          ctypes.byref(acc)))
        return bool(acc.value)


def verify_aggregate_proof(ip_verifier_srs, vk, public_inputs, proof):
    """verify_aggregate_proof (groth16_aggregation.rs:162-231).  vk = (alpha_g1 (12,), beta_g2, gamma_g2, delta_g2 (24,), gamma_abc_g1 (m+1,12));
    public_inputs (n, m, 4)."""
    vs = _vsrs(ip_verifier_srs)
    alpha, beta, gamma, delta, abc = vk
    abc = _c(abc, 12); k = Groth16VKStruct()
    for name, arr in (("alpha_g1", _a(alpha, 12)), ("beta_g2", _a(beta, 24)), ("gamma_g2", _a(gamma, 24)), ("delta_g2", _a(delta, 24))):
        ctypes.memmove(getattr(k, name), arr.ctypes.data, arr.nbytes)
    k.gamma_abc_g1 = abc.ctypes.data; k.gamma_abc_len = len(abc)
    pub = np.ascontiguousarray(public_inputs, dtype=np.uint64); n, m = pub.shape[0], pub.shape[1]
    acc = ctypes.c_int32(0)
    _check(lib().ripp_verify_aggregate_proof(ctypes.byref(vs), ctypes.byref(k), _p(pub), ctypes.c_size_t(n), ctypes.c_size_t(m), proof.ref(), ctypes.byref(acc)))
    return bool(acc.value)


def aggregate_proofs(ip_srs, a, b, c):
    """aggregate_proofs (groth16_aggregation.rs:77-160).  a, c: (n,12) G1Affine; b: (n,24) G2Affine -- the members of n Groth16 proofs.
    Returns (AggregateProof, stats)."""
    a, b, c = _c(a, 12), _c(b, 24), _c(c, 12)
    assert len(a) == len(b) == len(c)
    pf = AggregateProof(len(a)); st = RippStats()
    _check(lib().ripp_aggregate_proofs(ip_srs._h, _p(a), _p(b), _p(c), ctypes.c_size_t(len(a)), pf.ref(), ctypes.byref(st)))
    return pf, st.as_dict()


def aggregate_proofs_sharded(ip_srs, a_shard, b_shard, c_shard):
    """aggregate_proofs across the library's communicator (ripp_amd.sharded.NativeComm): this rank's shard of the proofs (global index
    j * world + rank), ip_srs built for the GLOBAL n.  Returns (AggregateProof, stats), identical on every rank."""
    a, b, c = _c(a_shard, 12), _c(b_shard, 24), _c(c_shard, 12)
    assert len(a) == len(b) == len(c)
    world = int(lib().ripp_comm_world())
    pf = AggregateProof(len(a) * world); st = RippStats()
    _check(lib().ripp_aggregate_proofs_sharded(ip_srs._h, _p(a), _p(b), _p(c), ctypes.c_size_t(len(a)), pf.ref(), ctypes.byref(st)))
    return pf, st.as_dict()


def gipa_tipp_prove_sharded(m_a, m_b, ck_a, ck_b):
    """ripp_gipa_tipp_prove_sharded: every argument is this rank's shard.  Returns (round-order steps (rounds*6,72), transcript (rounds,4),
    (base_a, base_b), (ck_base_a, ck_base_b)), identical on every rank."""
    m_a, m_b, ck_a, ck_b = _c(m_a, 18), _c(m_b, 36), _c(ck_a, 36), _c(ck_b, 18)
    nl = len(m_a); world = int(lib().ripp_comm_world()); n = nl * world
    rounds = max(n.bit_length() - 1, 1)
    steps = np.zeros((rounds * 6, 72), dtype=np.uint64); tr = np.zeros((rounds, 4), dtype=np.uint64)
    ba = np.zeros(18, dtype=np.uint64); bb = np.zeros(36, dtype=np.uint64); ka = np.zeros(36, dtype=np.uint64); kb = np.zeros(18, dtype=np.uint64)
    st = RippStats()
    _check(lib().ripp_gipa_tipp_prove_sharded(_p(m_a), _p(m_b), _p(ck_a), _p(ck_b), ctypes.c_size_t(nl), _p(steps), _p(tr), _p(ba), _p(bb), _p(ka), _p(kb), ctypes.byref(st)))
    r = n.bit_length() - 1
    return steps[: 6 * r], tr[:r], (ba, bb), (ka, kb)


# ------------------------------------------------------------------ wire format (CanonicalSerialize images of the proof structs)
def ser_tipa_tipp_proof(proof, compress=True, with_tipa=True):
    """TIPAProof (tipa/mod.rs:41-65) -- or GIPAProof (gipa.rs:24-51) with with_tipa=False -- from a prove_with_srs_shift dict."""
    steps = np.ascontiguousarray(proof["steps"], dtype=np.uint64).reshape(-1, 72); rounds = len(steps) // 6
    z = ctypes.c_void_p(None)
    args = [_p(steps), ctypes.c_size_t(rounds), _p(_a(proof["base_a"], 18)), _p(_a(proof["base_b"], 36))]
    args += [_p(_a(proof["final_ck_a"], 36)), _p(_a(proof["final_ck_b"], 18)), _p(_a(proof["opening_a"], 36)), _p(_a(proof["opening_b"], 18))] if with_tipa else [z, z, z, z]
    lib().ripp_ser_tipa_tipp_proof.restype = ctypes.c_size_t
    n = lib().ripp_ser_tipa_tipp_proof(*args, ctypes.c_int32(int(compress)), z, ctypes.c_size_t(0))
    out = np.zeros(n, dtype=np.uint8)
    assert lib().ripp_ser_tipa_tipp_proof(*args, ctypes.c_int32(int(compress)), _p(out), ctypes.c_size_t(n)) == n
    return bytes(out)


def de_tipa_tipp_proof(data, compress=True, with_tipa=True, max_rounds=64):
    buf = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    o = dict(steps=np.zeros((max_rounds * 6, 72), dtype=np.uint64), base_a=np.zeros(18, dtype=np.uint64), base_b=np.zeros(36, dtype=np.uint64),
             final_ck_a=np.zeros(36, dtype=np.uint64), final_ck_b=np.zeros(18, dtype=np.uint64), opening_a=np.zeros(36, dtype=np.uint64), opening_b=np.zeros(18, dtype=np.uint64))
    rounds = ctypes.c_size_t(0)
    _check(lib().ripp_de_tipa_tipp_proof(_p(buf), ctypes.c_size_t(len(buf)), ctypes.c_int32(int(compress)), ctypes.c_int32(int(with_tipa)), ctypes.c_size_t(max_rounds),
                                         ctypes.byref(rounds), _p(o["steps"]), _p(o["base_a"]), _p(o["base_b"]), _p(o["final_ck_a"]), _p(o["final_ck_b"]), _p(o["opening_a"]), _p(o["opening_b"])))
    o["steps"] = o["steps"][: rounds.value * 6].copy(); return o


def ser_tipa_ssm_proof(proof, compress=True):
    """TIPAWithSSMProof (tipa/structured_scalar_message.rs:138-156) from a prove_with_structured_scalar_message dict."""
    com_gt = np.ascontiguousarray(proof["com_gt"], dtype=np.uint64).reshape(-1, 72); com_g1 = np.ascontiguousarray(proof["com_g1"], dtype=np.uint64).reshape(-1, 18)
    rounds = len(com_gt) // 2; z = ctypes.c_void_p(None)
    args = [_p(com_gt), _p(com_g1), ctypes.c_size_t(rounds), _p(_a(proof["base_a"], 18)), _p(_a(proof["base_b"], 4)), _p(_a(proof["final_ck_a"], 36)), _p(_a(proof["opening_a"], 36)), ctypes.c_int32(int(compress))]
    lib().ripp_ser_tipa_ssm_proof.restype = ctypes.c_size_t
    n = lib().ripp_ser_tipa_ssm_proof(*args, z, ctypes.c_size_t(0))
    out = np.zeros(n, dtype=np.uint8)
    assert lib().ripp_ser_tipa_ssm_proof(*args, _p(out), ctypes.c_size_t(n)) == n
    return bytes(out)


def de_tipa_ssm_proof(data, compress=True, max_rounds=64):
    buf = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    o = dict(com_gt=np.zeros((max_rounds * 2, 72), dtype=np.uint64), com_g1=np.zeros((max_rounds * 2, 18), dtype=np.uint64), base_a=np.zeros(18, dtype=np.uint64),
             base_b=np.zeros(4, dtype=np.uint64), final_ck_a=np.zeros(36, dtype=np.uint64), opening_a=np.zeros(36, dtype=np.uint64))
    rounds = ctypes.c_size_t(0)
    _check(lib().ripp_de_tipa_ssm_proof(_p(buf), ctypes.c_size_t(len(buf)), ctypes.c_int32(int(compress)), ctypes.c_size_t(max_rounds), ctypes.byref(rounds),
                                        _p(o["com_gt"]), _p(o["com_g1"]), _p(o["base_a"]), _p(o["base_b"]), _p(o["final_ck_a"]), _p(o["opening_a"])))
    o["com_gt"] = o["com_gt"][: rounds.value * 2].copy(); o["com_g1"] = o["com_g1"][: rounds.value * 2].copy(); return o


def ser_g1_compressed(p): return _ser(lib().ripp_ser_g1_compressed, p, 48)
def ser_g2_compressed(p): return _ser(lib().ripp_ser_g2_compressed, p, 96)


# ------------------------------------------------------------------ host helpers / synthetic inputs
def final_exponentiation(f):
    f = np.ascontiguousarray(f, dtype=np.uint64).reshape(72); out = np.zeros(72, dtype=np.uint64)
    _check(lib().ripp_final_exp(_p(f), _p(out))); return out


def gt_mul(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(72); b = np.ascontiguousarray(b, dtype=np.uint64).reshape(72)
    out = np.zeros(72, dtype=np.uint64); _check(lib().ripp_gt_mul(_p(a), _p(b), _p(out))); return out


def _ser(fn, x, nbytes):
    x = np.ascontiguousarray(x, dtype=np.uint64); out = np.zeros(nbytes, dtype=np.uint8); fn(_p(x), _p(out)); return bytes(out)


def ser_gt(f): return _ser(lib().ripp_ser_gt, f, 576)
def ser_g1(p): return _ser(lib().ripp_ser_g1, p, 96)
def ser_g2(p): return _ser(lib().ripp_ser_g2, p, 192)
def ser_fr(s): return _ser(lib().ripp_ser_fr, s, 32)


def sipp_seed_digest(a, b, r, value):
    a, b, r = _c(a, 12), _c(b, 24), _c(r, 4); value = np.ascontiguousarray(value, dtype=np.uint64).reshape(72)
    out = np.zeros(32, dtype=np.uint8)
    _check(lib().ripp_sipp_seed_digest(_p(a), _p(b), _p(r), ctypes.c_size_t(len(a)), _p(value), _p(out))); return bytes(out)


def statement_hash_times():
    """(Blake2s ms, ms waiting for the serialisation workers) of the last statement hash in this process."""
    h, w = ctypes.c_double(), ctypes.c_double()
    lib().ripp_statement_hash_times(ctypes.byref(h), ctypes.byref(w)); return h.value, w.value


def synth_g1(start, n, first=0, stride=1):
    out = np.zeros((n, 12), dtype=np.uint64)
    _check(lib().ripp_synth_g1(ctypes.c_uint64(start), ctypes.c_size_t(first), ctypes.c_size_t(stride), ctypes.c_size_t(n), _p(out))); return out


def synth_g2(start, n, first=0, stride=1):
    out = np.zeros((n, 24), dtype=np.uint64)
    _check(lib().ripp_synth_g2(ctypes.c_uint64(start), ctypes.c_size_t(first), ctypes.c_size_t(stride), ctypes.c_size_t(n), _p(out))); return out


def synth_fr(seed, n, first=0, stride=1):
    out = np.zeros((n, 4), dtype=np.uint64)
    _check(lib().ripp_synth_fr(ctypes.c_uint64(seed), ctypes.c_size_t(first), ctypes.c_size_t(stride), ctypes.c_size_t(n), _p(out))); return out
