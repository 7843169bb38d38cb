"""ctypes binding of the CPU oracle (oracle/_build/libripp_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Arrays are numpy uint64 in the flat C-ABI layouts (little-endian Montgomery limbs):
  Fr (n,4)  G1 affine (n,12)  G1 Jacobian (n,18)  G2 affine (n,24)  G2 Jacobian (n,36)  GT (72,)
"""
import ctypes
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libripp_oracle.so")

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
RP = 1 << 384
RR = 1 << 256


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return LIB_PATH


def effective_cpus():
    """CPUs this process may really use: min(affinity mask, cgroup CPU quota).  The GPU boxes expose 256 hardware
    threads but cap the job at a 16-CPU quota; oversubscribing OpenMP beyond the quota is 2-10x SLOWER."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


_lib = None


def lib():
    global _lib
    if _lib is None:
        # libomp in this image otherwise pins every worker onto one core
        os.environ.setdefault("KMP_AFFINITY", "disabled")
        _lib = ctypes.CDLL(build())
        _lib.orc_set_num_threads(effective_cpus())
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def u64(shape):
    return np.zeros(shape, dtype=np.uint64)


# ---------------------------------------------------------------- int <-> limb conversions (host side, python ints)
def fp_to_limbs(v):
    m = (v * RP) % P
    return [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]


def limbs_to_fp(l):
    m = sum(int(x) << (64 * i) for i, x in enumerate(l))
    return (m * pow(RP, -1, P)) % P


def fr_to_limbs(v):
    m = (v * RR) % R
    return [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def limbs_to_fr(l):
    m = sum(int(x) << (64 * i) for i, x in enumerate(l))
    return (m * pow(RR, -1, R)) % R


def fr_array(vals):
    return np.array([fr_to_limbs(v % R) for v in vals], dtype=np.uint64).reshape(len(vals), 4)


def g1_array(points):  # model affine points (x, y) or None
    out = u64((len(points), 12))
    for i, pt in enumerate(points):
        if pt is not None:
            out[i, :6] = fp_to_limbs(pt[0]); out[i, 6:] = fp_to_limbs(pt[1])
    return out


def g2_array(points):  # ((x0,x1),(y0,y1)) or None
    out = u64((len(points), 24))
    for i, pt in enumerate(points):
        if pt is not None:
            out[i, 0:6] = fp_to_limbs(pt[0][0]); out[i, 6:12] = fp_to_limbs(pt[0][1])
            out[i, 12:18] = fp_to_limbs(pt[1][0]); out[i, 18:24] = fp_to_limbs(pt[1][1])
    return out


def g1_from_row(row):
    if not row.any():
        return None
    return (limbs_to_fp(row[:6]), limbs_to_fp(row[6:12]))


def g2_from_row(row):
    if not row.any():
        return None
    return ((limbs_to_fp(row[0:6]), limbs_to_fp(row[6:12])), (limbs_to_fp(row[12:18]), limbs_to_fp(row[18:24])))


def gt_from_model(f_flat):
    """model flat Fp12 (6 Fp2 in w-powers) -> (72,) uint64 tower-ordered Montgomery limbs."""
    tower = [f_flat[0], f_flat[2], f_flat[4], f_flat[1], f_flat[3], f_flat[5]]
    out = u64(72)
    for i, c in enumerate(tower):
        out[12 * i:12 * i + 6] = fp_to_limbs(c[0]); out[12 * i + 6:12 * i + 12] = fp_to_limbs(c[1])
    return out


# ---------------------------------------------------------------- oracle entry points
def gen_g1(start, n):
    out = u64((n, 12)); lib().orc_gen_g1(ctypes.c_uint64(start), ctypes.c_size_t(n), _p(out)); return out


def gen_g2(start, n):
    out = u64((n, 24)); lib().orc_gen_g2(ctypes.c_uint64(start), ctypes.c_size_t(n), _p(out)); return out


def gen_scalars(seed, n):
    out = u64((n, 4)); lib().orc_gen_scalars(ctypes.c_uint64(seed), ctypes.c_size_t(n), _p(out)); return out


def blind_g1(a, seed):
    out = u64((len(a), 18)); lib().orc_jacobian_blind_g1(_p(a), ctypes.c_size_t(len(a)), ctypes.c_uint64(seed), _p(out)); return out


def blind_g2(b, seed):
    out = u64((len(b), 36)); lib().orc_jacobian_blind_g2(_p(b), ctypes.c_size_t(len(b)), ctypes.c_uint64(seed), _p(out)); return out


def pairing_product_a(a, b):
    out = u64(72); lib().orc_pairing_product_a(_p(a), _p(b), ctypes.c_size_t(len(a)), _p(out)); return out


def miller_product_a(a, b):
    out = u64(72); lib().orc_miller_product_a(_p(a), _p(b), ctypes.c_size_t(len(a)), _p(out)); return out


def final_exp(f):
    out = u64(72); lib().orc_final_exp(_p(f), _p(out)); return out


def pairing_product_j(l, r):
    out = u64(72)
    rc = lib().orc_pairing_product_j(_p(l), ctypes.c_size_t(len(l)), _p(r), ctypes.c_size_t(len(r)), _p(out))
    return rc, out


def msm_g1_j(bases, scalars):
    out = u64(18); rc = lib().orc_msm_g1_j(_p(bases), ctypes.c_size_t(len(bases)), _p(scalars), ctypes.c_size_t(len(scalars)), _p(out)); return rc, out


def msm_g2_j(bases, scalars):
    out = u64(36); rc = lib().orc_msm_g2_j(_p(bases), ctypes.c_size_t(len(bases)), _p(scalars), ctypes.c_size_t(len(scalars)), _p(out)); return rc, out


def msm_g1_a(bases, scalars):
    out = u64(18); lib().orc_msm_g1_a(_p(bases), _p(scalars), ctypes.c_size_t(len(bases)), _p(out)); return out


def msm_g2_a(bases, scalars):
    out = u64(36); lib().orc_msm_g2_a(_p(bases), _p(scalars), ctypes.c_size_t(len(bases)), _p(out)); return out


def msm_g1_naive(bases, scalars):
    out = u64(18); lib().orc_msm_g1_naive(_p(bases), _p(scalars), ctypes.c_size_t(len(bases)), _p(out)); return out


def msm_g2_naive(bases, scalars):
    out = u64(36); lib().orc_msm_g2_naive(_p(bases), _p(scalars), ctypes.c_size_t(len(bases)), _p(out)); return out


def g1_to_affine(pj):
    out = u64(12); lib().orc_g1_to_affine(_p(pj), _p(out)); return out


def g2_to_affine(pj):
    out = u64(24); lib().orc_g2_to_affine(_p(pj), _p(out)); return out


def normalize_g1(pj):
    out = u64((len(pj), 12)); lib().orc_normalize_g1(_p(pj), ctypes.c_size_t(len(pj)), _p(out)); return out


def normalize_g2(pj):
    out = u64((len(pj), 24)); lib().orc_normalize_g2(_p(pj), ctypes.c_size_t(len(pj)), _p(out)); return out


def fold_g1_a(hi, lo, s):
    out = u64((len(hi), 12)); lib().orc_fold_g1_a(_p(hi), _p(lo), ctypes.c_size_t(len(hi)), _p(s), _p(out)); return out


def fold_g2_a(hi, lo, s):
    out = u64((len(hi), 24)); lib().orc_fold_g2_a(_p(hi), _p(lo), ctypes.c_size_t(len(hi)), _p(s), _p(out)); return out


def fold_g1_j(hi, lo, s):
    out = u64((len(hi), 18)); lib().orc_fold_g1_j(_p(hi), _p(lo), ctypes.c_size_t(len(hi)), _p(s), _p(out)); return out


def fold_g2_j(hi, lo, s):
    out = u64((len(hi), 36)); lib().orc_fold_g2_j(_p(hi), _p(lo), ctypes.c_size_t(len(hi)), _p(s), _p(out)); return out


def scale_g1_a(a, r):
    out = u64((len(a), 12)); lib().orc_scale_g1_a(_p(a), _p(r), ctypes.c_size_t(len(a)), _p(out)); return out


def product_of_pairings_with_coeffs(a, b, r):
    out = u64(72); lib().orc_product_of_pairings_with_coeffs(_p(a), _p(b), _p(r), ctypes.c_size_t(len(a)), _p(out)); return out


def sipp_prove(a, b, r, value):
    n = len(a); lg = max(n.bit_length() - 1, 0)
    proof = u64((2 * lg, 72)); ch = u64((lg, 4))
    rc = lib().orc_sipp_prove(_p(a), _p(b), _p(r), ctypes.c_size_t(n), _p(value), _p(proof), _p(ch))
    return rc, proof, ch


def sipp_verify(a, b, r, value, proof):
    return lib().orc_sipp_verify(_p(a), _p(b), _p(r), ctypes.c_size_t(len(a)), _p(value), _p(proof), ctypes.c_size_t(len(proof) // 2))


def sipp_seed_digest(a, b, r, value):
    out = np.zeros(32, dtype=np.uint8); lib().orc_sipp_seed_digest(_p(a), _p(b), _p(r), ctypes.c_size_t(len(a)), _p(value), _p(out)); return bytes(out)


def ser_gt(f):
    out = np.zeros(576, dtype=np.uint8); lib().orc_ser_gt(_p(f), _p(out)); return bytes(out)


def ser_g1(p):
    out = np.zeros(96, dtype=np.uint8); lib().orc_ser_g1(_p(p), _p(out)); return bytes(out)


def ser_g2(p):
    out = np.zeros(192, dtype=np.uint8); lib().orc_ser_g2(_p(p), _p(out)); return bytes(out)


def ser_fr(s):
    out = np.zeros(32, dtype=np.uint8); lib().orc_ser_fr(_p(s), _p(out)); return bytes(out)


def gt_pow(f, k):
    out = u64(72); lib().orc_gt_pow(_p(f), _p(k), _p(out)); return out


def gt_one():
    out = u64(72); lib().orc_fp12_one(_p(out)); return out


def gt_mul(f, g):
    out = u64(72); lib().orc_gt_mul(_p(f), _p(g), _p(out)); return out


def blake2s(data):
    buf = np.frombuffer(bytes(data), dtype=np.uint8).copy() if len(data) else np.zeros(1, dtype=np.uint8)
    out = np.zeros(32, dtype=np.uint8); lib().orc_blake2s(_p(buf), ctypes.c_size_t(len(data)), _p(out)); return bytes(out)


def blake2b(data):
    buf = np.frombuffer(bytes(data), dtype=np.uint8).copy() if len(data) else np.zeros(1, dtype=np.uint8)
    out = np.zeros(64, dtype=np.uint8); lib().orc_blake2b(_p(buf), ctypes.c_size_t(len(data)), _p(out)); return bytes(out)


def chacha20_block(key, counter):
    k = np.frombuffer(bytes(key), dtype=np.uint8).copy(); out = np.zeros(64, dtype=np.uint8)
    lib().orc_chacha20_block(_p(k), ctypes.c_uint64(counter), _p(out)); return bytes(out)


def gipa_tipp_prove(m_a, m_b, ck_a, ck_b):
    n = len(m_a); rounds = n.bit_length() - 1
    steps = u64((rounds * 6, 72)); tr = u64((rounds, 4)); ba = u64(18); bb = u64(36); ka = u64(36); kb = u64(18)
    rc = lib().orc_gipa_tipp_prove(_p(m_a), _p(m_b), _p(ck_a), _p(ck_b), ctypes.c_size_t(n), _p(steps), _p(tr), _p(ba), _p(bb), _p(ka), _p(kb))
    return rc, steps, tr, ba, bb, ka, kb


def gipa_tipp_verify(ck_a, ck_b, com, steps, base_a, base_b):
    com = np.ascontiguousarray(np.stack(com), dtype=np.uint64)
    return lib().orc_gipa_tipp_verify(_p(ck_a), _p(ck_b), ctypes.c_size_t(len(ck_a)), _p(com), _p(steps), ctypes.c_size_t(len(steps) // 6), _p(base_a), _p(base_b))


# ---------------------------------------------------------------- TIPA / TIPAWithSSM / Groth16 aggregation (oracle/tipa.h)
def fr_from_random_bytes(digest):
    d = np.frombuffer(bytes(digest), dtype=np.uint8).copy(); out = u64(4)
    ok = lib().orc_fr_from_random_bytes(_p(d), _p(out)); return (out if ok else None)


def srs_powers_g1(s, num):
    out = u64((num, 18)); lib().orc_srs_powers_g1(_p(s), ctypes.c_size_t(num), _p(out)); return out


def srs_powers_g2(s, num):
    out = u64((num, 36)); lib().orc_srs_powers_g2(_p(s), ctypes.c_size_t(num), _p(out)); return out


def g1_mul_a(p, k):
    out = u64(12); lib().orc_g1_mul_a(_p(p), _p(k), _p(out)); return out


def g2_mul_a(p, k):
    out = u64(24); lib().orc_g2_mul_a(_p(p), _p(k), _p(out)); return out


def g1_generator():
    out = u64(12); lib().orc_g1_generator(_p(out)); return out


def g2_generator():
    out = u64(24); lib().orc_g2_generator(_p(out)); return out


def to_jac_g1(a):
    """affine (n,12) -> Jacobian (n,18) with Z = 1 (infinity -> Z = 0)."""
    a = np.atleast_2d(a); out = u64((len(a), 18)); out[:, :12] = a
    one = np.array(fp_to_limbs(1), dtype=np.uint64); inf = ~a.any(axis=1)
    out[:, 12:18] = one; out[inf, 0:6] = one; out[inf, 6:12] = one; out[inf, 12:18] = 0
    return out


def to_jac_g2(b):
    b = np.atleast_2d(b); out = u64((len(b), 36)); out[:, :24] = b
    one = np.array(fp_to_limbs(1), dtype=np.uint64); inf = ~b.any(axis=1)
    out[:, 24:30] = one; out[inf, 0:6] = one; out[inf, 12:18] = one; out[inf, 24:30] = 0
    return out


def tipa_tipp_prove(g_alpha_powers, h_beta_powers, m_a, m_b, ck_a, ck_b, r_shift):
    n = len(m_a); rounds = n.bit_length() - 1
    o = dict(steps=u64((max(rounds, 1) * 6, 72)), tr=u64((max(rounds, 1), 4)), base_a=u64(18), base_b=u64(36), final_ck_a=u64(36), final_ck_b=u64(18),
             opening_a=u64(36), opening_b=u64(18), kzg_c=u64(4))
    rc = lib().orc_tipa_tipp_prove(_p(g_alpha_powers), _p(h_beta_powers), _p(m_a), _p(m_b), _p(ck_a), _p(ck_b), ctypes.c_size_t(n), _p(r_shift),
                                   _p(o["steps"]), _p(o["tr"]), _p(o["base_a"]), _p(o["base_b"]), _p(o["final_ck_a"]), _p(o["final_ck_b"]),
                                   _p(o["opening_a"]), _p(o["opening_b"]), _p(o["kzg_c"]))
    return rc, o


def tipa_tipp_verify(g, h, g_beta, h_alpha, com, o, r_shift):
    com = np.ascontiguousarray(np.stack(com), dtype=np.uint64)
    return lib().orc_tipa_tipp_verify(_p(g), _p(h), _p(g_beta), _p(h_alpha), _p(com), _p(o["steps"]), ctypes.c_size_t(len(o["steps"]) // 6),
                                      _p(o["base_a"]), _p(o["base_b"]), _p(o["final_ck_a"]), _p(o["final_ck_b"]), _p(o["opening_a"]), _p(o["opening_b"]), _p(r_shift))


def tipa_ssm_prove(h_beta_powers, m_a, m_b, ck_a):
    n = len(m_a); rounds = n.bit_length() - 1
    o = dict(com_gt=u64((max(rounds, 1) * 2, 72)), com_g1=u64((max(rounds, 1) * 2, 18)), tr=u64((max(rounds, 1), 4)), base_a=u64(18), base_b=u64(4),
             final_ck_a=u64(36), opening_a=u64(36), kzg_c=u64(4))
    rc = lib().orc_tipa_ssm_prove(_p(h_beta_powers), _p(m_a), _p(m_b), _p(ck_a), ctypes.c_size_t(n), _p(o["com_gt"]), _p(o["com_g1"]), _p(o["tr"]),
                                  _p(o["base_a"]), _p(o["base_b"]), _p(o["final_ck_a"]), _p(o["opening_a"]), _p(o["kzg_c"]))
    return rc, o


def tipa_ssm_verify(g, h, g_beta, com_a, com_t, scalar_b, o):
    return lib().orc_tipa_ssm_verify(_p(g), _p(h), _p(g_beta), _p(com_a), _p(com_t), _p(scalar_b), _p(o["com_gt"]), _p(o["com_g1"]),
                                     ctypes.c_size_t(len(o["com_gt"]) // 2), _p(o["base_a"]), _p(o["final_ck_a"]), _p(o["opening_a"]))


def aggregate_proofs(g_alpha_powers, h_beta_powers, a, b, c):
    from ripp_amd._lib import AggregateProof
    pf = AggregateProof(len(a))
    rc = lib().orc_aggregate_proofs(_p(g_alpha_powers), _p(h_beta_powers), _p(a), _p(b), _p(c), ctypes.c_size_t(len(a)), pf.ref())
    return rc, pf


def verify_aggregate_proof(v_srs, vk, public_inputs, pf):
    g, h, g_beta, h_alpha = v_srs; alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1 = vk
    public_inputs = np.ascontiguousarray(public_inputs, dtype=np.uint64); n, m = public_inputs.shape[0], public_inputs.shape[1]
    return lib().orc_verify_aggregate_proof(_p(g), _p(h), _p(g_beta), _p(h_alpha), _p(alpha_g1), _p(beta_g2), _p(gamma_g2), _p(delta_g2), _p(gamma_abc_g1),
                                            _p(public_inputs), ctypes.c_size_t(n), ctypes.c_size_t(m), pf.ref())
