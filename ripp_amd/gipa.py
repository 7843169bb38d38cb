"""Generic GIPA over the trait mirror of ripp_amd.api -- ip_proofs/src/gipa.rs:97-415 restated for ANY instantiation
`GIPA<IP, LMC, RMC, IPC, D>` (D = Blake2b, as in all of the reference's tests and benches), plus `GIPAWithSSM`
(ip_proofs/src/tipa/structured_scalar_message.rs:56-128).

The fused device-resident provers of the C ABI (`ripp_gipa_tipp_prove`, `ripp_tipa_ssm_prove`) cover the two instantiations the
aggregation application uses; this module is the reference's own generic control flow on top of the SAME trait-level entry points
(`InnerProduct.inner_product`, `Commitment.commit`, the `mul_helper` folds), so that every instantiation of the reference's tests runs
on the GPU:
    pairing_inner_product_test            GIPA(PairingInnerProduct, AFGHOCommitmentG1, AFGHOCommitmentG2, IdentityCommitment(GT))   gipa.rs:470-497
    multiexponentiation_inner_product_test GIPA(MultiexponentiationInnerProductG1, AFGHOCommitmentG1, PedersenCommitmentG1, IdentityCommitment(G1))   :499-530
    scalar_inner_product_test             GIPA(ScalarInnerProduct, PedersenCommitmentG2, PedersenCommitmentG2, IdentityCommitment(Fr))   :532-561
Every data-parallel step is a call into libripp_hip.so (no CPU arithmetic on vectors here); the host does what the reference's host
code does between them: serialise, hash, invert one scalar.

A Rust host gets the same for free: the reference's GIPA is generic over the traits rust/ripp-hip implements.
"""
import ctypes
import hashlib

import numpy as np

from . import api
from ._lib import lib

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


# ---------------------------------------------------------------- scalars: canonical integer <-> Montgomery limbs (R = 2^256)
def fr_to_int(x):
    return int.from_bytes(api.ser_fr(np.ascontiguousarray(x, dtype=np.uint64).reshape(4)), "little")


def fr_from_int(v):
    m = (v % R_MOD) * (1 << 256) % R_MOD
    return np.array([(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


# ---------------------------------------------------------------- value kinds: what `Add`, `MulAssign<Scalar>`, `CanonicalSerialize` do
class G1:
    cols = 18; vec_kind = "G1"
    @staticmethod
    def ser(v): return api.ser_g1(api.normalize_batch_g1(np.asarray(v, dtype=np.uint64).reshape(1, 18))[0])     # projective serialises as affine
    @staticmethod
    def fold(hi, lo, s): return hi.fold(lo, s) if isinstance(hi, api.Vec) else api.fold_g1(hi, lo, s)
    @staticmethod
    def zero(): z = np.zeros(18, dtype=np.uint64); return z                                                      # Z = 0
    @staticmethod
    def canon(v): return api.normalize_batch_g1(np.asarray(v, dtype=np.uint64).reshape(-1, 18)).tobytes()
    @classmethod
    def mul(cls, v, s): return cls.fold(np.asarray(v, dtype=np.uint64).reshape(1, 18), cls.zero().reshape(1, 18), s)[0]
    @classmethod
    def add(cls, a, b): return cls.fold(np.asarray(a, dtype=np.uint64).reshape(1, 18), np.asarray(b, dtype=np.uint64).reshape(1, 18), fr_from_int(1))[0]


class G2:
    cols = 36; vec_kind = "G2"
    @staticmethod
    def ser(v): return api.ser_g2(api.normalize_batch_g2(np.asarray(v, dtype=np.uint64).reshape(1, 36))[0])
    @staticmethod
    def fold(hi, lo, s): return hi.fold(lo, s) if isinstance(hi, api.Vec) else api.fold_g2(hi, lo, s)
    @staticmethod
    def zero(): return np.zeros(36, dtype=np.uint64)
    @staticmethod
    def canon(v): return api.normalize_batch_g2(np.asarray(v, dtype=np.uint64).reshape(-1, 36)).tobytes()
    @classmethod
    def mul(cls, v, s): return cls.fold(np.asarray(v, dtype=np.uint64).reshape(1, 36), cls.zero().reshape(1, 36), s)[0]
    @classmethod
    def add(cls, a, b): return cls.fold(np.asarray(a, dtype=np.uint64).reshape(1, 36), np.asarray(b, dtype=np.uint64).reshape(1, 36), fr_from_int(1))[0]


class Fr:
    cols = 4; vec_kind = "Fr"
    @staticmethod
    def ser(v): return api.ser_fr(v)
    @staticmethod
    def fold(hi, lo, s):
        if isinstance(hi, api.Vec): return hi.fold(lo, s)
        hi, lo = api._c(hi, 4), api._c(lo, 4); out = np.zeros_like(hi)
        api._check(lib().ripp_fold_fr(api._p(hi), api._p(lo), ctypes.c_size_t(len(hi)), api._p(np.ascontiguousarray(s, dtype=np.uint64).reshape(4)), api._p(out)))
        return out
    @staticmethod
    def zero(): return np.zeros(4, dtype=np.uint64)
    @staticmethod
    def canon(v): return np.ascontiguousarray(v, dtype=np.uint64).tobytes()
    @staticmethod
    def mul(v, s): return fr_from_int(fr_to_int(v) * fr_to_int(s))
    @staticmethod
    def add(a, b): return fr_from_int(fr_to_int(a) + fr_to_int(b))


class GT:
    """PairingOutput: the group law is written additively by arkworks (`+` is the Fq12 product, `*= scalar` the power)."""
    cols = 72
    @staticmethod
    def ser(v): return api.ser_gt(v)
    @staticmethod
    def canon(v): return np.ascontiguousarray(v, dtype=np.uint64).tobytes()
    @staticmethod
    def mul(v, s):
        v = np.ascontiguousarray(v, dtype=np.uint64).reshape(72); out = np.zeros(72, dtype=np.uint64)
        api._check(lib().ripp_gt_pow(api._p(v), api._p(np.ascontiguousarray(s, dtype=np.uint64).reshape(4)), api._p(out))); return out
    @staticmethod
    def add(a, b): return api.gt_mul(a, b)


class Placeholder:
    """HomomorphicPlaceholderValue (dh_commitments/src/identity/mod.rs:17-30): serialises to nothing, absorbs every operation."""
    cols = 0
    @staticmethod
    def ser(v): return b""
    @staticmethod
    def fold(hi, lo, s): return hi
    @staticmethod
    def canon(v): return b""
    @staticmethod
    def mul(v, s): return v
    @staticmethod
    def add(a, b): return a


class _Vec:
    """IdentityOutput<T>(Vec<T>): u64-LE length prefix + items (ark-serialize Vec)."""
    def __init__(self, kind): self.kind = kind
    def ser(self, v): return len(v).to_bytes(8, "little") + b"".join(self.kind.ser(x) for x in v)
    def canon(self, v): return b"".join(self.kind.canon(x) for x in v)
    def mul(self, v, s): return [self.kind.mul(x, s) for x in v]
    def add(self, a, b): return [self.kind.add(x, y) for x, y in zip(a, b)]


# ---------------------------------------------------------------- commitments (Message / Key / Output kinds + commit)
class AFGHOCommitmentG1(api.AFGHOCommitmentG1):
    message, key, output = G1, G2, GT


class AFGHOCommitmentG2(api.AFGHOCommitmentG2):
    message, key, output = G2, G1, GT


class PedersenCommitmentG1(api.PedersenCommitmentG1):
    message, key, output = Fr, G1, G1


class PedersenCommitmentG2(api.PedersenCommitmentG2):
    message, key, output = Fr, G2, G2


def IdentityCommitment(kind):
    """IdentityCommitment<T, F> (identity/mod.rs:72-89): commit(_k, m) = IdentityOutput(m.to_vec())."""
    class _Identity:
        message, key, output = kind, Placeholder, _Vec(kind)
        @staticmethod
        def commit(k, m): return [np.ascontiguousarray(x, dtype=np.uint64) for x in m]
    return _Identity


class SSMPlaceholderCommitment:
    """structured_scalar_message.rs:28-47: the structured scalar message is not committed to -- commit() is Fr::zero()."""
    message, key, output = Fr, Placeholder, Fr
    @staticmethod
    def commit(k, m): return Fr.zero()


class _IP:
    def __init__(self, fn, left, right, out): self.inner_product, self.left, self.right, self.out = fn, left, right, out


PairingIP = _IP(api.PairingInnerProduct.inner_product, G1, G2, GT)
MultiexpIPG1 = _IP(api.MultiexponentiationInnerProductG1.inner_product, G1, Fr, G1)
MultiexpIPG2 = _IP(api.MultiexponentiationInnerProductG2.inner_product, G2, Fr, G2)
ScalarIP = _IP(api.ScalarInnerProduct.inner_product, Fr, Fr, Fr)


class GIPA:
    """GIPA<IP, LMC, RMC, IPC, Blake2b>.  Vectors are numpy arrays of the kinds' layouts (placeholder keys: any list of the right length)."""

    def __init__(self, ip, lmc, rmc, ipc, resident=True):
        self.ip, self.lmc, self.rmc, self.ipc = ip, lmc, rmc, ipc
        self.resident = resident         # prover vectors live in HBM (api.Vec) between rounds; False = the host-slice calls of round 1

    @staticmethod
    def _to_device(kind, v):
        """upload a message / key vector once (include/ripp_hip.h: ripp_vec_*); placeholders and already-resident vectors pass through"""
        if isinstance(v, api.Vec) or not hasattr(kind, "vec_kind") or len(v) < 2: return v
        return api.Vec.upload(kind.vec_kind, np.asarray(v, dtype=np.uint64).reshape(len(v), -1))

    # ---- Fiat-Shamir challenge (gipa.rs:233-258 and :331-356): returns (c, c_inv) AFTER the reference's swap
    def _challenge(self, prev, com_1, com_2):
        nonce = 0
        while True:
            h = nonce.to_bytes(8, "big") + Fr.ser(prev if prev is not None else Fr.zero())
            for com in (com_1, com_2):
                h += self.lmc.output.ser(com[0]) + self.rmc.output.ser(com[1]) + self.ipc.output.ser(com[2])
            c128 = int.from_bytes(hashlib.blake2b(h).digest()[:16], "big")
            if c128 != 0:
                return fr_from_int(pow(c128, -1, R_MOD)), fr_from_int(c128)          # (c, c_inv) = (c128^-1, c128)
            nonce += 1

    def prove_with_aux(self, values, ck):
        """gipa.rs:162-312.  values = (m_a, m_b); ck = (ck_a, ck_b, ck_t).  Returns (proof, aux) with BOTH vectors in the reference's
        (reversed) order: proof = {r_commitment_steps, r_base}, aux = {r_transcript, ck_base}."""
        m_a, m_b = values; ck_a, ck_b, ck_t = ck
        n = len(m_a)
        assert n & (n - 1) == 0 and n > 0, "assert!(m_a.len().is_power_of_two())  (gipa.rs:195)"
        L, Rk, KA, KB = self.lmc.message, self.rmc.message, self.lmc.key, self.rmc.key
        if self.resident:
            m_a, m_b, ck_a, ck_b = self._to_device(L, m_a), self._to_device(Rk, m_b), self._to_device(KA, ck_a), self._to_device(KB, ck_b)
        steps, transcript = [], []
        while len(m_a) > 1:
            split = len(m_a) // 2
            m_a_1, m_a_2, ck_a_1, ck_a_2 = m_a[split:], m_a[:split], ck_a[:split], ck_a[split:]          # gipa.rs:209-217
            m_b_1, m_b_2, ck_b_1, ck_b_2 = m_b[:split], m_b[split:], ck_b[split:], ck_b[:split]
            com_1 = (self.lmc.commit(ck_a_1, m_a_1), self.rmc.commit(ck_b_1, m_b_1), self.ipc.commit(ck_t, [self.ip.inner_product(m_a_1, m_b_1)]))
            com_2 = (self.lmc.commit(ck_a_2, m_a_2), self.rmc.commit(ck_b_2, m_b_2), self.ipc.commit(ck_t, [self.ip.inner_product(m_a_2, m_b_2)]))
            c, c_inv = self._challenge(transcript[-1] if transcript else None, com_1, com_2)
            m_a = L.fold(m_a_1, m_a_2, c)                     # m_a_1 * c + m_a_2          :262-267
            m_b = Rk.fold(m_b_2, m_b_1, c_inv)                # m_b_2 * c_inv + m_b_1      :270-275
            ck_a = KA.fold(ck_a_2, ck_a_1, c_inv)             # ck_a_2 * c_inv + ck_a_1    :278-283
            ck_b = KB.fold(ck_b_1, ck_b_2, c)                 # ck_b_1 * c + ck_b_2        :286-291
            steps.append((com_1, com_2)); transcript.append(c)
        proof = {"r_commitment_steps": steps[::-1], "r_base": (m_a[0], m_b[0])}
        aux = {"r_transcript": transcript[::-1], "ck_base": (ck_a[0], ck_b[0])}
        return proof, aux

    def prove(self, values, ck, com):
        """gipa.rs:108-133 (with the reference's pre-checks)."""
        m_a, m_b, t = values
        if self.ip.out.canon(self.ip.inner_product(m_a, m_b)) != self.ip.out.canon(t):
            raise ValueError("InnerProductArgumentError::InnerProductInvalid")
        if not (self._eq(self.lmc.output, self.lmc.commit(ck[0], m_a), com[0]) and self._eq(self.rmc.output, self.rmc.commit(ck[1], m_b), com[1])
                and self._eq(self.ipc.output, self.ipc.commit([ck[2]], [t]), com[2])):
            raise ValueError("InnerProductArgumentError::InnerProductInvalid")
        return self.prove_with_aux((m_a, m_b), (ck[0], ck[1], [ck[2]]))[0]

    @staticmethod
    def _eq(kind, a, b): return kind.canon(a) == kind.canon(b)

    # ---- verifier (gipa.rs:135-160, 322-415)
    def compute_recursive_challenges(self, com, proof):
        com_a, com_b, com_t = com
        transcript = []
        for com_1, com_2 in reversed(proof["r_commitment_steps"]):
            c, c_inv = self._challenge(transcript[-1] if transcript else None, com_1, com_2)
            O = (self.lmc.output, self.rmc.output, self.ipc.output)
            com_a, com_b, com_t = [o.add(o.add(o.mul(x1, c), cur), o.mul(x2, c_inv)) for o, x1, cur, x2 in zip(O, com_1, (com_a, com_b, com_t), com_2)]
            transcript.append(c)
        return (com_a, com_b, com_t), transcript[::-1]

    def compute_final_commitment_keys(self, ck_a, ck_b, transcript):
        """gipa.rs:374-403: sum_i ck_a[i] * e_a[i] (the reference folds sequentially and notes the MSM as a TODO: here it IS the device MSM)."""
        ea, eb = [1], [1]
        for i, c in enumerate(transcript):
            ci = fr_to_int(c); cinv = pow(ci, -1, R_MOD)
            for j in range(1 << i):
                ea.append(ea[j] * cinv % R_MOD); eb.append(eb[j] * ci % R_MOD)
        assert len(ea) == len(ck_a)
        out = []
        for kind, keys, ex in ((self.lmc.key, ck_a, ea), (self.rmc.key, ck_b, eb)):
            if kind is Placeholder:
                out.append(keys[0]); continue
            sc = np.stack([fr_from_int(x) for x in ex])
            msm = api.MultiexponentiationInnerProductG1 if kind is G1 else api.MultiexponentiationInnerProductG2
            out.append(msm.inner_product(keys, sc))
        return out

    def verify_base_commitment(self, base_ck, base_com, proof):
        ck_a_base, ck_b_base, ck_t = base_ck
        a_base, b_base = proof["r_base"]
        t_base = self.ip.inner_product(np.asarray(a_base)[None], np.asarray(b_base)[None])
        return (self._eq(self.lmc.output, self.lmc.commit(_one(ck_a_base), np.asarray(a_base)[None]), base_com[0])
                and self._eq(self.rmc.output, self.rmc.commit(_one(ck_b_base), np.asarray(b_base)[None]), base_com[1])
                and self._eq(self.ipc.output, self.ipc.commit(ck_t, [t_base]), base_com[2]))

    def verify(self, ck, com, proof):
        n = len(ck[0])
        assert n & (n - 1) == 0 and n == len(ck[1])
        base_com, transcript = self.compute_recursive_challenges(com, proof)
        ck_a_base, ck_b_base = self.compute_final_commitment_keys(ck[0], ck[1], transcript)
        return self.verify_base_commitment((ck_a_base, ck_b_base, [ck[2]]), base_com, proof)


def _one(x):
    return [x] if not isinstance(x, np.ndarray) else x[None]


class GIPAWithSSM:
    """structured_scalar_message.rs:56-128: GIPA with RMC = SSMPlaceholderCommitment, the verifier recomputing the final scalar from the
    structure b_i = b^i."""

    def __init__(self, ip, lmc, ipc):
        self.ip, self.lmc, self.ipc = ip, lmc, ipc
        self.gipa = GIPA(ip, lmc, SSMPlaceholderCommitment, ipc)

    def prove_with_structured_scalar_message(self, values, ck):
        return self.gipa.prove_with_aux(values, (ck[0], [None] * len(values[1]), [ck[1]]))[0]

    def verify_with_structured_scalar_message(self, ck, com, scalar_b, proof):
        g = self.gipa
        base_com, transcript = g.compute_recursive_challenges((com[0], Fr.zero(), com[1]), proof)
        ck_a_base, ck_b_base = g.compute_final_commitment_keys(ck[0], [None] * len(ck[0]), transcript)
        gipa_valid = g.verify_base_commitment((ck_a_base, ck_b_base, [ck[1]]), base_com, proof)
        p2b = fr_to_int(scalar_b); b_base = 1
        for x in transcript:                                                           # :108-114
            b_base = b_base * (1 + pow(fr_to_int(x), -1, R_MOD) * p2b) % R_MOD
            p2b = p2b * p2b % R_MOD
        a_base = np.asarray(proof["r_base"][0])[None]
        t_base = self.ip.inner_product(a_base, fr_from_int(b_base)[None])
        base_valid = (g._eq(self.lmc.output, self.lmc.commit(_one(ck_a_base), a_base), base_com[0])
                      and g._eq(self.ipc.output, self.ipc.commit([ck[1]], [t_base]), base_com[2]))
        return gipa_valid and base_valid
