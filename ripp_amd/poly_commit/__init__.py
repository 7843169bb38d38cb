"""Polynomial-commitment applications of the inner-product arguments -- the host-side mirror of
ip_proofs/src/applications/poly_commit/mod.rs over the C ABI of libripp_hip.so:

    KZG                              mod.rs:50-119   setup / commit / open / verify
    BivariatePolynomial              mod.rs:121-140
    BivariatePolynomialCommitment    mod.rs:142-296  second tier = TIPAWithSSM<MultiexpIP<G1>, AFGHO-G1, Identity<G1>> (the fused device prover
                                                     `ripp_tipa_ssm_prove`), first tier = KZG
    UnivariatePolynomialCommitment   mod.rs:298-388  the sqrt-split with the reference's skew factor
    transparent (sub-module)         transparent.rs  the same applications over GIPAWithSSM and Pedersen, no trusted setup

Every group operation on a vector -- the KZG multi-scalar multiplications, the AFGHO commitment (a pairing inner product), the
structured-generator powers, the TIPA / GIPA folds and openings -- is a call into the HIP library.  The host does what the reference's
host does between them (mod.rs:210-236): build the powers of x, combine the y-polynomials' coefficients, one synthetic division.

Conventions of this module: field elements (coefficients, points, evaluations) are plain Python integers in [0, r); group elements are
the limb arrays of ripp_amd.api (G1 projective (18,), G1 affine (12,), GT (72,)).  `setup` takes its two trapdoors instead of drawing them
(as `SRS.from_trapdoors`): the tests fix them, a deployment draws them from its own CSPRNG and forgets them.
"""
import ctypes

import numpy as np

from .. import api
from .._lib import lib
from ..gipa import G1, G2, R_MOD, fr_from_int, fr_to_int

_p, _check = api._p, api._check


# ---------------------------------------------------------------- scalars
def frs(values):
    """list of integers -> (n,4) Montgomery limbs"""
    if not len(values):
        return np.zeros((0, 4), dtype=np.uint64)
    return np.stack([fr_from_int(v) for v in values])


def _pad(coeffs, n):
    assert len(coeffs) <= n
    return list(coeffs) + [0] * (n - len(coeffs))


def evaluate(coeffs, point):
    """DensePolynomial::evaluate (Horner)"""
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * point + c) % R_MOD
    return acc


def structured_scalar_power(num, s):
    """structured_scalar_message.rs:333-341: (1, s, s^2, ...)"""
    out, cur = [], 1
    for _ in range(num):
        out.append(cur); cur = cur * s % R_MOD
    return out


def _msm_affine(powers, coeffs):
    """VariableBaseMSM::msm over affine bases (mod.rs:86, 107, 231) -> G1 projective (18,)"""
    powers = api._c(powers, 12); sc = frs(_pad(coeffs, len(powers)))
    out = np.zeros(18, dtype=np.uint64)
    _check(lib().ripp_msm_g1_a(_p(powers), _p(sc), ctypes.c_size_t(len(powers)), _p(out)))
    return out


def _powers_g1(s, num):
    out = np.zeros((num, 18), dtype=np.uint64)
    _check(lib().ripp_srs_powers_g1(_p(fr_from_int(s)), ctypes.c_size_t(num), _p(out))); return out


def _powers_g2(s, num):
    out = np.zeros((num, 36), dtype=np.uint64)
    _check(lib().ripp_srs_powers_g2(_p(fr_from_int(s)), ctypes.c_size_t(num), _p(out))); return out


def _degree(coeffs):
    d = len(coeffs) - 1
    while d > 0 and coeffs[d] % R_MOD == 0:
        d -= 1
    return max(d, 0)


# ---------------------------------------------------------------- KZG
class KZG:
    """mod.rs:50-119"""

    @staticmethod
    def setup(alpha, beta, degree):
        """mod.rs:56-76 with (alpha, beta) given: (g^{alpha^i} for i <= degree, normalised; VerifierSRS)"""
        powers = api.normalize_batch_g1(_powers_g1(alpha, degree + 1))
        g = _powers_g1(1, 1)[0]; h = _powers_g2(1, 1)[0]
        v_srs = {"g": g, "h": h, "g_beta": _powers_g1(beta, 2)[1], "h_alpha": _powers_g2(alpha, 2)[1]}
        return powers, v_srs

    @staticmethod
    def commit(powers, polynomial):
        """mod.rs:78-88"""
        assert len(powers) >= _degree(polynomial) + 1
        return _msm_affine(powers, polynomial[:_degree(polynomial) + 1])

    @staticmethod
    def open(powers, polynomial, point):
        """mod.rs:90-109: the quotient p(X) / (X - z), remainder p(z) dropped; one MSM"""
        assert len(powers) >= _degree(polynomial) + 1
        d = _degree(polynomial)
        q = [0] * d; carry = 0
        for i in range(d, 0, -1):
            carry = (polynomial[i] + carry * point) % R_MOD
            q[i - 1] = carry
        return _msm_affine(powers, q)

    @staticmethod
    def verify(v_srs, com, point, eval, proof):
        """mod.rs:111-119: e(com - g*eval, h) == e(proof, h_alpha - h*point)"""
        lhs_g1 = G1.add(com, G1.mul(v_srs["g"], fr_from_int(-eval)))
        rhs_g2 = G2.add(v_srs["h_alpha"], G2.mul(v_srs["h"], fr_from_int(-point)))
        lhs = api.PairingInnerProduct.inner_product(lhs_g1[None], np.asarray(v_srs["h"], dtype=np.uint64)[None])
        rhs = api.PairingInnerProduct.inner_product(np.asarray(proof, dtype=np.uint64)[None], rhs_g2[None])
        return bool(np.array_equal(lhs, rhs))


# ---------------------------------------------------------------- bivariate
class BivariatePolynomial:
    """mod.rs:121-140: sum_i x^i * y_polynomials[i](y)"""

    def __init__(self, y_polynomials):
        self.y_polynomials = [list(p) for p in y_polynomials]

    def evaluate(self, point):
        x, y = point
        acc, xp = 0, 1
        for yp in self.y_polynomials:
            acc = (acc + xp * evaluate(yp, y)) % R_MOD; xp = xp * x % R_MOD
        return acc


def _padded_rows(bp, rows, cols):
    """the y-polynomials' coefficients, zero-padded to `rows` polynomials of `cols` coefficients (mod.rs:217-227)"""
    assert rows >= len(bp.y_polynomials)
    out = [_pad(yp[:_degree(yp) + 1], cols) for yp in bp.y_polynomials]
    return out + [[0] * cols for _ in range(rows - len(out))]


def _y_eval_coeffs(rows, powers_of_x, cols):
    """coefficient j of the partial evaluation p(x, Y): sum_i x^i coeffs[i][j] (mod.rs:228-234)"""
    return [sum(px * row[j] for px, row in zip(powers_of_x, rows)) % R_MOD for j in range(cols)]


class _PCSRS(api.SRS):
    """SRS { g_alpha_powers: vec![g], h_beta_powers, g_beta, h_alpha } (mod.rs:165-170): only the G2 side has powers.  The device
    handle wants both tables of one length; the unused G1 side is filled with g."""

    def __init__(self, h_beta_powers, g_beta, h_alpha):
        g = _powers_g1(1, 1)
        super().__init__(np.repeat(g, len(h_beta_powers), axis=0), h_beta_powers, g_beta, h_alpha)
        self.g_alpha_powers = g


class BivariatePolynomialCommitment:
    """mod.rs:142-296"""

    @staticmethod
    def setup(alpha, beta, x_degree, y_degree):
        """mod.rs:148-172: (SRS of the second-tier argument, KZG powers of the first tier)"""
        kzg_srs = api.normalize_batch_g1(_powers_g1(alpha, y_degree + 1))
        srs = _PCSRS(_powers_g2(beta, 2 * x_degree + 1), _powers_g1(beta, 2)[1], _powers_g2(alpha, 2)[1])
        return srs, kzg_srs

    @staticmethod
    def commit(srs, bivariate_polynomial):
        """mod.rs:174-196 -> (AFGHO commitment in GT, the KZG commitments of the y-polynomials (n,18))"""
        ip_srs, kzg_srs = srs
        ck, _ = ip_srs.get_commitment_keys()
        rows = _padded_rows(bivariate_polynomial, len(ck), len(kzg_srs))
        y_polynomial_coms = np.stack([_msm_affine(kzg_srs, row) for row in rows])
        return api.AFGHOCommitmentG1.commit(ck, y_polynomial_coms), y_polynomial_coms

    @staticmethod
    def open(srs, bivariate_polynomial, y_polynomial_comms, point):
        """mod.rs:198-263"""
        x, y = point
        ip_srs, kzg_srs = srs
        ck_1, _ = ip_srs.get_commitment_keys()
        powers_of_x = structured_scalar_power(len(ck_1), x)
        rows = _padded_rows(bivariate_polynomial, len(ck_1), len(kzg_srs))
        y_eval_coeffs = _y_eval_coeffs(rows, powers_of_x, len(kzg_srs))
        y_eval_comm = _msm_affine(kzg_srs, y_eval_coeffs)
        ip_proof = api.TIPAWithSSM.prove_with_structured_scalar_message(ip_srs, (y_polynomial_comms, frs(powers_of_x)), (ck_1,))
        kzg_proof = KZG.open(kzg_srs, y_eval_coeffs, y)
        return {"ip_proof": ip_proof, "y_eval_comm": y_eval_comm, "kzg_proof": kzg_proof}

    @staticmethod
    def verify(v_srs, com, point, eval, proof):
        """mod.rs:265-285"""
        x, y = point
        ip_proof_valid = api.TIPAWithSSM.verify_with_structured_scalar_message(v_srs, (com, proof["y_eval_comm"]), fr_from_int(x), proof["ip_proof"])
        kzg_proof_valid = KZG.verify(v_srs, proof["y_eval_comm"], y, eval, proof["kzg_proof"])
        return ip_proof_valid and kzg_proof_valid


# ---------------------------------------------------------------- univariate through the bivariate form
def _isqrt_ceil(v):
    import math
    s = math.isqrt(v)
    return s if s * s == v else s + 1


def _next_power_of_two(v):
    return 1 if v <= 1 else 1 << (v - 1).bit_length()


def bivariate_form(bivariate_degrees, polynomial):
    """mod.rs:316-338: coefficient k of p goes to y_polynomials[k / (y_degree+1)][k % (y_degree+1)]"""
    x_degree, y_degree = bivariate_degrees
    flat = _pad(list(polynomial)[:(x_degree + 1) * (y_degree + 1)], (x_degree + 1) * (y_degree + 1))
    return BivariatePolynomial([flat[i * (y_degree + 1):(i + 1) * (y_degree + 1)] for i in range(x_degree + 1)])


class UnivariatePolynomialCommitment:
    """mod.rs:298-388"""
    SKEW_FROM, SKEW = 32, 16          # mod.rs:304: KZG is cheaper than the pairing argument

    @classmethod
    def bivariate_degrees(cls, univariate_degree):
        """mod.rs:299-306"""
        sqrt = _next_power_of_two(_isqrt_ceil(univariate_degree + 1))
        skew_factor = cls.SKEW if sqrt >= cls.SKEW_FROM else sqrt // 2
        return sqrt // skew_factor - 1, sqrt * skew_factor - 1

    @staticmethod
    def parse_bivariate_degrees_from_srs(srs):
        """mod.rs:308-312"""
        return (len(srs[0].h_beta_powers) - 1) // 2, len(srs[1]) - 1

    @classmethod
    def setup(cls, alpha, beta, degree):
        return BivariatePolynomialCommitment.setup(alpha, beta, *cls.bivariate_degrees(degree))

    @classmethod
    def commit(cls, srs, polynomial):
        return BivariatePolynomialCommitment.commit(srs, bivariate_form(cls.parse_bivariate_degrees_from_srs(srs), polynomial))

    @classmethod
    def open(cls, srs, polynomial, y_polynomial_comms, point):
        """mod.rs:355-371: (x, y) = (z^(y_degree+1), z)"""
        x_degree, y_degree = cls.parse_bivariate_degrees_from_srs(srs)
        return BivariatePolynomialCommitment.open(srs, bivariate_form((x_degree, y_degree), polynomial), y_polynomial_comms,
                                                  (pow(point, y_degree + 1, R_MOD), point))

    @staticmethod
    def verify(v_srs, max_degree, com, point, eval, proof):
        """mod.rs:373-387"""
        _, y_degree = UnivariatePolynomialCommitment.bivariate_degrees(max_degree)
        return BivariatePolynomialCommitment.verify(v_srs, com, (pow(point, y_degree + 1, R_MOD), point), eval, proof)


from . import transparent  # noqa: E402,F401
