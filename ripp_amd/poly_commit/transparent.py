"""Transparent (no trusted setup) polynomial commitment -- ip_proofs/src/applications/poly_commit/transparent.rs mirrored over
ripp_amd.gipa:

    second tier  GIPAWithSSM<MultiexponentiationInnerProduct<G1>, AFGHOCommitmentG1, IdentityCommitment<G1, Fr>>   transparent.rs:28-33
    first tier   GIPAWithSSM<ScalarInnerProduct, PedersenCommitment<G1>, IdentityCommitment<Fr, Fr>>                transparent.rs:43-48

ck = (first_tier_ck: G1 generators (y_degree+1, 18), second_tier_ck: G2 generators (x_degree+1, 36)).  `setup` takes the generators'
seeds instead of an RNG (the reference draws random group elements, transparent.rs:95-103): key i is a fixed multiple of the group
generator -- deterministic for the parity tests; a deployment hashes to the curve instead.
Field elements are Python integers, group elements limb arrays, as in ripp_amd.poly_commit."""
import numpy as np

from .. import api
from ..gipa import (AFGHOCommitmentG1, Fr, G1, GIPAWithSSM, IdentityCommitment, MultiexpIPG1, PedersenCommitmentG1, R_MOD, ScalarIP,
                    fr_from_int)
from . import (BivariatePolynomial, _padded_rows, _y_eval_coeffs, bivariate_form, frs, structured_scalar_power, _isqrt_ceil,
               _next_power_of_two)

SecondTierIPA = GIPAWithSSM(MultiexpIPG1, AFGHOCommitmentG1, IdentityCommitment(G1))          # transparent.rs:28-33
FirstTierIPA = GIPAWithSSM(ScalarIP, PedersenCommitmentG1, IdentityCommitment(Fr))            # transparent.rs:43-48


_P_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
_FP_ONE = np.array([(((1 << 384) % _P_MOD) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)], dtype=np.uint64)     # Fp::one(), Montgomery form


def _jac_g1(a):
    """affine (n,12) -> projective (n,18) with Z = 1 (the synthetic keys are never the point at infinity)"""
    out = np.zeros((len(a), 18), dtype=np.uint64); out[:, :12] = a; out[:, 12:] = _FP_ONE; return out


def _jac_g2(a):
    out = np.zeros((len(a), 36), dtype=np.uint64); out[:, :24] = a; out[:, 24:30] = _FP_ONE; return out


class BivariatePolynomialCommitment:
    """transparent.rs:86-269"""

    @staticmethod
    def setup(seed_g1, seed_g2, x_degree, y_degree):
        """transparent.rs:92-100: (first_tier_ck (y_degree+1) in G1, second_tier_ck (x_degree+1) in G2)"""
        return _jac_g1(api.synth_g1(seed_g1, y_degree + 1)), _jac_g2(api.synth_g2(seed_g2, x_degree + 1))

    @staticmethod
    def commit(ck, bivariate_polynomial):
        """transparent.rs:102-129: Pedersen commitments of the y-polynomials, AFGHO commitment of those"""
        first_tier_ck, second_tier_ck = ck
        rows = _padded_rows(bivariate_polynomial, len(second_tier_ck), len(first_tier_ck))
        y_polynomial_coms = np.stack([PedersenCommitmentG1.commit(first_tier_ck, frs(row)) for row in rows])
        return AFGHOCommitmentG1.commit(second_tier_ck, y_polynomial_coms), y_polynomial_coms

    @staticmethod
    def open(ck, bivariate_polynomial, y_polynomial_comms, point):
        """transparent.rs:131-195"""
        x, y = point
        first_tier_ck, second_tier_ck = ck
        powers_of_x = structured_scalar_power(len(second_tier_ck), x)
        rows = _padded_rows(bivariate_polynomial, len(second_tier_ck), len(first_tier_ck))
        y_eval_coeffs = _y_eval_coeffs(rows, powers_of_x, len(first_tier_ck))
        y_eval_comm = PedersenCommitmentG1.commit(first_tier_ck, frs(y_eval_coeffs))
        second = SecondTierIPA.prove_with_structured_scalar_message((np.asarray(y_polynomial_comms), frs(powers_of_x)), (second_tier_ck, None))
        powers_of_y = structured_scalar_power(len(first_tier_ck), y)
        first = FirstTierIPA.prove_with_structured_scalar_message((frs(y_eval_coeffs), frs(powers_of_y)), (first_tier_ck, None))
        return {"second_tier_ip_proof": second, "y_eval_comm": y_eval_comm, "first_tier_ip_proof": first}

    @staticmethod
    def verify(ck, com, point, eval, proof):
        """transparent.rs:197-225"""
        first_tier_ck, second_tier_ck = ck
        x, y = point
        second_ok = SecondTierIPA.verify_with_structured_scalar_message((second_tier_ck, None), (com, [proof["y_eval_comm"]]), fr_from_int(x),
                                                                       proof["second_tier_ip_proof"])
        first_ok = FirstTierIPA.verify_with_structured_scalar_message((first_tier_ck, None), (proof["y_eval_comm"], [fr_from_int(eval)]), fr_from_int(y),
                                                                     proof["first_tier_ip_proof"])
        return bool(second_ok and first_ok)


class UnivariatePolynomialCommitment:
    """transparent.rs:227-330"""

    @staticmethod
    def bivariate_degrees(univariate_degree):
        """transparent.rs:233-239: the scalar argument is cheaper than the multi-exponentiation one -- skew 4 from sqrt >= 8"""
        sqrt = _next_power_of_two(_isqrt_ceil(univariate_degree + 1))
        skew_factor = 4 if sqrt >= 8 else sqrt // 2
        return sqrt // skew_factor - 1, sqrt * skew_factor - 1

    @staticmethod
    def parse_bivariate_degrees_from_ck(ck):
        return len(ck[1]) - 1, len(ck[0]) - 1

    @classmethod
    def setup(cls, seed_g1, seed_g2, degree):
        return BivariatePolynomialCommitment.setup(seed_g1, seed_g2, *cls.bivariate_degrees(degree))

    @classmethod
    def commit(cls, ck, polynomial):
        return BivariatePolynomialCommitment.commit(ck, bivariate_form(cls.parse_bivariate_degrees_from_ck(ck), polynomial))

    @classmethod
    def open(cls, ck, polynomial, y_polynomial_comms, point):
        x_degree, y_degree = cls.parse_bivariate_degrees_from_ck(ck)
        return BivariatePolynomialCommitment.open(ck, bivariate_form((x_degree, y_degree), polynomial), y_polynomial_comms,
                                                  (pow(point, y_degree + 1, R_MOD), point))

    @classmethod
    def verify(cls, ck, com, point, eval, proof):
        _, y_degree = cls.parse_bivariate_degrees_from_ck(ck)
        return BivariatePolynomialCommitment.verify(ck, com, (pow(point, y_degree + 1, R_MOD), point), eval, proof)
