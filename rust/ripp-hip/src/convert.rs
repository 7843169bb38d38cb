//! arkworks value <-> flat C-ABI struct.  arkworks structs are not `repr(C)`: limbs are COPIED field by field, never transmuted.
//! `Fp<MontBackend<_, N>, N>(pub BigInt<N>, PhantomData)` holds the Montgomery representation in `.0 .0` ([u64; N]); `new_unchecked`
//! takes that representation back without a reduction.
use crate::ffi::*;
use ark_bls12_381::{Bls12_381, Fq, Fq12, Fq2, Fq6, Fr, G1Affine, G1Projective, G2Affine, G2Projective};
use ark_ec::pairing::PairingOutput;
use ark_ec::AffineRepr;
use ark_ff::{BigInt, Zero};

pub fn fp(x: &Fq) -> RippFp { RippFp { l: x.0 .0 } }
pub fn fr(x: &Fr) -> RippFr { RippFr { l: x.0 .0 } }
pub fn fp2(x: &Fq2) -> RippFp2 { RippFp2 { c0: fp(&x.c0), c1: fp(&x.c1) } }
pub fn un_fp(x: &RippFp) -> Fq { Fq::new_unchecked(BigInt(x.l)) }
pub fn un_fr(x: &RippFr) -> Fr { Fr::new_unchecked(BigInt(x.l)) }
pub fn un_fp2(x: &RippFp2) -> Fq2 { Fq2::new(un_fp(&x.c0), un_fp(&x.c1)) }

/// Jacobian (X, Y, Z) as `short_weierstrass::Projective` stores it; Z = 0 is the identity on both sides.
pub fn g1j(p: &G1Projective) -> RippG1J { RippG1J { x: fp(&p.x), y: fp(&p.y), z: fp(&p.z) } }
pub fn g2j(p: &G2Projective) -> RippG2J { RippG2J { x: fp2(&p.x), y: fp2(&p.y), z: fp2(&p.z) } }
pub fn un_g1j(p: &RippG1J) -> G1Projective { let z = un_fp(&p.z); if z.is_zero() { G1Projective::zero() } else { G1Projective::new_unchecked(un_fp(&p.x), un_fp(&p.y), z) } }
pub fn un_g2j(p: &RippG2J) -> G2Projective { let z = un_fp2(&p.z); if z.is_zero() { G2Projective::zero() } else { G2Projective::new_unchecked(un_fp2(&p.x), un_fp2(&p.y), z) } }
/// affine: the C ABI encodes the identity as (0, 0)
pub fn g1a(p: &G1Affine) -> RippG1A { if p.infinity { RippG1A::default() } else { RippG1A { x: fp(&p.x), y: fp(&p.y) } } }
pub fn g2a(p: &G2Affine) -> RippG2A { if p.infinity { RippG2A::default() } else { RippG2A { x: fp2(&p.x), y: fp2(&p.y) } } }
pub fn un_g1a(p: &RippG1A) -> G1Affine { let (x, y) = (un_fp(&p.x), un_fp(&p.y)); if x.is_zero() && y.is_zero() { G1Affine::identity() } else { G1Affine::new_unchecked(x, y) } }
pub fn un_g2a(p: &RippG2A) -> G2Affine { let (x, y) = (un_fp2(&p.x), un_fp2(&p.y)); if x.is_zero() && y.is_zero() { G2Affine::identity() } else { G2Affine::new_unchecked(x, y) } }

pub fn gt(f: &PairingOutput<Bls12_381>) -> RippGt {
    let q = &f.0;
    RippGt { c: [fp2(&q.c0.c0), fp2(&q.c0.c1), fp2(&q.c0.c2), fp2(&q.c1.c0), fp2(&q.c1.c1), fp2(&q.c1.c2)] }
}
pub fn un_gt(f: &RippGt) -> PairingOutput<Bls12_381> {
    PairingOutput(Fq12::new(Fq6::new(un_fp2(&f.c[0]), un_fp2(&f.c[1]), un_fp2(&f.c[2])), Fq6::new(un_fp2(&f.c[3]), un_fp2(&f.c[4]), un_fp2(&f.c[5]))))
}
