//! MI355X backend for arkworks-rs/ripp behind the reference's own trait surface.
//!
//! * `InnerProduct` (inner_products/src/lib.rs:40-49): `HipPairingInnerProduct`, `HipMultiexpInnerProductG1/G2`, `HipScalarInnerProduct`
//! * `DoublyHomomorphicCommitment` (dh_commitments/src/lib.rs:20-55): `HipAFGHOCommitmentG1/G2`, `HipPedersenCommitmentG1/G2`
//! * `TIPACompatibleSetup` (ip_proofs/src/tipa/mod.rs:25-29) for all of them
//! * `sipp`: `hip_sipp_prove` / `hip_sipp_verify` / `hip_product_of_pairings[_with_coeffs]` (sipp/src/lib.rs:42-224)
//! * `fused`: the library's one-call provers returning the REFERENCE's proof types -- `hip_tipa_prove_with_srs_shift`, `hip_tipa_ssm_prove`,
//!   `hip_aggregate_proofs[_sharded]`, the verifiers, `HipSrs`, `HipVec`, `HipSippJob`, `hip_sipp_prove_sharded` (fused.rs)
//!
//! GIPA / TIPA are generic over the two traits (ip_proofs/src/gipa.rs:79-96), so
//! `GIPA<HipPairingInnerProduct, HipAFGHOCommitmentG1, HipAFGHOCommitmentG2, IdentityCommitment<..>, Blake2b>` runs every pairing
//! product and MSM of the prover on the GPU with no change to the reference crates.  The traits are static (no `&self`), so the engine
//! is the library's process-global context (`ripp_init`, lazily created on first use).
#![cfg(feature = "ffi")]
pub mod convert;
pub mod ffi;
pub mod fused;
pub use fused::*;

use ark_bls12_381::{Bls12_381, Fr, G1Affine, G1Projective, G2Affine, G2Projective};
use ark_dh_commitments::{random_generators, DoublyHomomorphicCommitment};
use ark_ec::pairing::PairingOutput;
use ark_inner_products::{Error, InnerProduct, InnerProductError};
use ark_ip_proofs::tipa::TIPACompatibleSetup;
use ark_std::rand::Rng;
use convert::*;
use ffi::*;

fn status(rc: i32, left: usize, right: usize) -> Result<(), Error> {
    match rc {
        RIPP_OK => Ok(()),
        RIPP_ERR_LENGTH => Err(Box::new(InnerProductError::MessageLengthInvalid(left, right))),      // inner_products/src/lib.rs:65-70
        _ => {
            let msg = unsafe { std::ffi::CStr::from_ptr(ripp_last_error()) }.to_string_lossy().into_owned();
            Err(format!("libripp_hip status {rc}: {msg}").into())
        }
    }
}

#[derive(Copy, Clone)] pub struct HipPairingInnerProduct;
impl InnerProduct for HipPairingInnerProduct {
    type LeftMessage = G1Projective; type RightMessage = G2Projective; type Output = PairingOutput<Bls12_381>;
    fn inner_product(left: &[G1Projective], right: &[G2Projective]) -> Result<Self::Output, Error> {
        let l: Vec<RippG1J> = left.iter().map(g1j).collect();
        let r: Vec<RippG2J> = right.iter().map(g2j).collect();
        let mut out = RippGt::default();
        status(unsafe { ripp_pairing_product_j(l.as_ptr(), l.len(), r.as_ptr(), r.len(), &mut out) }, left.len(), right.len())?;
        Ok(un_gt(&out))
    }
}
#[derive(Copy, Clone)] pub struct HipMultiexpInnerProductG1;
impl InnerProduct for HipMultiexpInnerProductG1 {
    type LeftMessage = G1Projective; type RightMessage = Fr; type Output = G1Projective;
    fn inner_product(left: &[G1Projective], right: &[Fr]) -> Result<G1Projective, Error> {
        let l: Vec<RippG1J> = left.iter().map(g1j).collect();
        let r: Vec<RippFr> = right.iter().map(fr).collect();
        let mut out = RippG1J::default();
        status(unsafe { ripp_msm_g1_j(l.as_ptr(), l.len(), r.as_ptr(), r.len(), &mut out) }, left.len(), right.len())?;
        Ok(un_g1j(&out))
    }
}
#[derive(Copy, Clone)] pub struct HipMultiexpInnerProductG2;
impl InnerProduct for HipMultiexpInnerProductG2 {
    type LeftMessage = G2Projective; type RightMessage = Fr; type Output = G2Projective;
    fn inner_product(left: &[G2Projective], right: &[Fr]) -> Result<G2Projective, Error> {
        let l: Vec<RippG2J> = left.iter().map(g2j).collect();
        let r: Vec<RippFr> = right.iter().map(fr).collect();
        let mut out = RippG2J::default();
        status(unsafe { ripp_msm_g2_j(l.as_ptr(), l.len(), r.as_ptr(), r.len(), &mut out) }, left.len(), right.len())?;
        Ok(un_g2j(&out))
    }
}
#[derive(Copy, Clone)] pub struct HipScalarInnerProduct;
impl InnerProduct for HipScalarInnerProduct {
    type LeftMessage = Fr; type RightMessage = Fr; type Output = Fr;
    fn inner_product(left: &[Fr], right: &[Fr]) -> Result<Fr, Error> {
        let (l, r): (Vec<RippFr>, Vec<RippFr>) = (left.iter().map(fr).collect(), right.iter().map(fr).collect());
        let mut out = RippFr::default();
        status(unsafe { ripp_scalar_inner_product(l.as_ptr(), l.len(), r.as_ptr(), r.len(), &mut out) }, left.len(), right.len())?;
        Ok(un_fr(&out))
    }
}

/// dh_commitments/src/afgho16/mod.rs:20-32: Message G1, Key G2, commit(k, m) = IP(m, k)
#[derive(Clone)] pub struct HipAFGHOCommitmentG1;
impl DoublyHomomorphicCommitment for HipAFGHOCommitmentG1 {
    type Scalar = Fr; type Message = G1Projective; type Key = G2Projective; type Output = PairingOutput<Bls12_381>;
    fn setup<R: Rng>(rng: &mut R, size: usize) -> Result<Vec<Self::Key>, Error> { Ok(random_generators(rng, size)) }
    fn commit(k: &[Self::Key], m: &[Self::Message]) -> Result<Self::Output, Error> { HipPairingInnerProduct::inner_product(m, k) }
}
/// afgho16/mod.rs:35-47: Message G2, Key G1, commit(k, m) = IP(k, m)
#[derive(Clone)] pub struct HipAFGHOCommitmentG2;
impl DoublyHomomorphicCommitment for HipAFGHOCommitmentG2 {
    type Scalar = Fr; type Message = G2Projective; type Key = G1Projective; type Output = PairingOutput<Bls12_381>;
    fn setup<R: Rng>(rng: &mut R, size: usize) -> Result<Vec<Self::Key>, Error> { Ok(random_generators(rng, size)) }
    fn commit(k: &[Self::Key], m: &[Self::Message]) -> Result<Self::Output, Error> { HipPairingInnerProduct::inner_product(k, m) }
}
/// pedersen/mod.rs:14-26
#[derive(Clone)] pub struct HipPedersenCommitmentG1;
impl DoublyHomomorphicCommitment for HipPedersenCommitmentG1 {
    type Scalar = Fr; type Message = Fr; type Key = G1Projective; type Output = G1Projective;
    fn setup<R: Rng>(rng: &mut R, size: usize) -> Result<Vec<Self::Key>, Error> { Ok(random_generators(rng, size)) }
    fn commit(k: &[Self::Key], m: &[Self::Message]) -> Result<Self::Output, Error> { HipMultiexpInnerProductG1::inner_product(k, m) }
}
#[derive(Clone)] pub struct HipPedersenCommitmentG2;
impl DoublyHomomorphicCommitment for HipPedersenCommitmentG2 {
    type Scalar = Fr; type Message = Fr; type Key = G2Projective; type Output = G2Projective;
    fn setup<R: Rng>(rng: &mut R, size: usize) -> Result<Vec<Self::Key>, Error> { Ok(random_generators(rng, size)) }
    fn commit(k: &[Self::Key], m: &[Self::Message]) -> Result<Self::Output, Error> { HipMultiexpInnerProductG2::inner_product(k, m) }
}
impl TIPACompatibleSetup for HipAFGHOCommitmentG1 {}
impl TIPACompatibleSetup for HipAFGHOCommitmentG2 {}
impl TIPACompatibleSetup for HipPedersenCommitmentG1 {}
impl TIPACompatibleSetup for HipPedersenCommitmentG2 {}

// ---- sipp crate counterparts (sipp/src/lib.rs) ---------------------------------------------------------------------------------
/// `product_of_pairings(a, b)` (sipp/src/lib.rs:219-224)
pub fn hip_product_of_pairings(a: &[G1Affine], b: &[G2Affine]) -> Result<PairingOutput<Bls12_381>, Error> {
    let (la, lb): (Vec<RippG1A>, Vec<RippG2A>) = (a.iter().map(g1a).collect(), b.iter().map(g2a).collect());
    if la.len() != lb.len() { return Err(Box::new(InnerProductError::MessageLengthInvalid(la.len(), lb.len()))); }
    let mut out = RippGt::default();
    status(unsafe { ripp_pairing_product_a(la.as_ptr(), lb.as_ptr(), la.len(), &mut out) }, a.len(), b.len())?;
    Ok(un_gt(&out))
}
/// `product_of_pairings_with_coeffs(a, b, r)` (sipp/src/lib.rs:184-217)
pub fn hip_product_of_pairings_with_coeffs(a: &[G1Affine], b: &[G2Affine], r: &[Fr]) -> Result<PairingOutput<Bls12_381>, Error> {
    assert!(a.len() == b.len() && a.len() == r.len());
    let (la, lb, lr): (Vec<RippG1A>, Vec<RippG2A>, Vec<RippFr>) = (a.iter().map(g1a).collect(), b.iter().map(g2a).collect(), r.iter().map(fr).collect());
    let mut out = RippGt::default();
    status(unsafe { ripp_pairing_product_coeffs_a(la.as_ptr(), lb.as_ptr(), lr.as_ptr(), la.len(), &mut out) }, a.len(), b.len())?;
    Ok(un_gt(&out))
}
/// `SIPP::<Bls12_381, Blake2s>::prove` (sipp/src/lib.rs:42-106): returns `Proof::gt_elems` = (z_l, z_r) per round
pub fn hip_sipp_prove(a: &[G1Affine], b: &[G2Affine], r: &[Fr], value: PairingOutput<Bls12_381>) -> Result<Vec<(PairingOutput<Bls12_381>, PairingOutput<Bls12_381>)>, Error> {
    assert_eq!(a.len(), b.len()); assert_eq!(a.len(), r.len());                      // :48-49
    assert!(a.len().is_power_of_two());                                              // :50-53
    let rounds = a.len().trailing_zeros() as usize;
    let (la, lb, lr): (Vec<RippG1A>, Vec<RippG2A>, Vec<RippFr>) = (a.iter().map(g1a).collect(), b.iter().map(g2a).collect(), r.iter().map(fr).collect());
    let mut proof = vec![RippGt::default(); 2 * rounds];
    let v = gt(&value);
    status(unsafe { ripp_sipp_prove(la.as_ptr(), lb.as_ptr(), lr.as_ptr(), la.len(), &v, proof.as_mut_ptr(), core::ptr::null_mut(), core::ptr::null_mut()) }, 0, 0)?;
    Ok(proof.chunks(2).map(|p| (un_gt(&p[0]), un_gt(&p[1]))).collect())
}
/// `SIPP::verify` (sipp/src/lib.rs:109-180)
pub fn hip_sipp_verify(a: &[G1Affine], b: &[G2Affine], r: &[Fr], claimed: PairingOutput<Bls12_381>, proof: &[(PairingOutput<Bls12_381>, PairingOutput<Bls12_381>)]) -> Result<bool, Error> {
    assert_eq!(a.len(), b.len()); assert_eq!(a.len(), r.len());
    let (la, lb, lr): (Vec<RippG1A>, Vec<RippG2A>, Vec<RippFr>) = (a.iter().map(g1a).collect(), b.iter().map(g2a).collect(), r.iter().map(fr).collect());
    let flat: Vec<RippGt> = proof.iter().flat_map(|(l, r)| [gt(l), gt(r)]).collect();
    let mut accept = 0i32;
    status(unsafe { ripp_sipp_verify(la.as_ptr(), lb.as_ptr(), lr.as_ptr(), la.len(), &gt(&claimed), flat.as_ptr(), proof.len(), &mut accept) }, 0, 0)?;
    Ok(accept == 1)
}

#[cfg(test)]
mod tests {
    //! The reference's own tests, re-instantiated with the Hip types (dh_commitments/src/afgho16/mod.rs:61-94, ip_proofs/src/gipa.rs:470-497).
    use super::*;
    use ark_dh_commitments::identity::{HomomorphicPlaceholderValue, IdentityCommitment, IdentityOutput};
    use ark_ff::UniformRand;
    use ark_ip_proofs::gipa::GIPA;
    use ark_std::rand::{rngs::StdRng, SeedableRng};
    use blake2::Blake2b;

    type GC1 = HipAFGHOCommitmentG1; type GC2 = HipAFGHOCommitmentG2;
    type IPC = IdentityCommitment<PairingOutput<Bls12_381>, Fr>;
    type PairingGIPA = GIPA<HipPairingInnerProduct, GC1, GC2, IPC, Blake2b>;
    const TEST_SIZE: usize = 8;

    #[test]
    fn hip_matches_arkworks_on_random_vectors() {
        let mut rng = StdRng::seed_from_u64(0u64);
        let l: Vec<G1Projective> = (0..33).map(|_| G1Projective::rand(&mut rng)).collect();
        let r: Vec<G2Projective> = (0..33).map(|_| G2Projective::rand(&mut rng)).collect();
        let s: Vec<Fr> = (0..33).map(|_| Fr::rand(&mut rng)).collect();
        assert_eq!(HipPairingInnerProduct::inner_product(&l, &r).unwrap(), ark_inner_products::PairingInnerProduct::<Bls12_381>::inner_product(&l, &r).unwrap());
        assert_eq!(HipMultiexpInnerProductG1::inner_product(&l, &s).unwrap(), ark_inner_products::MultiexponentiationInnerProduct::<G1Projective>::inner_product(&l, &s).unwrap());
        assert_eq!(HipMultiexpInnerProductG2::inner_product(&r, &s).unwrap(), ark_inner_products::MultiexponentiationInnerProduct::<G2Projective>::inner_product(&r, &s).unwrap());
        assert!(HipPairingInnerProduct::inner_product(&l[..5], &r[..4]).is_err());
    }

    /// the fused TIPA prover's proof is a `TIPAProof` of the reference and its UNMODIFIED verifier accepts it (tipa/mod.rs:242-301; the test
    /// mirrors ip_proofs/src/tipa/mod.rs `pairing_inner_product_test`)
    #[test]
    fn fused_tipa_proof_is_accepted_by_the_reference_verifier() {
        use ark_dh_commitments::afgho16::{AFGHOCommitmentG1, AFGHOCommitmentG2};
        use ark_inner_products::PairingInnerProduct;
        use ark_ip_proofs::tipa::TIPA;
        type RefIPC = IdentityCommitment<PairingOutput<Bls12_381>, Fr>;
        type RefTIPA = TIPA<PairingInnerProduct<Bls12_381>, AFGHOCommitmentG1<Bls12_381>, AFGHOCommitmentG2<Bls12_381>, RefIPC, Bls12_381, Blake2b>;
        let mut rng = StdRng::seed_from_u64(0u64);
        let (srs, ck_t) = RefTIPA::setup(&mut rng, TEST_SIZE).unwrap();
        let (ck_a, ck_b) = srs.get_commitment_keys();
        let v_srs = srs.get_verifier_key();
        let m_a: Vec<G1Projective> = random_generators(&mut rng, TEST_SIZE);
        let m_b: Vec<G2Projective> = random_generators(&mut rng, TEST_SIZE);
        let com_a = HipAFGHOCommitmentG1::commit(&ck_a, &m_a).unwrap();
        let com_b = HipAFGHOCommitmentG2::commit(&ck_b, &m_b).unwrap();
        let t = vec![HipPairingInnerProduct::inner_product(&m_a, &m_b).unwrap()];
        let com_t = RefIPC::commit(&vec![ck_t.clone()], &t).unwrap();
        let hsrs = HipSrs::new(&srs).unwrap();
        let proof = hip_tipa_prove(&hsrs, (&m_a, &m_b), (&ck_a, &ck_b)).unwrap();
        assert!(RefTIPA::verify(&v_srs, &ck_t, (&com_a, &com_b, &com_t), &proof).unwrap());
        assert!(hip_tipa_verify_with_srs_shift(&v_srs, (&com_a, &com_b, &com_t.0[0]), &proof, &ark_ff::One::one()).unwrap());
    }

    #[test]
    fn pairing_inner_product_gipa_on_the_gpu() {
        let mut rng = StdRng::seed_from_u64(0u64);
        let (ck_a, ck_b, ck_t) = PairingGIPA::setup(&mut rng, TEST_SIZE).unwrap();
        let m_a = random_generators(&mut rng, TEST_SIZE);
        let m_b = random_generators(&mut rng, TEST_SIZE);
        let com_a = GC1::commit(&ck_a, &m_a).unwrap();
        let com_b = GC2::commit(&ck_b, &m_b).unwrap();
        let t = vec![HipPairingInnerProduct::inner_product(&m_a, &m_b).unwrap()];
        let com_t = IPC::commit(&vec![ck_t.clone()], &t).unwrap();
        let proof = PairingGIPA::prove((&m_a, &m_b, &t[0]), (&ck_a, &ck_b, &ck_t), (&com_a, &com_b, &com_t)).unwrap();
        assert!(PairingGIPA::verify((&ck_a, &ck_b, &ck_t), (&com_a, &com_b, &com_t), &proof).unwrap());
        let _ = (HomomorphicPlaceholderValue, IdentityOutput::<Fr>(vec![]));
    }
}
