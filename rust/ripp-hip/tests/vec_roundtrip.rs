//! Device-resident vectors against arkworks: upload -> fold -> download must be the reference's `mul_helper` fold (ip_proofs/src/gipa.rs:262-290),
//! element for element, and a download through the wrong type must be refused instead of overrunning its buffer.
//! Needs an MI355X and libripp_hip.so (`cargo test --release`); like the rest of this crate it has never been through a compiler in the build image.
use ark_bls12_381::{Fr, G1Projective, G2Projective};
use ark_ec::CurveGroup;
use ark_ff::UniformRand;
use ripp_hip::fused::HipVec;

#[test]
fn upload_fold_download_equals_arkworks() {
    let mut rng = ark_std::test_rng();
    let n = 64usize;
    let g1: Vec<G1Projective> = (0..n).map(|_| G1Projective::rand(&mut rng)).collect();
    let g2: Vec<G2Projective> = (0..n).map(|_| G2Projective::rand(&mut rng)).collect();
    let sc: Vec<Fr> = (0..n).map(|_| Fr::rand(&mut rng)).collect();
    let x = Fr::rand(&mut rng);

    let v1 = HipVec::upload_g1(&g1).unwrap();
    let (lo, hi) = v1.halves().unwrap();
    let folded = HipVec::fold(&hi, &lo, &x).unwrap();
    let expect: Vec<G1Projective> = (0..n / 2).map(|i| g1[n / 2 + i] * x + g1[i]).collect();
    assert_eq!(G1Projective::normalize_batch(&folded.download_g1().unwrap()), G1Projective::normalize_batch(&expect));
    assert_eq!(folded.download_g1a().unwrap(), G1Projective::normalize_batch(&expect));

    let v2 = HipVec::upload_g2(&g2).unwrap();
    assert_eq!(v2.download_g2a().unwrap(), G2Projective::normalize_batch(&g2));
    let vs = HipVec::upload_fr(&sc).unwrap();
    assert_eq!(vs.download_fr().unwrap(), sc);

    // the wrong type for the vector's kind: an error, not 96 n bytes written into a 32 n-byte buffer
    assert!(v1.download_fr().is_err() && v1.download_g2a().is_err() && v2.download_g1().is_err() && vs.download_g1a().is_err());
}
