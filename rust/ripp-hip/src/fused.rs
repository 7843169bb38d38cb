//! The library's FUSED provers behind the reference's own types.
//!
//! The trait-level path (`GIPA<HipPairingInnerProduct, ..>`, lib.rs) re-uploads host slices on every `commit` / `inner_product` call: at
//! n = 2^10 a TIPP proof takes 520 ms that way and 41 ms through `ripp_tipa_tipp_prove`, which keeps the four vectors on the device for
//! all log2 n rounds.  This module binds those entry points and hands back the REFERENCE's proof types:
//! `TIPAProof` / `TIPAWithSSMProof` derive `CanonicalDeserialize` (ip_proofs/src/tipa/mod.rs:41, structured_scalar_message.rs:138), so the
//! library's `CanonicalSerialize` image (`ripp_ser_tipa_*`, wire.hpp -- byte-identical to ark-serialize's, tests/test_wire_cpu.py) is all
//! that is needed to build them, private fields notwithstanding.  The proofs below are therefore accepted by the unmodified
//! `TIPA::verify_with_srs_shift` / `TIPAWithSSM::verify_with_structured_scalar_message` of the reference.
//!
//! * `HipSrs`                      -- `SRS<Bls12_381>` resident in HBM (`ripp_srs_*`; tipa/mod.rs:94-118)
//! * `hip_tipa_prove_with_srs_shift`, `hip_tipa_prove`                    -- tipa/mod.rs:166-231
//! * `hip_tipa_ssm_prove`                                                  -- structured_scalar_message.rs:211-268
//! * `hip_tipa_verify_with_srs_shift`, `hip_tipa_ssm_verify`               -- tipa/mod.rs:242-301, structured_scalar_message.rs:270-331
//! * `hip_aggregate_proofs`, `hip_verify_aggregate_proof`                  -- applications/groth16_aggregation.rs:77-231
//! * `HipVec`                       -- RAII handle of a device-resident vector (`ripp_vec_*`): upload once, halve / fold / multiply on views
//! * `HipSippJob`, `hip_sipp_prove_sharded`                                -- sipp/src/lib.rs:42-106 on a resident statement / across ranks
//!
//! SOURCE ONLY: this image has no Rust toolchain; the crate has never been through a compiler.
use crate::convert::*;
use crate::ffi::*;
use ark_bls12_381::{Bls12_381, Fr, G1Affine, G1Projective, G2Affine, G2Projective};
use ark_dh_commitments::{
    afgho16::{AFGHOCommitmentG1, AFGHOCommitmentG2},
    identity::IdentityCommitment,
};
use ark_ec::pairing::PairingOutput;
use ark_groth16::{Proof, VerifyingKey};
use ark_inner_products::{Error, MultiexponentiationInnerProduct, PairingInnerProduct};
use ark_ip_proofs::tipa::{structured_scalar_message::TIPAWithSSMProof, TIPAProof, VerifierSRS, SRS};
use ark_serialize::{CanonicalDeserialize, CanonicalSerialize, Compress, Validate};
use blake2::Blake2b;
use core::ptr::{null, null_mut};

type GT = PairingOutput<Bls12_381>;
/// `PairingInnerProductABProof<Bls12_381, Blake2b>` (groth16_aggregation.rs:33-40; the alias is private there)
pub type TippProof = TIPAProof<PairingInnerProduct<Bls12_381>, AFGHOCommitmentG1<Bls12_381>, AFGHOCommitmentG2<Bls12_381>, IdentityCommitment<GT, Fr>, Bls12_381, Blake2b>;
/// `MultiExpInnerProductCProof<Bls12_381, Blake2b>` (groth16_aggregation.rs:50-56)
pub type SsmProof = TIPAWithSSMProof<MultiexponentiationInnerProduct<G1Projective>, AFGHOCommitmentG1<Bls12_381>, IdentityCommitment<G1Projective, Fr>, Bls12_381, Blake2b>;

fn check(rc: i32) -> Result<(), Error> {
    if rc == RIPP_OK { return Ok(()); }
    let msg = unsafe { std::ffi::CStr::from_ptr(ripp_last_error()) }.to_string_lossy().into_owned();
    Err(format!("libripp_hip status {rc}: {msg}").into())
}
fn rounds_of(n: usize) -> usize { assert!(n.is_power_of_two() && n >= 2, "TIPA needs a power-of-two length >= 2"); n.trailing_zeros() as usize }

// ---- SRS resident in HBM ---------------------------------------------------------------------------------------------------------------
/// `SRS<Bls12_381>` (tipa/mod.rs:94-102) normalised once and kept on the device; provers take the commitment keys and the KZG bases from it.
pub struct HipSrs { h: *mut RippSrs, pub n: usize, pub g_beta: G1Projective, pub h_alpha: G2Projective, g: G1Projective, hh: G2Projective }
impl HipSrs {
    pub fn new(srs: &SRS<Bls12_381>) -> Result<Self, Error> {
        assert_eq!(srs.g_alpha_powers.len(), srs.h_beta_powers.len());
        let ga: Vec<RippG1J> = srs.g_alpha_powers.iter().map(g1j).collect();
        let hb: Vec<RippG2J> = srs.h_beta_powers.iter().map(g2j).collect();
        let mut h: *mut RippSrs = null_mut();
        check(unsafe { ripp_srs_create(ga.as_ptr(), hb.as_ptr(), ga.len(), &mut h) })?;
        Ok(HipSrs { h, n: (ga.len() + 1) / 2, g_beta: srs.g_beta, h_alpha: srs.h_alpha, g: srs.g_alpha_powers[0], hh: srs.h_beta_powers[0] })
    }
    /// `SRS::get_commitment_keys` (tipa/mod.rs:114-118)
    pub fn get_commitment_keys(&self) -> Result<(Vec<G2Projective>, Vec<G1Projective>), Error> {
        let (mut ck1, mut ck2) = (vec![RippG2J::default(); self.n], vec![RippG1J::default(); self.n]);
        check(unsafe { ripp_srs_commitment_keys(self.h, ck1.as_mut_ptr(), ck2.as_mut_ptr()) })?;
        Ok((ck1.iter().map(un_g2j).collect(), ck2.iter().map(un_g1j).collect()))
    }
    /// `SRS::get_verifier_key` (tipa/mod.rs:120-127)
    pub fn get_verifier_key(&self) -> VerifierSRS<Bls12_381> { VerifierSRS { g: self.g, h: self.hh, g_beta: self.g_beta, h_alpha: self.h_alpha } }
    pub fn raw(&self) -> *const RippSrs { self.h }
}
impl Drop for HipSrs { fn drop(&mut self) { unsafe { ripp_srs_destroy(self.h) } } }
/// `structured_generators_scalar_power(num, &G1::generator(), s)` (tipa/mod.rs:372-391) on the device
pub fn hip_srs_powers_g1(num: usize, s: &Fr) -> Result<Vec<G1Projective>, Error> {
    let mut out = vec![RippG1J::default(); num];
    check(unsafe { ripp_srs_powers_g1(&fr(s), num, out.as_mut_ptr()) })?;
    Ok(out.iter().map(un_g1j).collect())
}
pub fn hip_srs_powers_g2(num: usize, s: &Fr) -> Result<Vec<G2Projective>, Error> {
    let mut out = vec![RippG2J::default(); num];
    check(unsafe { ripp_srs_powers_g2(&fr(s), num, out.as_mut_ptr()) })?;
    Ok(out.iter().map(un_g2j).collect())
}

// ---- TIPA (TIPP instantiation) ----------------------------------------------------------------------------------------------------------
/// flat outputs of `ripp_tipa_tipp_prove` / inputs of `ripp_tipa_tipp_verify`
pub struct TippParts { pub com_steps: Vec<RippGt>, pub rounds: usize, pub base_a: RippG1J, pub base_b: RippG2J, pub final_ck_a: RippG2J, pub final_ck_b: RippG1J, pub opening_a: RippG2J, pub opening_b: RippG1J }
impl TippParts {
    /// the reference's `TIPAProof` from the library's CanonicalSerialize image of these members
    pub fn to_proof(&self) -> Result<TippProof, Error> {
        let cap = unsafe { ripp_ser_tipa_tipp_proof(self.com_steps.as_ptr(), self.rounds, &self.base_a, &self.base_b, &self.final_ck_a, &self.final_ck_b, &self.opening_a, &self.opening_b, 0, null_mut(), 0) };
        let mut bytes = vec![0u8; cap];
        let len = unsafe { ripp_ser_tipa_tipp_proof(self.com_steps.as_ptr(), self.rounds, &self.base_a, &self.base_b, &self.final_ck_a, &self.final_ck_b, &self.opening_a, &self.opening_b, 0, bytes.as_mut_ptr(), cap) };
        if len == 0 || len > cap { return Err("ripp_ser_tipa_tipp_proof failed".into()); }
        // (the members were produced by this process a moment ago: no subgroup checks)
        Ok(TippProof::deserialize_with_mode(&bytes[..len], Compress::No, Validate::No)?)
    }
    /// ... and back: any `TIPAProof` of this instantiation (GPU- or CPU-made) as flat members, through its own `serialize_uncompressed`
    pub fn from_proof(proof: &TippProof, max_rounds: usize) -> Result<Self, Error> {
        let mut bytes = Vec::with_capacity(proof.uncompressed_size());
        proof.serialize_uncompressed(&mut bytes)?;
        let mut p = TippParts { com_steps: vec![RippGt::default(); 6 * max_rounds], rounds: 0, base_a: RippG1J::default(), base_b: RippG2J::default(),
                                final_ck_a: RippG2J::default(), final_ck_b: RippG1J::default(), opening_a: RippG2J::default(), opening_b: RippG1J::default() };
        check(unsafe { ripp_de_tipa_tipp_proof(bytes.as_ptr(), bytes.len(), 0, 1, max_rounds, &mut p.rounds, p.com_steps.as_mut_ptr(), &mut p.base_a, &mut p.base_b,
                                               &mut p.final_ck_a, &mut p.final_ck_b, &mut p.opening_a, &mut p.opening_b) })?;
        p.com_steps.truncate(6 * p.rounds);
        Ok(p)
    }
}
/// `TIPA::prove_with_srs_shift` (tipa/mod.rs:176-231) for `PairingInnerProductAB<Bls12_381, Blake2b>`: ONE library call, the vectors stay on
/// the device for all rounds.  `ck.2` (the IPC key) is a placeholder in this instantiation and is not needed.
pub fn hip_tipa_prove_with_srs_shift(srs: &HipSrs, values: (&[G1Projective], &[G2Projective]), ck: (&[G2Projective], &[G1Projective]), r_shift: &Fr) -> Result<TippProof, Error> {
    hip_tipa_prove_parts(srs, values, ck, r_shift)?.to_proof()
}
/// `TIPA::prove` (tipa/mod.rs:166-172)
pub fn hip_tipa_prove(srs: &HipSrs, values: (&[G1Projective], &[G2Projective]), ck: (&[G2Projective], &[G1Projective])) -> Result<TippProof, Error> {
    hip_tipa_prove_with_srs_shift(srs, values, ck, &ark_ff::One::one())
}
pub fn hip_tipa_prove_parts(srs: &HipSrs, values: (&[G1Projective], &[G2Projective]), ck: (&[G2Projective], &[G1Projective]), r_shift: &Fr) -> Result<TippParts, Error> {
    let n = values.0.len();
    if values.1.len() != n || ck.0.len() != n || ck.1.len() != n { return Err(Box::new(ark_inner_products::InnerProductError::MessageLengthInvalid(n, values.1.len()))); }
    let rounds = rounds_of(n);
    let (ma, mb): (Vec<RippG1J>, Vec<RippG2J>) = (values.0.iter().map(g1j).collect(), values.1.iter().map(g2j).collect());
    let (ka, kb): (Vec<RippG2J>, Vec<RippG1J>) = (ck.0.iter().map(g2j).collect(), ck.1.iter().map(g1j).collect());
    let mut p = TippParts { com_steps: vec![RippGt::default(); 6 * rounds], rounds, base_a: RippG1J::default(), base_b: RippG2J::default(),
                            final_ck_a: RippG2J::default(), final_ck_b: RippG1J::default(), opening_a: RippG2J::default(), opening_b: RippG1J::default() };
    let mut transcript = vec![RippFr::default(); rounds];
    let mut kzg_c = RippFr::default();
    check(unsafe { ripp_tipa_tipp_prove(srs.raw(), ma.as_ptr(), mb.as_ptr(), ka.as_ptr(), kb.as_ptr(), n, &fr(r_shift), p.com_steps.as_mut_ptr(), transcript.as_mut_ptr(),
                                        &mut p.base_a, &mut p.base_b, &mut p.final_ck_a, &mut p.final_ck_b, &mut p.opening_a, &mut p.opening_b, &mut kzg_c, null_mut()) })?;
    Ok(p)
}
fn verifier_srs(v: &VerifierSRS<Bls12_381>) -> RippVerifierSrs { RippVerifierSrs { g: g1j(&v.g), h: g2j(&v.h), g_beta: g1j(&v.g_beta), h_alpha: g2j(&v.h_alpha) } }
/// `TIPA::verify_with_srs_shift` (tipa/mod.rs:242-301) on the GPU, for a proof made by either side
pub fn hip_tipa_verify_with_srs_shift(v_srs: &VerifierSRS<Bls12_381>, com: (&GT, &GT, &GT), proof: &TippProof, r_shift: &Fr) -> Result<bool, Error> {
    let p = TippParts::from_proof(proof, 64)?;
    let c = [gt(com.0), gt(com.1), gt(com.2)];
    let mut accept = 0i32;
    check(unsafe { ripp_tipa_tipp_verify(&verifier_srs(v_srs), c.as_ptr(), p.com_steps.as_ptr(), p.rounds, &p.base_a, &p.base_b, &p.final_ck_a, &p.final_ck_b,
                                         &p.opening_a, &p.opening_b, &fr(r_shift), &mut accept) })?;
    Ok(accept == 1)
}

// ---- TIPAWithSSM (MultiExpInnerProductC instantiation) ----------------------------------------------------------------------------------
pub struct SsmParts { pub com_gt: Vec<RippGt>, pub com_g1: Vec<RippG1J>, pub rounds: usize, pub base_a: RippG1J, pub base_b: RippFr, pub final_ck_a: RippG2J, pub opening_a: RippG2J }
impl SsmParts {
    pub fn to_proof(&self) -> Result<SsmProof, Error> {
        let cap = unsafe { ripp_ser_tipa_ssm_proof(self.com_gt.as_ptr(), self.com_g1.as_ptr(), self.rounds, &self.base_a, &self.base_b, &self.final_ck_a, &self.opening_a, 0, null_mut(), 0) };
        let mut bytes = vec![0u8; cap];
        let len = unsafe { ripp_ser_tipa_ssm_proof(self.com_gt.as_ptr(), self.com_g1.as_ptr(), self.rounds, &self.base_a, &self.base_b, &self.final_ck_a, &self.opening_a, 0, bytes.as_mut_ptr(), cap) };
        if len == 0 || len > cap { return Err("ripp_ser_tipa_ssm_proof failed".into()); }
        Ok(SsmProof::deserialize_with_mode(&bytes[..len], Compress::No, Validate::No)?)
    }
    pub fn from_proof(proof: &SsmProof, max_rounds: usize) -> Result<Self, Error> {
        let mut bytes = Vec::with_capacity(proof.uncompressed_size());
        proof.serialize_uncompressed(&mut bytes)?;
        let mut p = SsmParts { com_gt: vec![RippGt::default(); 2 * max_rounds], com_g1: vec![RippG1J::default(); 2 * max_rounds], rounds: 0, base_a: RippG1J::default(),
                               base_b: RippFr::default(), final_ck_a: RippG2J::default(), opening_a: RippG2J::default() };
        check(unsafe { ripp_de_tipa_ssm_proof(bytes.as_ptr(), bytes.len(), 0, max_rounds, &mut p.rounds, p.com_gt.as_mut_ptr(), p.com_g1.as_mut_ptr(), &mut p.base_a, &mut p.base_b,
                                              &mut p.final_ck_a, &mut p.opening_a) })?;
        p.com_gt.truncate(2 * p.rounds); p.com_g1.truncate(2 * p.rounds);
        Ok(p)
    }
}
/// `TIPAWithSSM::prove_with_structured_scalar_message` (structured_scalar_message.rs:211-268): values = (m_a in G1, the structured scalars), ck = the G2 keys
pub fn hip_tipa_ssm_prove(srs: &HipSrs, values: (&[G1Projective], &[Fr]), ck_a: &[G2Projective]) -> Result<SsmProof, Error> {
    let n = values.0.len();
    if values.1.len() != n || ck_a.len() != n { return Err(Box::new(ark_inner_products::InnerProductError::MessageLengthInvalid(n, values.1.len()))); }
    let rounds = rounds_of(n);
    let (ma, mb, ka): (Vec<RippG1J>, Vec<RippFr>, Vec<RippG2J>) = (values.0.iter().map(g1j).collect(), values.1.iter().map(fr).collect(), ck_a.iter().map(g2j).collect());
    let mut p = SsmParts { com_gt: vec![RippGt::default(); 2 * rounds], com_g1: vec![RippG1J::default(); 2 * rounds], rounds, base_a: RippG1J::default(), base_b: RippFr::default(),
                           final_ck_a: RippG2J::default(), opening_a: RippG2J::default() };
    let mut transcript = vec![RippFr::default(); rounds];
    let mut kzg_c = RippFr::default();
    check(unsafe { ripp_tipa_ssm_prove(srs.raw(), ma.as_ptr(), mb.as_ptr(), ka.as_ptr(), n, p.com_gt.as_mut_ptr(), p.com_g1.as_mut_ptr(), transcript.as_mut_ptr(),
                                       &mut p.base_a, &mut p.base_b, &mut p.final_ck_a, &mut p.opening_a, &mut kzg_c, null_mut()) })?;
    p.to_proof()
}
/// `TIPAWithSSM::verify_with_structured_scalar_message` (structured_scalar_message.rs:270-331): com = (com_a, com_t), scalar_b = the structure's base
pub fn hip_tipa_ssm_verify(v_srs: &VerifierSRS<Bls12_381>, com: (&GT, &G1Projective), scalar_b: &Fr, proof: &SsmProof) -> Result<bool, Error> {
    let p = SsmParts::from_proof(proof, 64)?;
    let mut accept = 0i32;
    check(unsafe { ripp_tipa_ssm_verify(&verifier_srs(v_srs), &gt(com.0), &g1j(com.1), &fr(scalar_b), p.com_gt.as_ptr(), p.com_g1.as_ptr(), p.rounds, &p.base_a, &p.final_ck_a,
                                        &p.opening_a, &mut accept) })?;
    Ok(accept == 1)
}

// ---- aggregate_proofs (groth16_aggregation.rs:77-231) ------------------------------------------------------------------------------------
/// The members of the reference's `AggregateProof<Bls12_381, Blake2b>` (groth16_aggregation.rs:59-69).  That struct's fields are private and it
/// derives no deserialiser, so a foreign crate cannot build one; the two sub-proofs below ARE the reference's own types.  (A maintainer who adds
/// `pub(crate)` constructors to `AggregateProof` can forward these members one to one: INTEGRATION.md.)
pub struct HipAggregateProof { pub com_a: GT, pub com_b: GT, pub com_c: GT, pub ip_ab: GT, pub agg_c: G1Projective, pub proof_ab: TippProof, pub proof_c: SsmProof }
struct AggBuffers { rounds: usize, ab_steps: Vec<RippGt>, ab_tr: Vec<RippFr>, c_gt: Vec<RippGt>, c_g1: Vec<RippG1J>, c_tr: Vec<RippFr>, s: RippAggregateProof }
fn agg_buffers(n: usize) -> AggBuffers {
    let rounds = rounds_of(n);
    let mut b = AggBuffers { rounds, ab_steps: vec![RippGt::default(); 6 * rounds], ab_tr: vec![RippFr::default(); rounds], c_gt: vec![RippGt::default(); 2 * rounds],
                             c_g1: vec![RippG1J::default(); 2 * rounds], c_tr: vec![RippFr::default(); rounds], s: unsafe { core::mem::zeroed() } };
    b.s.ab_com_steps = b.ab_steps.as_mut_ptr(); b.s.ab_transcript = b.ab_tr.as_mut_ptr();
    b.s.c_com_gt = b.c_gt.as_mut_ptr(); b.s.c_com_g1 = b.c_g1.as_mut_ptr(); b.s.c_transcript = b.c_tr.as_mut_ptr();
    b
}
fn agg_to_proof(b: &AggBuffers) -> Result<HipAggregateProof, Error> {
    let s = &b.s;
    let ab = TippParts { com_steps: b.ab_steps.clone(), rounds: b.rounds, base_a: s.ab_base_a, base_b: s.ab_base_b, final_ck_a: s.ab_final_ck_a, final_ck_b: s.ab_final_ck_b,
                         opening_a: s.ab_opening_a, opening_b: s.ab_opening_b };
    let c = SsmParts { com_gt: b.c_gt.clone(), com_g1: b.c_g1.clone(), rounds: b.rounds, base_a: s.c_base_a, base_b: s.c_base_b, final_ck_a: s.c_final_ck_a, opening_a: s.c_opening_a };
    Ok(HipAggregateProof { com_a: un_gt(&s.com_a), com_b: un_gt(&s.com_b), com_c: un_gt(&s.com_c), ip_ab: un_gt(&s.ip_ab), agg_c: un_g1j(&s.agg_c),
                           proof_ab: ab.to_proof()?, proof_c: c.to_proof()? })
}
/// `aggregate_proofs::<Bls12_381, Blake2b>(ip_srs, proofs)` (groth16_aggregation.rs:77-160): 0.100 s at n = 2^14 on one MI355X
pub fn hip_aggregate_proofs(srs: &HipSrs, proofs: &[Proof<Bls12_381>]) -> Result<HipAggregateProof, Error> {
    let n = proofs.len();
    let (a, b, c): (Vec<RippG1A>, Vec<RippG2A>, Vec<RippG1A>) = (proofs.iter().map(|p| g1a(&p.a)).collect(), proofs.iter().map(|p| g2a(&p.b)).collect(), proofs.iter().map(|p| g1a(&p.c)).collect());
    let mut buf = agg_buffers(n);
    check(unsafe { ripp_aggregate_proofs(srs.raw(), a.as_ptr(), b.as_ptr(), c.as_ptr(), n, &mut buf.s, null_mut()) })?;
    agg_to_proof(&buf)
}
/// the same across the library's communicator (`ripp_comm_init`): `proofs` = this rank's shard (local j <-> global j * world + rank); every rank gets the proof
pub fn hip_aggregate_proofs_sharded(srs: &HipSrs, proofs: &[Proof<Bls12_381>]) -> Result<HipAggregateProof, Error> {
    let world = unsafe { ripp_comm_world() } as usize;
    let (a, b, c): (Vec<RippG1A>, Vec<RippG2A>, Vec<RippG1A>) = (proofs.iter().map(|p| g1a(&p.a)).collect(), proofs.iter().map(|p| g2a(&p.b)).collect(), proofs.iter().map(|p| g1a(&p.c)).collect());
    let mut buf = agg_buffers(proofs.len() * world);
    check(unsafe { ripp_aggregate_proofs_sharded(srs.raw(), a.as_ptr(), b.as_ptr(), c.as_ptr(), proofs.len(), &mut buf.s, null_mut()) })?;
    agg_to_proof(&buf)
}
/// `verify_aggregate_proof` (groth16_aggregation.rs:162-231): `public_inputs[i]` are the inputs of proof i (all of one length m)
pub fn hip_verify_aggregate_proof(v_srs: &VerifierSRS<Bls12_381>, vk: &VerifyingKey<Bls12_381>, public_inputs: &[Vec<Fr>], proof: &HipAggregateProof) -> Result<bool, Error> {
    let n = public_inputs.len();
    let m = public_inputs.first().map_or(0, |v| v.len());
    assert!(public_inputs.iter().all(|v| v.len() == m));
    let flat: Vec<RippFr> = public_inputs.iter().flat_map(|v| v.iter().map(fr)).collect();
    let abc: Vec<RippG1A> = vk.gamma_abc_g1.iter().map(g1a).collect();
    let cvk = RippGroth16Vk { alpha_g1: g1a(&vk.alpha_g1), beta_g2: g2a(&vk.beta_g2), gamma_g2: g2a(&vk.gamma_g2), delta_g2: g2a(&vk.delta_g2), gamma_abc_g1: abc.as_ptr(), gamma_abc_len: abc.len() };
    let ab = TippParts::from_proof(&proof.proof_ab, 64)?;
    let c = SsmParts::from_proof(&proof.proof_c, 64)?;
    let mut buf = agg_buffers(n);
    assert_eq!(ab.rounds, buf.rounds); assert_eq!(c.rounds, buf.rounds);
    buf.ab_steps.copy_from_slice(&ab.com_steps); buf.c_gt.copy_from_slice(&c.com_gt); buf.c_g1.copy_from_slice(&c.com_g1);
    let s = &mut buf.s;
    s.com_a = gt(&proof.com_a); s.com_b = gt(&proof.com_b); s.com_c = gt(&proof.com_c); s.ip_ab = gt(&proof.ip_ab); s.agg_c = g1j(&proof.agg_c);
    s.ab_base_a = ab.base_a; s.ab_base_b = ab.base_b; s.ab_final_ck_a = ab.final_ck_a; s.ab_final_ck_b = ab.final_ck_b; s.ab_opening_a = ab.opening_a; s.ab_opening_b = ab.opening_b;
    s.c_base_a = c.base_a; s.c_base_b = c.base_b; s.c_final_ck_a = c.final_ck_a; s.c_opening_a = c.opening_a;
    let mut accept = 0i32;
    check(unsafe { ripp_verify_aggregate_proof(&verifier_srs(v_srs), &cvk, flat.as_ptr(), n, m, &buf.s, &mut accept) })?;
    Ok(accept == 1)
}

// ---- device-resident vectors (ripp_vec_*) --------------------------------------------------------------------------------------------------
/// RAII handle of a vector in HBM: upload once, then halve, fold and take inner products on views -- what a GIPA-style round loop written by
/// the caller needs to stay off the PCIe bus (gipa.rs:196-297).  Views borrow the parent's storage; the library keeps it alive until the last
/// handle is freed.
pub struct HipVec { h: *mut RippVec }
impl HipVec {
    pub fn upload_g1a(p: &[G1Affine]) -> Result<Self, Error> { let v: Vec<RippG1A> = p.iter().map(g1a).collect(); let mut h = null_mut(); check(unsafe { ripp_vec_upload_g1a(v.as_ptr(), v.len(), &mut h) })?; Ok(HipVec { h }) }
    pub fn upload_g2a(p: &[G2Affine]) -> Result<Self, Error> { let v: Vec<RippG2A> = p.iter().map(g2a).collect(); let mut h = null_mut(); check(unsafe { ripp_vec_upload_g2a(v.as_ptr(), v.len(), &mut h) })?; Ok(HipVec { h }) }
    pub fn upload_g1(p: &[G1Projective]) -> Result<Self, Error> { let v: Vec<RippG1J> = p.iter().map(g1j).collect(); let mut h = null_mut(); check(unsafe { ripp_vec_upload_g1j(v.as_ptr(), v.len(), &mut h) })?; Ok(HipVec { h }) }
    pub fn upload_g2(p: &[G2Projective]) -> Result<Self, Error> { let v: Vec<RippG2J> = p.iter().map(g2j).collect(); let mut h = null_mut(); check(unsafe { ripp_vec_upload_g2j(v.as_ptr(), v.len(), &mut h) })?; Ok(HipVec { h }) }
    pub fn upload_fr(p: &[Fr]) -> Result<Self, Error> { let v: Vec<RippFr> = p.iter().map(fr).collect(); let mut h = null_mut(); check(unsafe { ripp_vec_upload_fr(v.as_ptr(), v.len(), &mut h) })?; Ok(HipVec { h }) }
    pub fn len(&self) -> usize { unsafe { ripp_vec_len(self.h) } }
    pub fn is_empty(&self) -> bool { self.len() == 0 }
    pub fn kind(&self) -> i32 { unsafe { ripp_vec_kind(self.h) } }
    pub fn slice(&self, off: usize, len: usize) -> Result<HipVec, Error> { let mut h = null_mut(); check(unsafe { ripp_vec_slice(self.h, off, len, &mut h) })?; Ok(HipVec { h }) }
    /// (lo, hi) = the two halves of a round (gipa.rs:207-217)
    pub fn halves(&self) -> Result<(HipVec, HipVec), Error> { let (mut lo, mut hi) = (null_mut(), null_mut()); check(unsafe { ripp_vec_halves(self.h, &mut lo, &mut hi) })?; Ok((HipVec { h: lo }, HipVec { h: hi })) }
    /// out[i] = s * hi[i] + lo[i] (`mul_helper` folds, gipa.rs:262-290), a new resident vector
    pub fn fold(hi: &HipVec, lo: &HipVec, s: &Fr) -> Result<HipVec, Error> { let mut h = null_mut(); check(unsafe { ripp_vec_fold(hi.h, lo.h, &fr(s), &mut h) })?; Ok(HipVec { h }) }
    /// `PairingInnerProduct::inner_product` on resident vectors
    pub fn pairing_product(left_g1: &HipVec, right_g2: &HipVec) -> Result<GT, Error> { let mut out = RippGt::default(); check(unsafe { ripp_vec_pairing_product(left_g1.h, right_g2.h, &mut out) })?; Ok(un_gt(&out)) }
    /// `MultiexponentiationInnerProduct::<G1>::inner_product` on resident vectors
    pub fn msm_g1(bases: &HipVec, scalars: &HipVec) -> Result<G1Projective, Error> { let mut out = RippG1J::default(); check(unsafe { ripp_vec_msm(bases.h, scalars.h, &mut out as *mut RippG1J as *mut core::ffi::c_void) })?; Ok(un_g1j(&out)) }
    pub fn msm_g2(bases: &HipVec, scalars: &HipVec) -> Result<G2Projective, Error> { let mut out = RippG2J::default(); check(unsafe { ripp_vec_msm(bases.h, scalars.h, &mut out as *mut RippG2J as *mut core::ffi::c_void) })?; Ok(un_g2j(&out)) }
    pub fn scalar_inner_product(l: &HipVec, r: &HipVec) -> Result<Fr, Error> { let mut out = RippFr::default(); check(unsafe { ripp_vec_scalar_inner_product(l.h, r.h, &mut out) })?; Ok(un_fr(&out)) }
    /// `ripp_vec_download` writes AFFINE elements (96 / 192 bytes) or scalars (32 bytes) according to the vector's kind: the buffer is typed by kind, and a
    /// handle of another kind is refused HERE (a 32 n-byte buffer handed over for a G1 vector would be overrun by 96 n bytes).
    fn want_kind(&self, kind: i32, what: &str) -> Result<(), Error> {
        if self.kind() == kind { Ok(()) } else { Err(format!("HipVec::{what}: the vector holds kind {} (1 = G1, 2 = G2, 3 = Fr), not kind {kind}", self.kind()).into()) }
    }
    pub fn download_g1a(&self) -> Result<Vec<G1Affine>, Error> { self.want_kind(RIPP_VEC_G1, "download_g1a")?; let mut v = vec![RippG1A::default(); self.len()]; check(unsafe { ripp_vec_download(self.h, v.as_mut_ptr() as *mut core::ffi::c_void) })?; Ok(v.iter().map(un_g1a).collect()) }
    pub fn download_g2a(&self) -> Result<Vec<G2Affine>, Error> { self.want_kind(RIPP_VEC_G2, "download_g2a")?; let mut v = vec![RippG2A::default(); self.len()]; check(unsafe { ripp_vec_download(self.h, v.as_mut_ptr() as *mut core::ffi::c_void) })?; Ok(v.iter().map(un_g2a).collect()) }
    /// the projective types the reference's GIPA carries (`.into()` of the affine elements the device holds)
    pub fn download_g1(&self) -> Result<Vec<G1Projective>, Error> { Ok(self.download_g1a()?.into_iter().map(Into::into).collect()) }
    pub fn download_g2(&self) -> Result<Vec<G2Projective>, Error> { Ok(self.download_g2a()?.into_iter().map(Into::into).collect()) }
    pub fn download_fr(&self) -> Result<Vec<Fr>, Error> { self.want_kind(RIPP_VEC_FR, "download_fr")?; let mut v = vec![RippFr::default(); self.len()]; check(unsafe { ripp_vec_download(self.h, v.as_mut_ptr() as *mut core::ffi::c_void) })?; Ok(v.iter().map(un_fr).collect()) }
}
impl Drop for HipVec { fn drop(&mut self) { unsafe { ripp_vec_free(self.h) } } }

// ---- SIPP on a resident statement / across ranks ------------------------------------------------------------------------------------------
/// A SIPP statement (shard) resident in HBM (`ripp_sipp_job_*`): repeated proofs pay no upload.  rank / world: element i of the statement lives
/// on rank i mod world at local index i div world.
pub struct HipSippJob { h: *mut RippSippJob, n_local: usize, world: usize }
impl HipSippJob {
    pub fn new(a: &[G1Affine], b: &[G2Affine], r: &[Fr], rank: i32, world: i32) -> Result<Self, Error> {
        assert!(a.len() == b.len() && a.len() == r.len());
        let (la, lb, lr): (Vec<RippG1A>, Vec<RippG2A>, Vec<RippFr>) = (a.iter().map(g1a).collect(), b.iter().map(g2a).collect(), r.iter().map(fr).collect());
        let mut h = null_mut();
        check(unsafe { ripp_sipp_job_create(la.as_ptr(), lb.as_ptr(), lr.as_ptr(), la.len(), rank, world, &mut h) })?;
        Ok(HipSippJob { h, n_local: la.len(), world: world as usize })
    }
    /// `SIPP::prove` (sipp/src/lib.rs:42-106) on the resident statement (world == 1)
    pub fn prove(&mut self, value: &GT) -> Result<Vec<(GT, GT)>, Error> {
        let rounds = (self.n_local * self.world).trailing_zeros() as usize;
        let mut proof = vec![RippGt::default(); 2 * rounds.max(1)];
        check(unsafe { ripp_sipp_job_prove(self.h, &gt(value), proof.as_mut_ptr(), null_mut(), null_mut()) })?;
        Ok(proof[..2 * rounds].chunks(2).map(|p| (un_gt(&p[0]), un_gt(&p[1]))).collect())
    }
    /// the same across the communicator; rank 0 passes the full statement (hashed while round 0 runs), the other ranks `None`
    pub fn prove_sharded(&mut self, value: &GT, full: Option<(&[G1Affine], &[G2Affine], &[Fr])>) -> Result<Vec<(GT, GT)>, Error> {
        let rounds = (self.n_local * self.world).trailing_zeros() as usize;
        let mut proof = vec![RippGt::default(); 2 * rounds.max(1)];
        let conv = full.map(|(a, b, r)| (a.iter().map(g1a).collect::<Vec<_>>(), b.iter().map(g2a).collect::<Vec<_>>(), r.iter().map(fr).collect::<Vec<_>>()));
        let (fa, fb, fr_) = conv.as_ref().map_or((null(), null(), null()), |(a, b, r)| (a.as_ptr(), b.as_ptr(), r.as_ptr()));
        check(unsafe { ripp_sipp_job_prove_sharded(self.h, &gt(value), fa, fb, fr_, null(), proof.as_mut_ptr(), null_mut(), null_mut()) })?;
        Ok(proof[..2 * rounds].chunks(2).map(|p| (un_gt(&p[0]), un_gt(&p[1]))).collect())
    }
}
impl Drop for HipSippJob { fn drop(&mut self) { unsafe { ripp_sipp_job_destroy(self.h) } } }
/// `SIPP::prove` across the communicator's ranks on HOST slices (`ripp_sipp_prove_sharded`): a, b, r = this rank's shard, `full` on rank 0 only
pub fn hip_sipp_prove_sharded(a: &[G1Affine], b: &[G2Affine], r: &[Fr], value: &GT, full: Option<(&[G1Affine], &[G2Affine], &[Fr])>) -> Result<Vec<(GT, GT)>, Error> {
    assert!(a.len() == b.len() && a.len() == r.len() && a.len().is_power_of_two());
    let world = unsafe { ripp_comm_world() } as usize;
    let rounds = (a.len() * world).trailing_zeros() as usize;
    let (la, lb, lr): (Vec<RippG1A>, Vec<RippG2A>, Vec<RippFr>) = (a.iter().map(g1a).collect(), b.iter().map(g2a).collect(), r.iter().map(fr).collect());
    let conv = full.map(|(a, b, r)| (a.iter().map(g1a).collect::<Vec<_>>(), b.iter().map(g2a).collect::<Vec<_>>(), r.iter().map(fr).collect::<Vec<_>>()));
    let (fa, fb, fr_) = conv.as_ref().map_or((null(), null(), null()), |(a, b, r)| (a.as_ptr(), b.as_ptr(), r.as_ptr()));
    let mut proof = vec![RippGt::default(); 2 * rounds.max(1)];
    check(unsafe { ripp_sipp_prove_sharded(la.as_ptr(), lb.as_ptr(), lr.as_ptr(), la.len(), &gt(value), fa, fb, fr_, null(), proof.as_mut_ptr(), null_mut(), null_mut()) })?;
    Ok(proof[..2 * rounds].chunks(2).map(|p| (un_gt(&p[0]), un_gt(&p[1]))).collect())
}
/// Use `n` devices of THIS process for the stateless trait calls (`HipPairingInnerProduct`, `HipMultiexpInnerProductG1/G2` on host slices): the library cuts the index
/// range into `n` parts, one per device (bound device + d), and combines the partial results on the host -- no communicator, no change to the caller (`ripp_config.n_devices`).
/// Proofs (SIPP / GIPA / TIPA) shard across processes instead (`hip_comm_init`).  `hip_device_slots_used()` tells how many parts the last such call used.
pub fn hip_set_devices(n: u32) -> Result<(), Error> {
    let mut c = core::mem::MaybeUninit::<RippConfig>::zeroed();
    check(unsafe { ripp_config_get(c.as_mut_ptr()) })?;
    let mut c = unsafe { c.assume_init() };
    c.n_devices = n;
    check(unsafe { ripp_configure(&c) })
}
pub fn hip_device_slots_used() -> i32 { unsafe { ripp_device_slots_used() } }
/// bring the library's RCCL communicator up: rank 0 creates the 128-byte id (`hip_comm_unique_id`), the host's own rendezvous hands it to the others
pub fn hip_comm_unique_id() -> Result<[u8; 128], Error> { let mut id = [0u8; 128]; check(unsafe { ripp_comm_unique_id(id.as_mut_ptr()) })?; Ok(id) }
pub fn hip_comm_init(id: &[u8; 128], rank: i32, world: i32) -> Result<(), Error> { check(unsafe { ripp_comm_init(id.as_ptr(), rank, world) }) }
pub fn hip_comm_destroy() { unsafe { ripp_comm_destroy() } }
