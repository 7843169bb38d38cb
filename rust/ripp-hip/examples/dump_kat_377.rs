//! BLS12-377 counterpart of dump_kat.rs -- the curve the reference's own SIPP test and `scaling-ipp` harness run on
//! (sipp/src/lib.rs:229, sipp/examples/scaling-ipp.rs:2).  Prints, from REAL arkworks 0.4 + the reference's `ark-sipp`:
//!   * every known answer tests/golden/bls12_377_vectors.json holds (generic SW point encoding with SWFlags in the last byte, e(G1, G2),
//!     bilinearity, an 8-pair product with infinities, MSMs, a SIPP proof of n = 4 with its seed digest and challenges);
//!   * `base_case`: the reference's `prove_and_verify_base_case` (sipp/src/lib.rs:232-254) -- seed b"falafel", 32 points and scalars drawn from
//!     FiatShamirRng<Blake2s>, `SIPP::prove` + `SIPP::verify` as the test runs them -- with its inputs, the value z and the proof elements.
//!     (`Proof::gt_elems` is private, sipp/src/lib.rs:32-34: the elements come from `sipp_replay`, the body of `SIPP::prove` on the crate's
//!     public pieces, and are checked against the verifier's equation.)  tools/compare_kat.py feeds those inputs to this
//!     repository's BLS12-377 oracle and compares z and the proof: the reference's only SIPP test becomes a byte-level known answer.
//!
//!   cargo run --release --no-default-features --example dump_kat_377 -- ../../tests/golden/bls12_377_vectors.json > kat_arkworks_377.json
//!   python3 ../../tools/compare_kat.py ../../tests/golden/bls12_377_vectors.json kat_arkworks_377.json --curve bls12_377
//!
//! Needs no GPU and does not link libripp_hip.
use ark_bls12_377::{Bls12_377, Fq, Fq2, Fr, G1Affine, G1Projective, G2Affine, G2Projective};
use ark_ec::{pairing::{Pairing, PairingOutput}, AffineRepr, CurveGroup, VariableBaseMSM};
use ark_ff::{BigInteger, Field, One, PrimeField, UniformRand};
use ark_serialize::CanonicalSerialize;
use ark_sipp::{product_of_pairings, product_of_pairings_with_coeffs, rng::FiatShamirRng, SIPP};
use blake2::Blake2s;
use digest::Digest;
use serde_json::{json, Value};

fn le_bytes(s: &str) -> Vec<u8> {
    let h = s.trim_start_matches("0x"); let h = if h.len() % 2 == 1 { format!("0{h}") } else { h.to_string() };
    let mut b = hex::decode(h).unwrap(); b.reverse(); b
}
fn fq(s: &str) -> Fq { Fq::from_le_bytes_mod_order(&le_bytes(s)) }
fn fr(s: &str) -> Fr { Fr::from_le_bytes_mod_order(&le_bytes(s)) }
fn g1(v: &Value) -> G1Affine { if v.is_null() { G1Affine::identity() } else { G1Affine::new(fq(v[0].as_str().unwrap()), fq(v[1].as_str().unwrap())) } }
fn g2(v: &Value) -> G2Affine {
    if v.is_null() { return G2Affine::identity(); }
    let c = |w: &Value| Fq2::new(fq(w[0].as_str().unwrap()), fq(w[1].as_str().unwrap()));
    G2Affine::new(c(&v[0]), c(&v[1]))
}
fn ser<T: CanonicalSerialize>(x: &T) -> String { let mut b = Vec::new(); x.serialize_uncompressed(&mut b).unwrap(); hex::encode(b) }
fn ser_c<T: CanonicalSerialize>(x: &T) -> String { let mut b = Vec::new(); x.serialize_compressed(&mut b).unwrap(); hex::encode(b) }
fn hexint<F: PrimeField>(x: &F) -> String { let h = hex::encode(x.into_bigint().to_bytes_be()); let t = h.trim_start_matches('0'); format!("0x{}", if t.is_empty() { "0" } else { t }) }
fn pg1(p: &G1Affine) -> Value { if p.infinity { Value::Null } else { json!([hexint(&p.x), hexint(&p.y)]) } }
fn pg2(p: &G2Affine) -> Value { if p.infinity { Value::Null } else { json!([[hexint(&p.x.c0), hexint(&p.x.c1)], [hexint(&p.y.c0), hexint(&p.y.c1)]]) } }

type GT = PairingOutput<Bls12_377>;

/// value, seed digest, proof and challenges of one SIPP statement (sipp/src/lib.rs:42-106)
fn sipp_section(a: &[G1Affine], b: &[G2Affine], r: &[Fr]) -> Value {
    let value: GT = product_of_pairings_with_coeffs::<Bls12_377>(a, b, r);
    // the reference's prover and verifier on this statement: it accepts its own proof (whose elements are private, see sipp_replay)
    let ref_proof = SIPP::<Bls12_377, Blake2s>::prove(a, b, r, value).unwrap();
    assert!(SIPP::<Bls12_377, Blake2s>::verify(a, b, r, value, &ref_proof).unwrap());
    let (seed, proof, chs) = sipp_replay::<Bls12_377>(a, b, r, value);
    assert!(sipp_equation_holds::<Bls12_377>(a, b, r, value, &proof, &chs));
    json!({"value": ser(&value), "seed_digest": hex::encode(Blake2s::digest(&seed)),
           "proof": proof.iter().map(|(l, r)| json!([ser(l), ser(r)])).collect::<Vec<_>>(), "challenges": chs.iter().map(hexint).collect::<Vec<_>>()})
}
/// `SIPP::prove`'s body (sipp/src/lib.rs:56-101) replayed statement by statement on the crate's PUBLIC pieces (`product_of_pairings`,
/// `rng::FiatShamirRng`, arkworks group operations).  `Proof::gt_elems` is a PRIVATE field (sipp/src/lib.rs:32-34) and `Proof` derives no
/// serialisation, so the elements of `SIPP::prove`'s result cannot be read from outside the crate; `main` still runs `SIPP::prove` +
/// `SIPP::verify` (the reference accepts its own proof of the statement) and checks the replayed elements against the verifier's equation.
/// Returns (the bytes the prover hashes, (z_l, z_r) per round, the challenges).
fn sipp_replay<E: Pairing>(a: &[E::G1Affine], b: &[E::G2Affine], r: &[E::ScalarField], value: PairingOutput<E>)
    -> (Vec<u8>, Vec<(PairingOutput<E>, PairingOutput<E>)>, Vec<E::ScalarField>) {
    let mut seed = Vec::new();
    (a, b, r, value).serialize_uncompressed(&mut seed).unwrap();                                     // :56-59, the reference's own expression
    let mut rng = FiatShamirRng::<Blake2s>::from_seed(&seed);
    let a = a.iter().zip(r).map(|(&a, r)| a * r).collect::<Vec<_>>();                                 // :61-65
    let mut a = E::G1::normalize_batch(&a);                                                           // :66
    let mut b = b.to_vec();
    let mut length = a.len();
    let (mut proof_vec, mut challenges) = (Vec::new(), Vec::new());
    while length != 1 {                                                                               // :69-104
        length /= 2;
        let (a_l, a_r) = (&a[..length], &a[length..]);
        let (b_l, b_r) = (&b[..length], &b[length..]);
        let z_l = product_of_pairings::<E>(a_r, b_l);
        let z_r = product_of_pairings::<E>(a_l, b_r);
        proof_vec.push((z_l, z_r));
        {
            let mut buf = Vec::new();
            (z_l, z_r).serialize_uncompressed(&mut buf).unwrap();
            rng.absorb(&buf);
        }
        let x: E::ScalarField = u128::rand(&mut rng).into();
        challenges.push(x);
        let a_proj = a_l.iter().zip(a_r).map(|(a_l, &a_r)| a_r * x + a_l).collect::<Vec<_>>();
        let a_next = E::G1::normalize_batch(&a_proj);
        let x_inv = x.inverse().unwrap();
        let b_proj = b_l.iter().zip(b_r).map(|(b_l, &b_r)| b_r * x_inv + b_l).collect::<Vec<_>>();
        let b_next = E::G2::normalize_batch(&b_proj);
        a = a_next;
        b = b_next;
    }
    (seed, proof_vec, challenges)
}

/// The equation `SIPP::verify` checks (sipp/src/lib.rs:151-177) on the replayed elements and challenges, with arkworks arithmetic.
fn sipp_equation_holds<E: Pairing>(a: &[E::G1Affine], b: &[E::G2Affine], r: &[E::ScalarField], claimed_value: PairingOutput<E>,
                                   proof: &[(PairingOutput<E>, PairingOutput<E>)], x_s: &[E::ScalarField]) -> bool {
    let length = a.len();
    let proof_len = proof.len();
    let mut z_prime = claimed_value;
    for ((z_l, z_r), x) in proof.iter().zip(x_s) {
        z_prime = z_prime + (*z_l * *x) + (*z_r * x.inverse().unwrap());
    }
    let mut s = vec![E::ScalarField::one(); length];
    let mut s_invs = vec![E::ScalarField::one(); length];
    for (j, x) in x_s.iter().enumerate() {
        let x_inv = x.inverse().unwrap();
        for i in 0..length {
            if i & (1 << (proof_len - j - 1)) != 0 {
                s[i] *= x;
                s_invs[i] *= &x_inv;
            }
        }
    }
    let s = s.into_iter().zip(r).map(|(x, r)| x * r).collect::<Vec<_>>();
    let a_prime = E::G1::msm(a, &s).unwrap();
    let b_prime = E::G2::msm(b, &s_invs).unwrap();
    E::pairing(a_prime, b_prime) == z_prime
}

fn main() {
    let path = std::env::args().nth(1).expect("path of tests/golden/bls12_377_vectors.json");
    let gold: Value = serde_json::from_str(&std::fs::read_to_string(path).unwrap()).unwrap();
    let mut out = serde_json::Map::new();
    // ---- generators and the generic short-Weierstrass encoding (flags in the LAST byte; the y-sign bit also in the uncompressed form)
    let (g, h) = (G1Affine::generator(), G2Affine::generator());
    out.insert("generators".into(), json!({"g1": pg1(&g), "g2": pg2(&h), "ser_g1": ser(&g), "ser_g2": ser(&h),
        "ser_g1_neg": ser(&(-g)), "ser_g2_neg": ser(&(-h)), "ser_g1_inf": ser(&G1Affine::identity()), "ser_g2_inf": ser(&G2Affine::identity()),
        "ser_g1_compressed": ser_c(&g), "ser_g2_compressed": ser_c(&h), "ser_g1_neg_compressed": ser_c(&(-g)), "ser_g2_neg_compressed": ser_c(&(-h)),
        "ser_g1_inf_compressed": ser_c(&G1Affine::identity()), "ser_g2_inf_compressed": ser_c(&G2Affine::identity())}));
    // ---- e(G1, G2) and bilinearity
    let e = Bls12_377::pairing(g, h);
    out.insert("pairing_generators".into(), json!({"gt": ser(&e)}));
    let (a, b) = (fr(gold["bilinearity"]["a"].as_str().unwrap()), fr(gold["bilinearity"]["b"].as_str().unwrap()));
    out.insert("bilinearity".into(), json!({"gt": ser(&Bls12_377::pairing(g * a, h * b)), "gt_pow": ser(&(e * (a * b)))}));
    // ---- 8-pair product with infinities
    let pa: Vec<G1Affine> = gold["product8"]["a"].as_array().unwrap().iter().map(g1).collect();
    let pb: Vec<G2Affine> = gold["product8"]["b"].as_array().unwrap().iter().map(g2).collect();
    out.insert("product8".into(), json!({"gt": ser(&product_of_pairings::<Bls12_377>(&pa, &pb))}));
    // ---- MSM n = 8
    let sc: Vec<Fr> = gold["msm8"]["scalars"].as_array().unwrap().iter().map(|s| fr(s.as_str().unwrap())).collect();
    let b1: Vec<G1Affine> = gold["msm8"]["g1_bases"].as_array().unwrap().iter().map(g1).collect();
    let b2: Vec<G2Affine> = gold["msm8"]["g2_bases"].as_array().unwrap().iter().map(g2).collect();
    out.insert("msm8".into(), json!({"g1": pg1(&G1Projective::msm(&b1, &sc).unwrap().into_affine()), "g2": pg2(&G2Projective::msm(&b2, &sc).unwrap().into_affine())}));
    // ---- SIPP n = 4 on the golden file's statement
    let a4: Vec<G1Affine> = gold["sipp4"]["a"].as_array().unwrap().iter().map(g1).collect();
    let b4: Vec<G2Affine> = gold["sipp4"]["b"].as_array().unwrap().iter().map(g2).collect();
    let r4: Vec<Fr> = gold["sipp4"]["r"].as_array().unwrap().iter().map(|s| fr(s.as_str().unwrap())).collect();
    out.insert("sipp4".into(), sipp_section(&a4, &b4, &r4));
    // ---- the reference's own test, verbatim (sipp/src/lib.rs:232-254): inputs drawn from FiatShamirRng::<Blake2s>::from_seed(b"falafel")
    let mut rng = FiatShamirRng::<Blake2s>::from_seed(b"falafel");
    let (mut ta, mut tb, mut tr) = (Vec::with_capacity(32), Vec::with_capacity(32), Vec::with_capacity(32));
    for _ in 0..32 {
        ta.push(G1Projective::rand(&mut rng).into_affine());
        tb.push(G2Projective::rand(&mut rng).into_affine());
        tr.push(Fr::rand(&mut rng));
    }
    let mut base = sipp_section(&ta, &tb, &tr);
    base["a"] = Value::Array(ta.iter().map(pg1).collect()); base["b"] = Value::Array(tb.iter().map(pg2).collect());
    base["r"] = Value::Array(tr.iter().map(|x| Value::String(hexint(x))).collect());
    out.insert("base_case".into(), base);
    println!("{}", serde_json::to_string_pretty(&Value::Object(out)).unwrap());
}
