//! Fixture emitter: runs the REAL reference (bp-pp 0.1.1 on k256 0.13.3 / merlin 3.0.0) with a seeded RNG and prints, as JSON on
//! stdout, everything needed to pin this repository's oracle and HIP path to it:
//!
//!   cargo run --release --no-default-features --bin gen_fixtures > ../tests/golden/ref_u64.json
//!
//! (no GPU and no libbppp_hip.so needed).  tests/test_ref_fixtures.py and tests/test_gpu_ref_fixtures.py consume the file when it
//! exists and skip -- saying so -- when it does not.  UNCOMPILED: written without a Rust toolchain (see lib.rs).
//!
//! Per case the file records the prover's inputs INCLUDING every byte it pulled from the RNG, so the consumers can (a) replay the
//! prover and demand byte-identical proofs, (b) check the model of `Scalar::generate_biased` (64 bytes, big-endian, reduced mod
//! n), (c) check accept bits, (d) check the transcript state the reference leaves behind (`t: &mut Transcript`), and (e) settle
//! the serde / hex conventions (`proof_json`, `commitment_json`, `identity_json`).
use bp_pp::range_proof::reciprocal::{Proof, SerializableProof};
use bp_pp::range_proof::u64_proof::{U64RangeProofProtocol, G_VEC_FULL_SZ, H_VEC_FULL_SZ};
use bp_pp_gpu::{conv, tstate};
use k256::elliptic_curve::group::GroupEncoding;
use k256::elliptic_curve::Group;
use k256::{AffinePoint, ProjectivePoint, Scalar};
use merlin::Transcript;
use rand_chacha::ChaCha20Rng;
use rand_core::{CryptoRng, RngCore, SeedableRng};
use serde_json::json;

/// ChaCha20 with a log of every byte handed out (and of the size of each request).
struct RecordingRng {
    inner: ChaCha20Rng,
    bytes: Vec<u8>,
    calls: Vec<usize>,
}
impl RecordingRng {
    fn take(&mut self) -> (Vec<u8>, Vec<usize>) {
        (std::mem::take(&mut self.bytes), std::mem::take(&mut self.calls))
    }
}
impl RngCore for RecordingRng {
    fn next_u32(&mut self) -> u32 {
        let mut b = [0u8; 4];
        self.fill_bytes(&mut b);
        u32::from_le_bytes(b)
    }
    fn next_u64(&mut self) -> u64 {
        let mut b = [0u8; 8];
        self.fill_bytes(&mut b);
        u64::from_le_bytes(b)
    }
    fn fill_bytes(&mut self, dest: &mut [u8]) {
        self.inner.fill_bytes(dest);
        self.bytes.extend_from_slice(dest);
        self.calls.push(dest.len());
    }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), rand_core::Error> {
        self.fill_bytes(dest);
        Ok(())
    }
}
impl CryptoRng for RecordingRng {}

/// An RNG that replays recorded bytes: `Scalar::generate_biased(&mut Replay(..))` is k256's own reduction of those bytes.
struct Replay<'a>(&'a [u8], usize);
impl<'a> RngCore for Replay<'a> {
    fn next_u32(&mut self) -> u32 { let mut b = [0u8; 4]; self.fill_bytes(&mut b); u32::from_le_bytes(b) }
    fn next_u64(&mut self) -> u64 { let mut b = [0u8; 8]; self.fill_bytes(&mut b); u64::from_le_bytes(b) }
    fn fill_bytes(&mut self, dest: &mut [u8]) { dest.copy_from_slice(&self.0[self.1..self.1 + dest.len()]); self.1 += dest.len(); }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), rand_core::Error> { self.fill_bytes(dest); Ok(()) }
}
impl<'a> CryptoRng for Replay<'a> {}

fn abi_point(p: &ProjectivePoint) -> String {
    let mut v = Vec::new();
    conv::put_point(&mut v, p);
    hex::encode(v)
}
fn abi_proof(p: &Proof) -> String {
    let mut v = Vec::new();
    conv::put_u64_proof(&mut v, p).expect("the honest prover emits the u64 shape");
    hex::encode(v)
}

fn main() {
    let seed = *b"bppp-ref-fixtures-v1-seed-000001";
    let mut rng = RecordingRng { inner: ChaCha20Rng::from_seed(seed), bytes: vec![], calls: vec![] };
    let label: &'static [u8] = b"u64 range proof"; // benches/range_proof.rs:32

    // generators exactly as the reference's own test and bench make them (tests.rs:22-24, benches/range_proof.rs:18-20)
    let g = ProjectivePoint::random(&mut rng);
    let g_vec: Vec<ProjectivePoint> = (0..G_VEC_FULL_SZ).map(|_| ProjectivePoint::random(&mut rng)).collect();
    let h_vec: Vec<ProjectivePoint> = (0..H_VEC_FULL_SZ).map(|_| ProjectivePoint::random(&mut rng)).collect();
    let (gen_bytes, _) = rng.take();
    let public = U64RangeProofProtocol { g, g_vec: g_vec.clone(), h_vec: h_vec.clone() };
    let mut gens = abi_point(&g);
    g_vec.iter().chain(h_vec.iter()).for_each(|p| gens.push_str(&abi_point(p)));

    let values: [u64; 6] = [123456, 0, u64::MAX, 1, 0x0123_4567_89AB_CDEF, 1 << 63];
    let mut cases = vec![];
    let mut negatives = vec![];
    for (j, x) in values.iter().enumerate() {
        let s = Scalar::generate_biased(&mut rng);
        let (s_bytes, _) = rng.take();
        // half of the cases bind context into the transcript first: the pre-loaded `t: &mut Transcript` contract
        let context: Vec<u8> = if j % 2 == 1 { format!("tx-{j}").into_bytes() } else { vec![] };
        let mut t0 = Transcript::new(label);
        if !context.is_empty() {
            t0.append_message(b"ctx", &context);
        }
        let commitment = public.commit_value(*x, &s);
        let mut pt = t0.clone();
        let proof = public.prove(*x, &s, &mut pt, &mut rng);
        let (rng_bytes, rng_calls) = rng.take();
        // the scalars the prover drew, by k256's own reduction of the recorded bytes
        let mut rnd = Vec::new();
        let mut rp = Replay(&rng_bytes, 0);
        for _ in 0..rng_calls.len() {
            conv::put_scalar(&mut rnd, &Scalar::generate_biased(&mut rp));
        }
        let mut vt = t0.clone();
        let ok = public.verify(&commitment, proof.clone(), &mut vt);
        assert!(ok);
        cases.push(json!({
            "x": x.to_string(), "s": hex::encode(s.to_bytes()), "s_rng_bytes": hex::encode(&s_bytes),
            "context": hex::encode(&context),
            "rng_bytes": hex::encode(&rng_bytes), "rng_calls": rng_calls, "rnd": hex::encode(&rnd),
            "commitment": abi_point(&commitment), "proof": abi_proof(&proof),
            "commitment_json": serde_json::to_value(&commitment.to_affine()).unwrap(),
            "proof_json": serde_json::to_value(&SerializableProof::from(&proof)).unwrap(),
            "state_before": hex::encode(tstate::to_bytes(&t0)),
            "state_after_prove": hex::encode(tstate::to_bytes(&pt)),
            "state_after_verify": hex::encode(tstate::to_bytes(&vt)),
            "accept": true,
        }));
        // a negative per case: one scalar bumped, one point swapped -- the reference's own verdict and transcript state
        let mut bad = proof.clone();
        if j % 2 == 0 {
            bad.circuit_proof.n[0] = bad.circuit_proof.n[0] + Scalar::ONE;
        } else {
            bad.circuit_proof.x.swap(0, 1);
        }
        let mut bt = t0.clone();
        let bad_ok = public.verify(&commitment, bad.clone(), &mut bt);
        negatives.push(json!({
            "commitment": abi_point(&commitment), "proof": abi_proof(&bad), "state_before": hex::encode(tstate::to_bytes(&t0)),
            "state_after_verify": hex::encode(tstate::to_bytes(&bt)), "accept": bad_ok,
        }));
    }
    let merlin_kat = {
        let mut t = Transcript::new(b"test protocol");
        t.append_message(b"some label", b"some data");
        let mut c = [0u8; 32];
        t.challenge_bytes(b"challenge", &mut c);
        hex::encode(c)
    };
    let doc = json!({
        "source": "distributed-lab/bp-pp 0.1.1 (k256 0.13.3, merlin 3.0.0), facade/src/bin/gen_fixtures.rs",
        "seed": hex::encode(seed), "label": hex::encode(label), "generators": gens, "generator_rng_bytes": hex::encode(&gen_bytes),
        "cases": cases, "negative_cases": negatives,
        // encodings the oracle only "believes" (SURVEY appendix A): settled here by the library itself
        "identity_to_bytes": hex::encode(ProjectivePoint::IDENTITY.to_bytes()),
        "identity_json": serde_json::to_value(&AffinePoint::IDENTITY).unwrap(),
        "scalar_json_example": serde_json::to_value(&Scalar::from(0xABCDEFu32)).unwrap(),
        "merlin_kat": { "challenge": merlin_kat },
    });
    println!("{}", serde_json::to_string_pretty(&doc).unwrap());
}
