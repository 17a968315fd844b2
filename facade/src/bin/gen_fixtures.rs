//! Fixture emitter: runs the REAL reference (bp-pp 0.1.1 on k256 0.13.3 / merlin 3.0.0) with a seeded RNG and prints, as JSON on
//! stdout, everything needed to pin this repository's oracle and HIP path to it:
//!
//!   cargo run --release --no-default-features --bin gen_fixtures > ../tests/golden/ref_u64.json
//!   cargo run --release --no-default-features --bin gen_fixtures -- generic ../tests/golden/statements_generic.json > ../tests/golden/ref_generic.json
//!
//! The second form pins the crate's `circuit` and `wnla` modules: it reads STATEMENTS (dimensions, W_m / W_l / a_m / a_l, the partition
//! as index tables, a witness; WNLA: l and n) written by tests/golden/make_statements_generic.py -- the reference's own `ac_works` and
//! `wnla_works` (tests.rs:45-171) and the k > 1 / f_m / f_l-and-f_m shapes of tests/circuit_cases.py --, draws generators and blindings
//! from the seeded RNG, runs the reference's prover and the reference's OWN verifier on its output, and records both: for the f_l-and-f_m
//! shape this repository's oracle predicts `accept: false` (circuit.rs:559-614); only this run can confirm or refute that.
//!
//! (no GPU and no libbppp_hip.so needed).  tests/test_ref_fixtures.py and tests/test_gpu_ref_fixtures.py consume the file when it
//! exists and skip -- saying so -- when it does not.  UNCOMPILED: written without a Rust toolchain (see lib.rs).
//!
//! Per case the file records the prover's inputs INCLUDING every byte it pulled from the RNG, so the consumers can (a) replay the
//! prover and demand byte-identical proofs, (b) check the model of `Scalar::generate_biased` (64 bytes, big-endian, reduced mod
//! n), (c) check accept bits, (d) check the transcript state the reference leaves behind (`t: &mut Transcript`), and (e) settle
//! the serde / hex conventions (`proof_json`, `commitment_json`, `identity_json`).
use bp_pp::circuit::{ArithmeticCircuit, PartitionType, Witness};
use bp_pp::range_proof::reciprocal::{Proof, SerializableProof};
use bp_pp::wnla::WeightNormLinearArgument;
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

fn hex_scalar(v: &serde_json::Value) -> Scalar {
    let b = hex::decode(v.as_str().expect("a hex string")).expect("hex");
    conv::get_scalar(&b).expect("a canonical scalar")
}
fn hex_scalars(v: &serde_json::Value) -> Vec<Scalar> {
    v.as_array().expect("an array of hex scalars").iter().map(hex_scalar).collect()
}
fn abi_points(ps: &[ProjectivePoint]) -> String {
    let mut v = Vec::new();
    ps.iter().for_each(|p| conv::put_point(&mut v, p));
    hex::encode(v)
}
fn abi_scalars(ss: &[Scalar]) -> String {
    let mut v = Vec::new();
    ss.iter().for_each(|x| conv::put_scalar(&mut v, x));
    hex::encode(v)
}
/// the scalars a prover drew, by k256's own reduction of the recorded bytes (one `generate_biased` per recorded request)
fn replay_scalars(bytes: &[u8], calls: &[usize]) -> Vec<Scalar> {
    let mut rp = Replay(bytes, 0);
    calls.iter().map(|_| Scalar::generate_biased(&mut rp)).collect()
}
fn pow2_at_least(n: usize) -> usize {
    let mut p = 1;
    while p < n {
        p *= 2;
    }
    p
}

/// `generic` mode: circuit.rs and wnla.rs through the reference itself (see the module comment).
fn generic(statements_path: &str) {
    let seed = *b"bppp-ref-fixtures-v1-seed-000002";
    let mut rng = RecordingRng { inner: ChaCha20Rng::from_seed(seed), bytes: vec![], calls: vec![] };
    let st: serde_json::Value = serde_json::from_str(&std::fs::read_to_string(statements_path).expect("the statements file")).expect("JSON");
    let mut circuits = vec![];
    for c in st["circuits"].as_array().expect("circuits") {
        let usz = |k: &str| c[k].as_u64().expect(k) as usize;
        let (dim_nm, dim_no, dim_nv, k) = (usz("dim_nm"), usz("dim_no"), usz("dim_nv"), usz("k"));
        let (dim_nl, dim_nw) = (dim_nv * k, 2 * dim_nm + dim_no);
        let label: &'static [u8] = Box::leak(hex::decode(c["label"].as_str().unwrap()).unwrap().into_boxed_slice());
        let rows = |key: &str| -> Vec<Vec<Scalar>> { c[key].as_array().unwrap().iter().map(hex_scalars).collect() };
        let table = |t: &str| -> Vec<i64> { c["partition"][t].as_array().unwrap().iter().map(|v| v.as_i64().unwrap()).collect() };
        let (lo, ll, lr, no) = (table("LO"), table("LL"), table("LR"), table("NO"));
        // generators as the reference's own tests draw them (tests.rs:78-80); padded to the powers of two the WNLA stage needs
        let (ng, nh) = (pow2_at_least(dim_nm), pow2_at_least(dim_nv + 9));
        let g = ProjectivePoint::random(&mut rng);
        let g_all: Vec<ProjectivePoint> = (0..ng).map(|_| ProjectivePoint::random(&mut rng)).collect();
        let h_all: Vec<ProjectivePoint> = (0..nh).map(|_| ProjectivePoint::random(&mut rng)).collect();
        rng.take();
        let partition = move |typ: PartitionType, index: usize| -> Option<usize> {
            let t = match typ {
                PartitionType::LO => &lo,
                PartitionType::LL => &ll,
                PartitionType::LR => &lr,
                PartitionType::NO => &no,
            };
            t.get(index).and_then(|v| if *v < 0 { None } else { Some(*v as usize) })
        };
        let circuit = ArithmeticCircuit {
            dim_nm, dim_no, k, dim_nl, dim_nv, dim_nw,
            g,
            g_vec: g_all[..dim_nm].to_vec(),
            h_vec: h_all[..9 + dim_nv].to_vec(),
            W_m: rows("W_m"),
            W_l: rows("W_l"),
            a_m: hex_scalars(&c["a_m"]),
            a_l: hex_scalars(&c["a_l"]),
            f_l: c["f_l"].as_bool().unwrap(),
            f_m: c["f_m"].as_bool().unwrap(),
            g_vec_: g_all[dim_nm..].to_vec(),
            h_vec_: h_all[9 + dim_nv..].to_vec(),
            partition,
        };
        let v_rows = rows("v");
        let mut instances = vec![];
        for _ in 0..c["instances"].as_u64().unwrap_or(1) {
            let s_v: Vec<Scalar> = (0..k).map(|_| Scalar::generate_biased(&mut rng)).collect();
            rng.take();
            let witness = Witness { v: v_rows.clone(), s_v: s_v.clone(), w_l: hex_scalars(&c["w_l"]), w_r: hex_scalars(&c["w_r"]), w_o: hex_scalars(&c["w_o"]) };
            let v: Vec<ProjectivePoint> = (0..k).map(|i| circuit.commit(&witness.v[i], &witness.s_v[i])).collect();
            let mut pt = Transcript::new(label);
            let proof = circuit.prove(&v, witness, &mut pt, &mut rng);
            let (rng_bytes, rng_calls) = rng.take();
            let rnd = replay_scalars(&rng_bytes, &rng_calls);
            let mut vt = Transcript::new(label);
            let accept = circuit.verify(&v, &mut vt, proof.clone());
            // the C ABI's layout: c_l, c_r, c_o, c_s | r[rounds] | x[rounds] | l[nl] | n[nn]
            let mut pb = Vec::new();
            for q in [&proof.c_l, &proof.c_r, &proof.c_o, &proof.c_s] {
                conv::put_point(&mut pb, q);
            }
            proof.r.iter().chain(proof.x.iter()).for_each(|q| conv::put_point(&mut pb, q));
            proof.l.iter().chain(proof.n.iter()).for_each(|x| conv::put_scalar(&mut pb, x));
            instances.push(json!({
                "s_v": abi_scalars(&s_v), "rng_bytes": hex::encode(&rng_bytes), "rng_calls": rng_calls, "rnd": abi_scalars(&rnd),
                "commitments": abi_points(&v), "proof": hex::encode(&pb), "rounds": proof.r.len(), "nl": proof.l.len(), "nn": proof.n.len(),
                "state_after_prove": hex::encode(tstate::to_bytes(&pt)), "state_after_verify": hex::encode(tstate::to_bytes(&vt)),
                "accept": accept,
            }));
        }
        circuits.push(json!({
            "name": c["name"], "label": c["label"], "g": abi_points(&[g]), "g_vec": abi_points(&g_all[..dim_nm]), "h_vec": abi_points(&h_all[..9 + dim_nv]),
            "g_vec_": abi_points(&g_all[dim_nm..]), "h_vec_": abi_points(&h_all[9 + dim_nv..]), "instances": instances,
        }));
    }
    let mut wnlas = vec![];
    for w in st["wnla"].as_array().expect("wnla") {
        let (ng, nh) = (w["ng"].as_u64().unwrap() as usize, w["nh"].as_u64().unwrap() as usize);
        let label: &'static [u8] = Box::leak(hex::decode(w["label"].as_str().unwrap()).unwrap().into_boxed_slice());
        // tests.rs:141-150: generators, c and rho from the RNG, mu = rho^2
        let g = ProjectivePoint::random(&mut rng);
        let g_vec: Vec<ProjectivePoint> = (0..ng).map(|_| ProjectivePoint::random(&mut rng)).collect();
        let h_vec: Vec<ProjectivePoint> = (0..nh).map(|_| ProjectivePoint::random(&mut rng)).collect();
        let c: Vec<Scalar> = (0..nh).map(|_| Scalar::generate_biased(&mut rng)).collect();
        let rho = Scalar::generate_biased(&mut rng);
        rng.take();
        let mu = rho * rho;
        let arg = WeightNormLinearArgument { g, g_vec: g_vec.clone(), h_vec: h_vec.clone(), c: c.clone(), rho, mu };
        let (l, n) = (hex_scalars(&w["l"]), hex_scalars(&w["n"]));
        let commit = arg.commit(&l, &n);
        let mut pt = Transcript::new(label);
        let proof = arg.prove(&commit, &mut pt, l.clone(), n.clone());
        let mut vt = Transcript::new(label);
        let accept = arg.verify(&commit, &mut vt, proof.clone());
        wnlas.push(json!({
            "name": w["name"], "label": w["label"], "ng": ng, "nh": nh, "g": abi_points(&[g]), "g_vec": abi_points(&g_vec), "h_vec": abi_points(&h_vec),
            "c": abi_scalars(&c), "rho": abi_scalars(&[rho]), "mu": abi_scalars(&[mu]), "l": abi_scalars(&l), "n": abi_scalars(&n),
            "commitment": abi_points(&[commit]), "proof_r": abi_points(&proof.r), "proof_x": abi_points(&proof.x),
            "proof_l": abi_scalars(&proof.l), "proof_n": abi_scalars(&proof.n),
            "state_after_prove": hex::encode(tstate::to_bytes(&pt)), "state_after_verify": hex::encode(tstate::to_bytes(&vt)), "accept": accept,
        }));
    }
    let doc = json!({
        "source": "distributed-lab/bp-pp 0.1.1 (k256 0.13.3, merlin 3.0.0), facade/src/bin/gen_fixtures.rs generic",
        "seed": hex::encode(seed), "statements": statements_path, "circuits": circuits, "wnla": wnlas,
    });
    println!("{}", serde_json::to_string_pretty(&doc).unwrap());
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    if args.len() >= 3 && args[1] == "generic" {
        return generic(&args[2]);
    }
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
