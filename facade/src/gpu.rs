//! GPU twin of `bp_pp::range_proof::u64_proof::U64RangeProofProtocol` (u64_proof.rs:19-82): same public parameters, same method
//! names and argument meaning, batch-first.  UNCOMPILED (see lib.rs).
use crate::conv::*;
use crate::ffi::*;
use crate::tstate;
use bp_pp::range_proof::reciprocal::Proof;
use bp_pp::range_proof::u64_proof::U64RangeProofProtocol;
use k256::{ProjectivePoint, Scalar};
use merlin::Transcript;
use rand_core::{CryptoRng, RngCore};
use std::ffi::CStr;

#[derive(Debug)]
pub enum GpuError {
    /// no gfx950 device / HIP failure / RCCL failure: the library has NO CPU fallback, the caller decides what to do
    Library { code: i32, detail: String },
    /// the reference would have panicked on this proof (`unwrap()` of a challenge >= n or of a zero inverse:
    /// transcript.rs:13, circuit.rs:192,196, reciprocal.rs:181, util.rs:119)
    ReferenceWouldPanic { index: usize },
}

fn check(rc: i32) -> Result<(), GpuError> {
    if rc >= 0 {
        return Ok(());
    }
    let detail = unsafe { CStr::from_ptr(bppp_last_error()) }.to_string_lossy().into_owned();
    Err(GpuError::Library { code: rc, detail })
}

pub struct U64RangeProofProtocolGpu {
    ctx: *mut BpppCtx,
    /// the crate's own protocol object: serves the proof shapes the 928-byte form cannot carry (CPU path of the CRATE, chosen
    /// here in the facade -- never inside libbppp_hip.so)
    cpu: U64RangeProofProtocol,
}

// the batched calls hold the context's lock (include/bppp.h, Threading): from several threads they run one after the other; the
// single-proof calls (verify / prove below) do not hold it while they wait -- they are gathered into shared batched calls
unsafe impl Send for U64RangeProofProtocolGpu {}
unsafe impl Sync for U64RangeProofProtocolGpu {}

impl U64RangeProofProtocolGpu {
    /// fb_window_bits: 0 = library default; see bppp_ctx_create.
    pub fn new(p: &U64RangeProofProtocol, device: i32, fb_window_bits: i32) -> Result<Self, GpuError> {
        assert_eq!(p.g_vec.len(), 16); // u64_proof.rs:12
        assert_eq!(p.h_vec.len(), 32); // u64_proof.rs:14
        let (mut g, mut gv, mut hv) = (Vec::new(), Vec::new(), Vec::new());
        put_point(&mut g, &p.g);
        p.g_vec.iter().for_each(|q| put_point(&mut gv, q));
        p.h_vec.iter().for_each(|q| put_point(&mut hv, q));
        // The single-proof calls hand `merlin::Transcript` to the library as 203 serialized bytes read through a pointer cast
        // (tstate.rs): before anything relies on that layout, compare it with a state computed WITHOUT the cast -- the library's own
        // host-side `Transcript::new(label)` -- and refuse to construct the object if merlin's struct ever stops looking like that.
        const CHECK_LABEL: &[u8] = b"bp-pp-gpu layout self-check";
        let mut independent = [0u8; tstate::STATE_BYTES];
        check(unsafe { bppp_transcript_new(CHECK_LABEL.as_ptr(), CHECK_LABEL.len(), independent.as_mut_ptr()) })?;
        if !tstate::self_check(CHECK_LABEL, &independent) {
            return Err(GpuError::Library { code: BPPP_ERR_INVALID_ARG, detail: "merlin::Transcript is not laid out as 200 state bytes + pos, pos_begin, cur_flags: \
                                                                          the transcript bridge of this facade does not apply to this merlin version".into() });
        }
        let mut ctx = std::ptr::null_mut();
        check(unsafe { bppp_ctx_create(&mut ctx, g.as_ptr(), gv.as_ptr(), hv.as_ptr(), device, fb_window_bits) })?;
        Ok(Self { ctx, cpu: p.clone() })
    }

    // ------------------------------------------------------------------------------------------------------------------------
    // The reference's own three methods, with the reference's exact signatures (u64_proof.rs:37, :42, :57-59): ONE proof per call,
    // `&self`, callable from any number of threads at once (the type is Send + Sync like the crate's).  They go through the
    // library's coalescing front end (bppp_u64_verify_one / bppp_u64_prove_one): the calling thread sleeps while its request rides
    // in a batched GPU call together with whatever the other threads submitted -- 64 threads: > 20,000 verifies/s instead of the
    // 600 that 64 separate one-proof launches gave.  A library failure (no device, out of memory) cannot be a `false`, so -- like
    // the reference's own `unwrap()`s -- it panics; the try_ forms return it as an error instead.

    /// `U64RangeProofProtocol::verify(&self, v, proof, t) -> bool` (u64_proof.rs:42-54).
    pub fn verify(&self, v: &ProjectivePoint, proof: Proof, t: &mut Transcript) -> bool {
        self.try_verify(v, proof, t).expect("bppp: GPU verify failed")
    }
    pub fn try_verify(&self, v: &ProjectivePoint, proof: Proof, t: &mut Transcript) -> Result<bool, GpuError> {
        let mut pb = Vec::with_capacity(928);
        if put_u64_proof(&mut pb, &proof).is_none() {
            return Ok(self.cpu.verify(v, proof, t)); // non-standard proof shape: the crate's CPU verifier (chosen here, not in the library)
        }
        let mut cb = Vec::with_capacity(64);
        put_point(&mut cb, v);
        let mut st8 = tstate::to_bytes(t);
        let (mut acc, mut st) = (0u8, 0i32);
        check(unsafe { bppp_u64_verify_one_transcript(self.ctx, st8.as_mut_ptr(), cb.as_ptr(), pb.as_ptr(), &mut acc, &mut st) })?;
        if st & BPPP_ST_DEGENERATE != 0 {
            return Err(GpuError::ReferenceWouldPanic { index: 0 });
        }
        *t = tstate::from_bytes(&st8); // advanced exactly as the reference's verify leaves it (untouched for a proof k256 would not deserialize)
        Ok(acc == 1)
    }

    /// `U64RangeProofProtocol::prove(&self, x, s, t, rng) -> Proof` (u64_proof.rs:57-82): the 52 `Scalar::generate_biased(rng)` draws
    /// are made here, in the reference's order, so the proof is the one the CPU prover emits from the same RNG stream.
    pub fn prove<R: RngCore + CryptoRng>(&self, x: u64, s: &Scalar, t: &mut Transcript, rng: &mut R) -> Proof {
        self.try_prove(x, s, t, rng).expect("bppp: GPU prove failed")
    }
    pub fn try_prove<R: RngCore + CryptoRng>(&self, x: u64, s: &Scalar, t: &mut Transcript, rng: &mut R) -> Result<Proof, GpuError> {
        let mut rnd = Vec::with_capacity(52 * 32);
        for _ in 0..52 {
            put_scalar(&mut rnd, &Scalar::generate_biased(&mut *rng));
        }
        let mut sb = Vec::with_capacity(32);
        put_scalar(&mut sb, s);
        let mut st8 = tstate::to_bytes(t);
        let (mut proof, mut com, mut st) = ([0u8; 928], [0u8; 64], 0i32);
        check(unsafe {
            bppp_u64_prove_one_transcript(self.ctx, st8.as_mut_ptr(), x, sb.as_ptr(), rnd.as_ptr(), proof.as_mut_ptr(), com.as_mut_ptr(), &mut st)
        })?;
        if st != 0 {
            return Err(GpuError::ReferenceWouldPanic { index: 0 });
        }
        *t = tstate::from_bytes(&st8);
        Ok(get_u64_proof(&proof).expect("library emitted an invalid proof"))
    }

    /// `U64RangeProofProtocol::commit_value(&self, x, s) -> ProjectivePoint` (u64_proof.rs:37-39).
    pub fn commit_value(&self, x: u64, s: &Scalar) -> ProjectivePoint {
        self.commit_value_batch(&[x], std::slice::from_ref(s)).expect("bppp: GPU commit failed").remove(0)
    }

    /// `commit_value(x, s)` (u64_proof.rs:37-39) for a batch.
    pub fn commit_value_batch(&self, xs: &[u64], ss: &[Scalar]) -> Result<Vec<ProjectivePoint>, GpuError> {
        assert_eq!(xs.len(), ss.len());
        let mut sb = Vec::with_capacity(32 * xs.len());
        ss.iter().for_each(|s| put_scalar(&mut sb, s));
        let mut out = vec![0u8; 64 * xs.len()];
        check(unsafe { bppp_u64_commit_value_batch(self.ctx, xs.len(), xs.as_ptr(), sb.as_ptr(), out.as_mut_ptr()) })?;
        Ok(out.chunks(64).map(|b| get_point(b).expect("library emitted a point off the curve")).collect())
    }

    /// n x `verify(v, proof, &mut Transcript::new(label))` (u64_proof.rs:42-54; the call pattern of benches/range_proof.rs:47).
    pub fn verify_batch(&self, label: &'static [u8], vs: &[ProjectivePoint], proofs: &[Proof]) -> Result<Vec<bool>, GpuError> {
        let t = Transcript::new(label);
        let mut ts: Vec<Transcript> = vec![t; vs.len()];
        self.verify_batch_transcripts(vs, proofs, &mut ts)
    }

    /// n x `verify(v, proof, t)` with the CALLER's transcripts: each `ts[i]` may already hold appended context, and comes back
    /// advanced exactly as the reference's verify would leave it (Ownership).
    pub fn verify_batch_transcripts(&self, vs: &[ProjectivePoint], proofs: &[Proof], ts: &mut [Transcript]) -> Result<Vec<bool>, GpuError> {
        assert!(vs.len() == proofs.len() && vs.len() == ts.len());
        let n = vs.len();
        let (mut cb, mut pb, mut sb) = (Vec::with_capacity(64 * n), Vec::with_capacity(928 * n), Vec::with_capacity(203 * n));
        let mut on_gpu = vec![true; n];
        for i in 0..n {
            put_point(&mut cb, &vs[i]);
            sb.extend_from_slice(&tstate::to_bytes(&ts[i]));
            if put_u64_proof(&mut pb, &proofs[i]).is_none() {
                on_gpu[i] = false; // non-standard shape: the crate's CPU verifier below
                pb.extend_from_slice(&[0u8; 928]);
            }
        }
        let (mut acc, mut st, mut out) = (vec![0u8; n], vec![0i32; n], vec![0u8; 203 * n]);
        check(unsafe {
            bppp_u64_verify_batch_transcript(self.ctx, n, sb.as_ptr(), n, cb.as_ptr(), pb.as_ptr(), acc.as_mut_ptr(), st.as_mut_ptr(), out.as_mut_ptr())
        })?;
        let mut res = Vec::with_capacity(n);
        for i in 0..n {
            if !on_gpu[i] {
                res.push(self.cpu.verify(&vs[i], proofs[i].clone(), &mut ts[i]));
                continue;
            }
            if st[i] & BPPP_ST_DEGENERATE != 0 {
                return Err(GpuError::ReferenceWouldPanic { index: i });
            }
            let mut s = [0u8; 203];
            s.copy_from_slice(&out[203 * i..203 * i + 203]);
            ts[i] = tstate::from_bytes(&s);
            res.push(acc[i] == 1);
        }
        Ok(res)
    }

    /// n x `prove(x, s, &mut Transcript::new(label), rng)` (u64_proof.rs:57-82).  The 52 `Scalar::generate_biased(rng)` draws per
    /// proof are made HERE, proof after proof in the reference's order (reciprocal.rs:121; circuit.rs:264-298; :371-372), so the
    /// GPU emits the proofs the CPU prover would have emitted from the same RNG stream.
    pub fn prove_batch<R: RngCore + CryptoRng>(&self, label: &'static [u8], xs: &[u64], ss: &[Scalar], rng: &mut R) -> Result<(Vec<Proof>, Vec<ProjectivePoint>), GpuError> {
        assert_eq!(xs.len(), ss.len());
        let n = xs.len();
        let mut rnd = Vec::with_capacity(n * 52 * 32);
        for _ in 0..n * 52 {
            put_scalar(&mut rnd, &Scalar::generate_biased(&mut *rng));
        }
        let mut sb = Vec::with_capacity(n * 32);
        ss.iter().for_each(|s| put_scalar(&mut sb, s));
        let (mut proofs, mut coms, mut st) = (vec![0u8; 928 * n], vec![0u8; 64 * n], vec![0i32; n]);
        check(unsafe {
            bppp_u64_prove_batch(self.ctx, label.as_ptr(), label.len(), n, xs.as_ptr(), sb.as_ptr(), rnd.as_ptr(), proofs.as_mut_ptr(), coms.as_mut_ptr(), st.as_mut_ptr())
        })?;
        if let Some(i) = st.iter().position(|s| *s != 0) {
            return Err(GpuError::ReferenceWouldPanic { index: i });
        }
        Ok((proofs.chunks(928).map(|b| get_u64_proof(b).expect("library emitted an invalid proof")).collect(),
            coms.chunks(64).map(|b| get_point(b).expect("library emitted a point off the curve")).collect()))
    }
}

impl Drop for U64RangeProofProtocolGpu {
    fn drop(&mut self) {
        unsafe { bppp_ctx_destroy(self.ctx) }
    }
}

/// One batch over several GPUs of a node (bppp_group_*): contiguous split by proof index, one 4-byte RCCL all-reduce of the reject
/// count.  Returns the accept bits and the global number of rejected proofs.
pub struct U64RangeProofGroupGpu {
    grp: *mut BpppGroup,
}

impl U64RangeProofGroupGpu {
    pub fn new(p: &U64RangeProofProtocol, devices: &[i32], fb_window_bits: i32) -> Result<Self, GpuError> {
        let (mut g, mut gv, mut hv) = (Vec::new(), Vec::new(), Vec::new());
        put_point(&mut g, &p.g);
        p.g_vec.iter().for_each(|q| put_point(&mut gv, q));
        p.h_vec.iter().for_each(|q| put_point(&mut hv, q));
        let mut grp = std::ptr::null_mut();
        check(unsafe { bppp_group_create(&mut grp, g.as_ptr(), gv.as_ptr(), hv.as_ptr(), devices.as_ptr(), devices.len() as i32, fb_window_bits) })?;
        Ok(Self { grp })
    }

    pub fn verify_batch(&self, label: &'static [u8], vs: &[ProjectivePoint], proofs: &[Proof]) -> Result<(Vec<bool>, i32), GpuError> {
        let n = vs.len();
        let (mut cb, mut pb) = (Vec::with_capacity(64 * n), Vec::with_capacity(928 * n));
        for i in 0..n {
            put_point(&mut cb, &vs[i]);
            put_u64_proof(&mut pb, &proofs[i]).expect("non-standard proof shape: use U64RangeProofProtocolGpu::verify_batch");
        }
        let (mut acc, mut st, mut rej) = (vec![0u8; n], vec![0i32; n], 0i32);
        check(unsafe {
            bppp_u64_verify_batch_sharded(self.grp, label.as_ptr(), label.len(), n, cb.as_ptr(), pb.as_ptr(), acc.as_mut_ptr(), st.as_mut_ptr(), &mut rej)
        })?;
        if let Some(i) = st.iter().position(|s| s & BPPP_ST_DEGENERATE != 0) {
            return Err(GpuError::ReferenceWouldPanic { index: i });
        }
        Ok((acc.iter().map(|a| *a == 1).collect(), rej))
    }

    /// `U64RangeProofProtocol::prove` for one batch over the group's GPUs (bppp_u64_prove_batch_sharded): the 52 scalars per proof
    /// are drawn here, in the reference's order, so the proofs equal the reference prover's for the same RNG stream.
    pub fn prove_batch<R: RngCore + CryptoRng>(&self, label: &'static [u8], xs: &[u64], ss: &[Scalar], rng: &mut R) -> Result<(Vec<Proof>, Vec<ProjectivePoint>), GpuError> {
        assert_eq!(xs.len(), ss.len());
        let n = xs.len();
        let mut rnd = Vec::with_capacity(n * 52 * 32);
        for _ in 0..n * 52 {
            put_scalar(&mut rnd, &Scalar::generate_biased(&mut *rng));
        }
        let mut sb = Vec::with_capacity(n * 32);
        ss.iter().for_each(|s| put_scalar(&mut sb, s));
        let (mut proofs, mut coms, mut st) = (vec![0u8; 928 * n], vec![0u8; 64 * n], vec![0i32; n]);
        check(unsafe {
            bppp_u64_prove_batch_sharded(self.grp, label.as_ptr(), label.len(), n, xs.as_ptr(), sb.as_ptr(), rnd.as_ptr(), proofs.as_mut_ptr(), coms.as_mut_ptr(), st.as_mut_ptr())
        })?;
        if let Some(i) = st.iter().position(|s| *s != 0) {
            return Err(GpuError::ReferenceWouldPanic { index: i });
        }
        Ok((proofs.chunks(928).map(|b| get_u64_proof(b).expect("library emitted an invalid proof")).collect(),
            coms.chunks(64).map(|b| get_point(b).expect("library emitted a point off the curve")).collect()))
    }
}

impl Drop for U64RangeProofGroupGpu {
    fn drop(&mut self) {
        unsafe { bppp_group_destroy(self.grp) }
    }
}
