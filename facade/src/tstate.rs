//! `merlin::Transcript` <-> the 203 serialized bytes the C ABI exchanges (include/bppp.h, BPPP_TRANSCRIPT_STATE_BYTES):
//! 200 bytes of Keccak-f[1600] state, then pos, pos_begin, cur_flags -- exactly the fields of merlin 3.0.0's `Strobe128`
//! (`state: AlignedKeccakState([u8; 200])`, `pos: u8`, `pos_begin: u8`, `cur_flags: u8`; src/strobe.rs), which is the only field
//! of `Transcript`.  merlin does not expose them, so the bytes are read through a pointer cast; the layout assumption is
//! checked at compile time (size) and at run time (`self_check`: a fresh transcript's bytes must equal the independently
//! computed `Transcript::new(label)` state, and a round trip must reproduce a challenge).
use merlin::Transcript;

pub const STATE_BYTES: usize = 203;

// Strobe128 is 200 + 3 bytes with 8-byte alignment -> 208; Transcript wraps exactly one Strobe128.
const _: () = assert!(std::mem::size_of::<Transcript>() == 208);

pub fn to_bytes(t: &Transcript) -> [u8; STATE_BYTES] {
    let mut out = [0u8; STATE_BYTES];
    // SAFETY: Transcript is plain old data of 208 bytes (asserted above); only the first 203 are meaningful.
    unsafe { std::ptr::copy_nonoverlapping(t as *const Transcript as *const u8, out.as_mut_ptr(), STATE_BYTES) };
    out
}

pub fn from_bytes(b: &[u8; STATE_BYTES]) -> Transcript {
    let mut t = Transcript::new(b"");
    // SAFETY: as above; the bytes come from `to_bytes` or from libbppp_hip.so, which only emits states merlin can be in.
    unsafe { std::ptr::copy_nonoverlapping(b.as_ptr(), &mut t as *mut Transcript as *mut u8, STATE_BYTES) };
    t
}

/// Layout check against a state computed WITHOUT this cast (e.g. by `bppp_transcript_new`): call once at start-up.
pub fn self_check(label: &'static [u8], independent_state: &[u8; STATE_BYTES]) -> bool {
    let t = Transcript::new(label);
    if &to_bytes(&t) != independent_state {
        return false;
    }
    let (mut a, mut b) = (t.clone(), from_bytes(&to_bytes(&t)));
    let (mut ca, mut cb) = ([0u8; 32], [0u8; 32]);
    a.challenge_bytes(b"self-check", &mut ca);
    b.challenge_bytes(b"self-check", &mut cb);
    ca == cb
}
