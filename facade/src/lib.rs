//! GPU batch backend for `bp-pp` (distributed-lab/bp-pp 0.1.1) behind the crate's own types.
//!
//! * [`ffi`]   -- every `extern "C"` entry point of `include/bppp.h` (generated from the header);
//! * [`conv`]  -- k256 / bp-pp values <-> the byte layouts of the C ABI (and the serde forms);
//! * [`tstate`] -- `merlin::Transcript` <-> the 203 serialized bytes of the pre-loaded-transcript entry points;
//! * [`gpu`]   -- `U64RangeProofProtocolGpu`: the same method names and argument meaning as
//!               `bp_pp::range_proof::u64_proof::U64RangeProofProtocol`, batch-first.
//!
//! STATUS: this tree has never been compiled -- the image it was written in has no Rust toolchain and no crate registry.
//! It is kept in the repository so that the parity pin can close the day one exists: `cargo run --no-default-features --bin
//! gen_fixtures > ../tests/golden/ref_u64.json` runs the REAL reference with a seeded RNG and writes the fixture that
//! `tests/test_ref_fixtures.py` (CPU oracle) and `tests/test_gpu_ref_fixtures.py` (HIP path) consume when present.
pub mod conv;
pub mod tstate;
#[cfg(feature = "gpu")]
pub mod ffi;
#[cfg(feature = "gpu")]
pub mod gpu;
