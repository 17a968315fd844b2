//! `extern "C"` declarations of EVERY entry point of include/bppp.h (libbppp_hip.so).  Generated from the header by
//! tools/gen_facade_ffi.py and checked against it by tests/test_facade_tree.py.  UNCOMPILED (no Rust toolchain in the build
//! image); the identical prototypes are exercised through ctypes in bp_pp_amd/_capi.py.
#![allow(dead_code)]
use std::os::raw::{c_char, c_int, c_long, c_void};

#[repr(C)] pub struct BpppCtx { _private: [u8; 0] }
#[repr(C)] pub struct BpppCircuit { _private: [u8; 0] }
#[repr(C)] pub struct BpppGroup { _private: [u8; 0] }

pub const BPPP_OK: c_int = 0;
pub const BPPP_ERR_NO_DEVICE: c_int = -1;
pub const BPPP_ERR_INVALID_ARG: c_int = -2;
pub const BPPP_ERR_HIP: c_int = -3;
pub const BPPP_ERR_ENCODING: c_int = -4;
pub const BPPP_ERR_NOMEM: c_int = -5;
pub const BPPP_ERR_RCCL: c_int = -6;
pub const BPPP_ST_BAD_ENCODING: i32 = 1;
pub const BPPP_ST_DEGENERATE: i32 = 2;
pub const POINT_BYTES: usize = 64;
pub const SCALAR_BYTES: usize = 32;
pub const U64_PROOF_BYTES: usize = 928;
pub const U64_PROOF_SEC1_BYTES: usize = 525;
pub const TRANSCRIPT_STATE_BYTES: usize = 203;

extern "C" {
    pub fn bppp_ctx_create(out: *mut *mut BpppCtx, g: *const u8, g_vec: *const u8, h_vec: *const u8, device: c_int, fb_window_bits: c_int) -> c_int;
    pub fn bppp_ctx_destroy(ctx: *mut BpppCtx);
    pub fn bppp_ctx_set_stream(ctx: *mut BpppCtx, hip_stream: *mut c_void) -> c_int;
    pub fn bppp_ctx_set_option(ctx: *mut BpppCtx, name: *const c_char, value: c_long) -> c_int;
    pub fn bppp_ctx_get_option(ctx: *mut BpppCtx, name: *const c_char) -> c_long;
    pub fn bppp_u64_plan(prove: c_int, n: usize, n_simds: c_int, flags: c_int) -> c_long;
    pub fn bppp_plan_describe(code: c_long, prove: c_int, buf: *mut c_char, cap: usize) -> c_int;
    pub fn bppp_ctx_synchronize(ctx: *mut BpppCtx) -> c_int;
    pub fn bppp_u64_verify_batch(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, commitments: *const u8, proofs: *const u8, accept: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_u64_verify_batch_device(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, d_commitments: *const c_void, d_proofs: *const c_void, d_accept: *mut c_void, d_status: *mut c_void, d_trace: *mut c_void, d_reject_count: *mut c_void) -> c_int;
    pub fn bppp_u64_verify_batch_rlc_device(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, d_commitments: *const c_void, d_proofs: *const c_void, d_accept: *mut c_void, d_status: *mut c_void, d_reject_count: *mut c_void, seed: *const u8) -> c_int;
    pub fn bppp_u64_verify_batch_rlc(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, commitments: *const u8, proofs: *const u8, accept: *mut u8, status: *mut i32, seed: *const u8) -> c_int;
    pub fn bppp_u64_verify_batch_sec1(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, commitments: *const u8, proofs: *const u8, accept: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_u64_verify_batch_sec1_device(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, d_commitments: *const c_void, d_proofs: *const c_void, d_accept: *mut c_void, d_status: *mut c_void, d_trace: *mut c_void, d_reject_count: *mut c_void) -> c_int;
    pub fn bppp_u64_prove_batch(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, x: *const u64, s: *const u8, rnd: *const u8, proofs: *mut u8, commitments: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_u64_prove_batch_device(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, d_x: *const c_void, d_s: *const c_void, d_rnd: *const c_void, d_proofs: *mut c_void, d_commitments: *mut c_void, d_status: *mut c_void) -> c_int;
    pub fn bppp_u64_prove_batch_sec1(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, x: *const u64, s: *const u8, rnd: *const u8, proofs525: *mut u8, commitments33: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_u64_prove_batch_sec1_device(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, d_x: *const c_void, d_s: *const c_void, d_rnd: *const c_void, d_proofs525: *mut c_void, d_commitments33: *mut c_void, d_status: *mut c_void) -> c_int;
    pub fn bppp_u64_verify_one(ctx: *mut BpppCtx, label: *const u8, label_len: usize, commitment: *const u8, proof: *const u8, accept: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_u64_verify_one_transcript(ctx: *mut BpppCtx, state: *mut u8, commitment: *const u8, proof: *const u8, accept: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_u64_prove_one(ctx: *mut BpppCtx, label: *const u8, label_len: usize, x: u64, s: *const u8, rnd: *const u8, proof: *mut u8, commitment: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_u64_prove_one_transcript(ctx: *mut BpppCtx, state: *mut u8, x: u64, s: *const u8, rnd: *const u8, proof: *mut u8, commitment: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_reciprocal_verify_one(ctx: *mut BpppCtx, label: *const u8, label_len: usize, dim_nd: usize, dim_np: usize, commitment: *const u8, proof: *const u8, rounds: usize, nl: usize, nn: usize, accept: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_reciprocal_verify_one_transcript(ctx: *mut BpppCtx, state: *mut u8, dim_nd: usize, dim_np: usize, commitment: *const u8, proof: *const u8, rounds: usize, nl: usize, nn: usize, accept: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_ctx_get_coalesce_stats(ctx: *mut BpppCtx, which: c_int, out: *mut u64) -> c_int;
    pub fn bppp_u64_commit_value_batch(ctx: *mut BpppCtx, n: usize, x: *const u64, s: *const u8, out: *mut u8) -> c_int;
    pub fn bppp_wnla_ctx_create(out: *mut *mut BpppCtx, g: *const u8, g_vec: *const u8, ng: usize, h_vec: *const u8, nh: usize, device: c_int, fb_window_bits: c_int) -> c_int;
    pub fn bppp_wnla_ctx_create_budget(out: *mut *mut BpppCtx, g: *const u8, g_vec: *const u8, ng: usize, h_vec: *const u8, nh: usize, device: c_int, fb_window_bits: c_int, fb_table_budget_bytes: u64) -> c_int;
    pub fn bppp_wnla_commit_batch(ctx: *mut BpppCtx, n: usize, c: *const u8, mu: *const u8, l: *const u8, nl: usize, nvec: *const u8, nn: usize, out: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_wnla_verify_batch(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, commitments: *const u8, c: *const u8, rho: *const u8, mu: *const u8, rounds: usize, proof_r: *const u8, proof_x: *const u8, proof_l: *const u8, nl: usize, proof_n: *const u8, nn: usize, accept: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_wnla_verify_batch_device(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, d_commitments: *const c_void, d_c: *const c_void, d_rho: *const c_void, d_mu: *const c_void, rounds: usize, d_proof_r: *const c_void, d_proof_x: *const c_void, d_proof_l: *const c_void, nl: usize, d_proof_n: *const c_void, nn: usize, d_accept: *mut c_void, d_status: *mut c_void) -> c_int;
    pub fn bppp_msm_batch(ctx: *mut BpppCtx, n: usize, nterms: usize, base_index: *const i32, scalars: *const u8, out: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_reciprocal_prove_batch(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, dim_nd: usize, dim_np: usize, commitments: *const u8, x: *const u8, s: *const u8, digits: *const u8, m: *const u8, rnd: *const u8, proofs: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_circuit_create(ctx: *mut BpppCtx, out: *mut *mut BpppCircuit, dims: *const usize, f_l: c_int, f_m: c_int, W_m: *const u8, W_l: *const u8, a_m: *const u8, a_l: *const u8, part_lo: *const i32, part_ll: *const i32, part_lr: *const i32, part_no: *const i32) -> c_int;
    pub fn bppp_circuit_destroy(circuit: *mut BpppCircuit);
    pub fn bppp_circuit_verify_batch(ctx: *mut BpppCtx, circuit: *const BpppCircuit, label: *const u8, label_len: usize, n: usize, commitments: *const u8, proofs: *const u8, rounds: usize, nl: usize, nn: usize, accept: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_circuit_verify_batch_device(ctx: *mut BpppCtx, circuit: *const BpppCircuit, label: *const u8, label_len: usize, n: usize, d_commitments: *const c_void, d_proofs: *const c_void, rounds: usize, nl: usize, nn: usize, d_accept: *mut c_void, d_status: *mut c_void) -> c_int;
    pub fn bppp_circuit_prove_batch(ctx: *mut BpppCtx, circuit: *const BpppCircuit, label: *const u8, label_len: usize, n: usize, v_commitments: *const u8, v: *const u8, s_v: *const u8, w_l: *const u8, w_r: *const u8, w_o: *const u8, rnd: *const u8, proofs: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_wnla_proof_shape(nl: usize, nn: usize, rounds: *mut usize, nl_out: *mut usize, nn_out: *mut usize);
    pub fn bppp_wnla_prove_batch(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, commitments: *const u8, c: *const u8, rho: *const u8, mu: *const u8, l: *const u8, nl: usize, n_vec: *const u8, nn: usize, proof_r: *mut u8, proof_x: *mut u8, proof_l: *mut u8, proof_n: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_reciprocal_verify_batch(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, dim_nd: usize, dim_np: usize, commitments: *const u8, proofs: *const u8, rounds: usize, nl: usize, nn: usize, accept: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_derive_generators(seed: *const u8, seed_len: usize, first_index: usize, n: usize, out: *mut u8) -> c_int;
    pub fn bppp_ctx_save_tables(ctx: *mut BpppCtx, path: *const c_char) -> c_int;
    pub fn bppp_ctx_create_from_tables(out: *mut *mut BpppCtx, path: *const c_char, device: c_int) -> c_int;
    pub fn bppp_ctx_create_shared(out: *mut *mut BpppCtx, parent: *mut BpppCtx) -> c_int;
    pub fn bppp_u64_verify_batch_transcript(ctx: *mut BpppCtx, n: usize, states: *const u8, n_states: usize, commitments: *const u8, proofs: *const u8, accept: *mut u8, status: *mut i32, states_out: *mut u8) -> c_int;
    pub fn bppp_u64_verify_batch_transcript_device(ctx: *mut BpppCtx, n: usize, d_states: *const c_void, n_states: usize, d_commitments: *const c_void, d_proofs: *const c_void, d_accept: *mut c_void, d_status: *mut c_void, d_reject_count: *mut c_void, d_states_out: *mut c_void) -> c_int;
    pub fn bppp_u64_prove_batch_transcript(ctx: *mut BpppCtx, n: usize, states: *const u8, n_states: usize, x: *const u64, s: *const u8, rnd: *const u8, proofs: *mut u8, commitments: *mut u8, status: *mut i32, states_out: *mut u8) -> c_int;
    pub fn bppp_u64_prove_batch_transcript_device(ctx: *mut BpppCtx, n: usize, d_states: *const c_void, n_states: usize, d_x: *const c_void, d_s: *const c_void, d_rnd: *const c_void, d_proofs: *mut c_void, d_commitments: *mut c_void, d_status: *mut c_void, d_states_out: *mut c_void) -> c_int;
    pub fn bppp_wnla_verify_batch_transcript(ctx: *mut BpppCtx, n: usize, states: *const u8, n_states: usize, commitments: *const u8, c: *const u8, rho: *const u8, mu: *const u8, rounds: usize, proof_r: *const u8, proof_x: *const u8, proof_l: *const u8, nl: usize, proof_n: *const u8, nn: usize, accept: *mut u8, status: *mut i32, states_out: *mut u8) -> c_int;
    pub fn bppp_reciprocal_verify_batch_transcript(ctx: *mut BpppCtx, n: usize, states: *const u8, n_states: usize, dim_nd: usize, dim_np: usize, commitments: *const u8, proofs: *const u8, rounds: usize, nl: usize, nn: usize, accept: *mut u8, status: *mut i32, states_out: *mut u8) -> c_int;
    pub fn bppp_circuit_verify_batch_transcript(ctx: *mut BpppCtx, circuit: *const BpppCircuit, n: usize, states: *const u8, n_states: usize, commitments: *const u8, proofs: *const u8, rounds: usize, nl: usize, nn: usize, accept: *mut u8, status: *mut i32, states_out: *mut u8) -> c_int;
    pub fn bppp_wnla_prove_batch_transcript(ctx: *mut BpppCtx, n: usize, states: *const u8, n_states: usize, commitments: *const u8, c: *const u8, rho: *const u8, mu: *const u8, l: *const u8, nl: usize, nvec: *const u8, nn: usize, proof_r: *mut u8, proof_x: *mut u8, proof_l: *mut u8, proof_n: *mut u8, status: *mut i32, states_out: *mut u8) -> c_int;
    pub fn bppp_reciprocal_prove_batch_transcript(ctx: *mut BpppCtx, n: usize, states: *const u8, n_states: usize, dim_nd: usize, dim_np: usize, commitments: *const u8, x: *const u8, s: *const u8, digits: *const u8, m: *const u8, rnd: *const u8, proofs: *mut u8, status: *mut i32, states_out: *mut u8) -> c_int;
    pub fn bppp_circuit_prove_batch_transcript(ctx: *mut BpppCtx, circuit: *const BpppCircuit, n: usize, states: *const u8, n_states: usize, v_commitments: *const u8, v: *const u8, s_v: *const u8, w_l: *const u8, w_r: *const u8, w_o: *const u8, rnd: *const u8, proofs: *mut u8, status: *mut i32, states_out: *mut u8) -> c_int;
    pub fn bppp_transcript_new(label: *const u8, label_len: usize, state_out: *mut u8) -> c_int;
    pub fn bppp_transcript_append_message(state: *mut u8, label: *const u8, label_len: usize, msg: *const u8, msg_len: usize) -> c_int;
    pub fn bppp_transcript_challenge_bytes(state: *mut u8, label: *const u8, label_len: usize, out: *mut u8, n: usize) -> c_int;
    pub fn bppp_shard_range(n_total: usize, rank: c_int, world: c_int, lo: *mut usize, hi: *mut usize);
    pub fn bppp_group_create(out: *mut *mut BpppGroup, g: *const u8, g_vec: *const u8, h_vec: *const u8, devices: *const c_int, n_devices: c_int, fb_window_bits: c_int) -> c_int;
    pub fn bppp_wnla_group_create(out: *mut *mut BpppGroup, g: *const u8, g_vec: *const u8, ng: usize, h_vec: *const u8, nh: usize, devices: *const c_int, n_devices: c_int, fb_window_bits: c_int) -> c_int;
    pub fn bppp_group_destroy(grp: *mut BpppGroup);
    pub fn bppp_group_size(grp: *const BpppGroup) -> c_int;
    pub fn bppp_group_ctx(grp: *mut BpppGroup, rank: c_int) -> *mut BpppCtx;
    pub fn bppp_group_set_option(grp: *mut BpppGroup, name: *const c_char, value: c_long) -> c_int;
    pub fn bppp_u64_verify_batch_sharded(grp: *mut BpppGroup, label: *const u8, label_len: usize, n: usize, commitments: *const u8, proofs: *const u8, accept: *mut u8, status: *mut i32, reject_count: *mut i32) -> c_int;
    pub fn bppp_u64_verify_batch_sharded_device(grp: *mut BpppGroup, label: *const u8, label_len: usize, n: usize, d_commitments: *const *const c_void, d_proofs: *const *const c_void, d_accept: *const *mut c_void, d_status: *const *mut c_void, d_reject_count: *const *mut c_void) -> c_int;
    pub fn bppp_u64_verify_batch_rlc_sharded(grp: *mut BpppGroup, label: *const u8, label_len: usize, n: usize, commitments: *const u8, proofs: *const u8, accept: *mut u8, status: *mut i32, reject_count: *mut i32, seed: *const u8) -> c_int;
    pub fn bppp_u64_verify_batch_rlc_sharded_device(grp: *mut BpppGroup, label: *const u8, label_len: usize, n: usize, d_commitments: *const *const c_void, d_proofs: *const *const c_void, d_accept: *const *mut c_void, d_status: *const *mut c_void, d_reject_count: *const *mut c_void, seed: *const u8) -> c_int;
    pub fn bppp_u64_verify_batch_sec1_sharded(grp: *mut BpppGroup, label: *const u8, label_len: usize, n: usize, commitments33: *const u8, proofs525: *const u8, accept: *mut u8, status: *mut i32, reject_count: *mut i32) -> c_int;
    pub fn bppp_u64_verify_batch_sec1_sharded_device(grp: *mut BpppGroup, label: *const u8, label_len: usize, n: usize, d_commitments33: *const *const c_void, d_proofs525: *const *const c_void, d_accept: *const *mut c_void, d_status: *const *mut c_void, d_reject_count: *const *mut c_void) -> c_int;
    pub fn bppp_u64_verify_batch_transcript_sharded(grp: *mut BpppGroup, n: usize, states: *const u8, n_states: usize, commitments: *const u8, proofs: *const u8, accept: *mut u8, status: *mut i32, states_out: *mut u8, reject_count: *mut i32) -> c_int;
    pub fn bppp_u64_verify_batch_transcript_sharded_device(grp: *mut BpppGroup, n: usize, d_states: *const *const c_void, n_states: usize, d_commitments: *const *const c_void, d_proofs: *const *const c_void, d_accept: *const *mut c_void, d_status: *const *mut c_void, d_reject_count: *const *mut c_void, d_states_out: *const *mut c_void) -> c_int;
    pub fn bppp_u64_prove_batch_sharded(grp: *mut BpppGroup, label: *const u8, label_len: usize, n: usize, x: *const u64, s: *const u8, rnd: *const u8, proofs: *mut u8, commitments: *mut u8, status: *mut i32) -> c_int;
    pub fn bppp_u64_prove_batch_sharded_device(grp: *mut BpppGroup, label: *const u8, label_len: usize, n: usize, d_x: *const *const c_void, d_s: *const *const c_void, d_rnd: *const *const c_void, d_proofs: *const *mut c_void, d_commitments: *const *mut c_void, d_status: *const *mut c_void) -> c_int;
    pub fn bppp_reciprocal_verify_batch_sharded(grp: *mut BpppGroup, label: *const u8, label_len: usize, n: usize, dim_nd: usize, dim_np: usize, commitments: *const u8, proofs: *const u8, rounds: usize, nl: usize, nn: usize, accept: *mut u8, status: *mut i32, reject_count: *mut i32) -> c_int;
    pub fn bppp_reciprocal_verify_batch_rlc_sharded(grp: *mut BpppGroup, label: *const u8, label_len: usize, n: usize, dim_nd: usize, dim_np: usize, commitments: *const u8, proofs: *const u8, rounds: usize, nl: usize, nn: usize, accept: *mut u8, status: *mut i32, reject_count: *mut i32, seed: *const u8) -> c_int;
    pub fn bppp_reciprocal_verify_batch_sharded_device(grp: *mut BpppGroup, label: *const u8, label_len: usize, n: usize, dim_nd: usize, dim_np: usize, d_commitments: *const *const c_void, d_proofs: *const *const c_void, rounds: usize, nl: usize, nn: usize, d_accept: *const *mut c_void, d_status: *const *mut c_void, d_reject_count: *const *mut c_void) -> c_int;
    pub fn bppp_reciprocal_verify_batch_rlc_sharded_device(grp: *mut BpppGroup, label: *const u8, label_len: usize, n: usize, dim_nd: usize, dim_np: usize, d_commitments: *const *const c_void, d_proofs: *const *const c_void, rounds: usize, nl: usize, nn: usize, d_accept: *const *mut c_void, d_status: *const *mut c_void, d_reject_count: *const *mut c_void, seed: *const u8) -> c_int;
    pub fn bppp_reciprocal_verify_batch_device(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, dim_nd: usize, dim_np: usize, d_commitments: *const c_void, d_proofs: *const c_void, rounds: usize, nl: usize, nn: usize, d_accept: *mut c_void, d_status: *mut c_void) -> c_int;
    pub fn bppp_reciprocal_verify_batch_rlc(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, dim_nd: usize, dim_np: usize, commitments: *const u8, proofs: *const u8, rounds: usize, nl: usize, nn: usize, accept: *mut u8, status: *mut i32, seed: *const u8) -> c_int;
    pub fn bppp_reciprocal_verify_batch_rlc_device(ctx: *mut BpppCtx, label: *const u8, label_len: usize, n: usize, dim_nd: usize, dim_np: usize, d_commitments: *const c_void, d_proofs: *const c_void, rounds: usize, nl: usize, nn: usize, d_accept: *mut c_void, d_status: *mut c_void, seed: *const u8) -> c_int;
    pub fn bppp_ctx_enable_timing(ctx: *mut BpppCtx, enable: c_int) -> c_int;
    pub fn bppp_ctx_get_timings(ctx: *mut BpppCtx, max_entries: c_int, names: *mut *const c_char, total_ms: *mut f64, launches: *mut i64, reset: c_int) -> c_int;
    pub fn bppp_ctx_device_bytes(ctx: *const BpppCtx) -> usize;
    pub fn bppp_strerror(code: c_int) -> *const c_char;
    pub fn bppp_last_error() -> *const c_char;
}
