// Kernel declarations shared by the kernel translation units (k_*.hip) and the host side (bppp_*.hip).
// The kernels are split over several translation units so that hipcc compiles them in parallel (the u64 verifier's
// shared-doubling kernels alone take minutes); a kernel is launched from the host TU through its declaration here.
#pragma once
#include <hip/hip_runtime.h>

#include "circuit_prove_core.h"
#include "prove_core.h"
#include "recip_core.h"
#include "recip_prove_core.h"
#include "bucket_core.h"
#include "rlc_core.h"
#include "wnla_rlc_core.h"
#include "wnla_prove_core.h"

#define BPPP_BLOCK 64
// One-lane-per-proof kernels of the u64 verifier: minimum waves per SIMD the register allocator must leave room for
// (2 => at most 256 VGPR + AGPR per lane, so two wavefronts share a SIMD and cover each other's table-gather latency).
// Measured on MI355X at 2^20 proofs (profiles/r02/r02_occupancy_ab.txt): 1 wave/SIMD (374 / 272 / 312 VGPRs) 182.8 ms per batch, 2 waves
// (256, the round kernel spilling 109 VGPRs into transcript-only code) 164.6 ms, 3 waves (168, spills inside the hot loops) 173.4 ms.
#ifndef BPPP_LANE_MIN_WAVES
#define BPPP_LANE_MIN_WAVES 2
#endif
// fixed-base MSM kernels: BPPP_FB_LANES lanes per proof, 256-thread workgroups; 2 waves/SIMD (212 VGPRs, no spills) -- forcing 3
// (168 VGPRs, 52 spilled) slows k_verify_final_check from 42.8 to 61.6 ms per 2^20 proofs (same measurement)
// k_verify_tables: chains of dependent loads and short arithmetic (five passes over a proof's 13 points).  Round 2 ran it at 3 waves per
// SIMD (168 VGPRs) with one running product per denominator; with one per block of four (straus_core.h: aff_push_block) the unwinding
// passes hold four denominators and their inverses at once: 229 spilled VGPRs at 168, 7 at 256.  Measured on 2^20 proofs
// (profiles/r03/r03_c_*): 3 waves 17.2-17.4 ms, 2 waves 14.5-14.6 ms (round 2's form: 14.6-14.75 ms at 3 waves).
#ifndef BPPP_TABLES_MIN_WAVES
#define BPPP_TABLES_MIN_WAVES 2
#endif
#define BPPP_FB_BLOCK 256
#ifndef BPPP_FB_MIN_WAVES
#define BPPP_FB_MIN_WAVES 2
#endif

namespace bppp {
// The transcript code keeps the sponge's byte position in a scalar register (merlin.h: st_uniform), which is right whenever
// every lane of a wavefront runs the same transcript schedule from the same position -- always, except when the caller passed
// PER-PROOF pre-loaded transcripts whose lengths differ (bppp_u64_verify_batch_transcript with n_states = n).  The kernels that
// touch the transcript therefore run their body once per distinct position present in the wavefront (one trip in every other
// case): lanes at the first active lane's position execute, the rest wait their turn.
template <typename F>
__device__ __forceinline__ void for_each_position_group(u32 key, F&& body) {
#if defined(__HIP_DEVICE_COMPILE__)
    // the leader is taken from a ballot of the lanes still pending, so the choice depends on the loop state: a bare
    // readfirstlane(key) is loop-invariant to the optimiser, which then (forward-progress rule) drops the loop altogether
    bool pending = true;
    while (true) {
        const unsigned long long todo = __ballot(pending);
        if (!todo) break;
        const int leader = __ffsll((long long)todo) - 1;
        const u32 k = (u32)__builtin_amdgcn_readlane((int)key, leader);
        if (pending && key == k) {
            body();
            pending = false;
        }
    }
#else
    (void)key;
    body();
#endif
}
// ---- generic fixed-base linear combination over the context's generators (the crate's commit functions)
struct MsmWs {
    size_t N;
    int nterms, nruns;
    const uint8_t* scalars;     // N x nterms x 32
    const int* runs;            // [nruns][3]: first scalar slot, first base, count
    u32* msc;                   // [nterms * 8][N]
    u32* pfix;                  // [30][N]
    int32_t* status;
    uint8_t* out;               // N x 64
    FbTable fb;
};
}  // namespace bppp
using bppp::MsmWs;

__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_verify_phase1(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_phase1_g16(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_c0_fixed(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_c0_fixed_l1(bppp::VerifyWs ws);      // one lane per proof (full batches)
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_TABLES_MIN_WAVES) void k_verify_tables(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_shared_inv2(const bppp::u32* in, bppp::u32* out, size_t N);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_shared_inv4(const bppp::u32* in, bppp::u32* out, size_t N);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_shared_inv8(const bppp::u32* in, bppp::u32* out, size_t N);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_shared_inv16(const bppp::u32* in, bppp::u32* out, size_t N);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_c0_join(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_TABLES_MIN_WAVES) void k_verify_tables_pass0(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_TABLES_MIN_WAVES) void k_verify_tables_pass1(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_TABLES_MIN_WAVES) void k_verify_tables_pass2(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_TABLES_MIN_WAVES) void k_verify_tables_pass3(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_TABLES_MIN_WAVES) void k_verify_tables_pass4(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_round_head_small(bppp::VerifyWs ws, int k);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_verify_c0_var(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_verify_round(bppp::VerifyWs ws, int k);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_phase1_small(bppp::VerifyWs ws);
#define BPPP_C0VAR_SMALL_BLOCK 256   // k_verify_var.hip: four wavefronts per workgroup, one per SIMD
__global__ __launch_bounds__(BPPP_C0VAR_SMALL_BLOCK) void k_verify_c0_var_small(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_C0VAR_SMALL_BLOCK, 2) void k_verify_round_tail(bppp::VerifyWs ws, int k);
__global__ __launch_bounds__(BPPP_C0VAR_SMALL_BLOCK, 2) void k_verify_tables_own(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_C0VAR_SMALL_BLOCK, 2) void k_verify_phase1_wg4(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_round_small(bppp::VerifyWs ws, int k);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_round_g4(bppp::VerifyWs ws, int k);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_c0_var_g4(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_round_g2(bppp::VerifyWs ws, int k);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_tables_split1(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_tables_split2(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_tables_split4(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_round_g8(bppp::VerifyWs ws, int k);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_round_g16(bppp::VerifyWs ws, int k);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_c0_var_g32(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_c0_var_g64(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_c0_fixed_l64(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_final_check_l64(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_final_scalars(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_final_scalars_g16(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_final_check(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_final_check_l1(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_accept(bppp::VerifyWs ws, int* reject_count);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_export_states(bppp::VerifyWs ws);
__global__ __launch_bounds__(BPPP_BLOCK) void k_rlc_lhs(bppp::VerifyWs ws, bppp::RlcWs r);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_rlc_chunk(bppp::VerifyWs ws, bppp::RlcWs r);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_rlc_chunk_c32(bppp::VerifyWs ws, bppp::RlcWs r);      // chunks of 32 proofs
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_final_check_flagged_l8(bppp::VerifyWs ws, bppp::RlcWs r);
__global__ __launch_bounds__(64) void k_verify_final_check_flagged(bppp::VerifyWs ws, bppp::RlcWs r);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_final_check_flagged_dense(bppp::VerifyWs ws, bppp::RlcWs r);
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_accept_flagged(bppp::VerifyWs ws, bppp::RlcWs r, int* reject_count, int* hist_count);
__global__ __launch_bounds__(BPPP_BLOCK) void k_fb_build_pass1(bppp::FbBuild fb, size_t nthreads);
__global__ __launch_bounds__(BPPP_BLOCK) void k_fb_build_pass2(bppp::FbBuild fb, size_t nthreads);
__global__ void k_decode_generators(const uint8_t* in, bppp::apt* out, int n, int* flags);
__global__ __launch_bounds__(BPPP_BLOCK) void k_commit_value(bppp::VerifyWs ws, const uint64_t* x, const uint8_t* s, uint8_t* out, int* flags, bppp::FbTable ct);
__global__ __launch_bounds__(256) void k_sec1_expand(uint8_t* commitments64, uint8_t* proofs928, const uint8_t* commitments33, const uint8_t* proofs525, size_t n);
__global__ __launch_bounds__(256) void k_sec1_compress(uint8_t* commitments33, uint8_t* proofs525, const uint8_t* commitments64,
                                                       const uint8_t* proofs928, size_t n);
// the u64 prover's stages that touch the transcript run once per distinct sponge position in the wavefront (for_each_position_group
// above; one trip unless the caller passed per-proof pre-loaded transcripts of different lengths) -- shared by k_prove.hip and k_prove_w2.hip
#if defined(__HIPCC__)
__device__ __forceinline__ bppp::u32 prove_position_key(const bppp::ProveWs& w, size_t t) {
    return (w.states && w.n_states != 1) ? w.tstate[(size_t)50 * w.N + t] : 0u;
}
#endif
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_stage_a(bppp::ProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_stage_b(bppp::ProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_stage_d(bppp::ProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_stage_d_g16(bppp::ProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_stage_f_g16(bppp::ProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_round_fold_g16(bppp::ProveWs w, int k);
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_stage_f(bppp::ProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_export_states(bppp::ProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_round_scalars(bppp::ProveWs w, int k);
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_round_scalars_wide(bppp::ProveWs w, int k);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_round_scalars_parts(bppp::ProveWs w, int k);
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_round_fold(bppp::ProveWs w, int k);
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_round_next(bppp::ProveWs w, int k);
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_round_next_g4(bppp::ProveWs w, int k);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_round_next_w2(bppp::ProveWs w, int k);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_stage_b_w2(bppp::ProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_stage_d_w2(bppp::ProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_stage_f_w2(bppp::ProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_round_fold_w2(bppp::ProveWs w, int k);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_stage_d_g16_w2(bppp::ProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_stage_f_g16_w2(bppp::ProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_round_fold_g16_w2(bppp::ProveWs w, int k);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_round_next_g4_w2(bppp::ProveWs w, int k);
template <int MINW> __global__ __launch_bounds__(BPPP_BLOCK, MINW) void k_prove_stage_d_g4(bppp::ProveWs w);
template <int MINW> __global__ __launch_bounds__(BPPP_BLOCK, MINW) void k_prove_stage_f_g4(bppp::ProveWs w);
template <int MINW> __global__ __launch_bounds__(BPPP_BLOCK, MINW) void k_prove_round_fold_g4(bppp::ProveWs w, int k);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_prove_msm_l64x(bppp::ProveWs w, bppp::MsmJobs jobs);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_prove_msm_x(bppp::ProveWs w, bppp::MsmJobs jobs);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_prove_msm_l1x(bppp::ProveWs w, bppp::MsmJobs jobs);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_prove_msm_l4x(bppp::ProveWs w, bppp::MsmJobs jobs);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_prove_msm_ct(bppp::ProveWs w, bppp::MsmJobs jobs);
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_commit_scalars(bppp::WnlaWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_begin(bppp::WnlaWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_round(bppp::WnlaWs w, int k);
__global__ __launch_bounds__(BPPP_BLOCK) void k_generic_export_states(bppp::WnlaWs w);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_TABLES_MIN_WAVES) void k_wnla_tables(bppp::WnlaWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_tables_split(bppp::WnlaWs w, int parts, int lp);
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_round_grp(bppp::WnlaWs w, int k, int group);
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_rlc_lhs(bppp::WnlaWs w, bppp::RlcWs r);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_wnla_rlc_chunk(bppp::WnlaWs w, bppp::RlcWs r);
__global__ __launch_bounds__(64) void k_wnla_rlc_check(bppp::WnlaWs w, bppp::RlcWs r);
__global__ __launch_bounds__(64) void k_wnla_msm_flagged(bppp::WnlaWs w, bppp::RlcWs r);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_wnla_msm_flagged_dense(bppp::WnlaWs w, bppp::RlcWs r);
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_accept_flagged(bppp::WnlaWs w, bppp::RlcWs r);
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_final_scalars(bppp::WnlaWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_final_scalars_grp(bppp::WnlaWs w, int lg);
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_final_scalars_join(bppp::WnlaWs w, int lg);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_wnla_msm(bppp::WnlaWs w, int commit_mode);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_wnla_msm_l1(bppp::WnlaWs w);               // one lane per instance (full batches)
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_recip_c0_fixed_l1(bppp::RecipWs w);
// a wavefront per instance (small calls: bppp_generic.hip: generic_fb_wide)
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_wnla_msm_l64(bppp::WnlaWs w);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_recip_c0_fixed_l64(bppp::RecipWs w);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_circuit_c0_fixed_l64(bppp::CircuitWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_circuit_c0_var_pts(bppp::CircuitWs w, int L);                 // a lane per C0 point (small calls)
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_commit_store(bppp::WnlaWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_accept(bppp::WnlaWs w);
__global__ __launch_bounds__(256) void k_count_rejects(const uint8_t* accept, size_t n, int* reject_count);
__global__ __launch_bounds__(BPPP_BLOCK) void k_wprove_init(bppp::WnlaProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_wprove_round_scalars(bppp::WnlaProveWs w, int k);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_wprove_msm(bppp::WnlaProveWs w, int set, int oddsh);
__global__ __launch_bounds__(BPPP_BLOCK) void k_wprove_round_fold(bppp::WnlaProveWs w, int k);
__global__ __launch_bounds__(BPPP_BLOCK) void k_gprove_export_states(bppp::TranscriptIo io, bppp::strobe base, const bppp::u32* tstate, size_t N, const int32_t* status);
__global__ __launch_bounds__(BPPP_BLOCK) void k_wprove_finish(bppp::WnlaProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_cprove_stage_a(bppp::CircuitProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_cprove_stage_b(bppp::CircuitProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_cprove_stage_c(bppp::CircuitProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_cprove_stage_d(bppp::CircuitProveWs w);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_cprove_msm(bppp::CircuitProveWs w, int set, int with_g);
__global__ __launch_bounds__(BPPP_BLOCK) void k_rprove_stage_r1(bppp::RecipProveWs w);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_rprove_msm(bppp::RecipProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_rprove_stage_r2(bppp::RecipProveWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_msm_scalars(bppp::MsmWs w);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_msm(bppp::MsmWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_msm_store(bppp::MsmWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_circuit_phase1(bppp::CircuitWs w);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_circuit_c0_fixed(bppp::CircuitWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_circuit_c0_tables(bppp::CircuitWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_circuit_c0_var(bppp::CircuitWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_circuit_c0_finish(bppp::CircuitWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_recip_phase1(bppp::RecipWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_recip_phase1_grp(bppp::RecipWs w, int G);
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_recip_c0_fixed(bppp::RecipWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_recip_c0_var(bppp::RecipWs w);
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_TABLES_MIN_WAVES) void k_recip_c0_tables(bppp::RecipWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_recip_c0_var_grp(bppp::RecipWs w, int group);
__global__ __launch_bounds__(BPPP_BLOCK) void k_recip_c0_finish(bppp::RecipWs w);
__global__ __launch_bounds__(BPPP_BLOCK) void k_bkt_prepare(bppp::BucketWs w);
__global__ __launch_bounds__(256) void k_bkt_accumulate(bppp::BucketWs w);
#define BPPP_BKT_SCALAR_GROUP 16
__global__ __launch_bounds__(256) void k_bkt_scalars(bppp::BucketWs w);
__global__ __launch_bounds__(64) void k_bkt_check(bppp::BucketWs w);
