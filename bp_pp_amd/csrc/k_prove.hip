// u64 batch prover kernels (prove_core.h).
// Part of libbppp_hip.so; per-lane work lives in the *_core.h headers, declarations in kernels.h.
#include "kernels.h"

using namespace bppp;

// ---- prover kernels (prove_core.h)
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_stage_a(ProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) prove_stage_a(w, t);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_stage_b(ProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    const u32 key = preloaded_position_key(w.states, w.n_states, t);
    for_each_position_group(key, [&]() { prove_stage_b(w, t); });
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_stage_d(ProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    for_each_position_group(prove_position_key(w, t), [&]() { prove_stage_d(w, t); });
}
// small calls: sixteen lanes per value, a lane per term of the stage's 16-term loop (prove_core.h: "lane forms")
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_stage_d_g16(ProveWs w) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 4;
    if (t >= w.N) return;
    const int lane = (int)(g & 15);
    for_each_position_group(prove_position_key(w, t), [&]() { prove_stage_d(w, t, lane); });
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_stage_f_g16(ProveWs w) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 4;
    if (t >= w.N) return;
    const int lane = (int)(g & 15);
    for_each_position_group(prove_position_key(w, t), [&]() { prove_stage_f(w, t, lane); });
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_stage_f(ProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    for_each_position_group(prove_position_key(w, t), [&]() { prove_stage_f(w, t); });
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_export_states(ProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) prove_export_state(w, t);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_round_scalars(ProveWs w, int k) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) prove_round_scalars(w, t, k);
}
// small calls: a wavefront per proof, a lane per generator (the loop over the 49 terms is a chain of dependent loads on one lane)
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_round_scalars_wide(ProveWs w, int k) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 6;
    const int lane = (int)(g & 63);
    if (t >= w.N) return;
    if (lane < 32) prove_round_scalars_h(w, t, k, lane);
    else if (lane < 48) prove_round_scalars_g(w, t, k, lane - 32);
    else if (lane == 48) prove_round_scalars_v(w, t, k);
}
// batches that leave the chip under-filled: the same scalars from FOUR workgroups per 64 values (blockIdx.y: the two leading scalars, h
// 0..15, h 16..31, g) -- the pieces are independent, every piece keeps the one-lane form's coalesced loads, and the chain one wavefront
// walks drops from 200 / 156 / 134 / 123 multiplications (rounds 1..4) to 88 / 48 / 48 / 48
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_round_scalars_parts(ProveWs w, int k) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    const int part = (int)blockIdx.y;
    if (part == 0) prove_round_scalars_v(w, t, k);
    else if (part == 3) {
#pragma nounroll
        for (int i = 0; i < 16; i++) prove_round_scalars_g(w, t, k, i);
    } else {
#pragma nounroll
        for (int i = 16 * (part - 1); i < 16 * part; i++) prove_round_scalars_h(w, t, k, i);
    }
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_round_fold(ProveWs w, int k) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    for_each_position_group(prove_position_key(w, t), [&]() { prove_round_fold(w, t, k); });
}
// small calls: sixteen lanes per value for part one of a round (prove_core.h: prove_round_fold_lanes; the next commitment is a fixed-base
// sum there).  Groups are whole or absent: a group's sixteen lanes share t, so the bounds test is uniform over the group.
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_round_fold_g16(ProveWs w, int k) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 4;
    if (t >= w.N) return;
    const int lane = (int)(g & 15);
    for_each_position_group(prove_position_key(w, t), [&]() { prove_round_fold_lanes(w, t, k, lane); });
}
// part two of a round (prove_core.h: prove_round_next): the next commitment by the variable-base path, no transcript in it
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_round_next(ProveWs w, int k) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) prove_round_next(w, t, k);
}
// small batches: four lanes per proof, the sum split into its four GLV streams (straus_core.h: straus_affine_g4)
__global__ __launch_bounds__(BPPP_BLOCK) void k_prove_round_next_g4(ProveWs w, int k) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 2;
    if (t < w.N) prove_round_next(w, t, k, (int)(g & 3));
}
// NL lanes per proof: 8 while the batch is small, 1 from the size at which one lane per proof fills the SIMDs twice over (as in the
// verifier's fixed-base kernels, k_verify_fixed.hip): no idle lanes in the short runs, no 3-step tree of complete additions per sum
template <int NL>
__device__ __forceinline__ void prove_msm_lanes(const ProveWs& w, const MsmJob& job) {
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / NL;
    int lane = (int)(g % NL);
    if (t >= w.N) return;
    pt part;
    FbRanges rg;
    prove_msm_ranges(rg, job);
    fb_group_sum<NL>(part, w.fb, t, lane, w.msc, rg);
    if (lane == 0) prove_msm_store(w, job, t, part);
}
// the independent sums of one stage in ONE launch (small calls: r_com | c_o | c_l | c_r, and X | R of a round)
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_prove_msm_l64x(ProveWs w, MsmJobs jobs) { prove_msm_lanes<64>(w, jobs.j[blockIdx.y]); }
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_prove_msm_x(ProveWs w, MsmJobs jobs) { prove_msm_lanes<BPPP_FB_LANES>(w, jobs.j[blockIdx.y]); }
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_prove_msm_l1x(ProveWs w, MsmJobs jobs) { prove_msm_lanes<1>(w, jobs.j[blockIdx.y]); }
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_prove_msm_l4x(ProveWs w, MsmJobs jobs) { prove_msm_lanes<4>(w, jobs.j[blockIdx.y]); }

// "ct_prover": the stage's sums over secret scalars, 8 lanes per proof, every window's entries read and masked (fb_lookup_add_ct)
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_prove_msm_ct(ProveWs w, MsmJobs jobs) {
    const MsmJob& job = jobs.j[blockIdx.y];
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / BPPP_FB_LANES;
    int lane = (int)(g % BPPP_FB_LANES);
    if (t >= w.N) return;
    pt part;
    FbRanges rg;
    prove_msm_ranges(rg, job);
    fb_group_sum_ct<BPPP_FB_LANES>(part, w.fb_ct, t, lane, w.msc, rg);
    if (lane == 0) prove_msm_store(w, job, t, part);
}
