// u64 batch prover: the lane kernels' 256-register builds (two wavefronts per SIMD) and the four-lanes-per-value forms.
// Part of libbppp_hip.so; split from k_prove.hip so that the two halves compile in parallel (this unit and its sibling were the longest
// pole of the build).  Per-lane work lives in prove_core.h, declarations in kernels.h.
#include "kernels.h"

using namespace bppp;

// ---- the lane kernels above at two wavefronts per SIMD (256 VGPR + AGPR), for prove batches that give every SIMD more than one
// wavefront (beyond 2^16 values; BASELINE configs[3]'s 2^14 values are 256 workgroups on 1024 SIMDs and keep the uncapped builds):
// uncapped they allocate 332-398 registers, i.e. ONE wavefront per SIMD, which is the cliff the u64 verifier's lane kernels fell off
// in round 1 (kernels.h, BPPP_LANE_MIN_WAVES).  Same per-lane code.
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_stage_b_w2(ProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    const u32 key = preloaded_position_key(w.states, w.n_states, t);
    for_each_position_group(key, [&]() { prove_stage_b(w, t); });
}
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_stage_d_w2(ProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    for_each_position_group(prove_position_key(w, t), [&]() { prove_stage_d(w, t); });
}
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_stage_f_w2(ProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    for_each_position_group(prove_position_key(w, t), [&]() { prove_stage_f(w, t); });
}
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_round_fold_w2(ProveWs w, int k) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    for_each_position_group(prove_position_key(w, t), [&]() { prove_round_fold(w, t, k); });
}
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_round_next_w2(ProveWs w, int k) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) prove_round_next(w, t, k);
}
// ... and the sixteen-lane forms at two wavefronts per SIMD, for calls whose groups give every SIMD more than one wavefront
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_stage_d_g16_w2(ProveWs w) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 4;
    if (t >= w.N) return;
    const int lane = (int)(g & 15);
    for_each_position_group(prove_position_key(w, t), [&]() { prove_stage_d(w, t, lane); });
}
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_stage_f_g16_w2(ProveWs w) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 4;
    if (t >= w.N) return;
    const int lane = (int)(g & 15);
    for_each_position_group(prove_position_key(w, t), [&]() { prove_stage_f(w, t, lane); });
}
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_round_fold_g16_w2(ProveWs w, int k) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 4;
    if (t >= w.N) return;
    const int lane = (int)(g & 15);
    for_each_position_group(prove_position_key(w, t), [&]() { prove_round_fold_lanes(w, t, k, lane); });
}
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_prove_round_next_g4_w2(ProveWs w, int k) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 2;
    if (t < w.N) prove_round_next(w, t, k, (int)(g & 3));
}
// batches of a few values per SIMD: FOUR lanes per value, a lane per run of four terms (prove_core.h: "lane forms", group = 4); MINW = 2
// is the 256-register build for launches that give every SIMD more than one wavefront
template <int MINW>
__global__ __launch_bounds__(BPPP_BLOCK, MINW) void k_prove_stage_d_g4(ProveWs w) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 2;
    if (t >= w.N) return;
    const int lane = (int)(g & 3);
    for_each_position_group(prove_position_key(w, t), [&]() { prove_stage_d(w, t, lane, 4); });
}
template <int MINW>
__global__ __launch_bounds__(BPPP_BLOCK, MINW) void k_prove_stage_f_g4(ProveWs w) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 2;
    if (t >= w.N) return;
    const int lane = (int)(g & 3);
    for_each_position_group(prove_position_key(w, t), [&]() { prove_stage_f(w, t, lane, 4); });
}
template <int MINW>
__global__ __launch_bounds__(BPPP_BLOCK, MINW) void k_prove_round_fold_g4(ProveWs w, int k) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 2;
    if (t >= w.N) return;
    const int lane = (int)(g & 3);
    for_each_position_group(prove_position_key(w, t), [&]() { prove_round_fold_lanes4(w, t, k, lane); });
}
template __global__ void k_prove_round_fold_g4<1>(ProveWs w, int k);
template __global__ void k_prove_round_fold_g4<2>(ProveWs w, int k);
template __global__ void k_prove_stage_d_g4<1>(ProveWs w);
template __global__ void k_prove_stage_d_g4<2>(ProveWs w);
template __global__ void k_prove_stage_f_g4<1>(ProveWs w);
template __global__ void k_prove_stage_f_g4<2>(ProveWs w);
