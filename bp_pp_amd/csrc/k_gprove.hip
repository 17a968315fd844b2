// generic WNLA / arithmetic circuit / reciprocal prover kernels.
// Part of libbppp_hip.so; per-lane work lives in the *_core.h headers, declarations in kernels.h.
#include "kernels.h"

using namespace bppp;

// ---- generic WNLA prover kernels (wnla_prove_core.h)
// Pre-loaded per-instance transcripts may sit at different sponge positions; the transcript code branches on the (wave-uniform)
// position, so the hashing kernels run once per distinct position present in the wavefront (kernels.h).  All instances absorb the
// same byte counts, so instances that start at one position stay together: the key is the position the caller handed in.
template <typename Ws>
__device__ __forceinline__ u32 gprove_position_key(const Ws& w, size_t t) {
    return w.divergent_positions ? preloaded_position_key(w.tio.states, w.tio.n_states, t) : 0u;
}
// the caller's `&mut Transcript` after a prove (any generic prover: they all end in the WNLA prover's state array)
__global__ __launch_bounds__(BPPP_BLOCK) void k_gprove_export_states(TranscriptIo io, strobe base, const u32* tstate, size_t N, const int32_t* status) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < N) tio_export(io, base, tstate, N, status, t);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_wprove_init(WnlaProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) wnla_prove_init(w, t);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_wprove_round_scalars(WnlaProveWs w, int k) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) wnla_prove_round_scalars(w, t, k);
}
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_wprove_msm(WnlaProveWs w, int set, int oddsh) {
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / BPPP_FB_LANES;
    int lane = (int)(g % BPPP_FB_LANES);
    if (t >= w.N) return;
    pt part;
    FbRanges rg;
    wnla_prove_msm_ranges(rg, w, oddsh);
    // (ct: wave-uniform -- the secret-scalar form reads every entry of every window and selects by mask)
    if (w.ct) fb_group_sum_ct<BPPP_FB_LANES>(part, w.fb_ct, t, lane, w.msc + (size_t)set * wp_set_words(w), rg);
    else fb_group_sum(part, w.fb, t, lane, w.msc + (size_t)set * wp_set_words(w), rg);
    if (lane == 0) ws_st_pt(w.pbuf + (size_t)set * 30 * w.N, w.N, t, part);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_wprove_round_fold(WnlaProveWs w, int k) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    for_each_position_group(gprove_position_key(w, t), [&]() { wnla_prove_round_fold(w, t, k); });
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_wprove_finish(WnlaProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) wnla_prove_finish(w, t);
}
// ---- generic circuit prover kernels (circuit_prove_core.h)
__global__ __launch_bounds__(BPPP_BLOCK) void k_cprove_stage_a(CircuitProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) circuit_prove_stage_a(w, t);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_cprove_stage_b(CircuitProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    for_each_position_group(gprove_position_key(w, t), [&]() { circuit_prove_stage_b(w, t); });
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_cprove_stage_c(CircuitProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    for_each_position_group(gprove_position_key(w, t), [&]() { circuit_prove_stage_c(w, t); });
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_cprove_stage_d(CircuitProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) circuit_prove_stage_d(w, t);
}
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_cprove_msm(CircuitProveWs w, int set, int with_g) {
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / BPPP_FB_LANES;
    int lane = (int)(g % BPPP_FB_LANES);
    if (t >= w.N) return;
    pt part;
    FbRanges rg;
    cp_ranges(rg, w, with_g != 0);
    if (w.ct) fb_group_sum_ct<BPPP_FB_LANES>(part, w.fb_ct, t, lane, w.msc + (size_t)set * cp_set_words(w), rg);
    else fb_group_sum(part, w.fb, t, lane, w.msc + (size_t)set * cp_set_words(w), rg);
    if (lane == 0) ws_st_pt(w.pbuf + (size_t)set * 30 * w.N, w.N, t, part);
}
// ---- generic reciprocal prover kernels (recip_prove_core.h)
__global__ __launch_bounds__(BPPP_BLOCK) void k_rprove_stage_r1(RecipProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    for_each_position_group(gprove_position_key(w, t), [&]() { recip_prove_stage_r1(w, t); });
}
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_rprove_msm(RecipProveWs w) {
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / BPPP_FB_LANES;
    int lane = (int)(g % BPPP_FB_LANES);
    if (t >= w.N) return;
    pt part;
    FbRanges rg;
    recip_prove_ranges(rg, w);
    if (w.ct) fb_group_sum_ct<BPPP_FB_LANES>(part, w.fb_ct, t, lane, w.msc, rg);
    else fb_group_sum(part, w.fb, t, lane, w.msc, rg);
    if (lane == 0) ws_st_pt(w.pbuf, w.N, t, part);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_rprove_stage_r2(RecipProveWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) recip_prove_stage_r2(w, t);
}
