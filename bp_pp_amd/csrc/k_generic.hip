// generic WNLA / arithmetic circuit / reciprocal verifier kernels and the generic MSM.
// Part of libbppp_hip.so; per-lane work lives in the *_core.h headers, declarations in kernels.h.
#include "kernels.h"

using namespace bppp;

// ---- generic WNLA kernels (wnla_core.h)
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_commit_scalars(WnlaWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) wnla_commit_scalars(w, t);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_begin(WnlaWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) wnla_verify_begin(w, t);
}
// kernels that run transcript operations go through for_each_position_group (kernels.h) when per-instance pre-loaded transcripts
// may sit at different sponge positions; one trip otherwise
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_round(WnlaWs w, int k) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    const u32 key = w.divergent_positions ? w.tstate[(size_t)50 * w.N + t] : 0u;
    for_each_position_group(key, [&]() { wnla_verify_round(w, t, k); });
}
// per-instance advanced transcripts back to the caller (any of the generic verifiers: they all end in the WNLA stage)
__global__ __launch_bounds__(BPPP_BLOCK) void k_generic_export_states(WnlaWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) tio_export(w.tio, w.base, w.tstate, w.N, w.status, t);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_final_scalars(WnlaWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) wnla_verify_final_scalars(w, t);
}
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_wnla_msm(WnlaWs w, int commit_mode) {
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / BPPP_FB_LANES;
    int lane = (int)(g % BPPP_FB_LANES);
    if (t >= w.N) return;
    pt part;
    FbRanges rg;
    wnla_msm_ranges(rg, w);
    fb_group_sum(part, w.fb, t, lane, w.msc, rg);
    if (lane == 0) wnla_verify_store(w, t, part);
    (void)commit_mode;
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_commit_store(WnlaWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) {
        pt total;
        ws_ld_pt(total, w.pfix, w.N, t);
        wnla_commit_store(w, t, total);
    }
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_accept(WnlaWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) wnla_verify_accept(w, t);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_msm_scalars(MsmWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    bool ok = true;
    sc zero;
    sc_set_u32(zero, 0);
#pragma nounroll
    for (int j = 0; j < w.nterms; j++) {
        sc k;
        const bool kok = sc_from_be(k, w.scalars + ((size_t)t * w.nterms + j) * 32);
        ok &= kok;
        ws_st8(w.msc, w.N, t, j, kok ? k.v : zero.v);
    }
    w.status[t] = ok ? ST_OK : ST_BAD_ENCODING;
}
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_msm(MsmWs w) {
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / BPPP_FB_LANES;
    int lane = (int)(g % BPPP_FB_LANES);
    if (t >= w.N) return;
    pt total;
    pt_set_identity(total);
#pragma nounroll
    for (int r = 0; r < w.nruns; r += 3) {      // up to three runs per pass of the 8-lane group sum
        FbRanges rg;
        rg.n = w.nruns - r < 3 ? w.nruns - r : 3;
        for (int q = 0; q < rg.n; q++) { rg.slot[q] = w.runs[3 * (r + q)]; rg.base[q] = w.runs[3 * (r + q) + 1]; rg.count[q] = w.runs[3 * (r + q) + 2]; }
        pt part;
        fb_group_sum(part, w.fb, t, lane, w.msc, rg);
        pt_add(total, total, part);
    }
    if (lane == 0) ws_st_pt(w.pfix, w.N, t, total);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_msm_store(MsmWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    pt total;
    ws_ld_pt(total, w.pfix, w.N, t);
    apt a;
    pt_to_affine(a, total);
    if (w.status[t] != ST_OK) { fe_set_u32(a.x, 0); fe_set_u32(a.y, 0); }
    apt_to_xy64(w.out + 64 * t, a);
}
// ---- generic arithmetic circuit kernels (circuit_core.h)
__global__ __launch_bounds__(BPPP_BLOCK) void k_circuit_phase1(CircuitWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    const u32 key = (w.tio.states && w.tio.n_states != 1) ? w.tio.states[(size_t)BPPP_TRANSCRIPT_STATE_BYTES * t + 200] : 0u;
    for_each_position_group(key, [&]() { circuit_phase1(w, t); });
}
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_circuit_c0_fixed(CircuitWs w) {
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / BPPP_FB_LANES;
    int lane = (int)(g % BPPP_FB_LANES);
    if (t >= w.N) return;
    pt part;
    FbRanges rg;
    circuit_c0_fixed_ranges(rg, w);
    fb_group_sum(part, w.fb, t, lane, w.sc0, rg);
    if (lane == 0) circuit_c0_fixed_store(w, t, part);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_circuit_c0_var(CircuitWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) circuit_c0_var(w, t);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_circuit_c0_finish(CircuitWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) circuit_c0_finish(w, t);
}
// ---- generic reciprocal range proof kernels (recip_core.h)
__global__ __launch_bounds__(BPPP_BLOCK) void k_recip_phase1(RecipWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    const u32 key = (w.tio.states && w.tio.n_states != 1) ? w.tio.states[(size_t)BPPP_TRANSCRIPT_STATE_BYTES * t + 200] : 0u;
    for_each_position_group(key, [&]() { recip_phase1(w, t); });
}
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_recip_c0_fixed(RecipWs w) {
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / BPPP_FB_LANES;
    int lane = (int)(g % BPPP_FB_LANES);
    if (t >= w.N) return;
    pt part;
    FbRanges rg;
    recip_c0_fixed_ranges(rg, w);
    fb_group_sum(part, w.fb, t, lane, w.sc0, rg);
    if (lane == 0) recip_c0_fixed_store(w, t, part);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_recip_c0_var(RecipWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) recip_c0_var(w, t);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_recip_c0_finish(RecipWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) recip_c0_finish(w, t);
}
