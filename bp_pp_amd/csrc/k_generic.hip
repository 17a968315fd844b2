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
// window tables of every round's X and R (fast path of the rounds; wnla_core.h)
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_TABLES_MIN_WAVES) void k_wnla_tables(WnlaWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) wnla_verify_tables(w, t);
}
// calls that leave the chip empty: a lane per (point, part) table (32 x parts lanes per instance, 2 x rounds x parts of them active)
// (lp = lanes per (instance, part): the power of two >= 2 x rounds -- round 6: 32 whatever the rounds left 4 lanes in 32 working for a
// 2-round argument, and the launch of 8 x the wavefronts took 1.9 ms at 512 instances where the 4-round argument's took 0.55)
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_tables_split(WnlaWs w, int parts, int lp) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g / ((size_t)lp * (size_t)parts);
    const int p = (int)(g % (size_t)lp), h = (int)((g / (size_t)lp) % (size_t)parts);
    if (t < w.N && p < 2 * w.rounds) wnla_verify_table_one(w, t, p, h, parts);
}
// the round on lane groups (group = 2 or 4 lanes per instance; 8 / 16 over tables in 2 / 4 parts), for batches that under-fill the chip; needs the fast path's tables
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_round_grp(WnlaWs w, int k, int group) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g / (size_t)group;
    if (t >= w.N) return;
    const int q = (int)(g % (size_t)group);
    const u32 key = w.divergent_positions ? w.tstate[(size_t)50 * w.N + t] : 0u;
    for_each_position_group(key, [&]() { wnla_verify_round(w, t, k, q, group); });
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
// 2^lg wavefronts per 64 instances, for batches that leave SIMDs empty with one: block b takes part b mod 2^lg of instances
// 64 (b >> lg) ..., so a wavefront's accesses stay as coalesced as the one-part kernel's and the part index is uniform in it; every
// part builds its own eighth (quarter, half) of the coefficient tables and the generator scalars that read it, k_wnla_final_scalars_join
// adds the shares of v
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_final_scalars_grp(WnlaWs w, int lg) {
    const size_t t = (size_t)(blockIdx.x >> lg) * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) wnla_final_scalars_part(w, t, (int)(blockIdx.x & ((1u << lg) - 1)), lg);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_final_scalars_join(WnlaWs w, int lg) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) wnla_final_scalars_join(w, t, lg);
}
// NL lanes per instance: 8, or ONE from the size at which one lane per instance fills every SIMD twice over (as in the u64 verifier's
// fixed-base kernels: the lane then walks each scalar's windows in order -- the recoded scalar is a shift register, the window's base
// address an increment -- and no tree of complete additions joins lane sums)
template <int NL>
__device__ __forceinline__ void wnla_msm_lanes(const WnlaWs& w) {
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / NL;
    int lane = (int)(g % NL);
    if (t >= w.N) return;
    pt part;
    FbRanges rg;
    wnla_msm_ranges(rg, w);
    fb_group_sum<NL>(part, w.fb, t, lane, w.msc, rg);
    if (lane == 0) wnla_verify_store(w, t, part);
}
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_wnla_msm(WnlaWs w, int commit_mode) { wnla_msm_lanes<BPPP_FB_LANES>(w); (void)commit_mode; }
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_wnla_msm_l1(WnlaWs w) { wnla_msm_lanes<1>(w); }
// a wavefront per instance: calls so small that 8 lanes per instance leave the chip empty and the call waits for one lane's chain of
// (1 + |g_vec| + |h_vec|) x windows / 8 dependent table additions (bppp_generic.hip: generic_fb_wide)
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_wnla_msm_l64(WnlaWs w) { wnla_msm_lanes<64>(w); }
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
// number of rejected instances of a finished batch (the sharded entry points all-reduce it): grid-stride, one ballot and one
// device-scope atomic per wavefront that saw a reject
__global__ __launch_bounds__(256) void k_count_rejects(const uint8_t* accept, size_t n, int* reject_count) {
    const size_t stride = (size_t)gridDim.x * 256;
    int mine = 0;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += stride) mine += accept[t] == 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mine += __shfl_xor(mine, m, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(reject_count, mine);
}
// ---- random-linear-combination mode of the final MSM (wnla_rlc_core.h)
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_rlc_lhs(WnlaWs w, RlcWs r) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N && w.status[t] == ST_OK && !rlc_done_by_bucket_stage(r, t)) wnla_rlc_lhs(w, r, t);
}
// combined scalars: one lane group (8 lanes) per chunk of 8 instances, lane j owns instance j's weight; A_i = sum_j w_j s_ji is
// summed across the group with shuffles and stored once per chunk (at the chunk's first instance)
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_wnla_rlc_chunk(WnlaWs w, RlcWs r) {
    const size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    const size_t chunk = g / BPPP_RLC_CHUNK;
    const int lane = (int)(g % BPPP_RLC_CHUNK);
    const size_t N = w.N, nchunks = (N + BPPP_RLC_CHUNK - 1) / BPPP_RLC_CHUNK;
    if (chunk >= nchunks) return;            // whole lane groups leave together
    const size_t first = chunk * BPPP_RLC_CHUNK, t = first + lane;
    if (rlc_done_by_bucket_stage(r, first)) {          // whole lane groups leave together (the superchunk size is a multiple of 8)
        if (lane == 0) r.flag[chunk] = 0;
        return;
    }
    int bad = (t < N) ? (w.status[t] != ST_OK) : 1;
#pragma unroll
    for (int m = 1; m < BPPP_RLC_CHUNK; m <<= 1) bad |= __shfl_xor(bad, m, 64);
    if (bad) {                               // incomplete chunk or a flagged instance: the exact kernels take it
        if (lane == 0) { r.flag[chunk] = 1; r.list[atomicAdd(r.count, 1)] = (u32)chunk; }
        return;
    }
    if (lane == 0) r.flag[chunk] = 2;        // pending: k_wnla_rlc_check decides
    u64 a, b;
    rlc_weight(a, b, r, t);
    sc wt;
    rlc_weight_scalar(wt, a, b);
    const int NB = 1 + w.ng + w.nh;
#pragma nounroll
    for (int i = 0; i < NB; i++) {
        sc s, p;
        ws_ld8(s.v, w.msc, N, t, i);
        sc_mul(p, s, wt);
#pragma unroll
        for (int m = 1; m < BPPP_RLC_CHUNK; m <<= 1) {
            sc o;
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] = __shfl_xor(p.v[k], m, 64);
            sc_add(p, p, o);
        }
        if (lane == 0) ws_st8(r.sc, N, first, i, p.v);
    }
}
// the chunk's MSM and verdict: a whole wavefront per chunk (the 1 + ng + nh bases x windows dealt over 64 lanes, 6-step tree) --
// there are only N / 8 chunks, and an 8-lane group per chunk would leave most SIMDs without a wavefront
__global__ __launch_bounds__(64) void k_wnla_rlc_check(WnlaWs w, RlcWs r) {
    const int lane = (int)threadIdx.x;
    const size_t N = w.N, nchunks = (N + BPPP_RLC_CHUNK - 1) / BPPP_RLC_CHUNK;
#pragma nounroll
    for (size_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (r.flag[chunk] != 2) continue;
        const size_t first = chunk * BPPP_RLC_CHUNK;
        FbRanges rg;
        wnla_msm_ranges(rg, w);
        pt rhs, lhs, L;
        fb_group_sum<64>(rhs, w.fb, first, lane, r.sc, rg);
        pt_set_identity(lhs);
#pragma nounroll
        for (int l = 0; l < BPPP_RLC_CHUNK; l++) { ws_ld_pt(L, r.lhs, N, first + l); pt_add(lhs, lhs, L); }
        const bool ok = pt_eq(lhs, rhs);
        if (ok && lane < BPPP_RLC_CHUNK) w.accept[first + lane] = 1;
        if (lane == 0) {
            r.flag[chunk] = ok ? 0 : 1;
            if (!ok) r.list[atomicAdd(r.count, 1)] = (u32)chunk;
        }
    }
}
// the exact final MSM for the instances of the chunks that did not pass: a whole wavefront per instance from the compacted list
// (only a few are expected) ...
__global__ __launch_bounds__(64) void k_wnla_msm_flagged(WnlaWs w, RlcWs r) {
    const int lane = (int)threadIdx.x;
    const size_t nchunks = (w.N + BPPP_RLC_CHUNK - 1) / BPPP_RLC_CHUNK;
    if ((size_t)(*r.count) * 8 > nchunks) return;   // many flagged chunks: k_wnla_msm_flagged_dense does them
    const size_t items = (size_t)(*r.count) * BPPP_RLC_CHUNK;
#pragma nounroll
    for (size_t item = blockIdx.x; item < items; item += gridDim.x) {
        const size_t t = (size_t)r.list[item / BPPP_RLC_CHUNK] * BPPP_RLC_CHUNK + item % BPPP_RLC_CHUNK;
        if (t >= w.N) continue;
        pt part;
        FbRanges rg;
        wnla_msm_ranges(rg, w);
        fb_group_sum<64>(part, w.fb, t, lane, w.msc, rg);
        if (lane == 0) wnla_verify_store(w, t, part);
    }
}
// ... and the regular 8-lane kernel, skipping the chunks that passed, when more than 1/8 of the chunks failed (a broken input stream)
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_wnla_msm_flagged_dense(WnlaWs w, RlcWs r) {
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / BPPP_FB_LANES;
    int lane = (int)(g % BPPP_FB_LANES);
    const size_t nchunks = (w.N + BPPP_RLC_CHUNK - 1) / BPPP_RLC_CHUNK;
    if ((size_t)(*r.count) * 8 <= nchunks) return;
    if (t >= w.N || !r.flag[t / BPPP_RLC_CHUNK]) return;
    pt part;
    FbRanges rg;
    wnla_msm_ranges(rg, w);
    fb_group_sum(part, w.fb, t, lane, w.msc, rg);
    if (lane == 0) wnla_verify_store(w, t, part);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_wnla_accept_flagged(WnlaWs w, RlcWs r) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N && r.flag[t / BPPP_RLC_CHUNK]) wnla_verify_accept(w, t);
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
    const u32 key = preloaded_position_key(w.tio.states, w.tio.n_states, t);
    for_each_position_group(key, [&]() { circuit_phase1(w, t); });
}
template <int NL>
__device__ __forceinline__ void circuit_c0_fixed_lanes(const CircuitWs& w) {
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / NL;
    int lane = (int)(g % NL);
    if (t >= w.N) return;
    pt part;
    FbRanges rg;
    circuit_c0_fixed_ranges(rg, w);
    fb_group_sum<NL>(part, w.fb, t, lane, w.sc0, rg);
    if (lane == 0) circuit_c0_fixed_store(w, t, part);
}
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_circuit_c0_fixed(CircuitWs w) { circuit_c0_fixed_lanes<BPPP_FB_LANES>(w); }
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_circuit_c0_fixed_l64(CircuitWs w) { circuit_c0_fixed_lanes<64>(w); }
__global__ __launch_bounds__(BPPP_BLOCK) void k_circuit_c0_tables(CircuitWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) circuit_c0_tables(w, t);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_circuit_c0_var(CircuitWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) circuit_c0_var(w, t);
}
// L lanes per instance, a lane per point (circuit_core.h: circuit_c0_var_points): tables and sum in one launch
__global__ __launch_bounds__(BPPP_BLOCK) void k_circuit_c0_var_pts(CircuitWs w, int L) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g / (size_t)L;
    if (t >= w.N) return;                     // whole groups leave together
    circuit_c0_var_points(w, t, (int)(g % (size_t)L), L);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_circuit_c0_finish(CircuitWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) circuit_c0_finish(w, t);
}
// ---- generic reciprocal range proof kernels (recip_core.h)
__global__ __launch_bounds__(BPPP_BLOCK) void k_recip_phase1(RecipWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= w.N) return;
    const u32 key = preloaded_position_key(w.tio.states, w.tio.n_states, t);
    for_each_position_group(key, [&]() { recip_phase1(w, t); });
}
// G = 2, 4 or 8 lanes per instance (recip_core.h: recip_phase1): calls whose one-lane kernels leave wavefront slots free
__global__ __launch_bounds__(BPPP_BLOCK) void k_recip_phase1_grp(RecipWs w, int G) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g / (size_t)G;
    if (t >= w.N) return;                     // whole groups leave together
    const int q = (int)(g % (size_t)G);
    const u32 key = preloaded_position_key(w.tio.states, w.tio.n_states, t);
    for_each_position_group(key, [&]() { recip_phase1(w, t, q, G); });
}
template <int NL>
__device__ __forceinline__ void recip_c0_fixed_lanes(const RecipWs& w) {
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / NL;
    int lane = (int)(g % NL);
    if (t >= w.N) return;
    pt part;
    FbRanges rg;
    recip_c0_fixed_ranges(rg, w);
    fb_group_sum<NL>(part, w.fb, t, lane, w.sc0, rg);
    if (lane == 0) recip_c0_fixed_store(w, t, part);
}
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_recip_c0_fixed(RecipWs w) { recip_c0_fixed_lanes<BPPP_FB_LANES>(w); }
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_recip_c0_fixed_l1(RecipWs w) { recip_c0_fixed_lanes<1>(w); }
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_recip_c0_fixed_l64(RecipWs w) { recip_c0_fixed_lanes<64>(w); }
__global__ __launch_bounds__(BPPP_BLOCK) void k_recip_c0_var(RecipWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) recip_c0_var(w, t);
}
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_TABLES_MIN_WAVES) void k_recip_c0_tables(RecipWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) recip_c0_tables(w, t);
}
// the sum on lane groups (2 or 4 lanes per instance), for batches that under-fill the chip; needs the fast path's tables
__global__ __launch_bounds__(BPPP_BLOCK) void k_recip_c0_var_grp(RecipWs w, int group) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g / (size_t)group;
    if (t >= w.N) return;
    recip_c0_var(w, t, (int)(g % (size_t)group), group);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_recip_c0_finish(RecipWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) recip_c0_finish(w, t);
}
