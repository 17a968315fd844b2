// u64 verifier: the 8-lanes-per-proof fixed-base kernels (C0 fixed half, final check, RLC chunks), table construction, commit_value.
// Part of libbppp_hip.so; per-lane work lives in the *_core.h headers, declarations in kernels.h.
#include "kernels.h"

using namespace bppp;

// fixed-base MSM kernels: NL lanes per proof (BPPP_FB_LANES = 8 while the batch is small enough for the extra lanes to fill the
// SIMDs; 1 from 2^17 proofs up, where one lane per proof already gives two wavefronts per SIMD and the 3-step tree of complete
// additions that joins the lane sums is 1.7 % of the batch: 162.3 -> 159.6 ms per 2^20 proofs, 21.75 -> 21.37 ms per 2^17),
// 256-thread workgroups, registers capped for 2 wavefronts per SIMD
template <int NL>
__device__ __forceinline__ void verify_c0_fixed_lanes(const VerifyWs& ws) {
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / NL;
    int lane = (int)(g % NL);
    if (t >= ws.N) return;   // whole lane groups leave together
    pt part;
    FbRanges rg;
    verify_c0_fixed_ranges(rg);
    fb_group_sum<NL>(part, fb_of(ws), t, lane, ws.sc0, rg);
    if (lane == 0) verify_c0_fixed_store(ws, t, part);
}
template <int NL>
__device__ __forceinline__ void verify_final_check_lanes(const VerifyWs& ws) {
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / NL;
    int lane = (int)(g % NL);
    if (t >= ws.N) return;
    pt part;
    FbRanges rg;
    verify_final_check_ranges(rg);
    fb_group_sum<NL>(part, fb_of(ws), t, lane, ws.fsc, rg);
    if (lane == 0) verify_final_check_store(ws, t, part);
}
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_c0_fixed(VerifyWs ws) { verify_c0_fixed_lanes<BPPP_FB_LANES>(ws); }
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_c0_fixed_l1(VerifyWs ws) { verify_c0_fixed_lanes<1>(ws); }
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_final_check(VerifyWs ws) { verify_final_check_lanes<BPPP_FB_LANES>(ws); }
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_final_check_l1(VerifyWs ws) { verify_final_check_lanes<1>(ws); }
// a whole wavefront per proof: calls of a handful of proofs (the chip is empty; 204 / 588 table additions over 64 lanes and a 6-step tree
// instead of 26 / 74 additions per lane and a 3-step tree)
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_c0_fixed_l64(VerifyWs ws) { verify_c0_fixed_lanes<64>(ws); }
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_final_check_l64(VerifyWs ws) { verify_final_check_lanes<64>(ws); }
// RLC mode, one combined final check per chunk of C proofs (rlc_core.h): C lanes per chunk, a lane per proof -- its 49 weighted
// scalars, summed over the chunk with shuffles; the chunk's ONE 49-base fixed-base sum on the same C lanes; the weighted commitments
// summed the same way.  C = 8 (round 1) or 32: the right-hand side's 588 table additions are then shared by four times the proofs
// (74 -> 18 per proof), and a chunk holds a bad proof four times as often -- bppp_u64.hip picks C from the previous call's reject rate.
template <int C>
__device__ __forceinline__ void rlc_chunk_lanes(const VerifyWs& ws, const RlcWs& r) {
    const size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    const size_t chunk = g / C;
    const int lane = (int)(g % C);
    const size_t N = ws.N, nchunks = (N + C - 1) / C;
    if (chunk >= nchunks) return;            // whole lane groups leave together
    const size_t t = chunk * C + lane;
    if (rlc_done_by_bucket_stage(r, chunk * C)) {     // whole lane groups leave together (super_m is a multiple of C)
        if (lane == 0) r.flag[chunk] = 0;
        return;
    }
    // a chunk with a missing or flagged proof goes to the exact kernels
    int bad = (t < N) ? (ws.status[t] != ST_OK) : 1;
#pragma unroll
    for (int m = 1; m < C; m <<= 1) bad |= __shfl_xor(bad, m, 64);
    if (bad) {
        if (lane == 0) { r.flag[chunk] = 1; r.list[atomicAdd(r.count, 1)] = (u32)chunk; }
        return;
    }
    u64 a, b;
    rlc_weight(a, b, r, t);
    sc w;
    rlc_weight_scalar(w, a, b);
#pragma nounroll
    for (int i = 0; i < BPPP_NG; i++) {
        sc p;
        rlc_product(p, ws, w, t, i);
#pragma unroll
        for (int m = 1; m < C; m <<= 1) {
            sc o;
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] = __shfl_xor(p.v[k], m, 64);
            sc_add(p, p, o);
        }
        ws_st8(r.sc, N, t, i, p.v);          // every lane keeps its own (identical) copy: no cross-lane memory traffic
    }
    FbRanges rg;
    rlc_ranges(rg);
    pt rhs, lhs;
    fb_group_sum<C>(rhs, fb_of(ws), t, lane, r.sc, rg);
    ws_ld_pt(lhs, r.lhs, N, t);
    lane_group_sum<C>(lhs);
    const bool ok = pt_eq(lhs, rhs);
    if (ok) ws.accept[t] = 1;
    if (lane == 0) {
        r.flag[chunk] = ok ? 0 : 1;
        if (!ok) r.list[atomicAdd(r.count, 1)] = (u32)chunk;
    }
}
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_rlc_chunk(VerifyWs ws, RlcWs r) { rlc_chunk_lanes<BPPP_RLC_CHUNK>(ws, r); }
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_rlc_chunk_c32(VerifyWs ws, RlcWs r) { rlc_chunk_lanes<32>(ws, r); }
// exact final check of the proofs of the flagged chunks.  Three forms by how many there are (the count lives on the device: all three are
// launched and two of them return at once):
//   a handful (< BPPP_FLAGGED_L8_FROM proofs)  a whole wavefront per proof -- 588 table additions over 64 lanes, 6-step tree: latency of one
//                                             proof's sum matters, not throughput
//   up to an eighth of the batch                8 lanes per proof over the compacted list (round 5: chunks of 32 flag 32 proofs per bad
//                                             proof -- 3 % of a batch at 1/1024 corrupted: 4.7 ms on a wavefront per proof, 1.4 here)
//   more                                        the regular 8-lane kernel over the whole batch, skipping the chunks that passed
#define BPPP_FLAGGED_L8_FROM 4096
__global__ __launch_bounds__(64) void k_verify_final_check_flagged(VerifyWs ws, RlcWs r) {
    const int lane = (int)threadIdx.x;
    const size_t C = rlc_chunk_of(r);
    const size_t items = (size_t)(*r.count) * C;
    if (items >= BPPP_FLAGGED_L8_FROM || items * 8 > ws.N) return;   // many: k_verify_final_check_flagged_l8 does them; dense (a small batch with
                                                                     // many bad chunks): k_verify_final_check_flagged_dense does -- exactly one of the three
#pragma nounroll
    for (size_t item = blockIdx.x; item < items; item += gridDim.x) {
        const size_t t = (size_t)r.list[item / C] * C + item % C;
        if (t >= ws.N) continue;
        pt part;
        FbRanges rg;
        verify_final_check_ranges(rg);
        fb_group_sum<64>(part, fb_of(ws), t, lane, ws.fsc, rg);
        if (lane == 0) verify_final_check_store(ws, t, part);
    }
}
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_final_check_flagged_l8(VerifyWs ws, RlcWs r) {
    const size_t C = rlc_chunk_of(r);
    const size_t items = (size_t)(*r.count) * C;
    if (items < BPPP_FLAGGED_L8_FROM || items * 8 > ws.N) return;
    const int lane = (int)(threadIdx.x % BPPP_FB_LANES);
    const size_t per_block = BPPP_FB_BLOCK / BPPP_FB_LANES;
#pragma nounroll
    for (size_t item = (size_t)blockIdx.x * per_block + threadIdx.x / BPPP_FB_LANES; item < items; item += (size_t)gridDim.x * per_block) {
        const size_t t = (size_t)r.list[item / C] * C + item % C;
        if (t >= ws.N) continue;                 // whole lane groups skip together
        pt part;
        FbRanges rg;
        verify_final_check_ranges(rg);
        fb_group_sum(part, fb_of(ws), t, lane, ws.fsc, rg);
        if (lane == 0) verify_final_check_store(ws, t, part);
    }
}
// the same for a batch where more than 1/8 of the proofs sit in chunks that failed (an adversarial or broken input stream): the regular
// 8-lane kernel over the whole batch, skipping the chunks that passed
__global__ __launch_bounds__(BPPP_FB_BLOCK, BPPP_FB_MIN_WAVES) void k_verify_final_check_flagged_dense(VerifyWs ws, RlcWs r) {
    size_t g = (size_t)blockIdx.x * BPPP_FB_BLOCK + threadIdx.x;
    size_t t = g / BPPP_FB_LANES;
    int lane = (int)(g % BPPP_FB_LANES);
    const size_t C = rlc_chunk_of(r);
    if ((size_t)(*r.count) * C * 8 <= ws.N) return;
    if (t >= ws.N || !r.flag[t / C]) return;
    pt part;
    FbRanges rg;
    verify_final_check_ranges(rg);
    fb_group_sum(part, fb_of(ws), t, lane, ws.fsc, rg);
    if (lane == 0) verify_final_check_store(ws, t, part);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_fb_build_pass1(FbBuild fb, size_t nthreads) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < nthreads) fb_build_pass1(fb, t);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_fb_build_pass2(FbBuild fb, size_t nthreads) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < nthreads) fb_build_pass2(fb, t);
}
// U64RangeProofProtocol::commit_value (u64_proof.rs:37-39): x*g + s*h_vec[0] through the fixed-base tables
__global__ __launch_bounds__(BPPP_BLOCK) void k_commit_value(VerifyWs ws, const uint64_t* x, const uint8_t* s, uint8_t* out,
                                                             int* flags, FbTable ct) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= ws.N) return;
    sc xs, ss;
    sc_set_u64(xs, x[t]);
    if (!sc_from_be(ss, s + 32 * t)) {
        atomicOr(flags, 1);
        sc_set_u32(ss, 0);
    }
    // scalars for bases 0 (g) and 17 (h_vec[0]) staged in the fsc scratch area, slots 0 and 1
    ws_st8(ws.fsc, ws.N, t, 0, xs.v);
    ws_st8(ws.fsc, ws.N, t, 1, ss.v);
    pt acc;
    pt_set_identity(acc);
    if (ct.table) {            // "ct_prover": x and s are the committer's secrets -- every entry of every window read, selected by mask
#pragma nounroll
        for (int w = 0; w < 16; w++) fb_lookup_add_ct(acc, ct, 0, w, xs.v);
#pragma nounroll
        for (int w = 0; w < 64; w++) fb_lookup_add_ct(acc, ct, 17, w, ss.v);
    } else {
        fixed_base_msm(acc, fb_of(ws), t, ws.fsc, 0, 0, 1, 64);     // x is a u64: 3 of the 12 windows at 22 bits
        fixed_base_msm(acc, fb_of(ws), t, ws.fsc, 1, 17, 1);
    }
    apt a;
    pt_to_affine(a, acc);
    apt_to_xy64(out + 64 * t, a);
}
