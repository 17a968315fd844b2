// u64 verifier: the one-lane-per-proof shared-doubling kernels (window tables, C0 variable-base half, WNLA rounds, RLC left-hand side).
// Part of libbppp_hip.so; per-lane work lives in the *_core.h headers, declarations in kernels.h.
#include "kernels.h"

using namespace bppp;

__global__ __launch_bounds__(BPPP_BLOCK, BPPP_TABLES_MIN_WAVES) void k_verify_tables(VerifyWs ws) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < ws.N) verify_tables(ws, t);
}
// the build as five kernels with the inversions between them shared by G proofs each (plan_core.h: shared_inv; k_verify_shared_inv*)
#define BPPP_TABLES_PASS_KERNEL(P)                                                                                      \
    __global__ __launch_bounds__(BPPP_BLOCK, BPPP_TABLES_MIN_WAVES) void k_verify_tables_pass##P(VerifyWs ws) {          \
        size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;                                                        \
        if (t < ws.N) verify_tables_pass<P>(ws, t);                                                                      \
    }
BPPP_TABLES_PASS_KERNEL(0)
BPPP_TABLES_PASS_KERNEL(1)
BPPP_TABLES_PASS_KERNEL(2)
BPPP_TABLES_PASS_KERNEL(3)
BPPP_TABLES_PASS_KERNEL(4)
// beside phase 1 on the helper stream (bppp_u64.hip: tables_beside): decodes the proof's points itself
__global__ __launch_bounds__(BPPP_C0VAR_SMALL_BLOCK, 2) void k_verify_tables_own(VerifyWs ws) {      // (256 threads: see k_verify_phase1_wg4)
    size_t t = (size_t)blockIdx.x * BPPP_C0VAR_SMALL_BLOCK + threadIdx.x;
    if (t < ws.N) verify_tables_own(ws, t);
}
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_verify_c0_var(VerifyWs ws) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < ws.N) verify_c0_var(ws, t);
}
__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_verify_round(VerifyWs ws, int k) {
    __shared__ u32 sponge[50 * BPPP_LDS_STRIDE];     // the wavefront's sponge states while they are hashed (merlin.h: strobe_lds)
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= ws.N) return;
    const u32 key = (ws.states && ws.n_states != 1) ? ws.tstate[(size_t)50 * ws.N + t] : 0u;   // the stored sponge position
#if defined(__HIP_DEVICE_COMPILE__)
    for_each_position_group(key, [&]() { verify_round_lds(ws, t, k, sponge + threadIdx.x); });
#else
    (void)key; (void)sponge;
#endif
}
// Small-batch variants (at most one wavefront per SIMD anyway: 2^16 proofs = 1024 workgroups on 1024 SIMDs): no register cap, so
// nothing spills -- the latency of the lone wavefront is what the batch takes.  The host picks them when the lane-kernel grid does
// not exceed the number of SIMDs.
// The 5-point sum needs 231 registers, so TWO of these wavefronts fit a SIMD -- and when a launch of exactly one wavefront per SIMD
// (2^16 proofs) arrives as 1,024 single-wavefront workgroups, the dispatcher puts two of them on some SIMDs of a CU and none on
// others: 5-10 % of the wavefronts then take 2.9 ms instead of 1.65 and the kernel waits for them (tools/probes/wave_timeline.py,
// profiles/r04/r04_f_wave_timeline.txt; reserving LDS so that a CU gets exactly four did not change it -- the imbalance is inside the CU).
// The four wavefronts of ONE workgroup, however, are dealt one to each SIMD (tools/probes/wgmap: 256 workgroups of 256 threads -> 1,024 SIMDs
// with one wavefront each), so this variant runs in 256-thread workgroups; the registers still leave room for the fixed-base half of
// C0 that runs beside it.  No barrier, no LDS: the workgroup size is placement only.
__global__ __launch_bounds__(BPPP_C0VAR_SMALL_BLOCK) void k_verify_c0_var_small(VerifyWs ws) {
    size_t t = (size_t)blockIdx.x * BPPP_C0VAR_SMALL_BLOCK + threadIdx.x;
    if (t < ws.N) verify_c0_var(ws, t);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_round_small(VerifyWs ws, int k) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= ws.N) return;
    const u32 key = (ws.states && ws.n_states != 1) ? ws.tstate[(size_t)50 * ws.N + t] : 0u;
    for_each_position_group(key, [&]() { verify_round(ws, t, k); });
}
// a round in two kernels (verify_core.h: verify_round_on, part): the head -- C_{k-1} to affine, transcript, challenge -- and the tail -- the
// two-point sum.  For the LAST round of batches whose one-lane kernels are a lone wavefront per SIMD the tail runs beside the final
// fixed-base sum, in a 256-register build so that the two share a SIMD (bppp_u64.hip: tail_beside).
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_round_head_small(VerifyWs ws, int k) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= ws.N) return;
    const u32 key = (ws.states && ws.n_states != 1) ? ws.tstate[(size_t)50 * ws.N + t] : 0u;
    for_each_position_group(key, [&]() { verify_round(ws, t, k, -1, 4, 1); });
}
// (256-thread workgroups for the reason given at k_verify_c0_var_small: their four wavefronts land one per SIMD, while single-wavefront
// workgroups of a kernel that fits a SIMD twice can land two on one SIMD and none on its neighbour)
__global__ __launch_bounds__(BPPP_C0VAR_SMALL_BLOCK, 2) void k_verify_round_tail(VerifyWs ws, int k) {
    size_t t = (size_t)blockIdx.x * BPPP_C0VAR_SMALL_BLOCK + threadIdx.x;
    if (t < ws.N) verify_round(ws, t, k, -1, 4, 2);
}
// ---- random-linear-combination batch mode (rlc_core.h)
__global__ __launch_bounds__(BPPP_BLOCK) void k_rlc_lhs(VerifyWs ws, RlcWs r) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < ws.N && !rlc_done_by_bucket_stage(r, t)) rlc_lhs(ws, r, t);
}

// small batches (4 n lanes still leave SIMDs empty): four lanes per proof, the round's two-point sum split into its four GLV
// streams (straus_core.h: straus_affine_g4); everything else is done identically by the four lanes
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_round_g4(VerifyWs ws, int k) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 2;
    if (t >= ws.N) return;                    // whole groups leave together
    const int q = (int)(g & 3);
    const u32 key = (ws.states && ws.n_states != 1) ? ws.tstate[(size_t)50 * ws.N + t] : 0u;
    for_each_position_group(key, [&]() { verify_round(ws, t, k, q); });
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_c0_var_g4(VerifyWs ws) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 2;
    if (t < ws.N) verify_c0_var(ws, t, (int)(g & 3));
}
// two lanes per proof: batches between a quarter and a half of the wavefront slots (one lane per proof would leave half the SIMDs empty)
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_round_g2(VerifyWs ws, int k) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 1;
    if (t >= ws.N) return;
    const int q = (int)(g & 1);
    const u32 key = (ws.states && ws.n_states != 1) ? ws.tstate[(size_t)50 * ws.N + t] : 0u;
    for_each_position_group(key, [&]() { verify_round(ws, t, k, q, 2); });
}

// ---- calls so small that the chip is empty (a few proofs per SIMD at most): what a call takes is the longest dependent chain of one
// proof, so the chains are cut further.  Window tables: a lane per table (13 points x PARTS tables: P and 2^65 P, or P, 2^35 P, 2^70 P,
// 2^100 P -- 16 PARTS lanes per proof, 13 PARTS active); sums: a lane per part of a GLV stream (straus_core.h: straus_affine_split) --
// 4 PARTS lanes per proof in a round, 16 PARTS (10 PARTS active) for C0.  Four parts up to one proof per SIMD, two up to four per SIMD.
template <int PARTS>
__device__ __forceinline__ void verify_tables_split(const VerifyWs& ws) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g / (16 * PARTS);
    const int p = (int)(g & 15), h = (int)((g >> 4) % PARTS);
    if (t < ws.N && p < BPPP_VPOINTS) verify_table_one(ws, t, p, h, PARTS, true);   // from the caller's bytes: runs beside phase 1
}
// one table per point (the lane groups' tables: 4,096 < n <= 16,384 proofs), a lane each, beside phase 1
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_tables_split1(VerifyWs ws) { verify_tables_split<1>(ws); }
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_tables_split2(VerifyWs ws) { verify_tables_split<2>(ws); }
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_tables_split4(VerifyWs ws) { verify_tables_split<4>(ws); }
template <int G>
__device__ __forceinline__ void verify_round_group(const VerifyWs& ws, int k) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g / G;
    if (t >= ws.N) return;                    // whole groups leave together
    const int q = (int)(g % G);
    const u32 key = (ws.states && ws.n_states != 1) ? ws.tstate[(size_t)50 * ws.N + t] : 0u;
    for_each_position_group(key, [&]() { verify_round(ws, t, k, q, G); });
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_round_g8(VerifyWs ws, int k) { verify_round_group<8>(ws, k); }
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_round_g16(VerifyWs ws, int k) { verify_round_group<16>(ws, k); }
template <int G>
__device__ __forceinline__ void verify_c0_var_group(const VerifyWs& ws) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g / G;
    if (t < ws.N) verify_c0_var(ws, t, (int)(g % G), G);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_c0_var_g32(VerifyWs ws) { verify_c0_var_group<32>(ws); }
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_c0_var_g64(VerifyWs ws) { verify_c0_var_group<64>(ws); }
