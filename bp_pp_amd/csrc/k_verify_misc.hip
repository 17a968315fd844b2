// u64 verifier: phase 1 (decode + transcript + scalars), final scalars, accept, generator decoding, SEC1 expansion.
// Part of libbppp_hip.so; per-lane work lives in the *_core.h headers, declarations in kernels.h.
#include "kernels.h"

using namespace bppp;

// One-lane-per-proof kernels of the u64 verifier: minimum waves per SIMD the register allocator must leave room for
// (2 => at most 256 VGPR + AGPR per lane, so two wavefronts share a SIMD and cover each other's table-gather latency).

__global__ __launch_bounds__(BPPP_BLOCK, BPPP_LANE_MIN_WAVES) void k_verify_phase1(VerifyWs ws) {
    __shared__ u32 sponge[50 * BPPP_LDS_STRIDE];     // the wavefront's 64 sponge states, word-major (merlin.h: strobe_lds): 12.5 KB
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= ws.N) return;
    // per-proof pre-loaded transcripts may sit at different byte positions (byte 200 of the serialized state)
    const u32 key = preloaded_position_key(ws.states, ws.n_states, t);
#if defined(__HIP_DEVICE_COMPILE__)
    for_each_position_group(key, [&]() { verify_phase1_lds(ws, t, sponge + threadIdx.x); });
#else
    (void)key; (void)sponge;
#endif
}
// the same in 256-thread workgroups (a sponge block per wavefront), for the batches that run the table kernel BESIDE phase 1 (bppp_u64.hip:
// tables_beside): the four wavefronts of a workgroup land one per SIMD, so a workgroup of each kernel gives every SIMD of a CU one
// wavefront of each (single-wavefront workgroups of two-per-SIMD kernels land unevenly: k_verify_var.hip)
__global__ __launch_bounds__(BPPP_C0VAR_SMALL_BLOCK, 2) void k_verify_phase1_wg4(VerifyWs ws) {
    __shared__ u32 sponge[(BPPP_C0VAR_SMALL_BLOCK / 64) * 50 * BPPP_LDS_STRIDE];
    size_t t = (size_t)blockIdx.x * BPPP_C0VAR_SMALL_BLOCK + threadIdx.x;
    if (t >= ws.N) return;
    const u32 key = preloaded_position_key(ws.states, ws.n_states, t);
#if defined(__HIP_DEVICE_COMPILE__)
    u32* col = sponge + (threadIdx.x >> 6) * 50 * BPPP_LDS_STRIDE + (threadIdx.x & 63);
    for_each_position_group(key, [&]() { verify_phase1_lds(ws, t, col); });
#else
    (void)key; (void)sponge;
#endif
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_phase1_small(VerifyWs ws) {     // see k_verify_round_small
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= ws.N) return;
    const u32 key = preloaded_position_key(ws.states, ws.n_states, t);
    for_each_position_group(key, [&]() { verify_phase1(ws, t); });
}
// small calls: sixteen lanes per proof for phase 1 too (verify_core.h: verify_phase1_on, lane) -- the scalar section's chain is spread
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_phase1_g16(VerifyWs ws) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 4;
    if (t >= ws.N) return;
    const int lane = (int)(g & 15);
    const u32 key = preloaded_position_key(ws.states, ws.n_states, t);
    for_each_position_group(key, [&]() {
        int32_t status = ST_OK;
        strobe tr;
        phase1_start_state(tr, status, ws, t);
        verify_phase1_on(ws, t, tr, status, lane);
    });
}
// small calls: sixteen lanes per proof (verify_core.h: verify_final_scalars_lane); whole groups are active or leave together
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_final_scalars_g16(VerifyWs ws) {
    const size_t g = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    const size_t t = g >> 4;
    if (t < ws.N) verify_final_scalars_lane(ws, t, (int)(g & 15));
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_final_scalars(VerifyWs ws) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < ws.N) verify_final_scalars(ws, t);
}
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_accept(VerifyWs ws, int* reject_count) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < ws.N) {
        verify_accept(ws, t);
        if (reject_count && !ws.accept[t]) atomicAdd(reject_count, 1);
    }
}
// (hist_count: the context's own reject counter of its RLC calls -- what the next call's choice of chunk sizes is made from)
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_accept_flagged(VerifyWs ws, RlcWs r, int* reject_count, int* hist_count) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t >= ws.N) return;
    if (r.flag[t / rlc_chunk_of(r)]) verify_accept(ws, t);
    if (!ws.accept[t]) {
        if (reject_count) atomicAdd(reject_count, 1);
        if (hist_count) atomicAdd(hist_count, 1);
    }
}
// generator decoding + validation (context creation): 64-B big-endian -> device affine; flags[0] |= 1 on a bad point
__global__ void k_decode_generators(const uint8_t* in, apt* out, int n, int* flags) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    apt a;
    if (!apt_from_xy64(a, in + 64 * i)) atomicOr(flags, 1);
    out[i] = a;
}
// wire format: 16 lanes per proof (14 points + scalar copy), SEC1 compressed -> the 64-byte form
__global__ __launch_bounds__(256) void k_sec1_expand(uint8_t* commitments64, uint8_t* proofs928, const uint8_t* commitments33,
                                                     const uint8_t* proofs525, size_t n) {
    size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t t = g / 16;
    int j = (int)(g % 16);
    if (t < n && j < 15) sec1_expand_lane(commitments64, proofs928, commitments33, proofs525, t, j);
}
// the prover's output in the wire format: 16 lanes per proof, 64-byte form -> SEC1 compressed
__global__ __launch_bounds__(256) void k_sec1_compress(uint8_t* commitments33, uint8_t* proofs525, const uint8_t* commitments64,
                                                       const uint8_t* proofs928, size_t n) {
    size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t t = g / 16;
    int j = (int)(g % 16);
    if (t < n && j < 15) sec1_compress_lane(commitments33, proofs525, commitments64, proofs928, t, j);
}
// pre-loaded transcript variant: each proof's advanced STROBE state back to the caller (203 bytes per proof)
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_export_states(VerifyWs ws) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < ws.N) verify_export_state(ws, t);
}

// ---- shared inversions of the large-batch verify (plan_core.h: shared_inv): lane i inverts for proofs i, i + L, ... (L = ceil(N / G))
#define BPPP_FE_BATCH_INV_KERNEL(G)                                                                                     \
    __global__ __launch_bounds__(BPPP_BLOCK) void k_verify_shared_inv##G(const u32* in, u32* out, size_t N) {                 \
        const size_t i = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;                                                  \
        if (i < (N + G - 1) / G) fe_batch_inv_lane<G>(in, out, N, i);                                                    \
    }
BPPP_FE_BATCH_INV_KERNEL(2)
BPPP_FE_BATCH_INV_KERNEL(4)
BPPP_FE_BATCH_INV_KERNEL(8)
BPPP_FE_BATCH_INV_KERNEL(16)
__global__ __launch_bounds__(BPPP_BLOCK) void k_verify_c0_join(VerifyWs ws) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < ws.N) verify_c0_join(ws, t);
}
