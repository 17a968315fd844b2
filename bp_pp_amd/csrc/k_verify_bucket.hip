// u64 verifier, optional RLC batch mode: the bucket (Pippenger) stage over superchunks of M proofs (bucket_core.h) -- LDS-staged
// counting sort per 8-bit window, lane-owned buckets, wavefront shuffle reductions.
// Part of libbppp_hip.so; per-lane work lives in the *_core.h headers, declarations in kernels.h.
#include "kernels.h"

using namespace bppp;

__global__ __launch_bounds__(BPPP_BLOCK) void k_bkt_prepare(BucketWs w) {
    size_t t = (size_t)blockIdx.x * BPPP_BLOCK + threadIdx.x;
    if (t < w.N) bkt_prepare(w, t);
}

__device__ __forceinline__ void pt_shfl_down(pt& o, const pt& a, int delta) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        o.X.v[i] = __shfl_down(a.X.v[i], delta, 64);
        o.Y.v[i] = __shfl_down(a.Y.v[i], delta, 64);
        o.Z.v[i] = __shfl_down(a.Z.v[i], delta, 64);
    }
}
__device__ __forceinline__ void pt_shfl_xor(pt& o, const pt& a, int mask) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        o.X.v[i] = __shfl_xor(a.X.v[i], mask, 64);
        o.Y.v[i] = __shfl_xor(a.Y.v[i], mask, 64);
        o.Z.v[i] = __shfl_xor(a.Z.v[i], mask, 64);
    }
}
// the group law leaves coordinates at magnitudes (5, 2, 2); pt_add wants <= 8 on its inputs, so sums of sums are fine as they are
__device__ __forceinline__ void wave_sum(pt& a) {       // every lane ends with the sum over the 64 lanes
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        pt o;
        pt_shfl_xor(o, a, m);
        pt_add(a, a, o);
    }
}

// One workgroup (4 wavefronts) per superchunk.  Dynamic LDS: per wave 512 words of bucket bookkeeping + M words of sorted item
// numbers (2 M x 16 bit), then 8 x 30 words for the window sums.
__global__ __launch_bounds__(256) void k_bkt_accumulate(BucketWs w) {
    extern __shared__ u32 lds[];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 M = w.M, items = 2 * M;
    const size_t chunk = blockIdx.x, first = chunk * (size_t)M;
    u32* cur = lds + (size_t)wid * (512 + M);      // scatter cursors: start of each bucket, then its end
    u32* beg = cur + 256;                          // start of each bucket
    unsigned short* sorted = (unsigned short*)(beg + 256);
    u32* winsum = lds + (size_t)4 * (512 + M);
    fe beta;
    glv_beta(beta);
#pragma nounroll
    for (int pass = 0; pass < 2; pass++) {
        const int win = wid + 4 * pass;
        for (int k = lane; k < 256; k += 64) cur[k] = 0;
        __syncthreads();
        // 1. histogram of the window's digits (digit 0 contributes nothing and is not sorted)
#pragma nounroll
        for (u32 it = lane; it < items; it += 64) {
            const u32 d = bkt_digit(w, first, it, win);
            if (d) atomicAdd(&cur[d], 1u);
        }
        __syncthreads();
        // 2. exclusive scan over the 256 counts: lane l scans d = 4 l .. 4 l + 3, lane totals combined with shuffles
        {
            const u32 c0 = cur[4 * lane], c1 = cur[4 * lane + 1], c2 = cur[4 * lane + 2], c3 = cur[4 * lane + 3];
            const u32 tot = c0 + c1 + c2 + c3;
            u32 inc = tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u32 o = __shfl_up(inc, off, 64);
                if (lane >= off) inc += o;
            }
            const u32 ex = inc - tot;
            __syncthreads();
            cur[4 * lane] = beg[4 * lane] = ex;
            cur[4 * lane + 1] = beg[4 * lane + 1] = ex + c0;
            cur[4 * lane + 2] = beg[4 * lane + 2] = ex + c0 + c1;
            cur[4 * lane + 3] = beg[4 * lane + 3] = ex + c0 + c1 + c2;
        }
        __syncthreads();
        // 3. scatter the item numbers into their buckets
#pragma nounroll
        for (u32 it = lane; it < items; it += 64) {
            const u32 d = bkt_digit(w, first, it, win);
            if (d) sorted[atomicAdd(&cur[d], 1u)] = (unsigned short)it;
        }
        __syncthreads();
        // 4. lane l owns the buckets l + 64 q; A = sum_q S_q, B = S_1 + 2 S_2 + 3 S_3 by running sums from q = 3 down
        pt run, B;
        pt_set_identity(run);
        pt_set_identity(B);
#pragma nounroll
        for (int q = 3; q >= 0; q--) {
            const int d = lane + 64 * q;
            const u32 b = beg[d], len = d ? cur[d] - b : 0u;
            u32 mx = len;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) { const u32 o = __shfl_xor(mx, m, 64); mx = o > mx ? o : mx; }
            pt S;
            pt_set_identity(S);
#pragma nounroll
            for (u32 k = 0; k < mx; k++) {
                if (k < len) {
                    const u32 it = sorted[b + k];
                    pt P;
                    bkt_load_point(P, w.c4[first + (it >> 1)], (it & 1) != 0, beta);
                    pt_add(S, S, P);
                }
            }
            pt_add(run, run, S);
            if (q) pt_add(B, B, run);       // wave-uniform
        }
        // 5. across the wavefront: T1 = sum_l l A_l (suffix sums of A, lanes 1..63 added up), T2 = sum_l B_l
        pt suf = run;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            pt o, s2;
            pt_shfl_down(o, suf, off);
            pt_add(s2, suf, o);
            pt_cmov(suf, lane + off < 64, s2);
        }
        pt id;
        pt_set_identity(id);
        pt_cmov(suf, lane == 0, id);
        wave_sum(suf);
        wave_sum(B);
#pragma nounroll
        for (int k = 0; k < 6; k++) pt_dbl(B, B);
        pt W;
        pt_add(W, suf, B);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 10; i++) { winsum[win * 30 + i] = W.X.v[i]; winsum[win * 30 + 10 + i] = W.Y.v[i]; winsum[win * 30 + 20 + i] = W.Z.v[i]; }
        }
        __syncthreads();
    }
    // 6. sum_w 2^(8 w) W_w: lane w doubles its window sum 8 w times, then a 3-step tree over the 8 lanes
    if (wid == 0) {
        pt W;
        pt_set_identity(W);
        if (lane < BPPP_BKT_WINDOWS) {
#pragma unroll
            for (int i = 0; i < 10; i++) { W.X.v[i] = winsum[lane * 30 + i]; W.Y.v[i] = winsum[lane * 30 + 10 + i]; W.Z.v[i] = winsum[lane * 30 + 20 + i]; }
        }
#pragma nounroll
        for (int k = 0; k < 8 * (BPPP_BKT_WINDOWS - 1); k++) {
            pt D;
            pt_dbl(D, W);
            pt_cmov(W, k < 8 * lane, D);
        }
#pragma unroll
        for (int m = 1; m < BPPP_BKT_WINDOWS; m <<= 1) {
            pt o;
            pt_shfl_xor(o, W, m);
            pt_add(W, W, o);
        }
        if (lane == 0) ws_st_pt(w.lhs, w.fb.N, chunk, W);
    }
}

// combined scalars A_i = sum_j w_j s_ji of one superchunk (blockIdx.x) for a group of BPPP_BKT_SCALAR_GROUP bases (blockIdx.y):
// thread t takes proofs first + t + 256 r; unreduced 12-limb sums per half-weight, shuffle tree inside a wavefront, LDS across the
// four, one reduction mod n per i.  The base groups give the generic verifiers' 769 bases a grid that fills the chip even when there
// are only a few superchunks.
__global__ __launch_bounds__(256) void k_bkt_scalars(BucketWs w) {
    __shared__ u32 part[4][24];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t chunk = blockIdx.x, first = chunk * (size_t)w.M;
    const int i0 = (int)blockIdx.y * BPPP_BKT_SCALAR_GROUP, i1 = i0 + BPPP_BKT_SCALAR_GROUP < w.nb ? i0 + BPPP_BKT_SCALAR_GROUP : w.nb;
#pragma nounroll
    for (int i = i0; i < i1; i++) {
        u32 aa[12], ab[12];
#pragma unroll
        for (int k = 0; k < 12; k++) aa[k] = ab[k] = 0;
#pragma nounroll
        for (size_t j = first + threadIdx.x; j < first + w.M && j < w.N; j += 256) {
            u32 s[8];
            ws_ld8(s, w.fsc, w.N, j, i);
            bkt_mac(aa, w.wab[2 * j], s);
            bkt_mac(ab, w.wab[2 * j + 1], s);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            u32 c = 0, c2 = 0;
#pragma unroll
            for (int k = 0; k < 12; k++) aa[k] = addc(aa[k], __shfl_down(aa[k], off, 64), c);
#pragma unroll
            for (int k = 0; k < 12; k++) ab[k] = addc(ab[k], __shfl_down(ab[k], off, 64), c2);
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 12; k++) { part[wid][k] = aa[k]; part[wid][12 + k] = ab[k]; }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma nounroll
            for (int q = 1; q < 4; q++) {
                u32 c = 0, c2 = 0;
#pragma unroll
                for (int k = 0; k < 12; k++) aa[k] = addc(aa[k], part[q][k], c);
#pragma unroll
                for (int k = 0; k < 12; k++) ab[k] = addc(ab[k], part[q][12 + k], c2);
            }
            sc A;
            bkt_finish_scalar(A, aa, ab);
            ws_st8(w.asc, w.fb.N, chunk, i, A.v);
        }
        __syncthreads();
    }
}

// right-hand side (one fixed-base MSM over the whole wavefront) and the verdict of one superchunk; a passing superchunk
// accepts every proof of it that carries no status flag
__global__ __launch_bounds__(64) void k_bkt_check(BucketWs w) {
    const int lane = (int)threadIdx.x;
    const size_t chunk = blockIdx.x, first = chunk * (size_t)w.M;
    FbRanges rg;
    fb_ranges_one(rg, 0, 0, w.nb);
    pt rhs, lhs;
    fb_group_sum<64>(rhs, w.fb, chunk, lane, w.asc, rg);
    ws_ld_pt(lhs, w.lhs, w.fb.N, chunk);
    const bool ok = pt_eq(lhs, rhs);
    if (lane == 0) w.sflag[chunk] = ok ? 0 : 1;
    if (ok) {
#pragma nounroll
        for (size_t j = first + lane; j < first + w.M && j < w.N; j += 64) w.accept[j] = w.status[j] == ST_OK ? 1 : 0;
    }
}
