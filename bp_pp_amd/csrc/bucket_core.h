// Bucket (Pippenger) stage of the optional random-linear-combination batch mode -- `util::vector_mul` with T = ProjectivePoint
// (util.rs:46-60) over the ONE sum where the bases are per-proof variables and the sum can be shared across proofs: the weighted
// commitments of the final check (wnla.rs:80-82).
//
// rlc_core.h checks chunks of 8 proofs, paying one 64-doubling scalar multiplication per proof for w_j C4_j.  In the regime a
// verifier normally lives in -- every proof valid -- far larger chunks pass, and for a large chunk the left-hand side
//        sum_j (a_j C4_j + b_j phi(C4_j)),      a_j, b_j 64-bit halves of the weight w_j = a_j + b_j lambda
// is a multi-scalar multiplication over 2 M variable points with SHORT scalars: exactly what bucket accumulation is for.  One
// workgroup takes a superchunk of M proofs (default 4096 -> 8192 items); the 64-bit scalars are cut into eight 8-bit windows; each
// of the four wavefronts owns two windows and, per window,
//   1. sorts the items by digit with a counting sort staged in LDS (histogram with LDS atomics, wave-parallel exclusive scan,
//      scatter of 16-bit item numbers),
//   2. lets lane l sum the buckets d = l, l + 64, l + 128, l + 192 (complete projective additions over the sorted lists, the
//      beta-multiplication of the phi stream folded into the load) while keeping only the lane-local running sums
//      A_l = sum_q S_{l+64q} and B_l = S_{l+64} + 2 S_{l+128} + 3 S_{l+192}, so that sum_d d S_d = sum_l l A_l + 64 sum_l B_l,
//   3. reduces across the wavefront with shuffles: a suffix scan of A (sum_l l A_l = sum_{m>=1} sum_{l>=m} A_l) and a tree sum.
// The eight window sums are combined by Horner (8 w doublings on lane w, 3-step shuffle tree).  Scalars: A_i = sum_j w_j s_ji is
// accumulated UNREDUCED in 12-limb integers per half-weight (64 x 256-bit products, 4096 terms) and reduced mod n once per
// superchunk; one fixed-base MSM over the bases (49 for the u64 protocol, 769 for the generic reciprocal verifier at BASELINE
// configs[4]'s shape: the stage is shared, BucketWs::nb) gives the right-hand side.  A superchunk whose check fails falls through to the chunk-of-8
// kernels (rlc_core.h) and from there to the exact per-proof check, so accept bits stay per proof.  Flagged proofs (status != 0)
// get weight zero and are rejected directly.
#pragma once
#include "rlc_core.h"

namespace bppp {

#define BPPP_BKT_WINDOWS 8       // 8-bit digits of a 64-bit half-weight
#define BPPP_BKT_MAX_M 8192      // items are numbered in 16 bits (2 M <= 65536) and LDS holds 4 waves x (2 KB + 2 M x 2 B) <= 160 KB

struct c4_packed { u32 x[8], y[8], z[8]; };   // C4 in canonical packed words, 96 B (projective: no inversion spent on it)

struct BucketWs {
    size_t N;
    u32 M;                  // proofs per superchunk
    u64 seed[4];
    const int32_t* status;
    const u32* acc;         // [30][N] C4 (projective limbs)
    int nb;                 // bases of the final check (49 for the u64 protocol; 1 + |g_vec| + |h_vec| for the generic verifiers)
    const u32* fsc;         // [nb*8][N] final-check scalars s_ji
    u64* wab;               // [N][2] half-weights a_j, b_j (0, 0 for a flagged proof)
    c4_packed* c4;          // [N]
    u32* lhs;               // [30][nsuper]
    u32* asc;               // [nb*8][nsuper] combined scalars A_i
    uint8_t* sflag;         // [nsuper] 1 = the superchunk's check failed (or could not be made): fall through to chunks of 8
    uint8_t* accept;
    FbTable fb;             // N = nsuper
};

// per proof: half-weights and the packed commitment
HD void bkt_prepare(const BucketWs& w, size_t t) {
    RlcWs r;
    r.seed[0] = w.seed[0]; r.seed[1] = w.seed[1]; r.seed[2] = w.seed[2]; r.seed[3] = w.seed[3];
    u64 a, b;
    rlc_weight(a, b, r, t);
    const bool ok = w.status[t] == ST_OK;
    w.wab[2 * t] = ok ? a : 0;
    w.wab[2 * t + 1] = ok ? b : 0;
    pt C;
    ws_ld_pt(C, w.acc, w.N, t);
    c4_packed k;
    fe_to_w8(k.x, C.X);
    fe_to_w8(k.y, C.Y);
    fe_to_w8(k.z, C.Z);
    w.c4[t] = k;
}
HD void bkt_load_point(pt& P, const c4_packed& k, bool phi, const fe& beta) {
    fe_from_w8(P.X, k.x);
    fe_from_w8(P.Y, k.y);
    fe_from_w8(P.Z, k.z);
    fe bx;
    fe_mul(bx, P.X, beta);
    fe_cmov(P.X, phi, bx);
}
// digit of item `it` (proof = first + it / 2, stream = it & 1) in window w; items past the end of the batch have digit 0
HD u32 bkt_digit(const BucketWs& w, size_t first, u32 it, int win) {
    const size_t j = first + (it >> 1);
    if (j >= w.N) return 0;
    const u64 v = w.wab[2 * j + (it & 1)];
    return (u32)(v >> (8 * win)) & 0xFFu;
}
// 64-bit x 256-bit product accumulated into a 12-limb integer (no reduction: 4096 terms of < 2^320 stay below 2^384)
HD void bkt_mac(u32 acc[12], u64 k, const u32 s[8]) {
    const u32 kk[2] = {(u32)k, (u32)(k >> 32)};
    u32 p[10];
    mul_limbs<8, 2>(p, s, kk);
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) acc[i] = addc(acc[i], i < 10 ? p[i] : 0u, c);
}
// (acc_a + lambda acc_b) mod n from the two unreduced 12-limb sums
HD void bkt_finish_scalar(sc& out, const u32 aa[12], const u32 ab[12]) {
    const sc LAM = {{0x1B23BD72u, 0xDF02967Cu, 0x20816678u, 0x122E22EAu, 0x8812645Au, 0xA5261C02u, 0xC05C30E0u, 0x5363AD4Cu}};
    u32 t[16];
    sc ra, rb;
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = i < 12 ? aa[i] : 0u;
    sc_reduce512(ra, t);
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = i < 12 ? ab[i] : 0u;
    sc_reduce512(rb, t);
    sc_mul(rb, rb, LAM);
    sc_add(out, ra, rb);
}
// ---- single-thread form of one superchunk (host emulation in tests/emul; the device kernels compute the same values with the
// wavefront algorithm described above): left-hand side by plain bucket sums, combined scalars, right-hand side, verdict
HD bool bkt_superchunk_serial(const BucketWs& w, size_t chunk) {
    const size_t first = chunk * w.M;
    fe beta;
    glv_beta(beta);
    pt total;
    pt_set_identity(total);
    for (int win = BPPP_BKT_WINDOWS - 1; win >= 0; win--) {
        for (int d = 0; d < 8; d++) pt_dbl(total, total);
        pt run, sum;
        pt_set_identity(run);
        pt_set_identity(sum);
        for (int d = 255; d >= 1; d--) {               // sum_d d S_d by running sums
            for (u32 it = 0; it < 2 * w.M; it++) {
                if (bkt_digit(w, first, it, win) != (u32)d) continue;
                pt P;
                bkt_load_point(P, w.c4[first + (it >> 1)], (it & 1) != 0, beta);
                pt_add(run, run, P);
            }
            pt_add(sum, sum, run);
        }
        pt_add(total, total, sum);
    }
    const size_t ns = w.fb.N;
    for (int i = 0; i < w.nb; i++) {
        u32 aa[12], ab[12];
        for (int k = 0; k < 12; k++) aa[k] = ab[k] = 0;
        for (size_t j = first; j < first + w.M && j < w.N; j++) {
            u32 s[8];
            ws_ld8(s, w.fsc, w.N, j, i);
            bkt_mac(aa, w.wab[2 * j], s);
            bkt_mac(ab, w.wab[2 * j + 1], s);
        }
        sc A;
        bkt_finish_scalar(A, aa, ab);
        ws_st8(w.asc, ns, chunk, i, A.v);
    }
    FbRanges rg;
    fb_ranges_one(rg, 0, 0, w.nb);
    pt rhs;
    fb_sum_serial(rhs, w.fb, chunk, w.asc, rg);
    return pt_eq(total, rhs);
}

}  // namespace bppp
