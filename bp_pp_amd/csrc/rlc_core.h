// Optional batch mode of the u64 verifier: random linear combination of the FINAL checks (the "fold many verifications into
// one multi-scalar multiplication" idea of BASELINE.json's north star, applied where it is sound and pays).
//
// Exact mode ends every proof with its own 49-base fixed-base MSM:  C4_j == sum_i s_ji B_i  (wnla.rs:80-82 unrolled), 637
// table additions, ~26 % of the whole pipeline.  Everything before that is forced per proof by the Fiat-Shamir chain and stays
// exactly as in exact mode.  Here proofs are grouped in chunks of 8 (one lane group of the fixed-base kernels) and a chunk is
// checked as
//        sum_j w_j C4_j  ==  sum_i (sum_j w_j s_ji) B_i,        w_j = a_j + b_j lambda  (a_j, b_j secret 64-bit values),
// i.e. ONE 49-base MSM per chunk plus one short (64 doublings) variable-base multiplication per proof: w_j C4_j =
// a_j C4_j + b_j phi(C4_j) with the GLV endomorphism phi(X : Y : Z) = (beta X : Y : Z).  The weights come from a keyed PRF
// (one Keccak-f[1600] of seed || index) of a caller-supplied 32-byte seed that must be chosen AFTER the proofs are fixed; a
// chunk containing an invalid proof then passes with probability <= 2^-128.  A chunk whose combined check fails -- or that
// contains a flagged or missing proof -- is re-checked with the exact per-proof kernels, so the caller still gets a per-proof
// accept bit, identical to exact mode except with that probability.
#pragma once
#include "verify_core.h"

namespace bppp {

#define BPPP_RLC_CHUNK BPPP_FB_LANES      // 8 proofs per chunk = one lane group

struct RlcWs {
    u64 seed[4];
    u32* lhs;          // [30][N]  w_j C4_j (projective limbs)
    u32* sc;           // [49*8][N] chunk scalars A_i = sum_j w_j s_ji, one identical copy per proof of the chunk
    uint8_t* flag;     // [ceil(N / 8)]  1 = chunk must be re-checked exactly
    u32* list;         // [ceil(N / 8)]  the flagged chunks, compacted (device: appended with an atomic counter)
    int* count;        // number of entries in `list`
    // bucket stage in front (bucket_core.h): superchunks of `super_m` proofs whose combined check PASSED (sflag = 0) are done --
    // their proofs are skipped here; null = no bucket stage, every chunk of 8 is checked
    const uint8_t* sflag;
    u32 super_m;
    // u64 verifier: proofs per chunk of THIS call, 8 or 32 (0 reads as 8; bppp_u64.hip picks it from the previous call's reject rate:
    // a chunk of 32 spreads the right-hand side's 49-base sum over four times the proofs, and fails four times as often)
    u32 chunk;
};
#define BPPP_RLC_CHUNK_MAX 32
HD u32 rlc_chunk_of(const RlcWs& r) { return r.chunk ? r.chunk : (u32)BPPP_RLC_CHUNK; }
HD bool rlc_done_by_bucket_stage(const RlcWs& r, size_t t) { return r.sflag && !r.sflag[t / r.super_m]; }

// weight halves (a, b): one permutation of  seed[0..3] | index | domain tag | 0...
HD void rlc_weight(u64& a, u64& b, const RlcWs& r, size_t t) {
    u64 st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
    st[0] = r.seed[0]; st[1] = r.seed[1]; st[2] = r.seed[2]; st[3] = r.seed[3];
    st[4] = (u64)t;
    st[5] = 0x434C525F50505042ULL;      // "BPPP_RLC"
    keccak_f1600(st);
    a = st[0];
    b = st[1];
}
// w = a + b lambda mod n
HD void rlc_weight_scalar(sc& w, u64 a, u64 b) {
    const u32 LAMBDA_W[8] = {0x1B23BD72u, 0xDF02967Cu, 0x20816678u, 0x122E22EAu, 0x8812645Au, 0xA5261C02u, 0xC05C30E0u, 0x5363AD4Cu};
    sc lam, bs, as;
#pragma unroll
    for (int i = 0; i < 8; i++) { lam.v[i] = LAMBDA_W[i]; bs.v[i] = 0; as.v[i] = 0; }
    bs.v[0] = (u32)b; bs.v[1] = (u32)(b >> 32);
    as.v[0] = (u32)a; as.v[1] = (u32)(a >> 32);
    sc_mul(w, bs, lam);
    sc_add(w, w, as);
}
// out = a C + b phi(C): projective window table of C in tbl[0..8] (complete formulas), 17 signed 4-bit windows, 64 doublings
HD void rlc_weighted_point(pt& out, pt C, pt_slot* tbl, u64 a, u64 b) {
    {   // tbl[e] = e C, e = 0..8
        pt cur;
        pt_set_identity(cur);
        tbl[0].p = cur;
        pt_normalize(C);
        tbl[1].p = C;
        cur = C;
#pragma nounroll
        for (int e = 2; e <= 8; e++) {
            pt d;
            if (e == 2) pt_dbl(d, cur);
            else pt_add(d, cur, C);
            pt_normalize(d);
            tbl[e].p = d;
            cur = d;
        }
    }
    // signed recoding of the two 64-bit halves: k' = k + 0x8...8 (17 nibbles), digit_i = nib_i(k') - 8
    u32 kp[2][3];
    {
        const u64 OFF = 0x8888888888888888ULL;
        const u64 a2 = a + OFF, b2 = b + OFF;
        kp[0][0] = (u32)a2; kp[0][1] = (u32)(a2 >> 32); kp[0][2] = 8u + (a2 < a ? 1u : 0u);
        kp[1][0] = (u32)b2; kp[1][1] = (u32)(b2 >> 32); kp[1][2] = 8u + (b2 < b ? 1u : 0u);
    }
    fe BETA;
    glv_beta(BETA);
    pt acc;
    pt_set_identity(acc);
#pragma nounroll
    for (int i = 16; i >= 0; i--) {
        if (i != 16) {
#pragma nounroll
            for (int d = 0; d < 4; d++) pt_dbl(acc, acc);
        }
#pragma nounroll
        for (int h = 0; h < 2; h++) {
            u32 word = 0;
#pragma unroll
            for (int l = 0; l < 3; l++) word = (l == (i >> 3)) ? (h ? kp[1][l] : kp[0][l]) : word;
            const int dg = (int)((word >> ((i & 7) * 4)) & 15) - 8;
            const int mag = dg < 0 ? -dg : dg;
            pt q = tbl[mag].p;
            fe bx, ny;
            fe_mul(bx, q.X, BETA);
            fe_cmov(q.X, h != 0, bx);
            fe_neg_m<1>(ny, q.Y);
            fe_cmov(q.Y, dg < 0, ny);
            pt_add(acc, acc, q);
        }
    }
    out = acc;
}
// L_j = a_j C4_j + b_j phi(C4_j)
HD void rlc_lhs(const VerifyWs& ws, const RlcWs& r, size_t t) {
    const size_t N = ws.N;
    u64 a, b;
    rlc_weight(a, b, r, t);
    pt C, L;
    ws_ld_pt(C, ws.acc, N, t);
    rlc_weighted_point(L, C, ws.straus + t * (5 * BPPP_STRAUS_ENTRIES), a, b);
    ws_st_pt(r.lhs, N, t, L);
}
// scalar side, lane work: P_i = w_j s_ji for proof j (i = 0..48)
HD void rlc_product(sc& out, const VerifyWs& ws, const sc& w, size_t t, int i) {
    sc s;
    ws_ld8(s.v, ws.fsc, ws.N, t, i);
    sc_mul(out, s, w);
}
HD void rlc_ranges(FbRanges& rg) { fb_ranges_one(rg, 0, 0, BPPP_NG); }
// host / single-thread form of the chunk check (the device kernel does the same with wavefront shuffles): returns the verdict
HD bool rlc_chunk_serial(const VerifyWs& ws, const RlcWs& r, size_t chunk) {
    const size_t C = rlc_chunk_of(r);
    const size_t N = ws.N, first = chunk * C;
    bool usable = first + C <= N;
    for (size_t j = first; usable && j < first + C; j++) usable &= ws.status[j] == ST_OK;
    if (!usable) return false;
    sc wv[BPPP_RLC_CHUNK_MAX];
    for (size_t l = 0; l < C; l++) {
        u64 a, b;
        rlc_weight(a, b, r, first + l);
        rlc_weight_scalar(wv[l], a, b);
    }
    for (int i = 0; i < BPPP_NG; i++) {
        sc A, p;
        sc_set_u32(A, 0);
        for (size_t l = 0; l < C; l++) { rlc_product(p, ws, wv[l], first + l, i); sc_add(A, A, p); }
        for (size_t l = 0; l < C; l++) ws_st8(r.sc, N, first + l, i, A.v);
    }
    FbRanges rg;
    rlc_ranges(rg);
    pt rhs, lhs, L;
    fb_sum_serial(rhs, fb_of(ws), first, r.sc, rg);
    pt_set_identity(lhs);
    for (size_t l = 0; l < C; l++) { ws_ld_pt(L, r.lhs, N, first + l); pt_add(lhs, lhs, L); }
    return pt_eq(lhs, rhs);
}

}  // namespace bppp
