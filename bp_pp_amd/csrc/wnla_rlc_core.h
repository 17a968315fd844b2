// The random-linear-combination batch mode (rlc_core.h) for the GENERIC verifiers: every generic protocol ends in the WNLA stage,
// whose last step is one fixed-base MSM over all 1 + |g_vec| + |h_vec| generators per instance (wnla.rs:80-82 after the
// recursion) -- 769 bases and 12 304 table additions for BASELINE configs[4]'s shape, half of that verifier's time.  Instances are
// grouped in chunks of 8 (one lane group); a chunk is checked as
//        sum_j w_j C_j  ==  sum_i (sum_j w_j s_ji) B_i,        w_j = a_j + b_j lambda   (a_j, b_j secret 64-bit values from the seed)
// i.e. ONE MSM per chunk and one 64-doubling multiplication per instance.  Chunks that fail, are incomplete or hold a flagged
// instance are re-checked by the exact kernels, so accept bits stay per instance (identical to exact mode except with
// probability <= 2^-128 per invalid chunk).  Everything before the final MSM is the exact pipeline, unchanged.
#pragma once
#include "rlc_core.h"
#include "wnla_core.h"

namespace bppp {

// L_j = a_j C_j + b_j phi(C_j)  (C_j = the instance's commitment after the last round, w.acc)
HD void wnla_rlc_lhs(const WnlaWs& w, const RlcWs& r, size_t t) {
    const size_t N = w.N;
    u64 a, b;
    rlc_weight(a, b, r, t);
    pt C, L;
    ws_ld_pt(C, w.acc, N, t);
    rlc_weighted_point(L, C, w.straus + t * (2 * BPPP_STRAUS_ENTRIES), a, b);
    ws_st_pt(r.lhs, N, t, L);
}
// single-thread form of the chunk check (host emulation; the device kernel does the same with wavefront shuffles)
HD bool wnla_rlc_chunk_serial(const WnlaWs& w, const RlcWs& r, size_t chunk) {
    const size_t N = w.N, first = chunk * BPPP_RLC_CHUNK;
    const int NB = 1 + w.ng + w.nh;
    bool usable = first + BPPP_RLC_CHUNK <= N;
    for (size_t j = first; usable && j < first + BPPP_RLC_CHUNK; j++) usable &= w.status[j] == ST_OK;
    if (!usable) return false;
    sc wv[BPPP_RLC_CHUNK];
    for (int l = 0; l < BPPP_RLC_CHUNK; l++) {
        u64 a, b;
        rlc_weight(a, b, r, first + l);
        rlc_weight_scalar(wv[l], a, b);
    }
    for (int i = 0; i < NB; i++) {
        sc A, s, p;
        sc_set_u32(A, 0);
        for (int l = 0; l < BPPP_RLC_CHUNK; l++) {
            ws_ld8(s.v, w.msc, N, first + l, i);
            sc_mul(p, s, wv[l]);
            sc_add(A, A, p);
        }
        for (int l = 0; l < BPPP_RLC_CHUNK; l++) ws_st8(r.sc, N, first + l, i, A.v);
    }
    FbRanges rg;
    wnla_msm_ranges(rg, w);
    pt rhs, lhs, L;
    fb_sum_serial(rhs, w.fb, first, r.sc, rg);
    pt_set_identity(lhs);
    for (int l = 0; l < BPPP_RLC_CHUNK; l++) { ws_ld_pt(L, r.lhs, N, first + l); pt_add(lhs, lhs, L); }
    return pt_eq(lhs, rhs);
}

}  // namespace bppp
