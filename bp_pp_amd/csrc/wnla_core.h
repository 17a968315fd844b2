// Generic batched `WeightNormLinearArgument::{commit, verify}` (wnla.rs:66-121) for arbitrary generator-vector sizes:
// the crate's `wnla` API surface (tests.rs:139-171 uses N = 4; an "aggregated" reciprocal proof with dim_nd = 256 uses
// |h_vec| = 512, |g_vec| = 256, 8 rounds).  n independent instances share one generator set (g, g_vec[ng], h_vec[nh]) and
// one shape (rounds = proof.x.len() = proof.r.len(), nl = proof.l.len(), nn = proof.n.len()); c, rho, mu, the commitment and
// the proof are per instance.
//
// Same restructuring as the u64 path: the per-round generator folds (wnla.rs:96-97) are never executed -- after k rounds,
// original generator i sits in folded slot i >> k with coefficient  ch(i) = prod_{t<k, bit t of i} y_{t+1}  (h_vec, c) or
// cg(i) = prod_{t<k} (bit t of i ? y_{t+1} : rho_{t+1})  (g_vec), for ANY lengths (reduce() = even/odd split, util.rs:7-22,
// and the zero-extension of vector_add, util.rs:69-76, make the odd-length cases fall out of the same bit rule).  Only
// com_ = com + y X + (y^2 - 1) R runs per round (the next challenge hashes it); the base case (wnla.rs:80-82) is one
// (1 + ng + nh)-term fixed-base MSM.  Length quirks are reproduced: extra entries of proof.l / proof.n beyond the folded
// generator vectors multiply identities (util.rs:24-26) but extra n entries still enter |n|^2_mu (wnla.rs:67).
#pragma once
#include "verify_core.h"

namespace bppp {

struct WnlaWs {
    size_t N;
    int ng, nh, rounds, nl, nn;
    const uint8_t *commitments, *c, *rho, *mu, *proof_r, *proof_x, *proof_l, *proof_n;   // C-ABI layouts (device memory)
    size_t stride_r, stride_x, stride_l, stride_n;   // bytes between instances (dense arrays: rounds*64, rounds*64, nl*32, nn*32)
    int transcript_preloaded;                        // 1: tstate already holds each instance's transcript (circuit stage ran before)
    uint8_t* accept;
    uint8_t* out_points;   // commit: n x 64
    int32_t* status;
    u32* tstate;           // [52][N]
    u32* acc;              // [30][N]
    u32* pfix;             // [30][N]
    u32* ys;               // [rounds*8][N]
    u32* tab;              // [2 * 2^rounds * 8][N]: ch table then cg table
    u32* msc;              // [(1+ng+nh)*8][N]
    pt_slot* straus;       // [N][2][9]
    // fast variable-base path for the rounds (null: the generic projective tables above are used): affine window tables of the
    // 2 x rounds round points of every instance, built by one kernel with four batched inversions (verify_core.h:
    // affine_tables_build), then Jacobian accumulators with mixed additions over signed 5-bit windows (straus_affine)
    apt_packed* atab;      // [2 * rounds * 16][N] 64-byte entries: point 2 (k - 1) = X of round k, 2 (k - 1) + 1 = R of round k
                           // (tab_parts > 1: that many such sets one after the other, set h over 2^(5 split_begin(tab_parts, h)) times the points)
    int tab_parts;         // 1, or 2 / 4 in calls so small that a round is walked by 8 / 16 lanes per instance (wnla_verify_table_one)
    u32* tscr;             // [BPPP_TSCR_PER_POINT * 2 * rounds * 10][N] running products of the table build
    u32* rpts;             // [2 * rounds * 16][N] the decoded round points (packed affine words)
    FbTable fb;
    strobe base;
    TranscriptIo tio;      // caller's transcripts (wnla.rs:75 `t: &mut Transcript`); the input side is ignored when transcript_preloaded
    int divergent_positions;   // 1: instances of one wavefront may sit at different sponge positions (per-instance input states)
};
HD size_t wnla_ceil_shift(size_t n, int k) { return (n + (((size_t)1 << k) - 1)) >> k; }

// ---- WeightNormLinearArgument::commit: scalars of  v g + <h_vec, l> + <g_vec, n>,  v = <c, l> + sum n_j^2 mu^(j+1)   (wnla.rs:66-72)
HD void wnla_commit_scalars(const WnlaWs& w, size_t t) {
    const size_t N = w.N;
    int32_t status = ST_OK;
    sc mu, v, mp, t1, zero;
    sc_set_u32(zero, 0);
    sc_set_u32(v, 0);
    bool ok = sc_from_be(mu, w.mu + 32 * t);
    mp = mu;
#pragma nounroll
    for (int j = 0; j < w.nn; j++) {
        sc nj;
        ok &= sc_from_be(nj, w.proof_n + (size_t)t * w.stride_n + (size_t)j * 32);
        sc_mul(t1, nj, nj); sc_mul(t1, t1, mp); sc_add(v, v, t1);
        sc_mul(mp, mp, mu);
        if (j < w.ng) ws_st8(w.msc, N, t, 1 + j, nj.v);
    }
#pragma nounroll
    for (int j = w.nn; j < w.ng; j++) ws_st8(w.msc, N, t, 1 + j, zero.v);
#pragma nounroll
    for (int i = 0; i < w.nl; i++) {
        sc li;
        ok &= sc_from_be(li, w.proof_l + (size_t)t * w.stride_l + (size_t)i * 32);
        if (i < w.nh) {
            sc ci;
            ok &= sc_from_be(ci, w.c + ((size_t)t * w.nh + i) * 32);
            sc_mul(t1, ci, li); sc_add(v, v, t1);
            ws_st8(w.msc, N, t, 1 + w.ng + i, li.v);
        }
    }
#pragma nounroll
    for (int i = w.nl; i < w.nh; i++) ws_st8(w.msc, N, t, 1 + w.ng + i, zero.v);
    if (!ok) { status |= ST_BAD_ENCODING; sc_set_u32(v, 0); }
    ws_st8(w.msc, N, t, 0, v.v);
    w.status[t] = status;
}
HD void wnla_commit_store(const WnlaWs& w, size_t t, const pt& total) {
    apt a;
    pt_to_affine(a, total);
    if (w.status[t] != ST_OK) { fe_set_u32(a.x, 0); fe_set_u32(a.y, 0); }
    apt_to_xy64(w.out_points + 64 * t, a);
}

// ---- verify: decode the commitment, start the transcript
HD void wnla_verify_begin(const WnlaWs& w, size_t t) {
    int32_t status = ST_OK;
    apt C;
    bool ok = apt_from_xy64(C, w.commitments + 64 * t);
    sc s;
    ok &= sc_from_be(s, w.rho + 32 * t);
    ok &= sc_from_be(s, w.mu + 32 * t);
    if (!ok) { status |= ST_BAD_ENCODING; fe_set_u32(C.x, 0); fe_set_u32(C.y, 0); }
    pt P;
    pt_from_affine(P, C);
    ws_st_pt(w.acc, w.N, t, P);
    if (!w.transcript_preloaded) {
        strobe tr;
        tio_begin(tr, status, w.tio, w.base, t);
        ws_st_strobe(w.tstate, w.N, t, tr);
    }
    w.status[t] = w.transcript_preloaded ? (w.status[t] | status) : status;
}
// ---- verify, fast path: decode the round points of all rounds and build their window tables.  A round whose X or R does not
// decode gets the identity for both (exactly what wnla_verify_round substitutes when it flags the instance).
HD void wnla_verify_tables(const WnlaWs& w, size_t t) {
    const size_t N = w.N;
#pragma nounroll
    for (int k = 1; k <= w.rounds; k++) {
        apt X, R;
        bool ok = apt_from_xy64(X, w.proof_x + (size_t)t * w.stride_x + (size_t)(w.rounds - k) * 64);
        ok &= apt_from_xy64(R, w.proof_r + (size_t)t * w.stride_r + (size_t)(w.rounds - k) * 64);
        if (!ok) { fe_set_u32(X.x, 0); fe_set_u32(X.y, 0); R = X; }
        ws_st_apt(w.rpts, N, t, 2 * (k - 1), X);
        ws_st_apt(w.rpts, N, t, 2 * (k - 1) + 1, R);
    }
    affine_tables_build(atab_of(w.atab, N, t, 2 * w.rounds * 16), w.tscr, w.rpts, N, t, 2 * w.rounds);
}
// ---- verify: round k = 1..rounds (wnla.rs:84-102), X = proof.x[rounds - k], R = proof.r[rounds - k]
// group_lane >= 0: one of group_size (2 or 4) consecutive lanes that all run the round for instance t and share its sum
// (straus_core.h: straus_affine_g4) -- batches that under-fill the chip with one lane per instance; -1: one lane per instance
// ... the same for calls that leave the chip EMPTY (a handful of instances: what such a call takes is the length of one instance's chain):
// a lane per (point, part) -- point p's table for part h of the 26 windows is the table of 2^(5 split_begin(parts, h)) P (straus_core.h:
// affine_table_one), so that a round's two-point sum can be walked by 4 x parts lanes (straus_affine_split), 13 or 7 windows each
HD void wnla_verify_table_one(const WnlaWs& w, size_t t, int p, int h, int parts) {
    const int k = p / 2 + 1;                              // point 2 (k - 1) = X of round k, 2 (k - 1) + 1 = R of round k
    apt X, R;
    bool ok = apt_from_xy64(X, w.proof_x + (size_t)t * w.stride_x + (size_t)(w.rounds - k) * 64);
    ok &= apt_from_xy64(R, w.proof_r + (size_t)t * w.stride_r + (size_t)(w.rounds - k) * 64);
    if (!ok) { fe_set_u32(X.x, 0); fe_set_u32(X.y, 0); R = X; }
    affine_table_one(atab_of(w.atab, w.N, t) + (h * 2 * w.rounds + p) * 16, (p & 1) ? R : X, 5 * split_begin(parts, h));
}
HD void wnla_verify_round(const WnlaWs& w, size_t t, int k, int group_lane = -1, int group_size = 4) {
    const size_t N = w.N;
    int32_t status = w.status[t];
    pt C;
    ws_ld_pt(C, w.acc, N, t);
    apt Ca, X, R;
    pt_to_affine(Ca, C);
    bool ok = apt_from_xy64(X, w.proof_x + (size_t)t * w.stride_x + (size_t)(w.rounds - k) * 64);
    ok &= apt_from_xy64(R, w.proof_r + (size_t)t * w.stride_r + (size_t)(w.rounds - k) * 64);
    if (!ok) { status |= ST_BAD_ENCODING; fe_set_u32(X.x, 0); fe_set_u32(X.y, 0); R = X; }
    strobe tr;
    ws_ld_strobe(tr, w.tstate, N, t);
    app_point(tr, "wnla_com", Ca);
    app_point(tr, "wnla_x", X);
    app_point(tr, "wnla_r", R);
    t_append_u64(tr, "l.sz", (u64)wnla_ceil_shift((size_t)w.nh, k - 1));     // self.h_vec.len() of this level
    t_append_u64(tr, "n.sz", (u64)wnla_ceil_shift((size_t)w.ng, k - 1));
    sc y;
    if (!t_get_challenge(tr, "wnla_challenge", y)) { status |= ST_DEGENERATE; sc_set_u32(y, 1); }
    ws_st_strobe(w.tstate, N, t, tr);
    ws_st8(w.ys, N, t, k - 1, y.v);
    sc y2m1, one;
    sc_set_u32(one, 1);
    sc_mul(y2m1, y, y);
    sc_sub(y2m1, y2m1, one);
    pt acc;
    if (w.atab) {
        const int pslot[2] = {2 * (k - 1), 2 * (k - 1) + 1};
        glv_words<2> g;
        glv_split sp;
        glv_decompose(sp, y);
        glv_words_set<2>(g, 0, sp);
        glv_decompose(sp, y2m1);
        glv_words_set<2>(g, 1, sp);
#if defined(__HIP_DEVICE_COMPILE__)
        if (group_lane >= 0 && group_size == 16) straus_affine_split<2, 16, 4>(acc, atab_of(w.atab, N, t), pslot, g, group_lane, 2 * w.rounds);
        else if (group_lane >= 0 && group_size == 8) straus_affine_split<2, 8, 2>(acc, atab_of(w.atab, N, t), pslot, g, group_lane, 2 * w.rounds);
        else if (group_lane >= 0 && group_size == 4) straus_affine_g4<2, 4>(acc, atab_of(w.atab, N, t, 2 * w.rounds * 16), pslot, g, group_lane);
        else if (group_lane >= 0) straus_affine_g4<2, 2>(acc, atab_of(w.atab, N, t, 2 * w.rounds * 16), pslot, g, group_lane);
        else
#endif
            straus_affine<2>(acc, atab_of(w.atab, N, t, 2 * w.rounds * 16), pslot, g);
        (void)group_lane; (void)group_size;
    } else {
        pt_slot* tbl = w.straus + t * (2 * BPPP_STRAUS_ENTRIES);
        glv_split rs[2];
        straus_build_table(tbl, X);
        straus_build_table(tbl + BPPP_STRAUS_ENTRIES, R);
        glv_decompose(rs[0], y);
        glv_decompose(rs[1], y2m1);
        straus_msm_glv(acc, tbl, rs, 2);
    }
    pt_madd(acc, acc, Ca, apt_is_identity(Ca));
    ws_st_pt(w.acc, N, t, acc);
    w.status[t] = status;
}
// ---- verify: base case scalars (wnla.rs:80-82 with :66-72), generators unrolled
// One of 2^lg parts of instance t (lg = 0: the whole instance).  Part q owns the coefficient-table indices whose top lg bits are q
// -- it starts its subtree at the product of those bits' factors, so no part reads what another wrote -- and with them every
// generator i whose low `rounds` bits fall in its range; its share of v is left in the table for wnla_final_scalars_join
// (arithmetic mod n: the order of the additions and of the factors does not show in the result).
HD void wnla_final_scalars_part(const WnlaWs& w, size_t t, int q, int lg) {
    const size_t N = w.N;
    const int k = w.rounds;
    const int T = 1 << k, kl = k - lg, Tl = 1 << kl, B = q << kl;
    sc rho, mu, one, zero, t1;
    sc_set_u32(one, 1);
    sc_set_u32(zero, 0);
    bool ok = sc_from_be(rho, w.rho + 32 * t);
    ok &= sc_from_be(mu, w.mu + 32 * t);
    // coefficient tables over the low k index bits: ch[b] = prod_{bit t of b} y_{t+1}; cg[b] = prod_t (bit ? y_{t+1} : rho_{t+1})
    sc pch = one, pcg = one;
    sc rt = rho, mt = mu;    // rho_{t+1}, mu_{t+1}
    if (lg > 0) {
#pragma nounroll
        for (int r = 0; r < k; r++) {
            if (r >= kl) {
                sc y;
                ws_ld8(y.v, w.ys, N, t, r);
                const bool bit = (q >> (r - kl)) & 1;
                sc_mul(t1, pch, y);
                if (bit) pch = t1;
                sc f = rt;
                if (bit) f = y;
                sc_mul(pcg, pcg, f);
            }
            rt = mt;
            sc_mul(mt, mt, mt);
        }
        rt = rho; mt = mu;
    }
    ws_st8(w.tab, N, t, B, pch.v);
    ws_st8(w.tab, N, t, T + B, pcg.v);
#pragma nounroll
    for (int r = 0; r < k; r++) {
        if (r < kl) {
            sc y;
            ws_ld8(y.v, w.ys, N, t, r);
            const int half = 1 << r;
#pragma nounroll
            for (int b = B; b < B + half; b++) {
                sc ch, cg;
                ws_ld8(ch.v, w.tab, N, t, b);
                ws_ld8(cg.v, w.tab, N, t, T + b);
                sc_mul(t1, ch, y);
                ws_st8(w.tab, N, t, b + half, t1.v);
                sc_mul(t1, cg, y);
                ws_st8(w.tab, N, t, T + b + half, t1.v);
                sc_mul(t1, cg, rt);
                ws_st8(w.tab, N, t, T + b, t1.v);
            }
        }
        rt = mt;                     // wnla.rs:109-110: rho <- mu, mu <- mu^2
        sc_mul(mt, mt, mt);
    }
    const sc mu_fin = mt;
    // v = <c', l> + sum_j n_j^2 mu_fin^(j+1);  c'_j = sum_{i >> k == j} c_i ch(i)
    sc v;
    sc_set_u32(v, 0);
    if (q == 0) {
        sc mp = mu_fin;
#pragma nounroll
        for (int j = 0; j < w.nn; j++) {
            sc nj;
            ok &= sc_from_be(nj, w.proof_n + (size_t)t * w.stride_n + (size_t)j * 32);
            sc_mul(t1, nj, nj); sc_mul(t1, t1, mp); sc_add(v, v, t1);
            sc_mul(mp, mp, mu_fin);
        }
        // proof.l entries are validated even when they multiply nothing
#pragma nounroll
        for (int j = 0; j < w.nl; j++) { sc lj; ok &= sc_from_be(lj, w.proof_l + (size_t)t * w.stride_l + (size_t)j * 32); }
    }
    const int nnf = (int)wnla_ceil_shift((size_t)w.ng, k);
#pragma nounroll
    for (int j = 0; (j << k) + B < w.nh; j++) {
        sc lj = zero;
        if (j < w.nl) ok &= sc_from_be(lj, w.proof_l + (size_t)t * w.stride_l + (size_t)j * 32);
#pragma nounroll
        for (int b = B; b < B + Tl; b++) {
            const int i = (j << k) + b;
            if (i >= w.nh) break;
            sc ci, ch, coef;
            ok &= sc_from_be(ci, w.c + ((size_t)t * w.nh + i) * 32);
            ws_ld8(ch.v, w.tab, N, t, b);
            sc_mul(coef, lj, ch);                       // scalar of h_i
            ws_st8(w.msc, N, t, 1 + w.ng + i, coef.v);
            sc_mul(t1, ci, coef);                       // c_i ch(i) l_j
            sc_add(v, v, t1);
        }
    }
#pragma nounroll
    for (int j = 0; (j << k) + B < w.ng; j++) {
        sc nj = zero;
        if (j < w.nn && j < nnf) ok &= sc_from_be(nj, w.proof_n + (size_t)t * w.stride_n + (size_t)j * 32);
#pragma nounroll
        for (int b = B; b < B + Tl; b++) {
            const int i = (j << k) + b;
            if (i >= w.ng) break;
            sc cg, coef;
            ws_ld8(cg.v, w.tab, N, t, T + b);
            sc_mul(coef, nj, cg);
            ws_st8(w.msc, N, t, 1 + i, coef.v);
        }
    }
    // the lane's share of v goes where its part of the ch table began (read for the last time above); the flag by OR
    ws_st8(w.tab, N, t, B, v.v);
    if (!ok) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr((int*)&w.status[t], (int)ST_BAD_ENCODING);
#else
        w.status[t] |= ST_BAD_ENCODING;
#endif
    }
}
HD void wnla_final_scalars_join(const WnlaWs& w, size_t t, int lg) {
    const int kl = w.rounds - lg;
    sc v, o;
    ws_ld8(v.v, w.tab, w.N, t, 0);
#pragma nounroll
    for (int q = 1; q < (1 << lg); q++) {
        ws_ld8(o.v, w.tab, w.N, t, q << kl);
        sc_add(v, v, o);
    }
    ws_st8(w.msc, w.N, t, 0, v.v);
}
HD void wnla_verify_final_scalars(const WnlaWs& w, size_t t) {
    wnla_final_scalars_part(w, t, 0, 0);
    wnla_final_scalars_join(w, t, 0);
}
// the lanes one instance's final scalars are dealt to (a power of two <= 8 that the table has room for; 1 = the whole instance)
HD int wnla_final_scalars_lg(int rounds, int want_lg) { return want_lg < rounds ? want_lg : (rounds > 0 ? rounds - 1 : 0); }
HD void wnla_msm_ranges(FbRanges& rg, const WnlaWs& w) { fb_ranges_one(rg, 0, 0, 1 + w.ng + w.nh); }
HD void wnla_verify_store(const WnlaWs& w, size_t t, const pt& rhs) { ws_st_pt(w.pfix, w.N, t, rhs); }
HD void wnla_verify_accept(const WnlaWs& w, size_t t) {
    pt C, rhs;
    ws_ld_pt(C, w.acc, w.N, t);
    ws_ld_pt(rhs, w.pfix, w.N, t);
    bool eq = pt_eq(C, rhs);
    w.accept[t] = (eq && w.status[t] == ST_OK) ? 1 : 0;
}

}  // namespace bppp
