// Generic batched `WeightNormLinearArgument::prove` (wnla.rs:125-190) for arbitrary sizes: n independent instances share one
// generator set (g, g_vec[ng], h_vec[nh]) and one shape (|l| = nl, |n| = nn, |c| = nh); c, rho, mu, the commitment and the
// witness vectors l, n are per instance.  The number of rounds and the final vector lengths follow from nl and nn alone
// (the recursion stops when |l| + |n| < 6, wnla.rs:126), so the host drives the round loop.
//
// Same restructuring as the verifier: folded generators are never materialised.  After k rounds original generator i sits
// in folded slot i >> k with coefficient ch[i] (h_vec) or cg[i] (g_vec), kept per instance and updated per round
// (reduce() = even/odd split, util.rs:7-22; vector_add zero-extends, util.rs:69-76).  The round's three group elements -- X, R
// (wnla.rs:147-157) and the next level's commitment (`wnla.commit(&l_, &n_)`, wnla.rs:186) -- are (1 + ng + nh)-term
// fixed-base MSMs over the ORIGINAL generators with scalars  coefficient x (vector entry of the partner / own slot).
#pragma once
#include "prove_core.h"
#include "wnla_core.h"

namespace bppp {

struct WnlaProveWs {
    size_t N;
    int ng, nh, nl, nn, rounds;
    const uint8_t *commitments, *c, *rho, *mu, *l_in, *n_in;     // C-ABI layouts (device memory): n x 64, n x nh x 32, n x 32, ...
    uint8_t *proof_r, *proof_x, *proof_l, *proof_n;              // outputs: n x rounds x 64 (x2), n x nl_f x 32, n x nn_f x 32
    int nl_f, nn_f;
    int transcript_preloaded;                                    // 1: tstate / status already hold each instance's transcript and flags (circuit prover)
    int32_t* status;
    u32* tstate;      // [52][N]
    u32* vl;          // [nl * 8][N]   current l (prefix of length ceil(nl / 2^k))
    u32* vn;          // [nn * 8][N]
    u32* vc;          // [nh * 8][N]   current c
    u32* ch;          // [nh * 8][N]   coefficient of h_i in its folded generator
    u32* cg;          // [ng * 8][N]
    u32* prm;         // [3 * 8][N]    rho_k, mu_k, rho_k^-1
    u32* com;         // [16][N]       current commitment, packed affine (hashed first thing in a round)
    u32* msc;         // [3][(1 + ng + nh) * 8][N]  scalar sets: X, R, next commitment
    u32* pbuf;        // [3][30][N]    X, R, next commitment (projective)
    pt_slot* straus;  // [N][2 * BPPP_STRAUS_ENTRIES]  window tables of X and R for the next commitment by the verifier's relation
    FbTable fb;
    FbTable fb_ct;    // "ct_prover": the 4-bit table the X | R sums scan in full (fb_core.h: fb_lookup_add_ct); used when ct != 0
    int ct;           // set by bppp_wnla_prove_batch when the option is on: l and n are the caller's secrets there (wnla.rs:152-160)
    strobe base;
    TranscriptIo tio;                                            // caller's transcripts (wnla.rs:125 `t: &mut Transcript`); input side ignored when transcript_preloaded
    int divergent_positions;                                     // 1: the instances of one wavefront may sit at different sponge positions
};
HD size_t wp_set_words(const WnlaProveWs& w) { return (size_t)(1 + w.ng + w.nh) * 8 * w.N; }
HD void wp_ld(sc& r, const u32* base, const WnlaProveWs& w, size_t t, int idx, int len) {   // zero-extended vector read
    if (idx < len) ws_ld8(r.v, base, w.N, t, idx);
    else sc_set_u32(r, 0);
}

HD void wnla_prove_init(const WnlaProveWs& w, size_t t) {
    const size_t N = w.N;
    int32_t status = ST_OK;
    apt C;
    bool ok = apt_from_xy64(C, w.commitments + 64 * t);
    if (!ok) { fe_set_u32(C.x, 0); fe_set_u32(C.y, 0); }
    ws_st_apt(w.com, N, t, 0, C);
    sc rho, mu, one, zero, x;
    sc_set_u32(one, 1);
    sc_set_u32(zero, 0);
    ok &= sc_from_be(rho, w.rho + 32 * t);
    ok &= sc_from_be(mu, w.mu + 32 * t);
#pragma nounroll
    for (int i = 0; i < w.nh; i++) {
        bool k = sc_from_be(x, w.c + ((size_t)t * w.nh + i) * 32);
        ok &= k;
        ws_st8(w.vc, N, t, i, k ? x.v : zero.v);
        ws_st8(w.ch, N, t, i, one.v);
    }
#pragma nounroll
    for (int i = 0; i < w.ng; i++) ws_st8(w.cg, N, t, i, one.v);
#pragma nounroll
    for (int i = 0; i < w.nl; i++) {
        bool k = sc_from_be(x, w.l_in + ((size_t)t * w.nl + i) * 32);
        ok &= k;
        ws_st8(w.vl, N, t, i, k ? x.v : zero.v);
    }
#pragma nounroll
    for (int i = 0; i < w.nn; i++) {
        bool k = sc_from_be(x, w.n_in + ((size_t)t * w.nn + i) * 32);
        ok &= k;
        ws_st8(w.vn, N, t, i, k ? x.v : zero.v);
    }
    if (!ok) { status |= ST_BAD_ENCODING; rho = one; mu = one; }
    ws_st8(w.prm, N, t, 0, rho.v);
    ws_st8(w.prm, N, t, 1, mu.v);
    if (!w.transcript_preloaded) {
        strobe tr;
        tio_begin(tr, status, w.tio, w.base, t);
        ws_st_strobe(w.tstate, N, t, tr);
    }
    w.status[t] = w.transcript_preloaded ? (w.status[t] | status) : status;
}
// round k (0-based): vx, vr and the scalar sets of X and R (wnla.rs:136-157)
HD void wnla_prove_round_scalars(const WnlaProveWs& w, size_t t, int k) {
    const size_t N = w.N;
    const int Lk = (int)wnla_ceil_shift((size_t)w.nl, k), Nk = (int)wnla_ceil_shift((size_t)w.nn, k), Ck = (int)wnla_ceil_shift((size_t)w.nh, k);
    sc rho, mu, rho_inv, mu2, zero, t1, t2;
    sc_set_u32(zero, 0);
    ws_ld8(rho.v, w.prm, N, t, 0);
    ws_ld8(mu.v, w.prm, N, t, 1);
    if (sc_is_zero(rho)) { w.status[t] |= ST_DEGENERATE; sc_set_u32(rho, 1); }   // rho.invert_vartime().unwrap()
    sc_inv(rho_inv, rho);
    ws_st8(w.prm, N, t, 2, rho_inv.v);
    sc_mul(mu2, mu, mu);
    // vx = 2 rho^-1 sum_j n[2j] n[2j+1] mu2^(j+1) + sum_j (c[2j] l[2j+1] + c[2j+1] l[2j]);  vr = sum_j n[2j+1]^2 mu2^(j+1) + sum_j c[2j+1] l[2j+1]
    sc vx, vr, mp = mu2;
    sc_set_u32(vx, 0);
    sc_set_u32(vr, 0);
#pragma nounroll
    for (int j = 0; 2 * j < Nk; j++) {
        sc a, b;
        wp_ld(a, w.vn, w, t, 2 * j, Nk);
        wp_ld(b, w.vn, w, t, 2 * j + 1, Nk);
        sc_mul(t1, a, b); sc_mul(t1, t1, mp); sc_add(vx, vx, t1);
        sc_mul(t1, b, b); sc_mul(t1, t1, mp); sc_add(vr, vr, t1);
        sc_mul(mp, mp, mu2);
    }
    sc_add(t1, rho_inv, rho_inv);
    sc_mul(vx, vx, t1);
    const int Mk = Lk > Ck ? Lk : Ck;
#pragma nounroll
    for (int j = 0; 2 * j < Mk; j++) {
        sc c0, c1, l0, l1;
        wp_ld(c0, w.vc, w, t, 2 * j, Ck);
        wp_ld(c1, w.vc, w, t, 2 * j + 1, Ck);
        wp_ld(l0, w.vl, w, t, 2 * j, Lk);
        wp_ld(l1, w.vl, w, t, 2 * j + 1, Lk);
        sc_mul(t1, c0, l1); sc_add(vx, vx, t1);
        sc_mul(t1, c1, l0); sc_add(vx, vx, t1);
        sc_mul(t1, c1, l1); sc_add(vr, vr, t1);
    }
    u32* mx = w.msc;
    u32* mr = w.msc + wp_set_words(w);
    ws_st8(mx, N, t, 0, vx.v);
    ws_st8(mr, N, t, 0, vr.v);
    // original generator i sits in folded slot p = i >> k: X takes the PARTNER slot's vector entry, R the own (odd) slot's
#pragma nounroll
    for (int i = 0; i < w.nh; i++) {
        const int p = i >> k;
        sc co, lp, lq;
        ws_ld8(co.v, w.ch, N, t, i);
        wp_ld(lq, w.vl, w, t, p ^ 1, Lk);
        sc_mul(t1, co, lq);
        ws_st8(mx, N, t, 1 + w.ng + i, t1.v);
        wp_ld(lp, w.vl, w, t, p, Lk);
        sc_mul(t2, co, lp);
        ws_st8(mr, N, t, 1 + w.ng + i, (p & 1) ? t2.v : zero.v);
    }
#pragma nounroll
    for (int i = 0; i < w.ng; i++) {
        const int p = i >> k;
        sc co, np, nq;
        ws_ld8(co.v, w.cg, N, t, i);
        wp_ld(nq, w.vn, w, t, p ^ 1, Nk);
        sc_mul(t1, co, nq);
        sc_mul(t1, t1, (p & 1) ? rho_inv : rho);      // <g0, n1 rho> + <g1, n0 rho^-1>
        ws_st8(mx, N, t, 1 + i, t1.v);
        wp_ld(np, w.vn, w, t, p, Nk);
        sc_mul(t2, co, np);
        ws_st8(mr, N, t, 1 + i, (p & 1) ? t2.v : zero.v);
    }
}
// terms of the three MSMs of a round.  X and the next commitment take every generator; R (oddsh = the round number k) only the
// generators whose folded slot i >> k is odd -- wnla_prove_round_scalars stores zero for the others -- so its two generator runs
// enumerate the odd blocks of 2^k terms only (fb_term_index)
HD int wnla_odd_block_terms(int n, int sh) { const int B = 1 << sh, rem = n & (2 * B - 1); return ((n >> (sh + 1)) << sh) + (rem > B ? rem - B : 0); }
HD void wnla_prove_msm_ranges(FbRanges& rg, const WnlaProveWs& w, int oddsh = -1) {
    if (oddsh < 0) { fb_ranges_one(rg, 0, 0, 1 + w.ng + w.nh); return; }
    rg.n = 3;
    rg.slot[0] = 0; rg.base[0] = 0; rg.count[0] = 1; rg.bits[0] = 0; rg.oddsh[0] = -1;
    rg.slot[1] = 1; rg.base[1] = 1; rg.count[1] = wnla_odd_block_terms(w.ng, oddsh); rg.bits[1] = 0; rg.oddsh[1] = oddsh;
    rg.slot[2] = 1 + w.ng; rg.base[2] = 1 + w.ng; rg.count[2] = wnla_odd_block_terms(w.nh, oddsh); rg.bits[2] = 0; rg.oddsh[2] = oddsh;
}
// round k: X, R (and, from round 1 on, this level's commitment) to affine, transcript, challenge, fold, next level's commitment.
// The reference recomputes every level's commitment from the folded vectors (wnla.rs:186 `wnla.commit(&l_, &n_)`).  Level 1 is
// done that way here too (its scalars -> the third MSM of round 0), because level 0's commitment is the CALLER's and need not be
// commit(l, n); from then on C_k IS commit(l_k, n_k), for which the argument's completeness gives
// commit(l_{k+1}, n_{k+1}) = C_k + y X_k + (y^2 - 1) R_k -- the verifier's own update (wnla.rs:84-102): two variable-base
// multiplications instead of an MSM over all 1 + |g_vec| + |h_vec| generators per round.
HD void wnla_prove_round_fold(const WnlaProveWs& w, size_t t, int k) {
    const size_t N = w.N;
    const int Lk = (int)wnla_ceil_shift((size_t)w.nl, k), Nk = (int)wnla_ceil_shift((size_t)w.nn, k), Ck = (int)wnla_ceil_shift((size_t)w.nh, k);
    pt P[3];
    apt A[3];
    ws_ld_pt(P[0], w.pbuf, N, t);
    ws_ld_pt(P[1], w.pbuf + 30 * N, N, t);
    if (k > 0) ws_ld_pt(P[2], w.pbuf + 60 * N, N, t);
    else pt_set_identity(P[2]);
    batch_to_affine<3>(A, P);
    apt Ca;
    if (k > 0) Ca = A[2];
    else ws_ld_apt(Ca, w.com, N, t, 0);
    const int slot = w.rounds - 1 - k;                  // proof.r / proof.x are pushed after the recursion returns (wnla.rs:187-188)
    apt_to_xy64(w.proof_x + ((size_t)t * w.rounds + slot) * 64, A[0]);
    apt_to_xy64(w.proof_r + ((size_t)t * w.rounds + slot) * 64, A[1]);
    strobe tr;
    ws_ld_strobe(tr, w.tstate, N, t);
    app_point(tr, "wnla_com", Ca);                      // wnla.rs:159-163
    app_point(tr, "wnla_x", A[0]);
    app_point(tr, "wnla_r", A[1]);
    t_append_u64(tr, "l.sz", (u64)Lk);
    t_append_u64(tr, "n.sz", (u64)Nk);
    sc y;
    if (!t_get_challenge(tr, "wnla_challenge", y)) { w.status[t] |= ST_DEGENERATE; sc_set_u32(y, 1); }
    ws_st_strobe(w.tstate, N, t, tr);
    sc rho, mu, rho_inv, mu2, t1, t2, zero;
    sc_set_u32(zero, 0);
    ws_ld8(rho.v, w.prm, N, t, 0);
    ws_ld8(mu.v, w.prm, N, t, 1);
    ws_ld8(rho_inv.v, w.prm, N, t, 2);
    if (sc_is_zero(rho)) sc_set_u32(rho, 1);
    sc_mul(mu2, mu, mu);
    // fold (wnla.rs:167-173): l' = l0 + y l1, n' = rho^-1 n0 + y n1, c' = c0 + y c1 (in place: slot j reads 2j, 2j+1 >= j)
    const int L1 = (Lk + 1) / 2, N1 = (Nk + 1) / 2, C1 = (Ck + 1) / 2;
#pragma nounroll
    for (int j = 0; j < L1; j++) {
        sc a, b;
        wp_ld(a, w.vl, w, t, 2 * j, Lk);
        wp_ld(b, w.vl, w, t, 2 * j + 1, Lk);
        sc_mul(t1, b, y); sc_add(t1, t1, a);
        ws_st8(w.vl, N, t, j, t1.v);
    }
#pragma nounroll
    for (int j = 0; j < N1; j++) {
        sc a, b;
        wp_ld(a, w.vn, w, t, 2 * j, Nk);
        wp_ld(b, w.vn, w, t, 2 * j + 1, Nk);
        sc_mul(t1, a, rho_inv); sc_mul(t2, b, y); sc_add(t1, t1, t2);
        ws_st8(w.vn, N, t, j, t1.v);
    }
#pragma nounroll
    for (int j = 0; j < C1; j++) {
        sc a, b;
        wp_ld(a, w.vc, w, t, 2 * j, Ck);
        wp_ld(b, w.vc, w, t, 2 * j + 1, Ck);
        sc_mul(t1, b, y); sc_add(t1, t1, a);
        ws_st8(w.vc, N, t, j, t1.v);
    }
    // generator coefficients: h' = h0 + y h1, g' = rho g0 + y g1
#pragma nounroll
    for (int i = 0; i < w.nh; i++) {
        if ((i >> k) & 1) { sc co; ws_ld8(co.v, w.ch, N, t, i); sc_mul(co, co, y); ws_st8(w.ch, N, t, i, co.v); }
    }
#pragma nounroll
    for (int i = 0; i < w.ng; i++) {
        sc co;
        ws_ld8(co.v, w.cg, N, t, i);
        sc_mul(co, co, ((i >> k) & 1) ? y : rho);
        ws_st8(w.cg, N, t, i, co.v);
    }
    ws_st8(w.prm, N, t, 0, mu.v);                       // rho <- mu, mu <- mu^2 (wnla.rs:180-181)
    ws_st8(w.prm, N, t, 1, mu2.v);
    if (k + 1 < w.rounds && k > 0) {
        sc y2m1, one;
        sc_set_u32(one, 1);
        sc_mul(y2m1, y, y);
        sc_sub(y2m1, y2m1, one);
        pt_slot* tbl = w.straus + t * (2 * BPPP_STRAUS_ENTRIES);
        glv_split rs[2];
        straus_build_table(tbl, A[0]);
        straus_build_table(tbl + BPPP_STRAUS_ENTRIES, A[1]);
        glv_decompose(rs[0], y);
        glv_decompose(rs[1], y2m1);
        pt acc;
        straus_msm_glv(acc, tbl, rs, 2);
        pt_madd(acc, acc, Ca, apt_is_identity(Ca));
        ws_st_pt(w.pbuf + 60 * N, N, t, acc);
    }
    if (k + 1 < w.rounds && k == 0) {
        // scalars of the next level's commitment  v g + <h', l'> + <g', n'>,  v = <c', l'> + |n'|^2_{mu'}   (wnla.rs:66-72 via :186)
        u32* mc = w.msc + 2 * wp_set_words(w);
        sc v, mp = mu2;
        sc_set_u32(v, 0);
#pragma nounroll
        for (int j = 0; j < N1; j++) {
            sc a;
            ws_ld8(a.v, w.vn, N, t, j);
            sc_mul(t1, a, a); sc_mul(t1, t1, mp); sc_add(v, v, t1);
            sc_mul(mp, mp, mu2);
        }
        const int M1 = L1 < C1 ? L1 : C1;
#pragma nounroll
        for (int j = 0; j < M1; j++) {
            sc a, b;
            ws_ld8(a.v, w.vc, N, t, j);
            ws_ld8(b.v, w.vl, N, t, j);
            sc_mul(t1, a, b); sc_add(v, v, t1);
        }
        ws_st8(mc, N, t, 0, v.v);
#pragma nounroll
        for (int i = 0; i < w.nh; i++) {
            sc co, lp;
            ws_ld8(co.v, w.ch, N, t, i);
            wp_ld(lp, w.vl, w, t, i >> (k + 1), L1);
            sc_mul(t1, co, lp);
            ws_st8(mc, N, t, 1 + w.ng + i, t1.v);
        }
#pragma nounroll
        for (int i = 0; i < w.ng; i++) {
            sc co, np;
            ws_ld8(co.v, w.cg, N, t, i);
            wp_ld(np, w.vn, w, t, i >> (k + 1), N1);
            sc_mul(t1, co, np);
            ws_st8(mc, N, t, 1 + i, t1.v);
        }
    }
}
// base case (wnla.rs:126-133): the remaining vectors are the proof's l and n
HD void wnla_prove_finish(const WnlaProveWs& w, size_t t) {
    const size_t N = w.N;
    const bool bad = w.status[t] != ST_OK;
    sc x, zero;
    sc_set_u32(zero, 0);
#pragma nounroll
    for (int j = 0; j < w.nl_f; j++) { ws_ld8(x.v, w.vl, N, t, j); sc_to_be(w.proof_l + ((size_t)t * w.nl_f + j) * 32, bad ? zero : x); }
#pragma nounroll
    for (int j = 0; j < w.nn_f; j++) { ws_ld8(x.v, w.vn, N, t, j); sc_to_be(w.proof_n + ((size_t)t * w.nn_f + j) * 32, bad ? zero : x); }
    if (bad) {      // a flagged instance has no proof: zero the round points too
#pragma nounroll
        for (size_t b = 0; b < (size_t)w.rounds * 64; b++) { w.proof_r[(size_t)t * w.rounds * 64 + b] = 0; w.proof_x[(size_t)t * w.rounds * 64 + b] = 0; }
    }
}
// shape of the proof for witness lengths (nl, nn): rounds and the lengths of the final l and n
inline void wnla_proof_shape(size_t nl, size_t nn, size_t& rounds, size_t& nl_f, size_t& nn_f) {
    rounds = 0;
    while (nl + nn >= 6) { nl = (nl + 1) / 2; nn = (nn + 1) / 2; rounds++; }
    nl_f = nl;
    nn_f = nn;
}

}  // namespace bppp
