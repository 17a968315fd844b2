// Generic batched `ArithmeticCircuit::prove` (circuit.rs:260-556) for a circuit shared by the batch (CircuitDev, circuit_core.h).
// Per instance: the witness (v: k vectors of dim_nv scalars, s_v, w_l, w_r, w_o), the k commitments of v (what the reference
// receives as `v`), and the prover's random scalars in the reference's draw order:
//     ro: 7 draws (slots 0,1,2,3,5,6,7) | rl: 6 (0,1,2,4,5,6) | rr: 5 (0,1,3,4,5) | ls: dim_nv | ns: dim_nm      (circuit.rs:264-298, 371-372)
// so that the proof bytes equal the CPU prover's for the same RNG stream.  Stages (each a kernel, MSMs in between through the
// batch-shared fixed-base tables): A  witness -> scalar sets of c_o, c_l, c_r;  B  transcript to delta, coefficient vectors,
// the f polynomial (circuit.rs:394-470), r_s -> scalar set of c_s;  C  tau, the WNLA witness l, n, the vector c and the WNLA
// commitment's scalar set;  then the generic WNLA prover (wnla_prove_core.h) on the same transcript.
//
// Proof layout (the verifier's): c_l, c_r, c_o, c_s | r[rounds] | x[rounds] | l | n,  shape from wnla_proof_shape(NH, NG).
#pragma once
#include "circuit_core.h"
#include "wnla_prove_core.h"

namespace bppp {

struct CircuitProveWs {
    size_t N;
    CircuitDev cd;
    int NG, NH, n_rnd;
    int transcript_preloaded;        // 1: tstate / status were prepared by an outer protocol stage (reciprocal prover)
    size_t rnd_stride;               // bytes between instances' draws (n_rnd * 32 when dense)
    const int* part;                 // [3 nv + nm]: LO | LL | LR | NO  (index into w_o or -1)
    const uint8_t *v_pts, *v, *s_v, *w_l, *w_r, *w_o, *rnd;   // C-ABI layouts (device): n x k x 64, n x k x nv x 32, n x k x 32, ...
    uint8_t* proof_head;             // n x 256: c_l, c_r, c_o, c_s
    int32_t* status;
    u32* tstate;
    // scalar vectors, limb-major with stride N
    u32 *ro, *rl, *rr, *rs;          // [9*8][N] each
    u32 *lo, *ll, *lr, *ls, *v1, *cl0;   // [nv*8][N] each (v1, cl0 use nv - 1 entries)
    u32 *no, *nl, *nr, *ns;          // [nm*8][N] each
    u32 *lamv, *muv, *coef;          // [nl*8], [nm*8], [(3nm+3nv)*8]
    u32* misc;                       // [8*8][N]: rho, lambda, beta, delta, mu, v0, rv0, delta^-1
    u32* msc;                        // [3][(1 + NG + NH)*8][N]  MSM scalar sets (slot = table base index)
    u32* pbuf;                       // [3][30][N]
    // outputs for the WNLA prover (C-ABI layouts, device)
    uint8_t *wn_commit, *wn_c, *wn_rho, *wn_mu, *wn_l, *wn_n;
    FbTable fb;
    FbTable fb_ct;                   // "ct_prover": the commitments to the witness and its blindings (circuit.rs:336-345, 469-470) in the full-scan form when ct != 0
    int ct;
    strobe base;
    TranscriptIo tio;                // caller's transcripts (circuit.rs:260 `t: &mut Transcript`); input side ignored when transcript_preloaded
    int divergent_positions;
};
enum { CM_RHO = 0, CM_LAMBDA, CM_BETA, CM_DELTA, CM_MU, CM_V0, CM_RV0, CM_DINV };

HD void cp_ld(sc& r, const u32* a, const CircuitProveWs& w, size_t t, int i) { ws_ld8(r.v, a, w.N, t, i); }
HD void cp_st(u32* a, const CircuitProveWs& w, size_t t, int i, const sc& x) { ws_st8(a, w.N, t, i, x.v); }
HD size_t cp_set_words(const CircuitProveWs& w) { return (size_t)(1 + w.NG + w.NH) * 8 * w.N; }
// scalar set `set` <- <h_vec, r9 || lvec> + <g_vec, nvec>  (slot = base index: h_vec[j] is base 1 + NG + j, g_vec[j] is base 1 + j)
HD void cp_fill_hg(const CircuitProveWs& w, size_t t, int set, const u32* r9, const u32* lvec, const u32* nvec) {
    u32* m = w.msc + (size_t)set * cp_set_words(w);
    sc x;
#pragma nounroll
    for (int i = 0; i < 9; i++) { cp_ld(x, r9, w, t, i); ws_st8(m, w.N, t, 1 + w.NG + i, x.v); }
#pragma nounroll
    for (int j = 0; j < w.cd.nv; j++) { cp_ld(x, lvec, w, t, j); ws_st8(m, w.N, t, 1 + w.NG + 9 + j, x.v); }
#pragma nounroll
    for (int j = 0; j < w.cd.nm; j++) { cp_ld(x, nvec, w, t, j); ws_st8(m, w.N, t, 1 + j, x.v); }
}
HD void cp_ranges(FbRanges& rg, const CircuitProveWs& w, bool with_g) {
    rg.n = 0;
    if (with_g) { rg.slot[rg.n] = 0; rg.base[rg.n] = 0; rg.count[rg.n] = 1; rg.n++; }
    rg.slot[rg.n] = 1; rg.base[rg.n] = 1; rg.count[rg.n] = w.cd.nm; rg.n++;
    rg.slot[rg.n] = 1 + w.NG; rg.base[rg.n] = 1 + w.NG; rg.count[rg.n] = 9 + w.cd.nv; rg.n++;
}

// ---- stage A: decode, partition w_o, scalar sets of c_o (set 0), c_l (set 1), c_r (set 2)          (circuit.rs:264-345)
HD void circuit_prove_stage_a(const CircuitProveWs& w, size_t t) {
    const CircuitDev& cd = w.cd;
    const int nm = cd.nm, nv = cd.nv, no_ = cd.no, k = cd.k;
    bool ok = true;
    sc zero, x;
    sc_set_u32(zero, 0);
    const uint8_t* rnd = w.rnd + (size_t)t * w.rnd_stride;
    int d = 0;
    auto draw = [&](sc& out) { ok &= sc_from_be(out, rnd + 32 * (size_t)d); d++; };
    const int ro_slots[7] = {0, 1, 2, 3, 5, 6, 7}, rl_slots[6] = {0, 1, 2, 4, 5, 6}, rr_slots[5] = {0, 1, 3, 4, 5};
#pragma nounroll
    for (int i = 0; i < 9; i++) { cp_st(w.ro, w, t, i, zero); cp_st(w.rl, w, t, i, zero); cp_st(w.rr, w, t, i, zero); }
#pragma nounroll
    for (int i = 0; i < 7; i++) { draw(x); cp_st(w.ro, w, t, ro_slots[i], x); }
#pragma nounroll
    for (int i = 0; i < 6; i++) { draw(x); cp_st(w.rl, w, t, rl_slots[i], x); }
#pragma nounroll
    for (int i = 0; i < 5; i++) { draw(x); cp_st(w.rr, w, t, rr_slots[i], x); }
#pragma nounroll
    for (int j = 0; j < nv; j++) { draw(x); cp_st(w.ls, w, t, j, x); }
#pragma nounroll
    for (int j = 0; j < nm; j++) { draw(x); cp_st(w.ns, w, t, j, x); }
    const uint8_t* wo = w.w_o + (size_t)t * no_ * 32;
#pragma nounroll
    for (int i = 0; i < no_; i++) ok &= sc_from_be(x, wo + 32 * (size_t)i);
    auto pick = [&](int idx, sc& out) {
        out = zero;
        if (idx >= 0) (void)sc_from_be(out, wo + 32 * (size_t)idx);
    };
#pragma nounroll
    for (int j = 0; j < nv; j++) {
        pick(w.part[j], x); cp_st(w.lo, w, t, j, x);
        pick(w.part[nv + j], x); cp_st(w.ll, w, t, j, x);
        pick(w.part[2 * nv + j], x); cp_st(w.lr, w, t, j, x);
    }
#pragma nounroll
    for (int j = 0; j < nm; j++) {
        pick(w.part[3 * nv + j], x); cp_st(w.no, w, t, j, x);
        ok &= sc_from_be(x, w.w_l + ((size_t)t * nm + j) * 32); cp_st(w.nl, w, t, j, x);
        ok &= sc_from_be(x, w.w_r + ((size_t)t * nm + j) * 32); cp_st(w.nr, w, t, j, x);
    }
#pragma nounroll
    for (int i = 0; i < k * nv; i++) ok &= sc_from_be(x, w.v + ((size_t)t * k * nv + i) * 32);
#pragma nounroll
    for (int i = 0; i < k; i++) ok &= sc_from_be(x, w.s_v + ((size_t)t * k + i) * 32);
    cp_fill_hg(w, t, 0, w.ro, w.lo, w.no);
    cp_fill_hg(w, t, 1, w.rl, w.ll, w.nl);
    cp_fill_hg(w, t, 2, w.rr, w.lr, w.nr);
    const int32_t st = ok ? ST_OK : ST_BAD_ENCODING;
    w.status[t] = w.transcript_preloaded ? (w.status[t] | st) : st;
}
// ---- stage B: c_o, c_l, c_r -> affine + transcript, challenges, coefficient vectors, f polynomial, r_s, scalar set of c_s (set 0)
HD void circuit_prove_stage_b(const CircuitProveWs& w, size_t t) {
    const size_t N = w.N;
    const CircuitDev& cd = w.cd;
    const int nm = cd.nm, nv = cd.nv, k = cd.k;
    pt P[3];
    apt A[3];
    for (int i = 0; i < 3; i++) ws_ld_pt(P[i], w.pbuf + (size_t)i * 30 * N, N, t);
    batch_to_affine<3>(A, P);                      // c_o, c_l, c_r
    uint8_t* ph = w.proof_head + 256 * t;
    apt_to_xy64(ph, A[1]);
    apt_to_xy64(ph + 64, A[2]);
    apt_to_xy64(ph + 128, A[0]);
    strobe tr;
    if (w.transcript_preloaded) ws_ld_strobe(tr, w.tstate, N, t);
    else {
        int32_t tst = ST_OK;
        tio_begin(tr, tst, w.tio, w.base, t);
        if (tst) w.status[t] |= tst;
    }
    app_point(tr, "commitment_cl", A[1]);          // circuit.rs:347-350
    app_point(tr, "commitment_cr", A[2]);
    app_point(tr, "commitment_co", A[0]);
    bool ok = true;
#pragma nounroll
    for (int i = 0; i < k; i++) {
        apt V;
        ok &= apt_from_xy64(V, w.v_pts + ((size_t)t * k + i) * 64);
        if (!ok) { fe_set_u32(V.x, 0); fe_set_u32(V.y, 0); }
        app_point(tr, "commitment_v", V);
    }
    if (!ok) w.status[t] |= ST_BAD_ENCODING;
    sc rho, lambda, beta, delta, mu, one, zero, two, t1, t2;
    sc_set_u32(one, 1);
    sc_set_u32(zero, 0);
    sc_set_u32(two, 2);
    bool cok = t_get_challenge(tr, "circuit_rho", rho);
    cok &= t_get_challenge(tr, "circuit_lambda", lambda);
    cok &= t_get_challenge(tr, "circuit_beta", beta);
    cok &= t_get_challenge(tr, "circuit_delta", delta);
    if (!cok) { w.status[t] |= ST_DEGENERATE; rho = one; lambda = one; beta = one; delta = one; }
    ws_st_strobe(w.tstate, N, t, tr);
    sc_mul(mu, rho, rho);
    // mu^-1, delta^-1, beta^-1 from one inversion
    if (sc_is_zero(mu) | sc_is_zero(delta) | sc_is_zero(beta)) w.status[t] |= ST_DEGENERATE;
    sc m_ = sc_is_zero(mu) ? one : mu, d_ = sc_is_zero(delta) ? one : delta, b_ = sc_is_zero(beta) ? one : beta;
    sc md, mdb, inv, mu_inv, delta_inv, beta_inv;
    sc_mul(md, m_, d_);
    sc_mul(mdb, md, b_);
    sc_inv(inv, mdb);
    sc_mul(beta_inv, inv, md);
    sc_mul(inv, inv, b_);                // (mu delta)^-1
    sc_mul(mu_inv, inv, d_);
    sc_mul(delta_inv, inv, m_);
    cp_st(w.misc, w, t, CM_RHO, rho); cp_st(w.misc, w, t, CM_LAMBDA, lambda); cp_st(w.misc, w, t, CM_BETA, beta);
    cp_st(w.misc, w, t, CM_DELTA, delta); cp_st(w.misc, w, t, CM_MU, mu); cp_st(w.misc, w, t, CM_DINV, delta_inv);
    sc lam_nv, mu_nv;
    circuit_collect(cd, w.lamv, w.muv, w.coef, N, t, lambda, mu, mu_inv, lam_nv, mu_nv);
    // v_0, r_v[0], v_1 = 2 sum_i coef_i (v_i[0], s_v[i], v_i[1..])                               (circuit.rs:376-392)
    sc v0 = zero, rv0 = zero;
#pragma nounroll
    for (int j = 0; j + 1 < nv; j++) cp_st(w.v1, w, t, j, zero);
    {
        sc lpow = one, mpow = mu;
#pragma nounroll
        for (int i = 0; i < k; i++) {
            sc cf = zero, x;
            if (cd.f_l) sc_add(cf, cf, lpow);
            if (cd.f_m) sc_add(cf, cf, mpow);
            sc_add(cf, cf, cf);
            (void)sc_from_be(x, w.v + ((size_t)t * k * nv + (size_t)i * nv) * 32);
            sc_mul(t1, x, cf); sc_add(v0, v0, t1);
            (void)sc_from_be(x, w.s_v + ((size_t)t * k + i) * 32);
            sc_mul(t1, x, cf); sc_add(rv0, rv0, t1);
#pragma nounroll
            for (int j = 0; j + 1 < nv; j++) {
                sc cur;
                (void)sc_from_be(x, w.v + ((size_t)t * k * nv + (size_t)i * nv + 1 + j) * 32);
                cp_ld(cur, w.v1, w, t, j);
                sc_mul(t1, x, cf); sc_add(cur, cur, t1);
                cp_st(w.v1, w, t, j, cur);
            }
            sc_mul(lpow, lpow, lam_nv);
            sc_mul(mpow, mpow, mu_nv);
        }
    }
    cp_st(w.misc, w, t, CM_V0, v0);
    cp_st(w.misc, w, t, CM_RV0, rv0);
    // c_l0 (collect_cl0, circuit.rs:572-582): nv - 1 entries
    {
        sc lp = lambda, mq;
        sc_mul(mq, mu, mu);
#pragma nounroll
        for (int j = 0; j + 1 < nv; j++) {
            sc c0 = zero;
            if (cd.f_l) sc_add(c0, c0, lp);
            if (cd.f_m) sc_sub(c0, c0, mq);
            cp_st(w.cl0, w, t, j, c0);
            sc_mul(lp, lp, lambda);
            sc_mul(mq, mq, mu);
        }
    }
    // the weighted sums over the n-type vectors (weight mu^(j+1)) and the dot products over the l-type vectors
    sc S_nsns = zero, S_nsno = zero, S_nsA = zero, S_nono = zero, S_nsB = zero, S_noA = zero, S_cRcR = zero, S_nscO = zero, S_noB = zero,
       S_AA = zero, S_cOcR = zero, S_cLcL = zero, S_AcO = zero, S_BB = zero, S_cOcL = zero, S_BcO = zero, S_cnOlr = zero;
    {
        sc wgt = mu;
#pragma nounroll
        for (int j = 0; j < nm; j++) {
            sc ns, no, nl, nr, cL, cR, cO, Aj, Bj, u;
            cp_ld(ns, w.ns, w, t, j); cp_ld(no, w.no, w, t, j); cp_ld(nl, w.nl, w, t, j); cp_ld(nr, w.nr, w, t, j);
            cp_ld(cL, w.coef, w, t, j); cp_ld(cR, w.coef, w, t, nm + j); cp_ld(cO, w.coef, w, t, 2 * nm + j);
            sc_add(Aj, nl, cR);
            sc_add(Bj, nr, cL);
#define BPPP_WACC(S, a, b) sc_mul(u, a, b); sc_mul(u, u, wgt); sc_add(S, S, u);
            BPPP_WACC(S_nsns, ns, ns) BPPP_WACC(S_nsno, ns, no) BPPP_WACC(S_nsA, ns, Aj) BPPP_WACC(S_nono, no, no) BPPP_WACC(S_nsB, ns, Bj)
            BPPP_WACC(S_noA, no, Aj) BPPP_WACC(S_cRcR, cR, cR) BPPP_WACC(S_nscO, ns, cO) BPPP_WACC(S_noB, no, Bj) BPPP_WACC(S_AA, Aj, Aj)
            BPPP_WACC(S_cOcR, cO, cR) BPPP_WACC(S_cLcL, cL, cL) BPPP_WACC(S_AcO, Aj, cO) BPPP_WACC(S_BB, Bj, Bj) BPPP_WACC(S_cOcL, cO, cL)
            BPPP_WACC(S_BcO, Bj, cO)
#undef BPPP_WACC
            if (j < nv) { sc lrj; cp_ld(lrj, w.lr, w, t, j); sc_mul(u, cO, lrj); sc_add(S_cnOlr, S_cnOlr, u); }   // vector_mul(&c_nO, &lr): circuit.rs:463
            sc_mul(wgt, wgt, mu);
        }
    }
    sc D_cl0ls = zero, D_cl0lo = zero, D_cl0ll = zero, D_cl0lr = zero, D_cRls = zero, D_cRlo = zero, D_cLls = zero, D_cLlo = zero,
       D_cRll = zero, D_cOls = zero, D_cOll = zero, D_cLlr = zero, D_cRv1 = zero, D_cLv1 = zero, D_cOv1 = zero;
#pragma nounroll
    for (int j = 0; j < nv; j++) {
        sc ls, lo, ll, lr, cL, cR, cO, u;
        cp_ld(ls, w.ls, w, t, j); cp_ld(lo, w.lo, w, t, j); cp_ld(ll, w.ll, w, t, j); cp_ld(lr, w.lr, w, t, j);
        cp_ld(cL, w.coef, w, t, 3 * nm + j); cp_ld(cR, w.coef, w, t, 3 * nm + nv + j); cp_ld(cO, w.coef, w, t, 3 * nm + 2 * nv + j);
#define BPPP_DACC(D, a, b) sc_mul(u, a, b); sc_add(D, D, u);
        BPPP_DACC(D_cRls, cR, ls) BPPP_DACC(D_cRlo, cR, lo) BPPP_DACC(D_cLls, cL, ls) BPPP_DACC(D_cLlo, cL, lo) BPPP_DACC(D_cRll, cR, ll)
        BPPP_DACC(D_cOls, cO, ls) BPPP_DACC(D_cOll, cO, ll) BPPP_DACC(D_cLlr, cL, lr)
        if (j + 1 < nv) {
            sc c0, v1;
            cp_ld(c0, w.cl0, w, t, j); cp_ld(v1, w.v1, w, t, j);
            BPPP_DACC(D_cl0ls, c0, ls) BPPP_DACC(D_cl0lo, c0, lo) BPPP_DACC(D_cl0ll, c0, ll) BPPP_DACC(D_cl0lr, c0, lr)
            BPPP_DACC(D_cRv1, cR, v1) BPPP_DACC(D_cLv1, cL, v1) BPPP_DACC(D_cOv1, cO, v1)
        }
#undef BPPP_DACC
    }
    // f(tau) coefficients at powers -2, -1, 0, 1, 2, 4, 5, 6                                       (circuit.rs:394-470)
    sc f[8], delta2, two_d, two_di;
    sc_mul(delta2, delta, delta);
    sc_add(two_d, delta, delta);
    sc_add(two_di, delta_inv, delta_inv);
    auto dbl = [&](sc& r, const sc& a) { sc_add(r, a, a); };
    sc_neg(f[0], S_nsns);
    sc_mul(t1, two_d, S_nsno); sc_add(f[1], D_cl0ls, t1);
    dbl(t1, D_cRls); sc_neg(f[2], t1);
    sc_mul(t1, D_cl0lo, delta); sc_sub(f[2], f[2], t1);
    dbl(t1, S_nsA); sc_sub(f[2], f[2], t1);
    sc_mul(t1, S_nono, delta2); sc_sub(f[2], f[2], t1);
    dbl(f[3], D_cLls);
    sc_mul(t1, D_cRlo, two_d); sc_add(f[3], f[3], t1);
    sc_add(f[3], f[3], D_cl0ll);
    dbl(t1, S_nsB); sc_add(f[3], f[3], t1);
    sc_mul(t1, S_noA, two_d); sc_add(f[3], f[3], t1);
    f[4] = S_cRcR;
    sc_mul(t1, D_cOls, two_di); sc_sub(f[4], f[4], t1);
    sc_mul(t1, D_cLlo, two_d); sc_sub(f[4], f[4], t1);
    dbl(t1, D_cRll); sc_sub(f[4], f[4], t1);
    sc_sub(f[4], f[4], D_cl0lr);
    sc_mul(t1, S_nscO, two_di); sc_sub(f[4], f[4], t1);
    sc_mul(t1, S_noB, two_d); sc_sub(f[4], f[4], t1);
    sc_sub(f[4], f[4], S_AA);
    sc_mul(f[5], S_cOcR, two_di);
    sc_add(f[5], f[5], S_cLcL);
    sc_mul(t1, D_cOll, two_di); sc_sub(f[5], f[5], t1);
    dbl(t1, D_cLlr); sc_sub(f[5], f[5], t1);
    dbl(t1, D_cRv1); sc_sub(f[5], f[5], t1);
    sc_mul(t1, S_AcO, two_di); sc_sub(f[5], f[5], t1);
    sc_sub(f[5], f[5], S_BB);
    sc_mul(t1, S_cOcL, two_di); sc_neg(f[6], t1);
    sc_mul(t1, S_cnOlr, two_di); sc_add(f[6], f[6], t1);
    dbl(t1, D_cLv1); sc_add(f[6], f[6], t1);
    sc_mul(t1, S_BcO, two_di); sc_add(f[6], f[6], t1);
    sc_mul(t1, D_cOv1, two_di); sc_neg(f[7], t1);
    // r_s (circuit.rs:472-484)
    sc ro[9], rl[9], rr[9], rs[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { cp_ld(ro[i], w.ro, w, t, i); cp_ld(rl[i], w.rl, w, t, i); cp_ld(rr[i], w.rr, w, t, i); }
    sc_mul(t1, ro[1], delta); sc_mul(t1, t1, beta); sc_add(rs[0], f[1], t1);
    sc_mul(rs[1], f[0], beta_inv);
    sc_mul(t1, ro[0], delta); sc_add(t1, t1, f[2]); sc_mul(t1, t1, beta_inv); sc_sub(rs[2], t1, rl[1]);
    sc_sub(t1, f[3], rl[0]); sc_mul(t1, t1, beta_inv); sc_mul(t2, ro[2], delta); sc_add(t2, t2, rr[1]); sc_add(rs[3], t1, t2);
    sc_add(t1, f[4], rr[0]); sc_mul(t1, t1, beta_inv); sc_mul(t2, ro[3], delta); sc_sub(t2, t2, rl[2]); sc_add(rs[4], t1, t2);
    sc_mul(t1, rv0, beta_inv); sc_neg(rs[5], t1);
    sc_mul(t1, f[5], beta_inv); sc_mul(t2, ro[5], delta); sc_add(t1, t1, t2); sc_add(t1, t1, rr[3]); sc_sub(rs[6], t1, rl[4]);
    sc_mul(t1, f[6], beta_inv); sc_add(t1, t1, rr[4]); sc_mul(t2, ro[6], delta); sc_add(t1, t1, t2); sc_sub(rs[7], t1, rl[5]);
    sc_mul(t1, f[7], beta_inv); sc_mul(t2, ro[7], delta); sc_add(t1, t1, t2); sc_sub(t1, t1, rl[6]); sc_add(rs[8], t1, rr[5]);
#pragma unroll
    for (int i = 0; i < 9; i++) cp_st(w.rs, w, t, i, rs[i]);
    cp_fill_hg(w, t, 0, w.rs, w.ls, w.ns);
}
// ---- stage C: c_s -> affine + transcript, tau, the WNLA witness (l, n), c, and the WNLA commitment's scalar set (set 0)
HD void circuit_prove_stage_c(const CircuitProveWs& w, size_t t) {
    const size_t N = w.N;
    const CircuitDev& cd = w.cd;
    const int nm = cd.nm, nv = cd.nv, nl = cd.nl;
    pt P;
    apt CS;
    ws_ld_pt(P, w.pbuf, N, t);
    pt_to_affine(CS, P);
    apt_to_xy64(w.proof_head + 256 * t + 192, CS);
    strobe tr;
    ws_ld_strobe(tr, w.tstate, N, t);
    app_point(tr, "commitment_cs", CS);            // circuit.rs:488
    sc tau, one, zero, t1, t2;
    sc_set_u32(one, 1);
    sc_set_u32(zero, 0);
    if (!t_get_challenge(tr, "circuit_tau", tau)) { w.status[t] |= ST_DEGENERATE; tau = one; }
    ws_st_strobe(w.tstate, N, t, tr);
    if (sc_is_zero(tau)) { w.status[t] |= ST_DEGENERATE; tau = one; }
    sc rho, beta, delta, mu, v0, rv0, delta_inv, tau_inv, tau2, tau3, t3di, two_tau3;
    cp_ld(rho, w.misc, w, t, CM_RHO); cp_ld(beta, w.misc, w, t, CM_BETA); cp_ld(delta, w.misc, w, t, CM_DELTA); cp_ld(mu, w.misc, w, t, CM_MU);
    cp_ld(v0, w.misc, w, t, CM_V0); cp_ld(rv0, w.misc, w, t, CM_RV0); cp_ld(delta_inv, w.misc, w, t, CM_DINV);
    sc_inv(tau_inv, tau);
    sc_mul(tau2, tau, tau);
    sc_mul(tau3, tau2, tau);
    sc_mul(t3di, tau3, delta_inv);
    sc_add(two_tau3, tau3, tau3);
    u32* m = w.msc;                                // set 0
    uint8_t* wl = w.wn_l + (size_t)t * w.NH * 32;
    uint8_t* wn = w.wn_n + (size_t)t * w.NG * 32;
    uint8_t* cw = w.wn_c + (size_t)t * w.NH * 32;
    // l = (r_s || l_s) tau^-1 - (r_o || l_o) delta + (r_l || l_l) tau - (r_r || l_r) tau^2 + (r_v || v_1) tau^3      (circuit.rs:497-501)
#pragma nounroll
    for (int i = 0; i < 9 + nv; i++) {
        sc a, b, c, d, e = zero, lv;
        if (i < 9) {
            cp_ld(a, w.rs, w, t, i); cp_ld(b, w.ro, w, t, i); cp_ld(c, w.rl, w, t, i); cp_ld(d, w.rr, w, t, i);
            if (i == 0) e = rv0;
        } else {
            const int j = i - 9;
            cp_ld(a, w.ls, w, t, j); cp_ld(b, w.lo, w, t, j); cp_ld(c, w.ll, w, t, j); cp_ld(d, w.lr, w, t, j);
        }
        // (r_v || v_1) has 9 + (nv - 1) entries: v_1[j] sits at index 9 + j
        if (i >= 9 && i - 9 + 1 < nv) cp_ld(e, w.v1, w, t, i - 9);
        sc_mul(lv, a, tau_inv);
        sc_mul(t1, b, delta); sc_sub(lv, lv, t1);
        sc_mul(t1, c, tau); sc_add(lv, lv, t1);
        sc_mul(t1, d, tau2); sc_sub(lv, lv, t1);
        sc_mul(t1, e, tau3); sc_add(lv, lv, t1);
        sc_to_be(wl + (size_t)i * 32, lv);
        ws_st8(m, N, t, 1 + w.NG + i, lv.v);
    }
#pragma nounroll
    for (int i = 9 + nv; i < w.NH; i++) sc_to_be(wl + (size_t)i * 32, zero);
    // n = pn_tau + n_tau, ps_tau                                                                   (circuit.rs:503-520)
    sc ps = zero, mp = mu;
#pragma nounroll
    for (int j = 0; j < nm; j++) {
        sc cL, cR, cO, pn, ns, no, nlj, nr, nt;
        cp_ld(cL, w.coef, w, t, j); cp_ld(cR, w.coef, w, t, nm + j); cp_ld(cO, w.coef, w, t, 2 * nm + j);
        sc_mul(pn, cO, t3di);
        sc_mul(t1, cL, tau2); sc_sub(pn, pn, t1);
        sc_mul(t1, cR, tau); sc_add(pn, pn, t1);
        sc_mul(t1, pn, pn); sc_mul(t1, t1, mp); sc_add(ps, ps, t1);
        cp_ld(ns, w.ns, w, t, j); cp_ld(no, w.no, w, t, j); cp_ld(nlj, w.nl, w, t, j); cp_ld(nr, w.nr, w, t, j);
        sc_mul(nt, ns, tau_inv);
        sc_mul(t1, no, delta); sc_sub(nt, nt, t1);
        sc_mul(t1, nlj, tau); sc_add(nt, nt, t1);
        sc_mul(t1, nr, tau2); sc_sub(nt, nt, t1);
        sc_add(nt, nt, pn);
        sc_to_be(wn + (size_t)j * 32, nt);
        ws_st8(m, N, t, 1 + j, nt.v);
        sc_mul(mp, mp, mu);
    }
#pragma nounroll
    for (int j = nm; j < w.NG; j++) sc_to_be(wn + (size_t)j * 32, zero);
    sc dl = zero, dm = zero;
#pragma nounroll
    for (int i = 0; i < nl; i++) { sc x, a; ws_ld8(x.v, w.lamv, N, t, i); cd_ld_sc(a, cd.a_l, i); sc_mul(x, x, a); sc_add(dl, dl, x); }
#pragma nounroll
    for (int i = 0; i < nm; i++) { sc x, a; ws_ld8(x.v, w.muv, N, t, i); cd_ld_sc(a, cd.a_m, i); sc_mul(x, x, a); sc_add(dm, dm, x); }
    sc_sub(t1, dl, dm);
    sc_mul(t1, t1, two_tau3);
    sc_add(ps, ps, t1);
    sc vv;
    sc_mul(vv, tau3, v0);
    sc_add(vv, vv, ps);                            // circuit.rs:540
    ws_st8(m, N, t, 0, vv.v);
    // c = cr_tau || cl_tau, zero-extended (circuit.rs:522-538, 544-551)
    sc_to_be(cw, one);
    sc_mul(t1, beta, tau_inv);
    sc_to_be(cw + 32, t1);
    sc bt = beta;
#pragma nounroll
    for (int i = 2; i < 9; i++) { sc_mul(bt, bt, tau); sc_to_be(cw + (size_t)i * 32, bt); }
#pragma nounroll
    for (int j = 0; j < nv; j++) {
        sc lL, lR, lO, cl;
        cp_ld(lL, w.coef, w, t, 3 * nm + j); cp_ld(lR, w.coef, w, t, 3 * nm + nv + j); cp_ld(lO, w.coef, w, t, 3 * nm + 2 * nv + j);
        sc_mul(cl, lO, t3di);
        sc_mul(t1, lL, tau2); sc_sub(cl, cl, t1);
        sc_mul(t1, lR, tau); sc_add(cl, cl, t1);
        sc_add(cl, cl, cl);
        if (j + 1 < nv) { sc c0; cp_ld(c0, w.cl0, w, t, j); sc_sub(cl, cl, c0); }
        sc_to_be(cw + (size_t)(9 + j) * 32, cl);
    }
#pragma nounroll
    for (int i = 9 + nv; i < w.NH; i++) sc_to_be(cw + (size_t)i * 32, zero);
    sc_to_be(w.wn_rho + 32 * t, rho);
    sc_to_be(w.wn_mu + 32 * t, mu);
    (void)t2;
}
// ---- stage D: the WNLA commitment to affine, for the WNLA prover's first transcript append
HD void circuit_prove_stage_d(const CircuitProveWs& w, size_t t) {
    pt P;
    apt C;
    ws_ld_pt(P, w.pbuf, w.N, t);
    pt_to_affine(C, P);
    apt_to_xy64(w.wn_commit + 64 * t, C);
}

}  // namespace bppp
