// Generic batched `ReciprocalRangeProofProtocol::verify` (reciprocal.rs:98-107 -> circuit.rs:154-256) for runtime dim_nd / dim_np
// (dim_np <= dim_nd + 1): dim_nd = dim_np = 16 is the u64 protocol (served by the specialised kernels of verify_core.h);
// dim_nd = 256, dim_np = 16 is the "aggregated" shape of BASELINE configs[4] (|g_vec| = 256, |h_vec| = 512, 8 WNLA rounds).
//
// This stage does reciprocal + circuit: transcript to tau, the closed-form scalars (make_circuit / collect_c collapse exactly
// as for u64, with S = sum_{i=1..nd} lambda^i, c_nL[j] = -np^j mu^-(j+1), c_nR[j] = (S - lambda^(j+1)) mu^-(j+1) + e,
// c_lL[j] = -S (e+j)^-1 for j < np and 0 above), and the circuit commitment C0.  It then hands C0, c, rho, mu and the
// transcript to the generic WNLA stage (wnla_core.h) as ordinary C-ABI-layout device buffers.
//
// Generic proof layout (per instance, `proof_bytes` = 64 (5 + 2 rounds) + 32 (nl + nn)):
//   c_l, c_r, c_o, c_s | r[rounds] | x[rounds] | reciprocal r | l[nl] | n[nn]
#pragma once
#include "wnla_core.h"

namespace bppp {

struct RecipWs {
    size_t N;
    int nd, np, rounds, nl, nn;
    int NG, NH;                      // lengths of g_vec || g_vec_ and h_vec || h_vec_ (the WNLA generator vectors)
    size_t proof_bytes;
    const uint8_t* commitments;      // N x 64
    const uint8_t* proofs;           // N x proof_bytes
    int32_t* status;
    u32* tstate;                     // [52][N]
    u32* sc0;                        // [(1 + nd + 5) * 8][N]: ps_tau | pn_tau[nd] | tau^-1, -delta, tau, -tau^2, 2 tau^3
    u32* pts;                        // [5 * 16][N] packed affine: c_s, c_o, c_l, c_r, V + r
    u32* acc;                        // [30][N]
    u32* pfix;                       // [30][N]
    u32* inv;                        // [np * 8][N]: (e + j)^-1
    pt_slot* straus;                 // [N][5][9]
    // fast variable-base path (as the WNLA rounds': wnla_core.h): window tables of the five points, slots atab_first .. + 5 x 16 of the
    // WNLA stage's table buffer (entry-major, atab_of); null = the Jacobian-table Straus sum on `straus`
    apt_packed* atab;
    u32* tscr;
    int atab_first;
    // outputs for the WNLA stage (C-ABI layouts)
    uint8_t* wn_commit;              // N x 64
    uint8_t* wn_c;                   // N x NH x 32
    uint8_t* wn_rho;                 // N x 32
    uint8_t* wn_mu;                  // N x 32
    FbTable fb;                      // bases: 0 g | 1..NG g_vec||g_vec_ | NG+1.. h_vec||h_vec_
    strobe base;
    TranscriptIo tio;                // caller's transcripts (reciprocal.rs:98 `t: &mut Transcript`)
};

// a^e for a public exponent (square and multiply; e = 0 gives 1)
HD void sc_pow_u32(sc& r, const sc& a, u32 e) {
    sc base = a;
    sc_set_u32(r, 1);
#pragma nounroll
    while (e) {
        if (e & 1u) sc_mul(r, r, base);
        e >>= 1;
        if (e) sc_mul(base, base, base);
    }
}
// q, G: lane q of the G (2, 4 or 8) consecutive lanes that run phase 1 for instance t together -- calls whose one-lane kernels leave
// wavefront slots free (k_recip_phase1_grp).  Decode, transcript, challenges and the np + 2 inversions are done by all lanes alike
// (identical values, identical stores); the two loops over the dim_nd digits -- 11 dependent multiplications per digit, 97 % of the
// kernel for configs[4]'s 256 -- are cut into G runs of consecutive digits: a lane starts its run from powers it raises itself
// (sc_pow_u32), the three sums meet by shuffles, lane 0 stores what belongs to the instance.  Every lane of a group must be active.
// G = 1: the one-lane form (also the host emulation's).
HD void recip_phase1(const RecipWs& w, size_t t, int q = 0, int G = 1) {
    const size_t N = w.N;
    const int nd = w.nd, np = w.np;
    int32_t status = ST_OK;
    const uint8_t* pv = w.commitments + 64 * t;
    const uint8_t* pp = w.proofs + w.proof_bytes * t;
    apt V, CL, CR, CO, CS, RR;
    bool ok = apt_from_xy64(V, pv);
    ok &= apt_from_xy64(CL, pp) & apt_from_xy64(CR, pp + 64) & apt_from_xy64(CO, pp + 128) & apt_from_xy64(CS, pp + 192);
    ok &= apt_from_xy64(RR, pp + 256 + (size_t)128 * w.rounds);
    if (!ok) {
        status |= ST_BAD_ENCODING;
        apt z;
        fe_set_u32(z.x, 0); fe_set_u32(z.y, 0);
        V = z; CL = z; CR = z; CO = z; CS = z; RR = z;
    }
    strobe tr;
    tio_begin(tr, status, w.tio, w.base, t);
    sc e, rho, lambda, beta, delta, tau;
    app_point(tr, "reciprocal_commitment", V);
    bool cok = t_get_challenge(tr, "reciprocal_challenge", e);
    apt Vr;
    {
        pt s;
        pt_from_affine(s, V);
        pt_madd(s, s, RR, apt_is_identity(RR));
        pt_to_affine(Vr, s);
    }
    app_point(tr, "commitment_cl", CL);
    app_point(tr, "commitment_cr", CR);
    app_point(tr, "commitment_co", CO);
    app_point(tr, "commitment_v", Vr);
    cok &= t_get_challenge(tr, "circuit_rho", rho);
    cok &= t_get_challenge(tr, "circuit_lambda", lambda);
    cok &= t_get_challenge(tr, "circuit_beta", beta);
    cok &= t_get_challenge(tr, "circuit_delta", delta);
    app_point(tr, "commitment_cs", CS);
    cok &= t_get_challenge(tr, "circuit_tau", tau);
    if (!cok) {
        status |= ST_DEGENERATE;
        sc_set_u32(e, 1); sc_set_u32(rho, 1); sc_set_u32(lambda, 1); sc_set_u32(beta, 1); sc_set_u32(delta, 1); sc_set_u32(tau, 1);
    }
    ws_st_strobe(w.tstate, N, t, tr);
    ws_st_apt(w.pts, N, t, 0, CS); ws_st_apt(w.pts, N, t, 1, CO); ws_st_apt(w.pts, N, t, 2, CL); ws_st_apt(w.pts, N, t, 3, CR);
    ws_st_apt(w.pts, N, t, 4, Vr);
    sc mu, one, zero, t1, t2;
    sc_set_u32(one, 1);
    sc_set_u32(zero, 0);
    sc_mul(mu, rho, rho);
    sc_to_be(w.wn_rho + 32 * t, rho);
    sc_to_be(w.wn_mu + 32 * t, mu);
    // inverses of mu, tau, e + j (j < np) with one Fn inversion: running products forward, peel backwards (values kept in HBM)
    bool zero_inv = sc_is_zero(delta) | sc_is_zero(mu) | sc_is_zero(tau);
    sc prod, m_ = sc_is_zero(mu) ? one : mu, t_ = sc_is_zero(tau) ? one : tau;
    sc_mul(prod, m_, t_);
#pragma nounroll
    for (int j = 0; j < np; j++) {
        sc js, a;
        sc_set_u32(js, (u32)j);
        sc_add(a, e, js);
        if (sc_is_zero(a)) { zero_inv = true; a = one; }
        ws_st8(w.inv, N, t, j, prod.v);          // prefix product BEFORE a_j
        sc_mul(prod, prod, a);
    }
    if (zero_inv) status |= ST_DEGENERATE;
    sc inv;
    sc_inv(inv, prod);
#pragma nounroll
    for (int j = np - 1; j >= 0; j--) {
        sc js, a, pre, aj;
        sc_set_u32(js, (u32)j);
        sc_add(a, e, js);
        if (sc_is_zero(a)) a = one;
        ws_ld8(pre.v, w.inv, N, t, j);
        sc_mul(aj, inv, pre);                    // (e + j)^-1
        sc_mul(inv, inv, a);
        ws_st8(w.inv, N, t, j, aj.v);
    }
    // inv = (mu tau)^-1
    sc mu_inv, tau_inv;
    sc_mul(mu_inv, inv, t_);
    sc_mul(tau_inv, inv, m_);
    sc tau2, tau3;
    sc_mul(tau2, tau, tau);
    sc_mul(tau3, tau2, tau);
    // this lane's run of digits: [ja, jb)
    const int seg = (nd + G - 1) / G, ja = q * seg < nd ? q * seg : nd, jb = ja + seg < nd ? ja + seg : nd;
    // S = sum_{i=1..nd} lambda^i, musum = sum_{i=1..nd} mu^i  (the run's terms lambda^(ja+1) .. lambda^jb, then the sum over the group)
    sc S, lp, mp, musum, mip, pw;
    if (ja == 0) { lp = lambda; mp = mu; }
    else { sc_pow_u32(lp, lambda, (u32)ja + 1); sc_pow_u32(mp, mu, (u32)ja + 1); }
    S = lp;
    musum = mp;
    if (ja >= jb) { S = zero; musum = zero; }
#pragma nounroll
    for (int i = ja + 1; i < jb; i++) { sc_mul(lp, lp, lambda); sc_add(S, S, lp); sc_mul(mp, mp, mu); sc_add(musum, musum, mp); }
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    if (G > 1) { sc_group_sum16(S, G); sc_group_sum16(musum, G); }
#endif
    sc tau_e, two_tau2_S, ps, base_np;
    sc_mul(tau_e, tau, e);
    sc_mul(two_tau2_S, tau2, S);
    sc_add(two_tau2_S, two_tau2_S, two_tau2_S);
    sc_set_u32(ps, 0);
    sc_set_u32(base_np, (u32)np);
    // np^j, mu^-(j+1), lambda^(j+1), mu^(j+1) at the run's first digit
    if (ja == 0) { pw = one; mip = mu_inv; lp = lambda; mp = mu; }
    else {
        sc_pow_u32(pw, base_np, (u32)ja);
        sc_pow_u32(mip, mu_inv, (u32)ja + 1);
        sc_pow_u32(lp, lambda, (u32)ja + 1);
        sc_pow_u32(mp, mu, (u32)ja + 1);
    }
    uint8_t* cw = w.wn_c + (size_t)t * w.NH * 32;
#pragma nounroll
    for (int j = ja; j < jb; j++) {
        sc pn, cl;
        sc_mul(t1, tau2, pw);
        sc_sub(t2, S, lp);
        sc_mul(t2, t2, tau);
        sc_add(t1, t1, t2);
        sc_mul(pn, t1, mip);
        sc_add(pn, pn, tau_e);
        ws_st8(w.sc0, N, t, 1 + j, pn.v);
        sc_mul(t1, pn, pn); sc_mul(t1, t1, mp); sc_add(ps, ps, t1);
        // cl_tau[j] = 2 tau^2 S (e+j)^-1 [j < np] - lambda^(j+1)
        cl = zero;
        if (j < np) { sc ej; ws_ld8(ej.v, w.inv, N, t, j); sc_mul(cl, two_tau2_S, ej); }
        sc_sub(cl, cl, lp);
        sc_to_be(cw + (size_t)(9 + j) * 32, cl);
        sc_mul(mip, mip, mu_inv);
        sc_mul(lp, lp, lambda);
        sc_mul(mp, mp, mu);
        sc_mul(pw, pw, base_np);
    }
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    if (G > 1) sc_group_sum16(ps, G);
#endif
    if (q != 0) return;          // (what follows belongs to the instance: lane 0 of the group)
    // (dim_np > dim_nd would leave c_lL entries beyond the lambda powers: dim_np <= dim_nd + 1 is required by the host)
    sc two_tau3;
    sc_add(two_tau3, tau3, tau3);
    sc_mul(t1, two_tau3, musum);
    sc_sub(ps, ps, t1);
    ws_st8(w.sc0, N, t, 0, ps.v);
    ws_st8(w.sc0, N, t, 1 + nd, tau_inv.v);
    sc_neg(t1, delta);
    ws_st8(w.sc0, N, t, 2 + nd, t1.v);
    ws_st8(w.sc0, N, t, 3 + nd, tau.v);
    sc_neg(t1, tau2);
    ws_st8(w.sc0, N, t, 4 + nd, t1.v);
    ws_st8(w.sc0, N, t, 5 + nd, two_tau3.v);
    // c = cr_tau (9) || cl_tau (nv = nd + 1, last entry 0) || zeros up to NH
    sc_to_be(cw, one);
    sc_mul(t1, beta, tau_inv);
    sc_to_be(cw + 32, t1);
    sc bt = beta;
#pragma nounroll
    for (int i = 2; i < 9; i++) { sc_mul(bt, bt, tau); sc_to_be(cw + (size_t)i * 32, bt); }
#pragma nounroll
    for (int i = 9 + nd; i < w.NH; i++) sc_to_be(cw + (size_t)i * 32, zero);
    w.status[t] = status;
}
// C0 fixed-base half: ps_tau g + <g_vec, pn_tau>  (bases 0..nd of the table)
HD void recip_c0_fixed_ranges(FbRanges& rg, const RecipWs& w) { fb_ranges_one(rg, 0, 0, 1 + w.nd); }
HD void recip_c0_fixed_store(const RecipWs& w, size_t t, const pt& total) { ws_st_pt(w.pfix, w.N, t, total); }
// C0 variable-base half + sum -> affine C0 for the WNLA stage (which hashes it first thing, wnla.rs:88)
// window tables of c_s, c_o, c_l, c_r, V + r (fast path; one inversion per instance)
HD void recip_c0_tables(const RecipWs& w, size_t t) {
    affine_tables_build(atab_of(w.atab, w.N, t) + w.atab_first, w.tscr, w.pts, w.N, t, 5);
}
// group_lane >= 0: one of group_size (2 or 4) consecutive lanes that all run the sum for instance t (straus_core.h: straus_affine_g4)
HD void recip_c0_var(const RecipWs& w, size_t t, int group_lane = -1, int group_size = 4) {
    const size_t N = w.N;
    pt acc;
    if (w.atab) {
        const int pslot[5] = {0, 1, 2, 3, 4};
        glv_words<5> g;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            sc k;
            ws_ld8(k.v, w.sc0, N, t, 1 + w.nd + j);
            glv_split sp;
            glv_decompose(sp, k);
            glv_words_set<5>(g, j, sp);
        }
        const atab_ref tab = atab_of(w.atab, N, t) + w.atab_first;
#if defined(__HIP_DEVICE_COMPILE__)
        if (group_lane >= 0 && group_size == 4) straus_affine_g4<5, 4>(acc, tab, pslot, g, group_lane);
        else if (group_lane >= 0) straus_affine_g4<5, 2>(acc, tab, pslot, g, group_lane);
        else
#endif
            straus_affine<5>(acc, tab, pslot, g);
        (void)group_lane; (void)group_size;
    } else {
        pt_slot* tbl = w.straus + t * (5 * BPPP_STRAUS_ENTRIES);
        glv_split rs[5];
#pragma nounroll
        for (int j = 0; j < 5; j++) {
            apt P;
            ws_ld_apt(P, w.pts, N, t, j);
            straus_build_table(tbl + j * BPPP_STRAUS_ENTRIES, P);
            sc k;
            ws_ld8(k.v, w.sc0, N, t, 1 + w.nd + j);
            glv_decompose(rs[j], k);
        }
        straus_msm_glv(acc, tbl, rs, 5);
    }
    ws_st_pt(w.acc, N, t, acc);
}
HD void recip_c0_finish(const RecipWs& w, size_t t) {
    pt a, f;
    ws_ld_pt(a, w.acc, w.N, t);
    ws_ld_pt(f, w.pfix, w.N, t);
    pt_add(a, a, f);
    apt c0;
    pt_to_affine(c0, a);
    apt_to_xy64(w.wn_commit + 64 * t, c0);
}

}  // namespace bppp
