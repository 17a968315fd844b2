// Batch u64 range-proof verification, exact per-proof mode: the per-proof work of
// `U64RangeProofProtocol::verify` (u64_proof.rs:42-54) -> `ReciprocalRangeProofProtocol::verify`
// (reciprocal.rs:98-107) -> `ArithmeticCircuit::verify` (circuit.rs:154-256) -> `WeightNormLinearArgument::verify`
// (wnla.rs:75-121), restructured for one lane per proof over a batch that shares one generator set.
//
// What is restructured (same group elements, same transcript bytes, same accept bit -- SURVEY.md 8a closed forms,
// checked numerically against the dense reference-shaped oracle in tests/):
//  * make_circuit + collect_c's dense 16x48 / 17x48 matrices (reciprocal.rs:150-214, circuit.rs:584-653) collapse to
//    closed forms for pn_tau / ps_tau / c; the 256 recomputed inversions become ONE Fn inversion (Montgomery trick).
//  * circuit.rs:206,230-235 (22 scalar multiplications) -> one 17-term fixed-base MSM over batch-shared tables plus
//    one 5-point shared-doubling (Straus) multi-scalar multiplication.
//  * each WNLA round's generator folding (wnla.rs:96-97, 68 scalar multiplications over 4 rounds) is NOT executed:
//    folded generators only matter in the base case (wnla.rs:80-82), where they unroll to a 49-term fixed-base MSM
//    over the ORIGINAL generators.  Only com_ = com + y*X + (y^2-1)*R (wnla.rs:100-102) runs per round, because the
//    next challenge hashes it (wnla.rs:88).
//
// Data layout in HBM (workspace): structure-of-arrays, limb-major -- word (slot*8 + limb) of proof t sits at
// base[(slot*8 + limb) * N + t], so a wavefront's 64 lanes read 256 contiguous bytes per load.
//
// This header holds the protocol phases (phase 1, C0, the WNLA rounds, the base case, the wire format); the workspace is verify_ws.h, the
// fixed-base sums fb_core.h, the variable-base sums straus_core.h.
#pragma once
#include "verify_ws.h"
#include "fb_core.h"
#include "straus_core.h"

namespace bppp {

// ---------------------------------------------------------------- phase 1: decode, transcript up to tau, scalar derivation
// reciprocal.rs:98-104 + circuit.rs:155-228 (closed forms of SURVEY.md 8a)
// TR = strobe (sponge state in registers: host emulation, small batches) or strobe_lds (state in the workgroup's LDS: k_verify_phase1).
// `tr` arrives holding the transcript every proof starts from (ws.base); status_in carries flags the caller already raised.
#if defined(__HIPCC__)
__device__ __forceinline__ void sc_group_sum16(sc& a, int group = 16);      // the sum over a group of lanes, on every lane (shuffles; below)
#endif
// a^e for 1 <= e <= 31, the same ten multiplications whatever e is (the lane forms: a lane's power of mu, lambda, ... directly)
HD void sc_pow_u5(sc& r, const sc& a, unsigned e) {      // a^e for 1 <= e <= 31, the same ten multiplications whatever e is
    sc acc, tmp;
    sc_set_u32(acc, 1);
#pragma unroll
    for (int bit = 4; bit >= 0; bit--) {
        sc_mul(acc, acc, acc);
        sc_mul(tmp, acc, a);
        const bool take = ((e >> bit) & 1u) != 0;
#pragma unroll
        for (int i = 0; i < 8; i++) acc.v[i] = take ? tmp.v[i] : acc.v[i];
    }
    r = acc;
}
// lane >= 0: one of SIXTEEN lanes that run phase 1 for proof t together (small calls: k_verify_phase1_g16).  Decode, transcript and
// challenges are done by all sixteen alike (identical values, identical stores); the scalar section -- a chain of ~245 dependent
// multiplications and 36 workspace round trips on one lane -- is spread: lane j inverts e + j itself (and every lane mu tau), takes term
// j of the 16-term loop with its powers by sc_pow_u5, and the three sums meet by shuffles.  Every lane of a group must be active.
template <typename TR>
HD void verify_phase1_on(const VerifyWs& ws, size_t t, TR& tr, int32_t status_in, int lane = -1) {
    const size_t N = ws.N;
    int32_t status = status_in;
    BPPP_STAMP(t, 0);
    const uint8_t* pv = ws.commitments + 64 * t;
    const uint8_t* pp = ws.proofs + (size_t)BPPP_U64_PROOF_BYTES * t;
    // The 13 proof points are decoded one at a time and go straight to the workspace (a 13-point array would live in scratch
    // memory); the transcript below reloads each one right before it is hashed.  Only V and proof.r (needed for V + r) stay in
    // registers.
    apt V, Pr;
    bool ok = apt_from_xy64(V, pv);
#pragma nounroll
    for (int i = 0; i < 12; i++) {
        apt Q;
        ok &= apt_from_xy64(Q, pp + 64 * i);
        ws_st_apt(ws.pts, N, t, i, Q);
    }
    ok &= apt_from_xy64(Pr, pp + 64 * 12);
    sc l0, l1, n0;
    ok &= sc_from_be(l0, pp + 832);
    ok &= sc_from_be(l1, pp + 864);
    ok &= sc_from_be(n0, pp + 896);
    if (!ok) {
        // keep control flow uniform: run on harmless values, the status forces accept = 0 at the end
        status |= ST_BAD_ENCODING;
        apt zero;
        fe_set_u32(zero.x, 0);
        fe_set_u32(zero.y, 0);
        V = zero;
        Pr = zero;
#pragma nounroll
        for (int i = 0; i < 12; i++) ws_st_apt(ws.pts, N, t, i, zero);
        sc_set_u32(l0, 0); sc_set_u32(l1, 0); sc_set_u32(n0, 0);
    }
    sc e, rho, lambda, beta, delta, tau;
    BPPP_STAMP(t, 1);
    app_point(tr, "reciprocal_commitment", V);                          // reciprocal.rs:99
    bool cok = t_get_challenge(tr, "reciprocal_challenge", e);          // reciprocal.rs:100
    // circuit_commitment = commitment + proof.r                         (reciprocal.rs:104)
    apt Vr;
    {
        pt s;
        pt_from_affine(s, V);
        pt_madd(s, s, Pr, apt_is_identity(Pr));
        pt_to_affine(Vr, s);
    }
    ws_st_apt(ws.pts, N, t, 12, Vr);
    BPPP_STAMP(t, 2);
    {
        apt Q;
        ws_ld_apt(Q, ws.pts, N, t, 0);
        app_point(tr, "commitment_cl", Q);                               // circuit.rs:155-159
        ws_ld_apt(Q, ws.pts, N, t, 1);
        app_point(tr, "commitment_cr", Q);
        ws_ld_apt(Q, ws.pts, N, t, 2);
        app_point(tr, "commitment_co", Q);
    }
    app_point(tr, "commitment_v", Vr);
    BPPP_STAMP(t, 3);
    cok &= t_get_challenge(tr, "circuit_rho", rho);                      // circuit.rs:161-164
    cok &= t_get_challenge(tr, "circuit_lambda", lambda);
    cok &= t_get_challenge(tr, "circuit_beta", beta);
    cok &= t_get_challenge(tr, "circuit_delta", delta);
    BPPP_STAMP(t, 4);
    {
        apt Q;
        ws_ld_apt(Q, ws.pts, N, t, 3);
        app_point(tr, "commitment_cs", Q);                               // circuit.rs:189
    }
    cok &= t_get_challenge(tr, "circuit_tau", tau);                      // circuit.rs:191
    if (!cok) {
        status |= ST_DEGENERATE;
        sc_set_u32(e, 1); sc_set_u32(rho, 1); sc_set_u32(lambda, 1); sc_set_u32(beta, 1); sc_set_u32(delta, 1); sc_set_u32(tau, 1);
    }
    ws_st_transcript(ws.tstate, N, t, tr);
    ws_st8(ws.chal, N, t, 0, e.v); ws_st8(ws.chal, N, t, 1, rho.v); ws_st8(ws.chal, N, t, 2, lambda.v);
    ws_st8(ws.chal, N, t, 3, beta.v); ws_st8(ws.chal, N, t, 4, delta.v); ws_st8(ws.chal, N, t, 5, tau.v);
    ws_st8(ws.lns, N, t, 0, l0.v); ws_st8(ws.lns, N, t, 1, l1.v); ws_st8(ws.lns, N, t, 2, n0.v);

    // ---- scalars.  One Fn inversion for {mu, tau, e+0..e+15} (the reference: util.rs:119, circuit.rs:192, reciprocal.rs:181 x256)
    sc mu;
    BPPP_STAMP(t, 5);
    sc_mul(mu, rho, rho);                                                // circuit.rs:166
    // Montgomery's trick over the 18 values a_0 = mu, a_1 = tau, a_{2+j} = e + j.  The values are recomputed where needed and the
    // running products / the inverses travel through workspace slots that are free at this point (cvec: products, fsc:
    // inverses) instead of two 18-element arrays in scratch memory.
    auto batch_value = [&](int i, sc& v) {
        if (i == 0) v = mu;
        else if (i == 1) v = tau;
        else {
            sc js;
            sc_set_u32(js, (u32)(i - 2));
            sc_add(v, e, js);
        }
    };
    bool zero_inv = sc_is_zero(delta);                                   // circuit.rs:196 unwraps delta^-1 although u64 never uses it
    sc one;
    sc_set_u32(one, 1);
    sc mu_inv, tau_inv, tau2, tau3, S, t1, t2, musum, ps;
    sc_mul(tau2, tau, tau);
    sc_mul(tau3, tau2, tau);
#if defined(__HIP_DEVICE_COMPILE__)
    if (lane >= 0) {
        // inverses: (mu tau)^-1 on every lane, (e + lane)^-1 on its lane; a zero is replaced by one, as in the one-lane form below
        sc m1 = mu, tq = tau, ej, js, inv;
        bool z = sc_is_zero(mu);
        zero_inv |= z;
        if (z) m1 = one;
        z = sc_is_zero(tau);
        zero_inv |= z;
        if (z) tq = one;
        sc_mul(t1, m1, tq);
        sc_inv(inv, t1);
        sc_mul(mu_inv, inv, tq);
        sc_mul(tau_inv, inv, m1);
        sc_set_u32(js, (u32)lane);
        sc_add(ej, e, js);
        z = sc_is_zero(ej);
        if (z) ej = one;
        int zany = z ? 1 : 0;
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) zany |= __shfl_xor(zany, m, 64);
        zero_inv |= zany != 0;
        if (zero_inv) status |= ST_DEGENERATE;
        sc einv;
        sc_inv(einv, ej);
        ws_st8(ws.fsc, N, t, lane, einv.v);                  // (e + j)^-1 at fsc slot j
        BPPP_STAMP(t, 6);
        // this lane's powers; S = sum lambda^i, musum = sum mu^i over the group
        sc lp, mp, mip;
        sc_pow_u5(lp, lambda, (unsigned)lane + 1);
        sc_pow_u5(mp, mu, (unsigned)lane + 1);
        sc_pow_u5(mip, mu_inv, (unsigned)lane + 1);
        S = lp;
        musum = mp;
        sc_group_sum16(S);
        sc_group_sum16(musum);
        sc tau_e, two_tau2_S, p16, pn;
        sc_mul(tau_e, tau, e);
        sc_mul(two_tau2_S, tau2, S);
        sc_add(two_tau2_S, two_tau2_S, two_tau2_S);
        // term j = lane of the loop below
        sc_set_u64(p16, (u64)1 << (4 * lane));
        sc_mul(t1, tau2, p16);
        sc_sub(t2, S, lp);
        sc_mul(t2, t2, tau);
        sc_add(t1, t1, t2);
        sc_mul(pn, t1, mip);
        sc_add(pn, pn, tau_e);
        ws_st8(ws.sc0, N, t, 1 + lane, pn.v);
        sc_mul(ps, pn, pn);
        sc_mul(ps, ps, mp);
        sc_group_sum16(ps);
        sc_mul(t1, two_tau2_S, einv);
        sc_sub(t1, t1, lp);
        ws_st8(ws.cvec, N, t, 9 + lane, t1.v);
    } else
#endif
    {
        sc run = one;
#pragma nounroll
        for (int i = 0; i < 18; i++) {
            sc v;
            batch_value(i, v);
            const bool z = sc_is_zero(v);
            zero_inv |= z;
            if (z) v = one;
            ws_st8(ws.cvec, N, t, i, run.v);      // product of a_0 .. a_{i-1}
            sc_mul(run, run, v);
        }
        if (zero_inv) status |= ST_DEGENERATE;
        sc inv;
        sc_inv(inv, run);
#pragma nounroll
        for (int i = 17; i >= 0; i--) {
            sc v, pre, ai;
            batch_value(i, v);
            if (sc_is_zero(v)) v = one;
            ws_ld8(pre.v, ws.cvec, N, t, i);
            sc_mul(ai, inv, pre);                 // a_i^-1
            sc_mul(inv, inv, v);
            if (i >= 2) ws_st8(ws.fsc, N, t, i - 2, ai.v);      // (e + j)^-1 at fsc slot j
            else if (i == 1) tau_inv = ai;
            else mu_inv = ai;
        }
        BPPP_STAMP(t, 6);
        // S = sum_{i=1..16} lambda^i ; musum = sum_{i=1..16} mu^i
        sc lp = lambda, mp = mu;
        S = lambda;
        musum = mu;
#pragma nounroll
        for (int i = 1; i < 16; i++) {
            sc_mul(lp, lp, lambda);
            sc_add(S, S, lp);
            sc_mul(mp, mp, mu);
            sc_add(musum, musum, mp);
        }
        sc tau_e, two_tau2_S;
        sc_mul(tau_e, tau, e);
        sc_mul(two_tau2_S, tau2, S);
        sc_add(two_tau2_S, two_tau2_S, two_tau2_S);
        sc_set_u32(ps, 0);
        sc mip = mu_inv;   // mu^-(j+1)
        lp = lambda;       // lambda^(j+1)
        mp = mu;           // mu^(j+1)
#pragma nounroll
        for (int j = 0; j < 16; j++) {
            // pn_tau[j] = mu^-(j+1) (tau^2 16^j + tau (S - lambda^(j+1))) + tau e          (circuit.rs:198-200)
            sc p16, pn;
            sc_set_u64(p16, (u64)1 << (4 * j));
            sc_mul(t1, tau2, p16);
            sc_sub(t2, S, lp);
            sc_mul(t2, t2, tau);
            sc_add(t1, t1, t2);
            sc_mul(pn, t1, mip);
            sc_add(pn, pn, tau_e);
            ws_st8(ws.sc0, N, t, 1 + j, pn.v);
            // ps_tau += mu^(j+1) pn^2                                                       (circuit.rs:202)
            sc_mul(t1, pn, pn);
            sc_mul(t1, t1, mp);
            sc_add(ps, ps, t1);
            // cl_tau[j] = 2 tau^2 S (e+j)^-1 - lambda^(j+1)                                 (circuit.rs:222-226)
            sc einv;
            ws_ld8(einv.v, ws.fsc, N, t, j);
            sc_mul(t1, two_tau2_S, einv);
            sc_sub(t1, t1, lp);
            ws_st8(ws.cvec, N, t, 9 + j, t1.v);
            sc_mul(mip, mip, mu_inv);
            sc_mul(lp, lp, lambda);
            sc_mul(mp, mp, mu);
        }
    }
    // ps_tau -= 2 tau^3 sum mu^i   (a_l = 0, a_m = 1s; circuit.rs:203-204)
    sc two_tau3;
    sc_add(two_tau3, tau3, tau3);
    sc_mul(t1, two_tau3, musum);
    sc_sub(ps, ps, t1);
    ws_st8(ws.sc0, N, t, 0, ps.v);
    ws_st8(ws.sc0, N, t, 17, tau_inv.v);
    sc_neg(t1, delta);
    ws_st8(ws.sc0, N, t, 18, t1.v);
    ws_st8(ws.sc0, N, t, 19, tau.v);
    sc_neg(t1, tau2);
    ws_st8(ws.sc0, N, t, 20, t1.v);
    ws_st8(ws.sc0, N, t, 21, two_tau3.v);   // v_ = 2 (V + r), times tau^3 (circuit.rs:182-187,235)
    BPPP_STAMP(t, 7);
    // cr_tau = [1, beta/tau, beta tau, ..., beta tau^7]                                  (circuit.rs:208-218)
    ws_st8(ws.cvec, N, t, 0, one.v);
    sc_mul(t1, beta, tau_inv);
    ws_st8(ws.cvec, N, t, 1, t1.v);
    sc bt = beta;
#pragma nounroll
    for (int i = 2; i < 9; i++) {
        sc_mul(bt, bt, tau);
        ws_st8(ws.cvec, N, t, i, bt.v);
    }
    BPPP_STAMP(t, 8);
    ws.status[t] = status;
    if (ws.trace) {
        uint8_t* tb = ws.trace + 704 * t;
        sc_to_be(tb, e); sc_to_be(tb + 32, rho); sc_to_be(tb + 64, lambda); sc_to_be(tb + 96, beta);
        sc_to_be(tb + 128, delta); sc_to_be(tb + 160, tau);
        apt_to_xy64(tb + 320, Vr);
    }
}
// the transcript a proof starts from: the caller's pre-loaded state if there is one (and merlin could be in it), else ws.base
HD void phase1_start_state(strobe& tr, int32_t& status, const VerifyWs& ws, size_t t) {
    tr = ws.base;
    if (ws.states) {
        strobe pre;
        const bool sok = strobe_from_bytes(pre, ws.states + (size_t)BPPP_TRANSCRIPT_STATE_BYTES * (ws.n_states == 1 ? 0 : t));
        if (sok) tr = pre;
        else status |= ST_BAD_ENCODING;      // not a state merlin could be in: flag the proof, run on the shared base
    }
}
HD void verify_phase1(const VerifyWs& ws, size_t t) {          // sponge state in registers
    int32_t status = ST_OK;
    strobe tr;
    phase1_start_state(tr, status, ws, t);
    verify_phase1_on(ws, t, tr, status);
}
#if defined(__HIP_DEVICE_COMPILE__)
// sponge state in LDS: lds_col = this lane's column of the workgroup's [50][64]-word block (merlin.h: strobe_lds)
__device__ __forceinline__ void verify_phase1_lds(const VerifyWs& ws, size_t t, u32* lds_col) {
    int32_t status = ST_OK;
    strobe_lds tl;
    tl.col = lds_col;
    {
        strobe tr;
        phase1_start_state(tr, status, ws, t);
        strobe_lds_load(tl, tr);
    }
    verify_phase1_on(ws, t, tl, status);
}
#endif

// ---------------------------------------------------------------- phase 2b: C0 fixed-base part: ps_tau*g + <g_vec, pn_tau>  (circuit.rs:206) -> pfix
// lane-group form: every lane of the proof's group computes a partial sum; the group total is stored by _store.
HD void verify_c0_fixed_ranges(FbRanges& rg) { fb_ranges_one(rg, 0, 0, 17); }
HD void verify_c0_fixed_store(const VerifyWs& ws, size_t t, const pt& total) { ws_st_pt(ws.pfix, ws.N, t, total); }   // added in round 1
// single-thread form (host emulation in tests/emul, thread order = lane order)
HD void verify_c0_fixed(const VerifyWs& ws, size_t t) {
    FbRanges rg;
    verify_c0_fixed_ranges(rg);
    pt acc;
    fb_sum_serial(acc, fb_of(ws), t, ws.sc0, rg);
    verify_c0_fixed_store(ws, t, acc);
}
// ---------------------------------------------------------------- phase 2a: C0 variable-base part (circuit.rs:230-235)
HD void verify_c0_var(const VerifyWs& ws, size_t t, int group_lane = -1, int group_size = 4) {
    const size_t N = ws.N;
    const int pslot[5] = {3, 2, 0, 1, 12};  // c_s, c_o, c_l, c_r, V+r  <->  sc0 slots 17..21
    glv_words<5> g;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        sc k;
        ws_ld8(k.v, ws.sc0, N, t, 17 + j);
        glv_split sp;
        glv_decompose(sp, k);
        glv_words_set<5>(g, j, sp);
    }
    BPPP_STAMP(t, 20);
    pt acc;
#if defined(__HIP_DEVICE_COMPILE__)
    if (group_lane >= 0 && group_size == 64) straus_affine_split<5, 64, 4>(acc, atab_of(ws.atab, ws.N, t), pslot, g, group_lane);
    else if (group_lane >= 0 && group_size == 32) straus_affine_split<5, 32, 2>(acc, atab_of(ws.atab, ws.N, t), pslot, g, group_lane);
    else if (group_lane >= 0) straus_affine_g4<5>(acc, atab_of(ws.atab, ws.N, t), pslot, g, group_lane);
    else
#endif
        straus_affine<5>(acc, atab_of(ws.atab, ws.N, t), pslot, g, ws.pace != 0);
    (void)group_lane; (void)group_size;
    BPPP_STAMP(t, 21);
    ws_st_pt(ws.acc, N, t, acc);   // the fixed-base part (pfix) is added at the top of round 1
}
// C0 = variable-base part + fixed-base part, ahead of round 1 when the rounds find 1 / Z ready (ws.zinv): the shared inversion needs C0's Z
HD void verify_c0_join(const VerifyWs& ws, size_t t) {
    pt C, F;
    ws_ld_pt(C, ws.acc, ws.N, t);
    ws_ld_pt(F, ws.pfix, ws.N, t);
    pt_add(C, C, F);
    ws_st_pt(ws.acc, ws.N, t, C);
}
// ---------------------------------------------------------------- phase 3 (k = 1..4): one WNLA round (wnla.rs:84-102)
// group_lane >= 0: this lane is one of four consecutive lanes that all run the round for proof t (identical work and identical
// stores, except the sum, which they share: straus_affine_g4) -- the small-batch kernels; -1: one lane per proof
// TR = strobe (sponge state in registers) or strobe_lds (in the workgroup's LDS: k_verify_round); tr arrives unloaded
// part: 0 the whole round; 1 its HEAD only (C_{k-1} to affine, transcript, challenge -- leaves the affine C_{k-1} in ws.acc and y_k in
// ws.chal); 2 its TAIL only (the two-point sum and C_k).  In two kernels the last round's tail runs beside the final fixed-base sum,
// which needs nothing but the challenges (bppp_u64.hip: batches whose one-lane kernels are a lone wavefront per SIMD).
template <typename TR>
HD void verify_round_on(const VerifyWs& ws, size_t t, int k, TR& tr, int group_lane = -1, int group_size = 4, int part = 0) {
    const size_t N = ws.N;
    pt C;
    ws_ld_pt(C, ws.acc, N, t);
    apt Ca;
    sc y;
    if (part == 2) {        // the head left C_{k-1} affine (Z = 1, or the identity) and y_k
        Ca.x = C.X; Ca.y = C.Y;
        if (fe_is_zero(C.Z)) { fe_set_u32(Ca.x, 0); fe_set_u32(Ca.y, 0); }
        ws_ld8(y.v, ws.chal, N, t, 5 + k);
    } else {
        if (k == 1 && !ws.zinv) {   // C0 = variable-base part (acc) + fixed-base part (pfix); the two kernels run concurrently on two streams
            pt F;
            ws_ld_pt(F, ws.pfix, N, t);
            pt_add(C, C, F);
        }
        BPPP_STAMP(t, 9);
        if (ws.zinv) {              // shared inversions: C0 was joined by verify_c0_join, 1 / Z is there (pt_to_affine's two products remain)
            fe zi;
            ws_ld_fe(zi, ws.zinv, N, t, 0, 1);
            fe_mul(Ca.x, C.X, zi);
            fe_mul(Ca.y, C.Y, zi);
        } else pt_to_affine(Ca, C);
        BPPP_STAMP(t, 10);
        ws_ld_transcript(tr, ws.tstate, N, t);
        app_point(tr, "wnla_com", Ca);                                       // wnla.rs:88-92
        {   // the round's proof points are only hashed here (the sum below reads their window tables): loaded one at a time, right
            // before their append, so that nothing but the sponge state and C is live across the permutations
            apt Q;
            ws_ld_apt(Q, ws.pts, N, t, 8 + (4 - k));   // proof.x.last()
            app_point(tr, "wnla_x", Q);
            ws_ld_apt(Q, ws.pts, N, t, 4 + (4 - k));   // proof.r.last()
            app_point(tr, "wnla_r", Q);
        }
        t_append_u64(tr, "l.sz", (u64)(32 >> (k - 1)));
        t_append_u64(tr, "n.sz", (u64)(16 >> (k - 1)));
        bool cok = t_get_challenge(tr, "wnla_challenge", y);                 // wnla.rs:94
        if (!cok) {
            ws.status[t] |= ST_DEGENERATE;
            sc_set_u32(y, 1);
        }
        BPPP_STAMP(t, 11);
        ws_st_transcript(ws.tstate, N, t, tr);
        ws_st8(ws.chal, N, t, 5 + k, y.v);
        if (ws.trace) {
            uint8_t* tb = ws.trace + 704 * t;
            sc_to_be(tb + 32 * (5 + k), y);
            apt_to_xy64(tb + 320 + 64 * k, Ca);
        }
        if (part == 1) {
            pt Cs;
            pt_from_affine(Cs, Ca);
            ws_st_pt(ws.acc, N, t, Cs);
            return;
        }
    }
    // com_ = com + y X + (y^2 - 1) R                                     (wnla.rs:100-102)
    sc y2m1, one;
    sc_set_u32(one, 1);
    sc_mul(y2m1, y, y);
    sc_sub(y2m1, y2m1, one);
    const int pslot[2] = {8 + (4 - k), 4 + (4 - k)};
    glv_words<2> g;
    glv_split sp;
    glv_decompose(sp, y);
    glv_words_set<2>(g, 0, sp);
    glv_decompose(sp, y2m1);
    glv_words_set<2>(g, 1, sp);
    BPPP_STAMP(t, 12);
    pt acc;
#if defined(__HIP_DEVICE_COMPILE__)
    if (group_lane >= 0 && group_size == 16) straus_affine_split<2, 16, 4>(acc, atab_of(ws.atab, ws.N, t), pslot, g, group_lane);
    else if (group_lane >= 0 && group_size == 8) straus_affine_split<2, 8, 2>(acc, atab_of(ws.atab, ws.N, t), pslot, g, group_lane);
    else if (group_lane >= 0 && group_size == 4) straus_affine_g4<2, 4>(acc, atab_of(ws.atab, ws.N, t), pslot, g, group_lane);
    else if (group_lane >= 0) straus_affine_g4<2, 2>(acc, atab_of(ws.atab, ws.N, t), pslot, g, group_lane);
    else
#endif
        straus_affine<2>(acc, atab_of(ws.atab, ws.N, t), pslot, g, ws.pace != 0);
    (void)group_lane; (void)group_size;
    BPPP_STAMP(t, 13);
    pt_madd(acc, acc, Ca, apt_is_identity(Ca));
    ws_st_pt(ws.acc, N, t, acc);
}
HD void verify_round(const VerifyWs& ws, size_t t, int k, int group_lane = -1, int group_size = 4, int part = 0) {
    strobe tr;
    verify_round_on(ws, t, k, tr, group_lane, group_size, part);
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void verify_round_lds(const VerifyWs& ws, size_t t, int k, u32* lds_col) {
    strobe_lds tl;
    tl.col = lds_col;
    verify_round_on(ws, t, k, tl);
}
#endif
// ---------------------------------------------------------------- phase 4: base case (wnla.rs:80-82 with :66-72), generators unrolled
// (Round 5 measured a form that keeps ch / cg as four register-resident quarter tables -- no store-to-load round trips through the
// workspace, the cause of this kernel's 40 % memory wait: 2.64 ms against 1.9.  The 128 VGPRs of tables cost the kernel the occupancy
// that hides its latency today.  Not kept.)
HD void verify_final_scalars(const VerifyWs& ws, size_t t) {
    const size_t N = ws.N;
    BPPP_STAMP(t, 22);
    sc rho, y[4], rk[4], l0, l1, n0;
    ws_ld8(rho.v, ws.chal, N, t, 1);
#pragma nounroll
    for (int k = 0; k < 4; k++) ws_ld8(y[k].v, ws.chal, N, t, 6 + k);
    ws_ld8(l0.v, ws.lns, N, t, 0);
    ws_ld8(l1.v, ws.lns, N, t, 1);
    ws_ld8(n0.v, ws.lns, N, t, 2);
    // rho_1 = rho, rho_{k+1} = mu_k, mu_{k+1} = mu_k^2 (wnla.rs:109-110): rho_k = rho^(2^(k-1)); final mu = rho^32
    rk[0] = rho;
#pragma nounroll
    for (int k = 1; k < 4; k++) sc_mul(rk[k], rk[k - 1], rk[k - 1]);
    sc mu5;
    sc_mul(mu5, rk[3], rk[3]);
    sc_mul(mu5, mu5, mu5);
    // ch[b] = prod_{k: bit k of b} y_{k+1}        (h_vec / c folding, wnla.rs:96,98 unrolled)
    // cg[b] = prod_k (bit k of b ? y_{k+1} : rho_{k+1})   (g_vec folding, wnla.rs:97 unrolled)
    // Built in place in the output slots (cg[b] at fsc slot 1 + b, ch[b] at slot 17 + b) instead of two 16-element arrays in
    // scratch memory; the final products overwrite them.
    sc one;
    sc_set_u32(one, 1);
    ws_st8(ws.fsc, N, t, 17, one.v);
    ws_st8(ws.fsc, N, t, 1, one.v);
#pragma nounroll
    for (int k = 0; k < 4; k++) {
        const int half = 1 << k;
#pragma nounroll
        for (int b = 0; b < half; b++) {
            sc chb, cgb, tmp;
            ws_ld8(chb.v, ws.fsc, N, t, 17 + b);
            ws_ld8(cgb.v, ws.fsc, N, t, 1 + b);
            sc_mul(tmp, chb, y[k]);
            ws_st8(ws.fsc, N, t, 17 + b + half, tmp.v);
            sc_mul(tmp, cgb, y[k]);
            ws_st8(ws.fsc, N, t, 1 + b + half, tmp.v);
            sc_mul(tmp, cgb, rk[k]);
            ws_st8(ws.fsc, N, t, 1 + b, tmp.v);
        }
    }
    // c'_0, c'_1 = folded c (c[25..31] = 0)
    sc c0f, c1f, tmp, cv, chv;
    sc_set_u32(c0f, 0);
    sc_set_u32(c1f, 0);
#pragma nounroll
    for (int i = 0; i < 25; i++) {
        ws_ld8(cv.v, ws.cvec, N, t, i);
        ws_ld8(chv.v, ws.fsc, N, t, 17 + (i & 15));
        sc_mul(tmp, cv, chv);
        if (i < 16) sc_add(c0f, c0f, tmp);
        else sc_add(c1f, c1f, tmp);
    }
    // v = <c', l> + n0^2 mu'   (wnla.rs:67 with weight_vector_mul exponent 1, util.rs:28-44)
    sc v, w;
    sc_mul(v, c0f, l0);
    sc_mul(w, c1f, l1);
    sc_add(v, v, w);
    sc_mul(w, n0, n0);
    sc_mul(w, w, mu5);
    sc_add(v, v, w);
    ws_st8(ws.fsc, N, t, 0, v.v);
#pragma nounroll
    for (int i = 0; i < 16; i++) {
        sc cgv;
        ws_ld8(cgv.v, ws.fsc, N, t, 1 + i);
        sc_mul(tmp, n0, cgv);
        ws_st8(ws.fsc, N, t, 1 + i, tmp.v);
        ws_ld8(chv.v, ws.fsc, N, t, 17 + i);
        sc_mul(tmp, l0, chv);
        ws_st8(ws.fsc, N, t, 17 + i, tmp.v);
        sc_mul(tmp, l1, chv);
        ws_st8(ws.fsc, N, t, 33 + i, tmp.v);
    }
}
#if defined(__HIPCC__)
// The same scalars by SIXTEEN lanes per proof, for calls that leave the chip empty (bppp_u64.hip: the small-call path): lane b forms
// ch[b] and cg[b] -- four conditional multiplications each, in registers -- and its three output scalars; the two folded c values are
// a sum over the group (shuffles).  The one-lane form above walks 118 multiplications whose operands travel through the workspace
// (a store-to-load round trip per step): 145 us for a lone proof, a seventh of it here.  Every lane of a group must be active.
__device__ __forceinline__ void sc_group_sum16(sc& a, int group) {      // group: 16 or a smaller power of two (declared above, default 16)
#pragma unroll
    for (int m = 1; m < group; m <<= 1) {
        sc o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.v[i] = __shfl_xor(a.v[i], m, 64);
        sc_add(a, a, o);
    }
}
__device__ __forceinline__ void verify_final_scalars_lane(const VerifyWs& ws, size_t t, int b) {
    const size_t N = ws.N;
    sc rho, y[4], rk[4], l0, l1, n0, mu5, tmp;
    ws_ld8(rho.v, ws.chal, N, t, 1);
#pragma unroll
    for (int k = 0; k < 4; k++) ws_ld8(y[k].v, ws.chal, N, t, 6 + k);
    ws_ld8(l0.v, ws.lns, N, t, 0);
    ws_ld8(l1.v, ws.lns, N, t, 1);
    ws_ld8(n0.v, ws.lns, N, t, 2);
    rk[0] = rho;
#pragma unroll
    for (int k = 1; k < 4; k++) sc_mul(rk[k], rk[k - 1], rk[k - 1]);
    sc_mul(mu5, rk[3], rk[3]);
    sc_mul(mu5, mu5, mu5);
    sc ch, cg;
    sc_set_u32(ch, 1);
    sc_set_u32(cg, 1);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const bool bit = ((b >> k) & 1) != 0;
        sc_mul(tmp, ch, y[k]);
#pragma unroll
        for (int i = 0; i < 8; i++) ch.v[i] = bit ? tmp.v[i] : ch.v[i];
        sc f;
#pragma unroll
        for (int i = 0; i < 8; i++) f.v[i] = bit ? y[k].v[i] : rk[k].v[i];
        sc_mul(cg, cg, f);
    }
    // c'_0 = sum_{i < 16} c[i] ch[i], c'_1 = sum_{16 <= i < 25} c[i] ch[i - 16]
    sc c0f, c1f, cv;
    ws_ld8(cv.v, ws.cvec, N, t, b);
    sc_mul(c0f, cv, ch);
    ws_ld8(cv.v, ws.cvec, N, t, b < 9 ? 16 + b : 16);
    sc_mul(c1f, cv, ch);
    if (b >= 9) sc_set_u32(c1f, 0);
    sc_group_sum16(c0f);
    sc_group_sum16(c1f);
    if (b == 0) {
        sc v, w;
        sc_mul(v, c0f, l0);
        sc_mul(w, c1f, l1);
        sc_add(v, v, w);
        sc_mul(w, n0, n0);
        sc_mul(w, w, mu5);
        sc_add(v, v, w);
        ws_st8(ws.fsc, N, t, 0, v.v);
    }
    sc_mul(tmp, n0, cg);
    ws_st8(ws.fsc, N, t, 1 + b, tmp.v);
    sc_mul(tmp, l0, ch);
    ws_st8(ws.fsc, N, t, 17 + b, tmp.v);
    sc_mul(tmp, l1, ch);
    ws_st8(ws.fsc, N, t, 33 + b, tmp.v);
}
#endif
HD void verify_final_check_ranges(FbRanges& rg) { fb_ranges_one(rg, 0, 0, BPPP_NG); }
HD void verify_final_check_store(const VerifyWs& ws, size_t t, const pt& rhs) { ws_st_pt(ws.pfix, ws.N, t, rhs); }
// accept bit: C4 == rhs as projective classes (wnla.rs:81), and no status flag
HD void verify_accept(const VerifyWs& ws, size_t t) {
    const size_t N = ws.N;
    pt C, rhs;
    ws_ld_pt(C, ws.acc, N, t);
    ws_ld_pt(rhs, ws.pfix, N, t);
    bool eq = pt_eq(C, rhs);                                             // wnla.rs:81
    ws.accept[t] = (eq && ws.status[t] == ST_OK) ? 1 : 0;
    if (ws.trace) {
        apt Ca;
        pt_to_affine(Ca, C);
        apt_to_xy64(ws.trace + 704 * t + 320 + 64 * 5, Ca);
    }
}
// The caller's `&mut Transcript` after verify (Ownership): the state after the last challenge (wnla.rs:94 in round 4,
// a PRF operation: cur_flags = I|A|C = 7).  A proof whose inputs k256 would have refused to deserialize never reaches the
// reference's verify, so its transcript comes back untouched.
HD void verify_export_state(const VerifyWs& ws, size_t t) {
    if (!ws.states_out) return;
    uint8_t* out = ws.states_out + (size_t)BPPP_TRANSCRIPT_STATE_BYTES * t;
    if (ws.status[t] & ST_BAD_ENCODING) {
        if (ws.states) {
            const uint8_t* in = ws.states + (size_t)BPPP_TRANSCRIPT_STATE_BYTES * (ws.n_states == 1 ? 0 : t);
#pragma nounroll
            for (int i = 0; i < BPPP_TRANSCRIPT_STATE_BYTES; i++) out[i] = in[i];
        } else {
            strobe_to_bytes(out, ws.base, 2);      // Transcript::new ends with an AD operation (the dom-sep message)
        }
        return;
    }
    strobe tr;
    ws_ld_strobe(tr, ws.tstate, ws.N, t);
    strobe_to_bytes(out, tr, 7);
}
HD void verify_final_check(const VerifyWs& ws, size_t t) {
    FbRanges rg;
    verify_final_check_ranges(rg);
    pt acc;
    fb_sum_serial(acc, fb_of(ws), t, ws.fsc, rg);
    verify_final_check_store(ws, t, acc);
    verify_accept(ws, t);
}

// ---------------------------------------------------------------- wire format: SEC1 compressed points
// The reference's SerializableProof (reciprocal.rs:37-41, circuit.rs:37-46, wnla.rs:33-38) holds k256 `AffinePoint`s, whose
// byte form is 33-byte SEC1 compressed (02|03 || x; the identity is 33 zero bytes), and 32-byte big-endian scalars: a u64
// proof is 13*33 + 3*32 = 525 bytes, its commitment 33.  One lane per point recovers y = sqrt(x^3 + 7) (p = 3 mod 4: one
// exponentiation) and writes the 64-byte x||y form the verify pipeline reads.  An undecodable point (bad tag, x >= p,
// x^3 + 7 a non-residue -- k256's from_bytes fails) becomes (1, 0), which is never on the curve, so verify_phase1 flags
// BPPP_ST_BAD_ENCODING for that proof.
#define BPPP_U64_PROOF_SEC1_BYTES 525
HD void sec1_decompress_to_xy64(uint8_t* out64, const uint8_t* in33) {
    u32 nz = 0;
#pragma nounroll
    for (int i = 0; i < 33; i++) nz |= in33[i];
    fe x, rhs, y, y2, seven;
    bool ok = fe_from_be(x, in33 + 1);
    const uint8_t tag = in33[0];
    ok &= (tag == 2) | (tag == 3);
    fe_sqr(rhs, x);
    fe_mul(rhs, rhs, x);
    fe_set_u32(seven, 7);
    fe_add(rhs, rhs, seven);
    fe_sqrt_candidate(y, rhs);
    fe_sqr(y2, y);
    ok &= fe_eq(y2, rhs);
    fe ny;
    fe_neg_m<1>(ny, y);
    fe_cmov(y, fe_is_odd(y) != ((tag & 1) != 0), ny);
    // an undecodable point must never alias the identity (0, 0): it becomes (1, 0), which is off the curve for every tag and x
    // (0 != 1 + 7), so verify_phase1 / apt_from_xy64 flag the proof.  (x, 0) would not do: x = 0 mod p -- 02||00..00, a bad tag
    // over x = 0, 02||p -- would come out as 64 zero bytes, the identity's encoding.
    fe zero, one;
    fe_set_u32(zero, 0);
    fe_set_u32(one, 1);
    fe_cmov(x, !ok, one);
    fe_cmov(y, !ok, zero);
    const bool identity = nz == 0;
    fe_cmov(x, identity, zero);
    fe_cmov(y, identity, zero);
    fe_to_be(out64, x);
    fe_to_be(out64 + 32, y);
}
// lane j of proof t: j = 0 commitment, 1..13 proof points, 14 copies the three scalars
HD void sec1_expand_lane(uint8_t* commitments64, uint8_t* proofs928, const uint8_t* commitments33, const uint8_t* proofs525,
                         size_t t, int j) {
    if (j == 0) sec1_decompress_to_xy64(commitments64 + 64 * t, commitments33 + 33 * t);
    else if (j <= 13) sec1_decompress_to_xy64(proofs928 + (size_t)BPPP_U64_PROOF_BYTES * t + 64 * (j - 1),
                                              proofs525 + (size_t)BPPP_U64_PROOF_SEC1_BYTES * t + 33 * (j - 1));
    else if (j == 14) {
#pragma nounroll
        for (int i = 0; i < 96; i++)
            proofs928[(size_t)BPPP_U64_PROOF_BYTES * t + 832 + i] = proofs525[(size_t)BPPP_U64_PROOF_SEC1_BYTES * t + 429 + i];
    }
}

// the other direction (the prover's output as the crate serialises it: GroupEncoding / serde of SerializableProof, wnla.rs:33-61,
// circuit.rs:36-76): lane j of proof t: j = 0 commitment, 1..13 proof points, 14 copies the three scalars.  The 64-byte points are the
// library's own output (canonical, on the curve): tag 02 / 03 by the parity of y, the identity (64 zero bytes) -> 33 zero bytes.
HD void sec1_compress_lane(uint8_t* commitments33, uint8_t* proofs525, const uint8_t* commitments64, const uint8_t* proofs928, size_t t, int j) {
    if (j <= 13) {
        const uint8_t* in = j == 0 ? commitments64 + 64 * t : proofs928 + (size_t)BPPP_U64_PROOF_BYTES * t + 64 * (j - 1);
        uint8_t* out = j == 0 ? commitments33 + 33 * t : proofs525 + (size_t)BPPP_U64_PROOF_SEC1_BYTES * t + 33 * (j - 1);
        uint8_t any = 0;
#pragma nounroll
        for (int i = 0; i < 64; i++) any |= in[i];
        out[0] = any ? (uint8_t)(2 + (in[63] & 1)) : 0;
#pragma nounroll
        for (int i = 0; i < 32; i++) out[1 + i] = in[i];
    } else if (j == 14) {
#pragma nounroll
        for (int i = 0; i < 96; i++)
            proofs525[(size_t)BPPP_U64_PROOF_SEC1_BYTES * t + 429 + i] = proofs928[(size_t)BPPP_U64_PROOF_BYTES * t + 832 + i];
    }
}


}  // namespace bppp
