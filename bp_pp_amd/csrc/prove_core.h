// Batch u64 range-proof PROVER, one lane per proof (+ 8-lane fixed-base MSM kernels): the per-proof work of
// `U64RangeProofProtocol::prove` (u64_proof.rs:57-82) -> `ReciprocalRangeProofProtocol::prove` (reciprocal.rs:110-146)
// -> `ArithmeticCircuit::prove` (circuit.rs:260-556) -> `WeightNormLinearArgument::prove` (wnla.rs:125-190).
//
// Byte-identical output to the reference prover for the same 52 `Scalar::generate_biased` draws (an input here, in the
// reference's draw order: reciprocal.rs:121 | circuit.rs:264-298 ro(7) rl(6) rr(5) | circuit.rs:371 ls(17) | :372 ns(16)).
//
// Restructuring (same group / field elements, hence same transcript bytes):
//  * the dense circuit coefficients collapse to the u64 closed forms (as in verify_core.h); with c_nO = c_lR = c_lO = 0,
//    no = lo = lr = 0 the polynomial coefficients f_[0..7] (circuit.rs:403-453) reduce to 7 short sums;
//  * every prover point is a fixed-base MSM over the 49 ORIGINAL generators through the batch-shared tables: the folded
//    generators of WNLA round k (wnla.rs:170-171) are linear combinations of the originals with coefficients
//    ch[i] = prod_{t<k-1, bit t of i} y_{t+1} (h) and cg[i] = prod_{t<k-1} (bit t of i ? y_{t+1} : rho_{t+1}) (g), so
//    X (wnla.rs:152-156) and R (wnla.rs:158-160) become 49-term MSMs with computed scalars;
//  * the next round's commitment `wnla.commit(l_, n_)` (wnla.rs:186) equals com + y X + (y^2 - 1) R (the verifier's
//    relation, wnla.rs:100-102) and is computed that way with one GLV Straus multiplication.
#pragma once
#include "verify_core.h"

namespace bppp {

// scalar slots (sv): [slot*8 + limb][N]
enum : int {
    SV_E = 0, SV_RHO, SV_MU, SV_RHOINV, SV_LAMBDA, SV_BETA, SV_DELTA, SV_MUINV, SV_SV /* s + r_blind */, SV_Y,
    SV_R0 = 10,      // 16 reciprocals r_j = (d_j + e)^-1
    SV_EINV0 = 26,   // 16 inverses (e + j)^-1
    SV_L0 = 42,      // l vector, 32
    SV_N0 = 74,      // n vector, 16
    SV_C0 = 90,      // c vector, 32
    SV_CH0 = 122,    // h-generator unrolling coefficients, 32
    SV_CG0 = 154,    // g-generator unrolling coefficients, 16
    SV_COUNT = 170
};
// point buffer slots (pbuf): [slot*30 + word][N], projective limbs
enum : int { PB_V = 0, PB_RCOM, PB_CO, PB_CL, PB_CR, PB_CS, PB_C, PB_X, PB_R, PB_COUNT };
// MSM scalar sets (msc): set s, base b -> slot s*49 + b
#define BPPP_MSC_SETS 4

struct ProveWs {
    size_t N;
    const uint64_t* x;      // N
    const uint8_t* s;       // N x 32
    const uint8_t* rnd;     // N x 52 x 32
    uint8_t* proofs;        // N x 928
    uint8_t* commitments;   // N x 64
    int32_t* status;        // N
    u32* tstate;            // [52][N]
    u32* sv;                // [SV_COUNT*8][N]
    u32* msc;               // [4*49*8][N]
    u32* pbuf;              // [PB_COUNT*30][N]
    pt_slot* straus;        // [N][5][9] projective slots: prove_round_fold re-uses the bytes as affine window tables + build scratch
    FbTable fb;
    strobe base;
    // pre-loaded transcripts (the reference's `t: &mut Transcript`, u64_proof.rs:57): as VerifyWs::states
    const uint8_t* states;
    size_t n_states;
    uint8_t* states_out;
    int next_by_msm;        // 1: prove_round_fold leaves the next commitment's scalars in set 0 for job_e (calls that leave SIMDs idle) instead of handing it to prove_round_next
    // "ct_prover": the sums over the witness and its blindings (V, r_com, c_o, c_l, c_r, c_s) read every entry of every window of this
    // 4-bit table and select by mask (fb_core.h: fb_lookup_add_ct); the WNLA stage's sums, whose vectors the argument reveals by
    // design (the circuit layer blinds them: circuit.rs:371-372), keep the fast gathers
    FbTable fb_ct;
    int ct;
};
struct MsmJob {             // one fixed-base MSM per proof: sum over runs of bases of scalar set `set` (fb_core.h: FbRanges)
    int set, out_slot, nranges;
    int first[BPPP_FB_MAX_RUNS], count[BPPP_FB_MAX_RUNS];
    int bits[BPPP_FB_MAX_RUNS];     // > 0: the run's scalars are below 2^bits (hex digits, multiplicities, the u64 value)
    int oddsh[BPPP_FB_MAX_RUNS];    // >= 0: only the odd blocks of 2^oddsh terms are present; count = present terms
};

struct MsmJobs { MsmJob j[4]; };   // independent sums of one stage, launched together in a small call (k_prove_msm_l64x: blockIdx.y = job)

HD void pw_ld_sc(sc& r, const ProveWs& w, size_t t, int slot) { ws_ld8(r.v, w.sv, w.N, t, slot); }
HD void pw_st_sc(const ProveWs& w, size_t t, int slot, const sc& r) { ws_st8(w.sv, w.N, t, slot, r.v); }
HD void pw_st_msc(const ProveWs& w, size_t t, int set, int base, const sc& r) { ws_st8(w.msc, w.N, t, set * BPPP_NG + base, r.v); }
HD void pw_ld_msc(sc& r, const ProveWs& w, size_t t, int set, int base) { ws_ld8(r.v, w.msc, w.N, t, set * BPPP_NG + base); }
HD void pw_ld_pt(pt& p, const ProveWs& w, size_t t, int slot) { ws_ld_pt(p, w.pbuf + (size_t)slot * 30 * w.N, w.N, t); }
HD void pw_st_pt(const ProveWs& w, size_t t, int slot, const pt& p) { ws_st_pt(w.pbuf + (size_t)slot * 30 * w.N, w.N, t, p); }
HD bool pw_rnd(sc& r, const ProveWs& w, size_t t, int i) { return sc_from_be(r, w.rnd + ((size_t)t * 52 + i) * 32); }

// MSM lane work (8 lanes per proof on the device; fb_group_sum tree-adds the partial sums)
HD void prove_msm_ranges(FbRanges& rg, const MsmJob& job) {
    rg.n = job.nranges;
    for (int r = 0; r < job.nranges; r++) {
        rg.slot[r] = job.set * BPPP_NG + job.first[r]; rg.base[r] = job.first[r]; rg.count[r] = job.count[r];
        rg.bits[r] = job.bits[r]; rg.oddsh[r] = job.oddsh[r];
    }
}
HD void prove_msm_store(const ProveWs& w, const MsmJob& job, size_t t, const pt& total) { pw_st_pt(w, t, job.out_slot, total); }
HD void prove_msm(const ProveWs& w, const MsmJob& job, size_t t) {   // single-thread form (host emulation)
    FbRanges rg;
    prove_msm_ranges(rg, job);
    pt acc;
    fb_sum_serial(acc, w.fb, t, w.msc, rg);
    prove_msm_store(w, job, t, acc);
}

HD void prove_msm_ct(const ProveWs& w, const MsmJob& job, size_t t) {   // the same sum in the secret-scalar form (single-thread form)
    FbRanges rg;
    prove_msm_ranges(rg, job);
    pt acc;
    fb_sum_serial_ct(acc, w.fb_ct, t, w.msc, rg);
    prove_msm_store(w, job, t, acc);
}

// K points -> affine with one field inversion (Montgomery trick); identity (Z = 0) -> (0, 0)
template <int K>
HD void batch_to_affine(apt* out, const pt* in) {
    fe pre[K], run;
    fe_set_u32(run, 1);
#pragma nounroll
    for (int i = 0; i < K; i++) {
        pre[i] = run;
        fe m;
        fe_mul(m, run, in[i].Z);
        fe_cmov(run, !fe_is_zero(in[i].Z), m);
    }
    fe inv;
    fe_inv(inv, run);
#pragma nounroll
    for (int i = K - 1; i >= 0; i--) {
        bool id = fe_is_zero(in[i].Z);
        fe zi, m;
        fe_mul(zi, inv, pre[i]);
        fe_mul(m, inv, in[i].Z);
        fe_cmov(inv, !id, m);
        fe_mul(out[i].x, in[i].X, zi);
        fe_mul(out[i].y, in[i].Y, zi);
        if (id) { fe_set_u32(out[i].x, 0); fe_set_u32(out[i].y, 0); }
    }
}
// hex digits and their multiplicities (u64_proof.rs:84-102)
HD u32 u64_digit(uint64_t x, int j) { return (u32)(x >> (4 * j)) & 15u; }
HD u32 u64_multiplicity(uint64_t x, int d) {
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) c += (((u32)(x >> (4 * j)) & 15u) == (u32)d) ? 1u : 0u;
    return c;
}

// ---------------------------------------------------------------- stage A: scalars of V = x*g + s*h_vec[0]   (reciprocal.rs:88-90)
HD void prove_stage_a(const ProveWs& w, size_t t) {
    sc xs, ss;
    int32_t status = ST_OK;
    sc_set_u64(xs, w.x[t]);
    if (!sc_from_be(ss, w.s + 32 * t)) { status |= ST_BAD_ENCODING; sc_set_u32(ss, 0); }
    pw_st_msc(w, t, 0, 0, xs);
    pw_st_msc(w, t, 0, 17, ss);
    w.status[t] = status;
}
// ---------------------------------------------------------------- stage B: V -> transcript -> e, reciprocals, scalars of r_com, c_o, c_l, c_r
HD void prove_stage_b(const ProveWs& w, size_t t) {
    const uint64_t x = w.x[t];
    int32_t status = w.status[t];
    pt V;
    pw_ld_pt(V, w, t, PB_V);
    apt Va;
    pt_to_affine(Va, V);
    apt_to_xy64(w.commitments + 64 * t, Va);
    strobe tr = w.base;
    if (w.states) {
        strobe pre;
        if (strobe_from_bytes(pre, w.states + (size_t)BPPP_TRANSCRIPT_STATE_BYTES * (w.n_states == 1 ? 0 : t))) tr = pre;
        else status |= ST_BAD_ENCODING;      // not a state merlin could be in
    }
    app_point(tr, "reciprocal_commitment", Va);                          // reciprocal.rs:114
    sc e;
    if (!t_get_challenge(tr, "reciprocal_challenge", e)) { status |= ST_DEGENERATE; sc_set_u32(e, 1); }
    ws_st_strobe(w.tstate, w.N, t, tr);
    pw_st_sc(w, t, SV_E, e);
    // (e + j)^-1 for j = 0..15 with one inversion; r_i = (digit_i + e)^-1 = inv[digit_i]   (reciprocal.rs:117-119)
    sc a[16], pre[16], one;
    sc_set_u32(one, 1);
    bool zero_inv = false;
#pragma nounroll
    for (int j = 0; j < 16; j++) {
        sc js;
        sc_set_u32(js, (u32)j);
        sc_add(a[j], e, js);
        bool z = sc_is_zero(a[j]);
        zero_inv |= z;
        if (z) a[j] = one;
        if (j == 0) pre[0] = a[0];
        else sc_mul(pre[j], pre[j - 1], a[j]);
    }
    if (zero_inv) status |= ST_DEGENERATE;
    sc inv;
    sc_inv(inv, pre[15]);
#pragma nounroll
    for (int j = 15; j >= 1; j--) {
        sc aj;
        sc_mul(aj, inv, pre[j - 1]);
        sc_mul(inv, inv, a[j]);
        a[j] = aj;
    }
    a[0] = inv;
#pragma nounroll
    for (int j = 0; j < 16; j++) pw_st_sc(w, t, SV_EINV0 + j, a[j]);
    sc rnd[19];
    bool rok = true;
#pragma nounroll
    for (int i = 0; i < 19; i++) rok &= pw_rnd(rnd[i], w, t, i);
    if (!rok) {
        status |= ST_BAD_ENCODING;
#pragma nounroll
        for (int i = 0; i < 19; i++) sc_set_u32(rnd[i], 0);
    }
    sc zero;
    sc_set_u32(zero, 0);
    // r_com = r_blind*h[0] + <h[9..], r>          (set 0: bases 17, 26..41)   reciprocal.rs:93-95,121-122
    pw_st_msc(w, t, 0, 17, rnd[0]);
    // c_o: ro over h[0..9)                         (set 1: bases 17..25)       circuit.rs:264-274,335-337
    // c_l: d over g_vec, rl ‖ m over h[0..25)      (set 2: bases 1..16, 17..41)  circuit.rs:276-286,339-341
    // c_r: r over g_vec, rr over h[0..9)           (set 3: bases 1..16, 17..25)  circuit.rs:288-298,343-345
    const int ro_idx[9] = {1, 2, 3, 4, -1, 5, 6, 7, -1};
    const int rl_idx[9] = {8, 9, 10, -1, 11, 12, 13, -1, -1};
    const int rr_idx[9] = {14, 15, -1, 16, 17, 18, -1, -1, -1};
#pragma nounroll
    for (int i = 0; i < 9; i++) {
        pw_st_msc(w, t, 1, 17 + i, ro_idx[i] >= 0 ? rnd[ro_idx[i]] : zero);
        pw_st_msc(w, t, 2, 17 + i, rl_idx[i] >= 0 ? rnd[rl_idx[i]] : zero);
        pw_st_msc(w, t, 3, 17 + i, rr_idx[i] >= 0 ? rnd[rr_idx[i]] : zero);
    }
#pragma nounroll
    for (int j = 0; j < 16; j++) {
        sc r = a[0];
        u32 d = u64_digit(x, j);
#pragma nounroll
        for (int q = 0; q < 16; q++)
            if ((u32)q == d) r = a[q];
        pw_st_sc(w, t, SV_R0 + j, r);
        pw_st_msc(w, t, 0, 26 + j, r);
        pw_st_msc(w, t, 3, 1 + j, r);
        sc ds, ms;
        sc_set_u32(ds, d);
        sc_set_u32(ms, u64_multiplicity(x, j));
        pw_st_msc(w, t, 2, 1 + j, ds);
        pw_st_msc(w, t, 2, 26 + j, ms);
    }
    sc ssc, svv;
    if (!sc_from_be(ssc, w.s + 32 * t)) sc_set_u32(ssc, 0);
    sc_add(svv, ssc, rnd[0]);                                            // s_v = s + r_blind  (reciprocal.rs:135)
    pw_st_sc(w, t, SV_SV, svv);
    w.status[t] = status;
}
// ---- lane forms of the stage kernels' 16-term loops (calls that leave SIMDs idle: bppp_u64.hip).  A call of a few values waits for ONE
// lane's chain of dependent multiplications -- 416 in stage D's loop over the 16 digits, 272 in stage F's --, and the terms are
// independent: lane j of a group of sixteen takes term j, or lane q of a group of four the run of terms 4 q .. 4 q + 3 (the powers
// mu^(j+1), mu^-(j+1), lambda^(j+1) of a lane's first term by ten multiplications each instead of the running products), the seven sums
// meet by shuffles, and everything outside the loop is done by all lanes of the group alike (identical values, identical stores).
// lane = -1 is the one-lane form (and the only one the host emulation runs); group = 16 up to 2^12 values, 4 up to 2^14.
HD void prove_group_sum16(sc& a, int group = 16) {
#if defined(__HIP_DEVICE_COMPILE__)
    sc_group_sum16(a, group);
#else
    (void)a; (void)group;
#endif
}
HD bool prove_group_all16(bool ok, int group = 16) {
#if defined(__HIP_DEVICE_COMPILE__)
    int r = ok ? 1 : 0;
#pragma unroll
    for (int m = 1; m < group; m <<= 1) r &= __shfl_xor(r, m, 64);
    return r != 0;
#else
    return ok;
#endif
}
// ---------------------------------------------------------------- stage D: transcript to delta; f_, rs; scalars of c_s
HD void prove_stage_d(const ProveWs& w, size_t t, int lane = -1, int group = 16) {
    const size_t N = w.N;
    const uint64_t x = w.x[t];
    int32_t status = w.status[t];
    pt P[5];   // cl, cr, co, Vc = V + r_com, r_com
    pt V;
    pw_ld_pt(P[0], w, t, PB_CL);
    pw_ld_pt(P[1], w, t, PB_CR);
    pw_ld_pt(P[2], w, t, PB_CO);
    pw_ld_pt(P[4], w, t, PB_RCOM);
    pw_ld_pt(V, w, t, PB_V);
    pt_add(P[3], V, P[4]);                                               // circuit.commit(v, s + r_blind) (reciprocal.rs:142) = V + r_com
    apt A[5];
    batch_to_affine<5>(A, P);
    uint8_t* pb = w.proofs + (size_t)BPPP_U64_PROOF_BYTES * t;
    apt_to_xy64(pb + 0, A[0]);          // c_l
    apt_to_xy64(pb + 64, A[1]);         // c_r
    apt_to_xy64(pb + 128, A[2]);        // c_o
    apt_to_xy64(pb + 64 * 12, A[4]);    // reciprocal proof.r
    strobe tr;
    ws_ld_strobe(tr, w.tstate, N, t);
    app_point(tr, "commitment_cl", A[0]);                                // circuit.rs:347-350
    app_point(tr, "commitment_cr", A[1]);
    app_point(tr, "commitment_co", A[2]);
    app_point(tr, "commitment_v", A[3]);
    sc rho, lambda, beta, delta;
    bool cok = t_get_challenge(tr, "circuit_rho", rho);                  // circuit.rs:352-355
    cok &= t_get_challenge(tr, "circuit_lambda", lambda);
    cok &= t_get_challenge(tr, "circuit_beta", beta);
    cok &= t_get_challenge(tr, "circuit_delta", delta);
    if (!cok) { status |= ST_DEGENERATE; sc_set_u32(rho, 1); sc_set_u32(lambda, 1); sc_set_u32(beta, 1); sc_set_u32(delta, 1); }
    ws_st_strobe(w.tstate, N, t, tr);
    sc e, mu;
    pw_ld_sc(e, w, t, SV_E);
    sc_mul(mu, rho, rho);
    // inverses of mu, beta, rho (delta^-1 is unwrapped by the reference, circuit.rs:401, but multiplies only zeros here)
    sc iv[3] = {mu, beta, rho}, pre[3], one;
    sc_set_u32(one, 1);
    bool zero_inv = sc_is_zero(delta);
#pragma nounroll
    for (int i = 0; i < 3; i++) {
        bool z = sc_is_zero(iv[i]);
        zero_inv |= z;
        if (z) iv[i] = one;
        if (i == 0) pre[0] = iv[0];
        else sc_mul(pre[i], pre[i - 1], iv[i]);
    }
    if (zero_inv) status |= ST_DEGENERATE;
    sc inv;
    sc_inv(inv, pre[2]);
#pragma nounroll
    for (int i = 2; i >= 1; i--) {
        sc ai;
        sc_mul(ai, inv, pre[i - 1]);
        sc_mul(inv, inv, iv[i]);
        iv[i] = ai;
    }
    iv[0] = inv;
    const sc mu_inv = iv[0], beta_inv = iv[1], rho_inv = iv[2];
    pw_st_sc(w, t, SV_RHO, rho); pw_st_sc(w, t, SV_MU, mu); pw_st_sc(w, t, SV_RHOINV, rho_inv);
    pw_st_sc(w, t, SV_LAMBDA, lambda); pw_st_sc(w, t, SV_BETA, beta); pw_st_sc(w, t, SV_DELTA, delta);
    pw_st_sc(w, t, SV_MUINV, mu_inv);
    // S = sum_{i=1..16} lambda^i
    sc S = lambda, lp = lambda;
#pragma nounroll
    for (int i = 1; i < 16; i++) { sc_mul(lp, lp, lambda); sc_add(S, S, lp); }
    // prover randomness ls (17), ns (16): draws 19..35, 36..51   (circuit.rs:371-372)
    bool rok = true;
    sc f0, f1, f2, f3, f4, f5, f6, t1, t2;
    sc_set_u32(f0, 0); sc_set_u32(f1, 0); sc_set_u32(f2, 0); sc_set_u32(f3, 0); sc_set_u32(f4, 0); sc_set_u32(f5, 0); sc_set_u32(f6, 0);
    sc mip = mu_inv, mp = mu;
    lp = lambda;
    const int per = 16 / group, j_begin = lane >= 0 ? lane * per : 0, j_end = lane >= 0 ? j_begin + per : 16;      // a lane's run of terms
    if (lane >= 0) {            // this lane's terms only: the powers of its first one directly
        sc_pow_u5(mip, mu_inv, (unsigned)j_begin + 1);
        sc_pow_u5(mp, mu, (unsigned)j_begin + 1);
        sc_pow_u5(lp, lambda, (unsigned)j_begin + 1);
    }
#pragma nounroll
    for (int j = j_begin; j < j_end; j++) {
        sc ls_j, ns_j, r_j, einv_j, d_j, m_j, p16;
        rok &= pw_rnd(ls_j, w, t, 19 + j);
        rok &= pw_rnd(ns_j, w, t, 36 + j);
        pw_ld_sc(r_j, w, t, SV_R0 + j);
        pw_ld_sc(einv_j, w, t, SV_EINV0 + j);
        sc_set_u32(d_j, u64_digit(x, j));
        sc_set_u32(m_j, u64_multiplicity(x, j));
        pw_st_msc(w, t, 0, 26 + j, ls_j);          // c_s: ls over h[9..]
        pw_st_msc(w, t, 0, 1 + j, ns_j);           // c_s: ns over g_vec
        // c_nL[j] = -16^j mu^-(j+1); c_nR[j] = (S - lambda^(j+1)) mu^-(j+1) + e; c_lL[j] = -S (e+j)^-1
        sc cnl, cnr, cll;
        sc_set_u64(p16, (u64)1 << (4 * j));
        sc_mul(cnl, p16, mip);
        sc_neg(cnl, cnl);
        sc_sub(cnr, S, lp);
        sc_mul(cnr, cnr, mip);
        sc_add(cnr, cnr, e);
        sc_mul(cll, S, einv_j);
        sc_neg(cll, cll);
        sc A_, B_;
        sc_add(A_, d_j, cnr);                      // nl + c_nR
        sc_add(B_, r_j, cnl);                      // nr + c_nL
        // f0 -= ns^2 mu^(j+1)                                              (circuit.rs:406)
        sc_mul(t1, ns_j, ns_j); sc_mul(t1, t1, mp); sc_sub(f0, f0, t1);
        // f1 += lambda^(j+1) ls_j                                          (circuit.rs:409)
        sc_mul(t1, lp, ls_j); sc_add(f1, f1, t1);
        // f2 -= 2 ns A mu^(j+1)                                            (circuit.rs:415)
        sc_mul(t1, ns_j, A_); sc_mul(t1, t1, mp); sc_add(t1, t1, t1); sc_sub(f2, f2, t1);
        // f3 += 2 c_lL ls + lambda^(j+1) m + 2 ns B mu^(j+1)               (circuit.rs:419-422)
        sc_mul(t1, cll, ls_j); sc_add(t1, t1, t1); sc_add(f3, f3, t1);
        sc_mul(t1, lp, m_j); sc_add(f3, f3, t1);
        sc_mul(t1, ns_j, B_); sc_mul(t1, t1, mp); sc_add(t1, t1, t1); sc_add(f3, f3, t1);
        // f4 += (c_nR^2 - A^2) mu^(j+1)                                    (circuit.rs:426,433)
        sc_mul(t1, cnr, cnr); sc_mul(t2, A_, A_); sc_sub(t1, t1, t2); sc_mul(t1, t1, mp); sc_add(f4, f4, t1);
        // f5 += (c_nL^2 - B^2) mu^(j+1)                                    (circuit.rs:439,444)
        sc_mul(t1, cnl, cnl); sc_mul(t2, B_, B_); sc_sub(t1, t1, t2); sc_mul(t1, t1, mp); sc_add(f5, f5, t1);
        // f6 += 2 c_lL v_1[j], v_1 = 2 r                                   (circuit.rs:449, 392-395)
        sc_mul(t1, cll, r_j); sc_add(t1, t1, t1); sc_add(t1, t1, t1); sc_add(f6, f6, t1);
        sc_mul(mip, mip, mu_inv);
        sc_mul(lp, lp, lambda);
        sc_mul(mp, mp, mu);
    }
    if (lane >= 0) {            // the group's seven sums, and whether every lane's draws decoded
        prove_group_sum16(f0, group); prove_group_sum16(f1, group); prove_group_sum16(f2, group); prove_group_sum16(f3, group);
        prove_group_sum16(f4, group); prove_group_sum16(f5, group); prove_group_sum16(f6, group);
        rok = prove_group_all16(rok, group);
    }
    sc ls16;
    rok &= pw_rnd(ls16, w, t, 19 + 16);
    pw_st_msc(w, t, 0, 26 + 16, ls16);
    sc ro[9], rl[9], rr[9];
#pragma nounroll
    for (int i = 0; i < 9; i++) { pw_ld_msc(ro[i], w, t, 1, 17 + i); pw_ld_msc(rl[i], w, t, 2, 17 + i); pw_ld_msc(rr[i], w, t, 3, 17 + i); }
    if (!rok) status |= ST_BAD_ENCODING;
    // rs (circuit.rs:457-467), f7 = 0, rv[0] = 2 (s + r_blind)
    sc rs[9], svv, rv0;
    pw_ld_sc(svv, w, t, SV_SV);
    sc_add(rv0, svv, svv);
    sc_mul(t1, ro[1], delta); sc_mul(t1, t1, beta); sc_add(rs[0], f1, t1);
    sc_mul(rs[1], f0, beta_inv);
    sc_mul(t1, ro[0], delta); sc_add(t1, t1, f2); sc_mul(t1, t1, beta_inv); sc_sub(rs[2], t1, rl[1]);
    sc_sub(t1, f3, rl[0]); sc_mul(t1, t1, beta_inv); sc_mul(t2, ro[2], delta); sc_add(t2, t2, rr[1]); sc_add(rs[3], t1, t2);
    sc_add(t1, f4, rr[0]); sc_mul(t1, t1, beta_inv); sc_mul(t2, ro[3], delta); sc_sub(t2, t2, rl[2]); sc_add(rs[4], t1, t2);
    sc_mul(t1, rv0, beta_inv); sc_neg(rs[5], t1);
    sc_mul(t1, f5, beta_inv); sc_mul(t2, ro[5], delta); sc_add(t1, t1, t2); sc_add(t1, t1, rr[3]); sc_sub(rs[6], t1, rl[4]);
    sc_mul(t1, f6, beta_inv); sc_add(t1, t1, rr[4]); sc_mul(t2, ro[6], delta); sc_add(t1, t1, t2); sc_sub(rs[7], t1, rl[5]);
    sc_mul(t1, ro[7], delta); sc_sub(t1, t1, rl[6]); sc_add(rs[8], t1, rr[5]);
#pragma nounroll
    for (int i = 0; i < 9; i++) pw_st_msc(w, t, 0, 17 + i, rs[i]);       // c_s: rs over h[0..9)   (circuit.rs:469-470)
    w.status[t] = status;
}
// ---------------------------------------------------------------- stage F: c_s -> tau; l, n, c, v; scalars of C0; WNLA state
HD void prove_stage_f(const ProveWs& w, size_t t, int lane = -1, int group = 16) {
    const size_t N = w.N;
    const uint64_t x = w.x[t];
    int32_t status = w.status[t];
    pt CS;
    pw_ld_pt(CS, w, t, PB_CS);
    apt csa;
    pt_to_affine(csa, CS);
    apt_to_xy64(w.proofs + (size_t)BPPP_U64_PROOF_BYTES * t + 192, csa);
    strobe tr;
    ws_ld_strobe(tr, w.tstate, N, t);
    app_point(tr, "commitment_cs", csa);                                 // circuit.rs:472
    sc tau;
    if (!t_get_challenge(tr, "circuit_tau", tau)) { status |= ST_DEGENERATE; sc_set_u32(tau, 1); }
    ws_st_strobe(w.tstate, N, t, tr);
    if (sc_is_zero(tau)) { status |= ST_DEGENERATE; sc_set_u32(tau, 1); }
    sc tau_inv, tau2, tau3, e, mu, mu_inv, lambda, beta, delta, svv;
    sc_inv(tau_inv, tau);
    sc_mul(tau2, tau, tau);
    sc_mul(tau3, tau2, tau);
    pw_ld_sc(e, w, t, SV_E); pw_ld_sc(mu, w, t, SV_MU); pw_ld_sc(mu_inv, w, t, SV_MUINV); pw_ld_sc(lambda, w, t, SV_LAMBDA);
    pw_ld_sc(beta, w, t, SV_BETA); pw_ld_sc(delta, w, t, SV_DELTA); pw_ld_sc(svv, w, t, SV_SV);
    sc zero, one, t1, t2;
    sc_set_u32(zero, 0);
    sc_set_u32(one, 1);
    // l[0..9) = tau^-1 rs - delta ro + tau rl - tau^2 rr + tau^3 rv                      (circuit.rs:479-483)
#pragma nounroll
    for (int i = 0; i < 9; i++) {
        sc rs_i, ro_i, rl_i, rr_i, li;
        pw_ld_msc(rs_i, w, t, 0, 17 + i); pw_ld_msc(ro_i, w, t, 1, 17 + i); pw_ld_msc(rl_i, w, t, 2, 17 + i); pw_ld_msc(rr_i, w, t, 3, 17 + i);
        sc_mul(li, tau_inv, rs_i);
        sc_mul(t1, delta, ro_i); sc_sub(li, li, t1);
        sc_mul(t1, tau, rl_i); sc_add(li, li, t1);
        sc_mul(t1, tau2, rr_i); sc_sub(li, li, t1);
        if (i == 0) { sc_add(t1, svv, svv); sc_mul(t1, t1, tau3); sc_add(li, li, t1); }   // rv[0] = 2 (s + r_blind)
        pw_st_sc(w, t, SV_L0 + i, li);
        if (lane >= 0) pw_st_msc(w, t, 0, 17 + i, li);      // (lane form: the scalars of C0 are written where they are made, see below)
    }
    // S, closed forms as in the verifier
    sc S = lambda, lp = lambda, mp = mu, musum = mu;
#pragma nounroll
    for (int i = 1; i < 16; i++) { sc_mul(lp, lp, lambda); sc_add(S, S, lp); sc_mul(mp, mp, mu); sc_add(musum, musum, mp); }
    sc tau_e, two_tau2_S, two_tau3, ps;
    sc_mul(tau_e, tau, e);
    sc_mul(two_tau2_S, tau2, S);
    sc_add(two_tau2_S, two_tau2_S, two_tau2_S);
    sc_add(two_tau3, tau3, tau3);
    sc_set_u32(ps, 0);
    sc mip = mu_inv;
    lp = lambda;
    mp = mu;
    const int per = 16 / group, j_begin = lane >= 0 ? lane * per : 0, j_end = lane >= 0 ? j_begin + per : 16;      // a lane's run of terms
    sc ls16_in;                                                  // c_s's scalar of h[25] = ls[16]: read before the C0 scalars overwrite set 0
    pw_ld_msc(ls16_in, w, t, 0, 26 + 16);
    if (lane >= 0) {
        sc_pow_u5(mip, mu_inv, (unsigned)j_begin + 1);
        sc_pow_u5(mp, mu, (unsigned)j_begin + 1);
        sc_pow_u5(lp, lambda, (unsigned)j_begin + 1);
    }
#pragma nounroll
    for (int j = j_begin; j < j_end; j++) {
        sc ls_j, ns_j, r_j, einv_j, d_j, m_j, p16, pn, lj, nj, cj;
        pw_ld_msc(ls_j, w, t, 0, 26 + j);
        pw_ld_msc(ns_j, w, t, 0, 1 + j);
        pw_ld_sc(r_j, w, t, SV_R0 + j);
        pw_ld_sc(einv_j, w, t, SV_EINV0 + j);
        sc_set_u32(d_j, u64_digit(x, j));
        sc_set_u32(m_j, u64_multiplicity(x, j));
        // l[9+j] = tau^-1 ls + tau ll + tau^3 v_1,  ll = m, v_1 = 2 r
        sc_mul(lj, tau_inv, ls_j);
        sc_mul(t1, tau, m_j); sc_add(lj, lj, t1);
        sc_mul(t1, two_tau3, r_j); sc_add(lj, lj, t1);
        pw_st_sc(w, t, SV_L0 + 9 + j, lj);
        if (lane >= 0) pw_st_msc(w, t, 0, 17 + 9 + j, lj);        // (read above as ls_j: this lane's own slot)
        // pn_tau[j] (circuit.rs:485-487)
        sc_set_u64(p16, (u64)1 << (4 * j));
        sc_mul(t1, tau2, p16);
        sc_sub(t2, S, lp);
        sc_mul(t2, t2, tau);
        sc_add(t1, t1, t2);
        sc_mul(pn, t1, mip);
        sc_add(pn, pn, tau_e);
        sc_mul(t1, pn, pn); sc_mul(t1, t1, mp); sc_add(ps, ps, t1);
        // n[j] = pn_tau + tau^-1 ns - delta no + tau nl - tau^2 nr   (circuit.rs:493-498), no = 0, nl = d, nr = r
        sc_mul(nj, tau_inv, ns_j);
        sc_mul(t1, tau, d_j); sc_add(nj, nj, t1);
        sc_mul(t1, tau2, r_j); sc_sub(nj, nj, t1);
        sc_add(nj, nj, pn);
        pw_st_sc(w, t, SV_N0 + j, nj);
        if (lane >= 0) pw_st_msc(w, t, 0, 1 + j, nj);             // (read above as ns_j: this lane's own slot)
        // c[9+j] = cl_tau[j] = 2 tau^2 S (e+j)^-1 - lambda^(j+1)
        sc_mul(cj, two_tau2_S, einv_j);
        sc_sub(cj, cj, lp);
        pw_st_sc(w, t, SV_C0 + 9 + j, cj);
        sc_mul(mip, mip, mu_inv);
        sc_mul(lp, lp, lambda);
        sc_mul(mp, mp, mu);
    }
    if (lane >= 0) prove_group_sum16(ps, group);
    {   // l[25] = tau^-1 ls[16]; l[26..32) = 0; c[25..32) = 0     (circuit.rs:526-529)
        sc l25;
        sc_mul(l25, tau_inv, ls16_in);
        pw_st_sc(w, t, SV_L0 + 25, l25);
        if (lane >= 0) pw_st_msc(w, t, 0, 17 + 25, l25);
        pw_st_sc(w, t, SV_C0 + 25, zero);
#pragma nounroll
        for (int i = 26; i < 32; i++) { pw_st_sc(w, t, SV_L0 + i, zero); pw_st_sc(w, t, SV_C0 + i, zero); }
    }
    // cr_tau
    pw_st_sc(w, t, SV_C0, one);
    sc_mul(t1, beta, tau_inv);
    pw_st_sc(w, t, SV_C0 + 1, t1);
    sc bt = beta;
#pragma nounroll
    for (int i = 2; i < 9; i++) { sc_mul(bt, bt, tau); pw_st_sc(w, t, SV_C0 + i, bt); }
    // v = ps_tau + tau^3 v_0, v_0 = 2 x        (circuit.rs:518, 376-381)
    sc_mul(t1, two_tau3, musum);
    sc_sub(ps, ps, t1);
    sc xs;
    sc_set_u64(xs, x);
    sc_mul(t1, two_tau3, xs);
    sc_add(ps, ps, t1);
    // scalars of C0 = v g + <h, l> + <g_vec, n>  (set 0: bases 0..42)   circuit.rs:520-524
    pw_st_msc(w, t, 0, 0, ps);
    if (lane < 0) {             // (the lane form has written these where it made them: no lane reads what another lane of its group stored)
#pragma nounroll
        for (int j = 0; j < 16; j++) { sc nj; pw_ld_sc(nj, w, t, SV_N0 + j); pw_st_msc(w, t, 0, 1 + j, nj); }
#pragma nounroll
        for (int i = 0; i < 26; i++) { sc li; pw_ld_sc(li, w, t, SV_L0 + i); pw_st_msc(w, t, 0, 17 + i, li); }
    }
    // generator unrolling coefficients start at 1
#pragma nounroll
    for (int i = 0; i < 32; i++) pw_st_sc(w, t, SV_CH0 + i, one);
#pragma nounroll
    for (int i = 0; i < 16; i++) pw_st_sc(w, t, SV_CG0 + i, one);
    w.status[t] = status;
}
// ---------------------------------------------------------------- WNLA round k: scalars of X (set 1) and R (set 2)   (wnla.rs:135-160)
// in three pieces, so that a small call can give every generator its own lane (k_prove_round_scalars_wide): the two leading scalars,
// the h_vec terms, the g_vec terms
HD void prove_round_scalars_v(const ProveWs& w, size_t t, int k) {
    const int sh = k - 1, nl = 32 >> sh, nn = 16 >> sh;
    sc rho_inv, mu, mu2, t1, vx, vr;
    pw_ld_sc(rho_inv, w, t, SV_RHOINV); pw_ld_sc(mu, w, t, SV_MU);
    sc_mul(mu2, mu, mu);
    // vx = wvm(n0, n1, mu2) * 2 rho^-1 + <c0, l1> + <c1, l0>;  vr = wvm(n1, n1, mu2) + <c1, l1>
    sc wx, wr, wpow = mu2;
    sc_set_u32(wx, 0);
    sc_set_u32(wr, 0);
#pragma nounroll
    for (int m = 0; m < nn / 2; m++) {
        sc n0, n1;
        pw_ld_sc(n0, w, t, SV_N0 + 2 * m);
        pw_ld_sc(n1, w, t, SV_N0 + 2 * m + 1);
        sc_mul(t1, n0, n1); sc_mul(t1, t1, wpow); sc_add(wx, wx, t1);
        sc_mul(t1, n1, n1); sc_mul(t1, t1, wpow); sc_add(wr, wr, t1);
        sc_mul(wpow, wpow, mu2);
    }
    sc_add(t1, rho_inv, rho_inv);
    sc_mul(vx, wx, t1);
    vr = wr;
#pragma nounroll
    for (int m = 0; m < nl / 2; m++) {
        sc c0, c1, l0, l1;
        pw_ld_sc(c0, w, t, SV_C0 + 2 * m); pw_ld_sc(c1, w, t, SV_C0 + 2 * m + 1);
        pw_ld_sc(l0, w, t, SV_L0 + 2 * m); pw_ld_sc(l1, w, t, SV_L0 + 2 * m + 1);
        sc_mul(t1, c0, l1); sc_add(vx, vx, t1);
        sc_mul(t1, c1, l0); sc_add(vx, vx, t1);
        sc_mul(t1, c1, l1); sc_add(vr, vr, t1);
    }
    pw_st_msc(w, t, 1, 0, vx);
    pw_st_msc(w, t, 2, 0, vr);
    if (w.next_by_msm || k == 1) {      // job_e: stage F / the previous fold left v of C_{k-1} in set 0; its odd-slot part is v_r, which R carries
        sc v;
        pw_ld_msc(v, w, t, 0, 0);
        sc_sub(v, v, vr);
        pw_st_msc(w, t, 0, 0, v);
    }
}
// original h_i sits in folded slot j = i >> (k-1) with coefficient ch[i]:  X gets ch[i] l[j^1], R gets (j odd) ch[i] l[j]
HD void prove_round_scalars_h(const ProveWs& w, size_t t, int k, int i) {
    const int j = i >> (k - 1);
    sc ch, lx, lr, t1, t2, zero;
    sc_set_u32(zero, 0);
    pw_ld_sc(ch, w, t, SV_CH0 + i);
    pw_ld_sc(lx, w, t, SV_L0 + (j ^ 1));
    sc_mul(t1, ch, lx);
    pw_st_msc(w, t, 1, 17 + i, t1);
    pw_ld_sc(lr, w, t, SV_L0 + j);
    sc_mul(t2, ch, lr);
    pw_st_msc(w, t, 2, 17 + i, (j & 1) ? t2 : zero);
}
// g_i: X gets cg[i] * (j even ? rho n[j+1] : rho^-1 n[j-1]);  R gets (j odd) cg[i] n[j]
HD void prove_round_scalars_g(const ProveWs& w, size_t t, int k, int i) {
    const int j = i >> (k - 1);
    sc rho, rho_inv, cg, nx, nr, t1, t2, zero;
    sc_set_u32(zero, 0);
    pw_ld_sc(rho, w, t, SV_RHO); pw_ld_sc(rho_inv, w, t, SV_RHOINV);
    pw_ld_sc(cg, w, t, SV_CG0 + i);
    pw_ld_sc(nx, w, t, SV_N0 + (j ^ 1));
    sc_mul(t1, nx, (j & 1) ? rho_inv : rho);
    sc_mul(t1, t1, cg);
    pw_st_msc(w, t, 1, 1 + i, t1);
    pw_ld_sc(nr, w, t, SV_N0 + j);
    sc_mul(t2, cg, nr);
    pw_st_msc(w, t, 2, 1 + i, (j & 1) ? t2 : zero);
}
HD void prove_round_scalars(const ProveWs& w, size_t t, int k) {
    prove_round_scalars_v(w, t, k);
#pragma nounroll
    for (int i = 0; i < 32; i++) prove_round_scalars_h(w, t, k, i);
#pragma nounroll
    for (int i = 0; i < 16; i++) prove_round_scalars_g(w, t, k, i);
}
// ---------------------------------------------------------------- WNLA round k: transcript, challenge, folds, next commitment (wnla.rs:162-188)
// In two parts, because only the first is on the prover's critical path:
//   prove_round_fold    transcript, challenge y_k, the folded l / n / c, the generator coefficients -- what round k + 1's scalars and
//                       sums (X, R) need.  Leaves y_k (SV_Y) and the affine X, R, C_{k-1} (the table scratch's point slots) for
//   prove_round_next    C_k = C_{k-1} + y X + (y^2 - 1) R by the verifier's variable-base path (window tables of X and R, GLV Straus):
//                       ~125 doublings of dependent work that nobody needs before round k + 1 appends C_k to the transcript, so the
//                       host runs it on the helper stream UNDER round k + 1's scalars and fixed-base sums (bppp_u64.hip).
// (small calls, next_by_msm: the next commitment is one more fixed-base sum whose scalars part one prepares; there is no part two.)
// The point slots behind the running products: 0 = X, 1 = R (the table builder's inputs), 2 = C_{k-1}.
HD u32* prove_fold_rpts(const ProveWs& w) { return (u32*)((uint8_t*)w.straus + (size_t)32 * sizeof(apt_packed) * w.N) + (size_t)28 * 10 * w.N; }
HD void prove_round_next(const ProveWs& w, size_t t, int k, int group_lane = -1);
// the part every form shares: C_{k-1}, X, R to affine (one inversion), X and R into the proof, the round's transcript, y_k
HD void prove_round_fold_head(const ProveWs& w, size_t t, int k, apt A[3], sc& y, int32_t& status) {
    const size_t N = w.N;
    const int sh = k - 1, nl = 32 >> sh, nn = 16 >> sh;
    pt P[3];   // C_{k-1}, X, R
    pw_ld_pt(P[0], w, t, PB_C);
    pw_ld_pt(P[1], w, t, PB_X);
    pw_ld_pt(P[2], w, t, PB_R);
    if (w.next_by_msm || k == 1) pt_add(P[0], P[0], P[2]);     // C_{k-1} = E + R (job_e)
    batch_to_affine<3>(A, P);
    uint8_t* pb = w.proofs + (size_t)BPPP_U64_PROOF_BYTES * t;
    apt_to_xy64(pb + 64 * (8 + (4 - k)), A[1]);     // proof.x is pushed innermost-first (wnla.rs:188): x[4-k] = X of round k
    apt_to_xy64(pb + 64 * (4 + (4 - k)), A[2]);
    strobe tr;
    ws_ld_strobe(tr, w.tstate, N, t);
    app_point(tr, "wnla_com", A[0]);
    app_point(tr, "wnla_x", A[1]);
    app_point(tr, "wnla_r", A[2]);
    t_append_u64(tr, "l.sz", (u64)nl);
    t_append_u64(tr, "n.sz", (u64)nn);
    if (!t_get_challenge(tr, "wnla_challenge", y)) { status |= ST_DEGENERATE; sc_set_u32(y, 1); }
    ws_st_strobe(w.tstate, N, t, tr);
}
HD void prove_round_fold(const ProveWs& w, size_t t, int k) {
    const size_t N = w.N;
    const int sh = k - 1, nl = 32 >> sh, nn = 16 >> sh;
    int32_t status = w.status[t];
    apt A[3];
    sc y;
    prove_round_fold_head(w, t, k, A, y, status);
    uint8_t* pb = w.proofs + (size_t)BPPP_U64_PROOF_BYTES * t;
    sc rho, rho_inv, mu, t1;
    pw_ld_sc(rho, w, t, SV_RHO); pw_ld_sc(rho_inv, w, t, SV_RHOINV); pw_ld_sc(mu, w, t, SV_MU);
    // l_ = l0 + y l1; c_ = c0 + y c1; n_ = rho^-1 n0 + y n1   (in place: slot m is written after slots 2m, 2m+1 are read)
#pragma nounroll
    for (int m = 0; m < nl / 2; m++) {
        sc a0, a1;
        pw_ld_sc(a0, w, t, SV_L0 + 2 * m); pw_ld_sc(a1, w, t, SV_L0 + 2 * m + 1);
        sc_mul(t1, a1, y); sc_add(a0, a0, t1);
        pw_st_sc(w, t, SV_L0 + m, a0);
        pw_ld_sc(a0, w, t, SV_C0 + 2 * m); pw_ld_sc(a1, w, t, SV_C0 + 2 * m + 1);
        sc_mul(t1, a1, y); sc_add(a0, a0, t1);
        pw_st_sc(w, t, SV_C0 + m, a0);
    }
#pragma nounroll
    for (int m = 0; m < nn / 2; m++) {
        sc a0, a1;
        pw_ld_sc(a0, w, t, SV_N0 + 2 * m); pw_ld_sc(a1, w, t, SV_N0 + 2 * m + 1);
        sc_mul(a0, a0, rho_inv);
        sc_mul(t1, a1, y); sc_add(a0, a0, t1);
        pw_st_sc(w, t, SV_N0 + m, a0);
    }
    if (k < 4) {
        // generator coefficients pick up this round's factor
#pragma nounroll
        for (int i = 0; i < 32; i++) {
            if ((i >> sh) & 1) { sc c; pw_ld_sc(c, w, t, SV_CH0 + i); sc_mul(c, c, y); pw_st_sc(w, t, SV_CH0 + i, c); }
        }
#pragma nounroll
        for (int i = 0; i < 16; i++) {
            sc c;
            pw_ld_sc(c, w, t, SV_CG0 + i);
            sc_mul(c, c, ((i >> sh) & 1) ? y : rho);
            pw_st_sc(w, t, SV_CG0 + i, c);
        }
        // rho <- mu, mu <- mu^2, rho^-1 <- (rho^-1)^2           (wnla.rs:180-181; mu = rho^2 at every level)
        pw_st_sc(w, t, SV_RHO, mu);
        sc_mul(t1, mu, mu);
        pw_st_sc(w, t, SV_MU, t1);
        sc_mul(t1, rho_inv, rho_inv);
        pw_st_sc(w, t, SV_RHOINV, t1);
        if (w.next_by_msm) {
            // next commitment = wnla.commit(l_, n_) as the reference computes it (wnla.rs:186, :66-72): v g + <h', l_> + <g', n_> over the
            // ORIGINAL generators (h'_j = sum ch[i] h_i, g'_j = sum cg[i] g_i over i >> k == j), v = <c_, l_> + |n_|^2_{mu'} -- as a
            // fixed-base sum.  Only its EVEN folded slots j are summed for it (job_e, riding with the next round's X | R): the odd ones are
            // that round's R.  25 table-driven terms and no dependent chain, against the 125 doublings of the variable-base form below,
            // which is the cheaper one once the chip is full.
            sc v, mun, mp, a, b;
            pw_ld_sc(mun, w, t, SV_MU);         // mu' (already advanced above)
            mp = mun;
            sc_set_u32(v, 0);
#pragma nounroll
            for (int m = 0; m < nn / 2; m++) {
                pw_ld_sc(a, w, t, SV_N0 + m);
                sc_mul(t1, a, a); sc_mul(t1, t1, mp); sc_add(v, v, t1);
                sc_mul(mp, mp, mun);
            }
#pragma nounroll
            for (int m = 0; m < nl / 2; m++) {
                pw_ld_sc(a, w, t, SV_C0 + m);
                pw_ld_sc(b, w, t, SV_L0 + m);
                sc_mul(t1, a, b); sc_add(v, v, t1);
            }
            pw_st_msc(w, t, 0, 0, v);
#pragma nounroll
            for (int i = 0; i < 16; i++) {
                if ((i >> k) & 1) continue;         // R's term
                pw_ld_sc(a, w, t, SV_CG0 + i);
                pw_ld_sc(b, w, t, SV_N0 + (i >> k));
                sc_mul(t1, a, b);
                pw_st_msc(w, t, 0, 1 + i, t1);
            }
#pragma nounroll
            for (int i = 0; i < 32; i++) {
                if ((i >> k) & 1) continue;
                pw_ld_sc(a, w, t, SV_CH0 + i);
                pw_ld_sc(b, w, t, SV_L0 + (i >> k));
                sc_mul(t1, a, b);
                pw_st_msc(w, t, 0, 17 + i, t1);
            }
            w.status[t] = status;
            return;
        }
        // what prove_round_next needs: the challenge and the three affine points
        pw_st_sc(w, t, SV_Y, y);
        u32* rpts = prove_fold_rpts(w);
        ws_st_apt(rpts, N, t, 0, A[1]);
        ws_st_apt(rpts, N, t, 1, A[2]);
        ws_st_apt(rpts, N, t, 2, A[0]);
    } else {
        // proof.l = [l0, l1], proof.n = [n0]   (wnla.rs:126-133)
        sc l0, l1, n0;
        pw_ld_sc(l0, w, t, SV_L0); pw_ld_sc(l1, w, t, SV_L0 + 1); pw_ld_sc(n0, w, t, SV_N0);
        sc_to_be(pb + 832, l0);
        sc_to_be(pb + 864, l1);
        sc_to_be(pb + 896, n0);
    }
    w.status[t] = status;
}
// next commitment = com + y X + (y^2 - 1) R             (= wnla.commit(l_, n_), wnla.rs:186), rounds 1 .. 3 of the chain form
// group_lane >= 0: one of four consecutive lanes that share the sum (straus_affine_g4; identical table build and stores) -- small
// batches; -1: one lane per proof
HD void prove_round_next(const ProveWs& w, size_t t, int k, int group_lane) {
    const size_t N = w.N;
    (void)k;
    sc y, y2m1, one;
    pw_ld_sc(y, w, t, SV_Y);
    sc_set_u32(one, 1);
    sc_mul(y2m1, y, y);
    sc_sub(y2m1, y2m1, one);
    // the verifier's fast variable-base path (verify_core.h): affine window tables 1..16 of X and R from one four-level pass,
    // Jacobian accumulator, mixed additions, signed 5-bit windows.  Its buffers are carved out of the window-table workspace
    // (45 projective slots = 5.4 KB per proof): 32 table entries (2 KB), 28 running products (1.1 KB), the three points (192 B).
    uint8_t* sb = (uint8_t*)w.straus;
    apt_packed* atab = (apt_packed*)sb;
    u32* tscr = (u32*)(sb + (size_t)32 * sizeof(apt_packed) * N);
    u32* rpts = prove_fold_rpts(w);
    apt C;
    ws_ld_apt(C, rpts, N, t, 2);
    const atab_ref tab = atab_of(atab, N, t, 32);
    affine_tables_build(tab, tscr, rpts, N, t, 2);
    const int pslot[2] = {0, 1};
    glv_words<2> g;
    glv_split sp;
    glv_decompose(sp, y);
    glv_words_set<2>(g, 0, sp);
    glv_decompose(sp, y2m1);
    glv_words_set<2>(g, 1, sp);
    pt acc;
#if defined(__HIP_DEVICE_COMPILE__)
    if (group_lane >= 0) straus_affine_g4<2>(acc, tab, pslot, g, group_lane);
    else
#endif
        straus_affine<2>(acc, tab, pslot, g);
    pt_madd(acc, acc, C, apt_is_identity(C));
    pw_st_pt(w, t, PB_C, acc);
    (void)group_lane;
}
#if defined(__HIPCC__)
// Small calls (next_by_msm): part one of a round on SIXTEEN lanes per value.  Every lane runs the head (identical values and stores);
// then lane q folds l, c (q < nl / 2), n (q < nn / 2), updates the generator coefficients ch[q], ch[q + 16], cg[q], forms its share of
// the next round's scalars -- cg[q] n_[q >> k], ch[i] l_[i >> k] for i = q, q + 16 (the level's commitment, even slots | R, odd slots) and
// the same with the partner slot (X), the folded entries of OTHER lanes read by shuffle, never through memory -- and its terms of the
// three leading scalars, which meet by group sums.  The host launches no scalar kernel for the next round.  One lane's chain is then the
// head plus a dozen multiplications instead of the head plus 150.  Every lane of a group must be active.
__device__ __forceinline__ void sc_group_xor1(sc& r, const sc& a) {          // the value of the neighbouring lane (lane ^ 1)
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = __shfl_xor(a.v[i], 1, 64);
}
__device__ __forceinline__ void sc_group_read16(sc& r, const sc& a, int src) {
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = __shfl(a.v[i], src, 16);
}
__device__ __forceinline__ void prove_round_fold_lanes(const ProveWs& w, size_t t, int k, int q) {
    const int sh = k - 1, nl = 32 >> sh, nn = 16 >> sh;
    int32_t status = w.status[t];
    apt A[3];
    sc y;
    prove_round_fold_head(w, t, k, A, y, status);
    uint8_t* pb = w.proofs + (size_t)BPPP_U64_PROOF_BYTES * t;
    sc rho, rho_inv, mu, t1, zero;
    sc_set_u32(zero, 0);
    pw_ld_sc(rho, w, t, SV_RHO); pw_ld_sc(rho_inv, w, t, SV_RHOINV); pw_ld_sc(mu, w, t, SV_MU);
    // folds, in place: every lane's loads are consumed before its store is issued, and a wavefront's requests are served in order
    sc lq = zero, cq = zero, nq = zero;
    const bool has_l = q < nl / 2, has_n = q < nn / 2;
    {
        sc a0, a1;
        const int m = has_l ? q : 0;
        pw_ld_sc(a0, w, t, SV_L0 + 2 * m); pw_ld_sc(a1, w, t, SV_L0 + 2 * m + 1);
        sc_mul(t1, a1, y); sc_add(lq, a0, t1);
        pw_ld_sc(a0, w, t, SV_C0 + 2 * m); pw_ld_sc(a1, w, t, SV_C0 + 2 * m + 1);
        sc_mul(t1, a1, y); sc_add(cq, a0, t1);
        const int mn = has_n ? q : 0;
        pw_ld_sc(a0, w, t, SV_N0 + 2 * mn); pw_ld_sc(a1, w, t, SV_N0 + 2 * mn + 1);
        sc_mul(a0, a0, rho_inv);
        sc_mul(t1, a1, y); sc_add(nq, a0, t1);
        if (has_l) { pw_st_sc(w, t, SV_L0 + q, lq); pw_st_sc(w, t, SV_C0 + q, cq); }
        if (has_n) pw_st_sc(w, t, SV_N0 + q, nq);
    }
    if (k < 4) {
        // generator coefficients pick up this round's factor: ch[q], ch[q + 16], cg[q]
        sc ch0, ch1, cg;
        pw_ld_sc(ch0, w, t, SV_CH0 + q); pw_ld_sc(ch1, w, t, SV_CH0 + q + 16); pw_ld_sc(cg, w, t, SV_CG0 + q);
        if ((q >> sh) & 1) { sc_mul(ch0, ch0, y); pw_st_sc(w, t, SV_CH0 + q, ch0); }
        if (((q + 16) >> sh) & 1) { sc_mul(ch1, ch1, y); pw_st_sc(w, t, SV_CH0 + q + 16, ch1); }
        sc_mul(cg, cg, ((q >> sh) & 1) ? y : rho);
        pw_st_sc(w, t, SV_CG0 + q, cg);
        // rho <- mu, mu <- mu^2, rho^-1 <- (rho^-1)^2           (wnla.rs:180-181)
        sc mun;
        sc_mul(mun, mu, mu);
        pw_st_sc(w, t, SV_RHO, mu);
        pw_st_sc(w, t, SV_MU, mun);
        sc_mul(t1, rho_inv, rho_inv);
        pw_st_sc(w, t, SV_RHOINV, t1);
        const sc rinv2 = t1;
        // What the next round needs, all of it from here (the host launches no scalar kernel after a lane-form fold): per original
        // generator i in folded slot j = i >> k, coefficient c (ch or cg, updated above), folded entry e_j (l_ or n_):
        //   c e_j        is the generator's term of the level's commitment wnla.commit(l_, n_) -- summed from set 0 if j is even (job_e),
        //                and IS its term of the next round's R if j is odd (set 2; wnla.rs:140-150);
        //   c e_{j ^ 1}  (times rho' for even, rho'^-1 for odd j in g_vec) is its term of the next round's X (set 1; prove_round_scalars_*).
        // Slot 0 of the three sets: v over the even slots, v_r = v over the odd ones, v_x.  Folded entries of other lanes come by shuffle.
        sc mp, l_p, n_p, vt = zero, wx = zero, vxl = zero;
        const bool odd = (q & 1) != 0;
        sc_pow_u5(mp, mun, (unsigned)q + 1);                  // mu'^(q+1)
        sc_group_read16(l_p, lq, q ^ 1);                      // the slot's partner
        sc_group_read16(n_p, nq, q ^ 1);
        sc_mul(t1, nq, nq); sc_mul(t1, t1, mp);
        if (has_n) vt = t1;
        sc_mul(t1, cq, lq);
        if (has_l) sc_add(vt, vt, t1);
        sc v_e = odd ? zero : vt, v_o = odd ? vt : zero;
        sc_mul(t1, nq, n_p); sc_mul(t1, t1, mp); sc_mul(t1, t1, mun);      // n_2m n_2m+1 (mu'^2)^(m+1), q = 2m
        if (has_n && !odd) wx = t1;
        sc_mul(t1, cq, l_p);
        if (has_l) vxl = t1;
        sc_group_sum16(v_e); sc_group_sum16(v_o); sc_group_sum16(wx); sc_group_sum16(vxl);
        sc_add(t1, rinv2, rinv2);
        sc_mul(wx, wx, t1);
        sc_add(wx, wx, vxl);                                  // v_x = wvm(n0, n1, mu'^2) 2 rho'^-1 + <c0, l1> + <c1, l0>
        pw_st_msc(w, t, 0, 0, v_e);
        pw_st_msc(w, t, 1, 0, wx);
        pw_st_msc(w, t, 2, 0, v_o);
        const int j0 = q >> k, j1 = (q + 16) >> k;            // slots of h_q | g_q, and of h_{q+16}
        sc src, srx;
        sc_group_read16(src, nq, j0);
        sc_group_read16(srx, nq, j0 ^ 1);
        sc_mul(t1, cg, src);
        pw_st_msc(w, t, (j0 & 1) ? 2 : 0, 1 + q, t1);
        sc_mul(t1, srx, (j0 & 1) ? rinv2 : mu);
        sc_mul(t1, t1, cg);
        pw_st_msc(w, t, 1, 1 + q, t1);
        sc_group_read16(src, lq, j0);
        sc_group_read16(srx, lq, j0 ^ 1);
        sc_mul(t1, ch0, src);
        pw_st_msc(w, t, (j0 & 1) ? 2 : 0, 17 + q, t1);
        sc_mul(t1, ch0, srx);
        pw_st_msc(w, t, 1, 17 + q, t1);
        sc_group_read16(src, lq, j1);
        sc_group_read16(srx, lq, j1 ^ 1);
        sc_mul(t1, ch1, src);
        pw_st_msc(w, t, (j1 & 1) ? 2 : 0, 17 + q + 16, t1);
        sc_mul(t1, ch1, srx);
        pw_st_msc(w, t, 1, 17 + q + 16, t1);
    } else {
        // proof.l = [l0, l1], proof.n = [n0]   (wnla.rs:126-133): lanes 0 and 1 hold them
        if (q == 0) { sc_to_be(pb + 832, lq); sc_to_be(pb + 896, nq); }
        if (q == 1) sc_to_be(pb + 864, lq);
    }
    w.status[t] = status;
}
// The same on FOUR lanes per value, for batches of a few values per SIMD (next_by_msm): lane q owns the folded slots j = q, q + 4, ... of
// l | c and of n, and everything that hangs off a slot -- the 2^k generator coefficients of its block (ch / cg pick up this round's
// factor), their products with the folded entry (the level's commitment in its even slots, job_e; the next round's R in its odd ones)
// and with the partner slot's entry (the next round's X; lane q ^ 1 holds it: one shuffle), the slot's terms of the three leading
// scalars -- so nothing one lane computes is read by another except through shuffles.  No scalar kernel for the next round.  In-place folds: pass `it` reads slots 8 it .. 8 it + 7 and
// writes 4 it .. 4 it + 3, every lane's loads of a pass are issued before any of its stores, and a later pass reads beyond what
// earlier ones wrote.  One lane's chain: the head plus ~40 multiplications instead of ~150.
__device__ __forceinline__ void prove_round_fold_lanes4(const ProveWs& w, size_t t, int k, int q) {
    const int sh = k - 1, nls = 16 >> sh, nns = 8 >> sh;       // folded lengths
    int32_t status = w.status[t];
    apt A[3];
    sc y;
    prove_round_fold_head(w, t, k, A, y, status);
    uint8_t* pb = w.proofs + (size_t)BPPP_U64_PROOF_BYTES * t;
    sc rho, rho_inv, mu, mun, rinv2, t1, v_e, v_o, wx, vxl;
    pw_ld_sc(rho, w, t, SV_RHO); pw_ld_sc(rho_inv, w, t, SV_RHOINV); pw_ld_sc(mu, w, t, SV_MU);
    sc_mul(mun, mu, mu);                                          // mu' of the next level
    sc_mul(rinv2, rho_inv, rho_inv);                              // rho'^-1
    sc_set_u32(v_e, 0); sc_set_u32(v_o, 0); sc_set_u32(wx, 0); sc_set_u32(vxl, 0);
    const int blk = 1 << k;                                       // original generators per folded slot
#pragma nounroll
    for (int it = 0; it < (nls + 3) / 4; it++) {
        const int j = 4 * it + q;
        const bool has = j < nls;
        const int jj = has ? j : 0;
        sc a0, a1, lj, cj;
        pw_ld_sc(a0, w, t, SV_L0 + 2 * jj); pw_ld_sc(a1, w, t, SV_L0 + 2 * jj + 1);
        sc_mul(t1, a1, y); sc_add(lj, a0, t1);
        pw_ld_sc(a0, w, t, SV_C0 + 2 * jj); pw_ld_sc(a1, w, t, SV_C0 + 2 * jj + 1);
        sc_mul(t1, a1, y); sc_add(cj, a0, t1);
        if (has) { pw_st_sc(w, t, SV_L0 + j, lj); pw_st_sc(w, t, SV_C0 + j, cj); }
        if (k < 4) {
            sc l_p;
            sc_group_xor1(l_p, lj);                               // the partner slot j ^ 1 (lane q ^ 1, same pass)
            sc_mul(t1, cj, lj);
            if (has) { if (j & 1) sc_add(v_o, v_o, t1); else sc_add(v_e, v_e, t1); }
            sc_mul(t1, cj, l_p);
            if (has) sc_add(vxl, vxl, t1);
#pragma nounroll
            for (int r = 0; r < blk; r++) {
                const int i = jj * blk + r;
                sc ch;
                pw_ld_sc(ch, w, t, SV_CH0 + i);
                if (r >> sh) { sc_mul(ch, ch, y); if (has) pw_st_sc(w, t, SV_CH0 + i, ch); }
                sc_mul(t1, ch, lj);
                if (has) pw_st_msc(w, t, (j & 1) ? 2 : 0, 17 + i, t1);
                sc_mul(t1, ch, l_p);
                if (has) pw_st_msc(w, t, 1, 17 + i, t1);
            }
        } else if (has) {
            sc_to_be(pb + 832 + 32 * j, lj);                      // proof.l = [l0, l1]   (wnla.rs:126-133)
        }
    }
    sc mp, m4;
    sc_pow_u5(mp, mun, (unsigned)q + 1);                          // mu'^(j + 1) for this lane's first slot; the next one is mu'^4 further
    sc_mul(m4, mun, mun); sc_mul(m4, m4, m4);
#pragma nounroll
    for (int it = 0; it < (nns + 3) / 4; it++) {
        const int j = 4 * it + q;
        const bool has = j < nns;
        const int jj = has ? j : 0;
        sc a0, a1, nj;
        pw_ld_sc(a0, w, t, SV_N0 + 2 * jj); pw_ld_sc(a1, w, t, SV_N0 + 2 * jj + 1);
        sc_mul(a0, a0, rho_inv);
        sc_mul(t1, a1, y); sc_add(nj, a0, t1);
        if (has) pw_st_sc(w, t, SV_N0 + j, nj);
        if (k < 4) {
            sc n_p, xn;
            sc_group_xor1(n_p, nj);
            sc_mul(t1, nj, nj); sc_mul(t1, t1, mp);
            if (has) { if (j & 1) sc_add(v_o, v_o, t1); else sc_add(v_e, v_e, t1); }
            sc_mul(t1, nj, n_p); sc_mul(t1, t1, mp); sc_mul(t1, t1, mun);      // n_2m n_2m+1 (mu'^2)^(m+1), j = 2m
            if (has && !(j & 1)) sc_add(wx, wx, t1);
            sc_mul(xn, n_p, (j & 1) ? rinv2 : mu);                 // X's folded entry for this slot: rho' n[j+1] | rho'^-1 n[j-1]
            sc_mul(mp, mp, m4);
#pragma nounroll
            for (int r = 0; r < blk; r++) {
                const int i = jj * blk + r;
                sc cg;
                pw_ld_sc(cg, w, t, SV_CG0 + i);
                sc_mul(cg, cg, (r >> sh) ? y : rho);
                if (has) pw_st_sc(w, t, SV_CG0 + i, cg);
                sc_mul(t1, cg, nj);
                if (has) pw_st_msc(w, t, (j & 1) ? 2 : 0, 1 + i, t1);
                sc_mul(t1, cg, xn);
                if (has) pw_st_msc(w, t, 1, 1 + i, t1);
            }
        } else if (has) {
            sc_to_be(pb + 896, nj);                               // proof.n = [n0]
        }
    }
    if (k < 4) {
        // rho <- mu, mu <- mu^2, rho^-1 <- (rho^-1)^2           (wnla.rs:180-181)
        pw_st_sc(w, t, SV_RHO, mu);
        pw_st_sc(w, t, SV_MU, mun);
        pw_st_sc(w, t, SV_RHOINV, rinv2);
        // slot 0 of the three sets: v over the even slots (job_e), v_x, v_r = v over the odd slots
        prove_group_sum16(v_e, 4); prove_group_sum16(v_o, 4); prove_group_sum16(wx, 4); prove_group_sum16(vxl, 4);
        sc_add(t1, rinv2, rinv2);
        sc_mul(wx, wx, t1);
        sc_add(wx, wx, vxl);
        pw_st_msc(w, t, 0, 0, v_e);
        pw_st_msc(w, t, 1, 0, wx);
        pw_st_msc(w, t, 2, 0, v_o);
    }
    w.status[t] = status;
}
#endif

// the MSM jobs of the pipeline, in launch order
// The MSMs of the prover, over exactly the terms that are there.  Round 2 summed every slot of a contiguous range at full width; a
// quarter of those table additions had a zero digit by construction: hexadecimal digits and multiplicities below 2^5, the u64 value,
// blinding slots the reference leaves at zero (circuit.rs:264-298), and the even halves of R's folded vectors (wnla.rs:140-150).
#define BPPP_JOB(...) MsmJob j = __VA_ARGS__; return j
#define NOODD {-1, -1, -1, -1, -1}
HD MsmJob job_v() { BPPP_JOB({0, PB_V, 2, {0, 17}, {1, 1}, {64, 0}, NOODD}); }                                    // x g + s h[0]
HD MsmJob job_rcom() { BPPP_JOB({0, PB_RCOM, 2, {17, 26}, {1, 16}, {0, 0}, NOODD}); }
HD MsmJob job_co() { BPPP_JOB({1, PB_CO, 2, {17, 22}, {4, 3}, {0, 0}, NOODD}); }                                  // ro: h[4], h[8] stay zero
HD MsmJob job_cl() { BPPP_JOB({2, PB_CL, 4, {1, 17, 21, 26}, {16, 3, 3, 16}, {4, 0, 0, 5}, NOODD}); }            // digits | rl: h[3], h[7], h[8] zero | multiplicities
HD MsmJob job_cr() { BPPP_JOB({3, PB_CR, 3, {1, 17, 20}, {16, 2, 3}, {0, 0, 0}, NOODD}); }                        // r | rr: h[2], h[6..8] zero
HD MsmJob job_cs() { BPPP_JOB({0, PB_CS, 1, {1}, {42}, {0}, NOODD}); }
HD MsmJob job_x() { BPPP_JOB({1, PB_X, 1, {0}, {49}, {0}, NOODD}); }
// R of round k: v_r g + the odd halves (blocks of 2^(k-1) original generators) of g_vec and h_vec
HD MsmJob job_r(int k) { BPPP_JOB({2, PB_R, 3, {0, 1, 17}, {1, 8, 16}, {0, 0, 0}, {-1, k - 1, k - 1}}); }
// E of round k: the EVEN halves of the commitment C_{k-1} = wnla.commit(l, n) the round starts from, over the original generators with
// the coefficients stage F (k = 1) or the previous fold (k >= 2, next_by_msm) left in set 0.  Its odd halves ARE R of round k, term by
// term (R takes the odd folded slots of l and n with the same generator coefficients, wnla.rs:140-150), so C_{k-1} = E + R once slot 0
// holds v - v_r (prove_round_scalars_v) -- 22 or 25 terms here instead of the 43 or 49 of the whole commitment, no launch of its own
// (it rides with X | R), and one point addition in the fold's head.  (Round 1: l[25..32) is zero, so 13 even slots of h.)
HD MsmJob job_e(int k) { BPPP_JOB({0, PB_C, 3, {0, 1, 17}, {1, 8, k == 1 ? 13 : 16}, {0, 0, 0}, {-1, (k - 1) | BPPP_FB_EVEN, (k - 1) | BPPP_FB_EVEN}}); }
#undef NOODD
#undef BPPP_JOB

// the caller's `&mut Transcript` after prove: the state after the last wnla_challenge (a PRF operation: cur_flags = 7); a flagged
// instance (malformed input scalar or state) gets its input state back
HD void prove_export_state(const ProveWs& w, size_t t) {
    if (!w.states_out) return;
    uint8_t* out = w.states_out + (size_t)BPPP_TRANSCRIPT_STATE_BYTES * t;
    if (w.status[t] & ST_BAD_ENCODING) {
        if (w.states) {
            const uint8_t* in = w.states + (size_t)BPPP_TRANSCRIPT_STATE_BYTES * (w.n_states == 1 ? 0 : t);
#pragma nounroll
            for (int i = 0; i < BPPP_TRANSCRIPT_STATE_BYTES; i++) out[i] = in[i];
        } else {
            strobe_to_bytes(out, w.base, 2);
        }
        return;
    }
    strobe tr;
    ws_ld_strobe(tr, w.tstate, w.N, t);
    strobe_to_bytes(out, tr, 7);
}

}  // namespace bppp
