// Generic batched `ArithmeticCircuit::verify` (circuit.rs:154-256) for a circuit SHARED by the batch: arbitrary dim_nm, dim_no,
// k, dim_nv (dim_nl = dim_nv k, dim_nw = 2 dim_nm + dim_no), dense W_m / W_l, a_m / a_l, f_l / f_m, and the partition closure
// as four index tables.  The u64 / reciprocal range proofs are NOT served from here (their matrices depend on a per-proof
// challenge and collapse to closed forms: verify_core.h, recip_core.h); this is the crate's general `circuit` API surface.
//
// What is restructured: the reference slices W into eight dense matrices per call (collect_m_rl / collect_m_o,
// circuit.rs:616-653) and multiplies each by lambda_vec / mu_vec (collect_c, :584-614).  Every entry of the six resulting
// vectors is  <lambda_vec, W_l[:, col]> - <mu_vec, W_m[:, col]>  for ONE column `col` of W (or zero where the partition
// returns None), so the host stores W_l and W_m once, column-compressed, plus the column index per output entry; the device
// walks the non-zeros.  Same field elements, same canonical bytes downstream.
//
// Per instance this stage produces C0 (circuit.rs:206,230-235), c, rho, mu and the transcript state, then hands them to the
// generic WNLA stage (wnla_core.h) exactly as recip_core.h does.
//
// Proof layout (per instance, proof_bytes = 64 (4 + 2 rounds) + 32 (nl + nn)):  c_l, c_r, c_o, c_s | r[rounds] | x[rounds] | l | n
#pragma once
#include "wnla_core.h"

namespace bppp {

struct CircuitDev {
    int nm, no, k, nl, nv, nw, f_l, f_m;
    const int* colptr_l;    // [nw + 1]
    const int* rows_l;      // [nnz_l]
    const u32* vals_l;      // [nnz_l][8]
    const int* colptr_m;
    const int* rows_m;
    const u32* vals_m;
    const int* colmap;      // [3 nm + 3 nv]: W column behind c_nL[j], c_nR[j], c_nO[j] (j < nm), c_lL[j], c_lR[j], c_lO[j] (j < nv); -1 = zero
    const u32* a_l;         // [nl][8]
    const u32* a_m;         // [nm][8]
    // optional per-instance matrix entries (a circuit whose VALUES depend on a per-proof challenge but whose sparsity pattern
    // does not -- the reciprocal range proof's, reciprocal.rs:150-214): inst_l[p] / inst_m[p] >= 0 selects scalar slot
    // inst_vals[slot] of the instance instead of the shared vals_l[p] / vals_m[p]
    const int* inst_l;
    const int* inst_m;
    const u32* inst_vals;   // [n_inst * 8][N]
};
struct CircuitWs {
    size_t N;
    CircuitDev cd;
    int rounds, NG, NH;
    size_t proof_bytes;
    const uint8_t* commitments;      // N x k x 64
    const uint8_t* proofs;           // N x proof_bytes
    int32_t* status;
    u32* tstate;                     // [52][N]
    u32* lamv;                       // [nl * 8][N]  lambda_vec
    u32* muv;                        // [nm * 8][N]  mu_vec
    u32* coef;                       // [(3 nm + 3 nv) * 8][N]
    u32* sc0;                        // [(1 + nm + 4 + k) * 8][N]: ps_tau | pn_tau[nm] | tau^-1, -delta, tau, -tau^2 | 2 tau^3 coef_i (k)
    u32* pts;                        // [(4 + k) * 16][N] packed affine: c_s, c_o, c_l, c_r, v_0 .. v_{k-1}
    u32* acc;                        // [30][N]
    u32* pfix;                       // [30][N]
    pt_slot* straus;                 // [N][5][9]
    // fast path of C0's variable-base part (round 6): affine window tables of the 4 + k points behind the WNLA stage's round-point tables
    // (bppp_generic.hip: wnla_fast_setup with extra_points = 4 + k), built by circuit_c0_tables; null = the projective tables above
    apt_packed* atab;
    u32* tscr;
    int atab_first;
    uint8_t* wn_commit;              // N x 64
    uint8_t* wn_c;                   // N x NH x 32
    uint8_t* wn_rho;                 // N x 32
    uint8_t* wn_mu;                  // N x 32
    FbTable fb;                      // bases: 0 g | 1..NG g_vec||g_vec_ | NG+1.. h_vec||h_vec_
    strobe base;
    TranscriptIo tio;                // caller's transcripts (circuit.rs:154 `t: &mut Transcript`)
};

}  // namespace bppp
#include <vector>
namespace bppp {
// Host side of the circuit description: validation, column compression of W_l / W_m, the output-entry -> column map
// (collect_m_rl / collect_m_o, circuit.rs:616-653).  Shared by the C-ABI library and the host emulation in tests/emul.
struct CircuitHostData {
    std::vector<int> cpl, rl, cpm, rm, colmap;
    std::vector<u32> vl, vm, al, am;
};
inline bool circuit_host_build(CircuitHostData& h, const size_t dims[6], const uint8_t* W_m, const uint8_t* W_l, const uint8_t* a_m,
                               const uint8_t* a_l, const int32_t* part_lo, const int32_t* part_ll, const int32_t* part_lr,
                               const int32_t* part_no) {
    const size_t nm = dims[0], no = dims[1], k = dims[2], nl = dims[3], nv = dims[4], nw = dims[5];
    if (nm == 0 || nv == 0 || k == 0 || nl != nv * k || nw != 2 * nm + no) return false;   // circuit.rs:100-106
    for (size_t j = 0; j < nv; j++)
        if (part_lo[j] >= (int32_t)no || part_ll[j] >= (int32_t)no || part_lr[j] >= (int32_t)no) return false;
    for (size_t j = 0; j < nm; j++)
        if (part_no[j] >= (int32_t)no) return false;
    auto parse = [](const uint8_t* b, u32 wv[8]) -> bool {    // canonical scalars only, as k256 deserialisation requires
        sc x;
        if (!sc_from_be(x, b)) return false;
        for (int i = 0; i < 8; i++) wv[i] = x.v[i];
        return true;
    };
    auto compress = [&](const uint8_t* W, size_t rows, std::vector<int>& colptr, std::vector<int>& ridx, std::vector<u32>& vals) -> bool {
        colptr.assign(nw + 1, 0);
        for (size_t col = 0; col < nw; col++) {
            colptr[col] = (int)ridx.size();
            for (size_t r = 0; r < rows; r++) {
                u32 wv[8];
                if (!parse(W + 32 * (r * nw + col), wv)) return false;
                u32 any = 0;
                for (int i = 0; i < 8; i++) any |= wv[i];
                if (!any) continue;
                ridx.push_back((int)r);
                vals.insert(vals.end(), wv, wv + 8);
            }
        }
        colptr[nw] = (int)ridx.size();
        return true;
    };
    if (!compress(W_l, nl, h.cpl, h.rl, h.vl) || !compress(W_m, nm, h.cpm, h.rm, h.vm)) return false;
    h.al.assign(nl * 8, 0);
    h.am.assign(nm * 8, 0);
    for (size_t i = 0; i < nl; i++) if (!parse(a_l + 32 * i, &h.al[i * 8])) return false;
    for (size_t i = 0; i < nm; i++) if (!parse(a_m + 32 * i, &h.am[i * 8])) return false;
    h.colmap.assign(3 * nm + 3 * nv, -1);
    auto ocol = [&](int32_t j_) { return j_ >= 0 ? (int)(2 * nm + (size_t)j_) : -1; };
    for (size_t j = 0; j < nm; j++) {
        h.colmap[j] = (int)j;
        h.colmap[nm + j] = (int)(nm + j);
        h.colmap[2 * nm + j] = ocol(part_no[j]);
    }
    for (size_t j = 0; j < nv; j++) {
        h.colmap[3 * nm + j] = ocol(part_ll[j]);
        h.colmap[3 * nm + nv + j] = ocol(part_lr[j]);
        h.colmap[3 * nm + 2 * nv + j] = ocol(part_lo[j]);
    }
    return true;
}

HD void cd_ld_sc(sc& r, const u32* base, int idx) {
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = base[(size_t)idx * 8 + i];
}

// lambda_vec, mu_vec and the six coefficient vectors c_nL, c_nR, c_nO, c_lL, c_lR, c_lO (collect_lambda, collect_c:
// circuit.rs:584-614) for one instance, into limb-major arrays of stride N.  Shared by the verifier and the prover.
HD void circuit_collect(const CircuitDev& cd, u32* lamv_, u32* muv_, u32* coef_, size_t N, size_t t, const sc& lambda, const sc& mu,
                        const sc& mu_inv, sc& lam_nv, sc& mu_nv) {
    const int nm = cd.nm, nv = cd.nv, nl = cd.nl, k = cd.k;
    struct { u32 *lamv, *muv, *coef; } w = {lamv_, muv_, coef_};
    sc one, zero, t1, t2;
    sc_set_u32(one, 1);
    sc_set_u32(zero, 0);
    // lambda_vec = e(lambda, nl) [- tensor terms when f_l && f_m]   (collect_lambda, circuit.rs:584-599)
    sc lp = one;
#pragma nounroll
    for (int i = 0; i < nl; i++) { ws_st8(w.lamv, N, t, i, lp.v); sc_mul(lp, lp, lambda); }
    // lambda^nv, mu^nv
    lam_nv = one;
    mu_nv = one;
#pragma nounroll
    for (int i = 0; i < nv; i++) { sc_mul(lam_nv, lam_nv, lambda); sc_mul(mu_nv, mu_nv, mu); }
    if (cd.f_l && cd.f_m) {
        sc mj = one, lj = one;      // (mu^nv)^j, (lambda^nv)^j
#pragma nounroll
        for (int j = 0; j < k; j++) {
            sc li = one, mi = one;  // lambda^i, mu^i
#pragma nounroll
            for (int i = 0; i < nv; i++) {
                const int idx = j * nv + i;
                if (idx < nl) {
                    sc cur;
                    ws_ld8(cur.v, w.lamv, N, t, idx);
                    sc_mul(t1, li, mu);
                    sc_mul(t1, t1, mj);
                    sc_mul(t2, mi, lj);
                    sc_add(t1, t1, t2);
                    sc_sub(cur, cur, t1);
                    ws_st8(w.lamv, N, t, idx, cur.v);
                }
                sc_mul(li, li, lambda);
                sc_mul(mi, mi, mu);
            }
            sc_mul(mj, mj, mu_nv);
            sc_mul(lj, lj, lam_nv);
        }
    }
    // mu_vec = e(mu, nm) * mu                                            (circuit.rs:169)
    sc mp = mu;
#pragma nounroll
    for (int i = 0; i < nm; i++) { ws_st8(w.muv, N, t, i, mp.v); sc_mul(mp, mp, mu); }
    // the six coefficient vectors (collect_c): column walks; n-type entries scaled by mu^-(j+1) (diag_inv)
    const int nout = 3 * nm + 3 * nv;
    sc mip = mu_inv;
#pragma nounroll
    for (int o = 0; o < nout; o++) {
        const int col = cd.colmap[o];
        sc a = zero;
        if (col >= 0) {      // circuit data: the same for every lane
#pragma nounroll
            for (int p = cd.colptr_l[col]; p < cd.colptr_l[col + 1]; p++) {
                sc x, val;
                ws_ld8(x.v, w.lamv, N, t, cd.rows_l[p]);
                if (cd.inst_l && cd.inst_l[p] >= 0) ws_ld8(val.v, cd.inst_vals, N, t, cd.inst_l[p]);
                else cd_ld_sc(val, cd.vals_l, p);
                sc_mul(x, x, val);
                sc_add(a, a, x);
            }
#pragma nounroll
            for (int p = cd.colptr_m[col]; p < cd.colptr_m[col + 1]; p++) {
                sc x, val;
                ws_ld8(x.v, w.muv, N, t, cd.rows_m[p]);
                if (cd.inst_m && cd.inst_m[p] >= 0) ws_ld8(val.v, cd.inst_vals, N, t, cd.inst_m[p]);
                else cd_ld_sc(val, cd.vals_m, p);
                sc_mul(x, x, val);
                sc_sub(a, a, x);
            }
        }
        if (o < 3 * nm) {
            const int j = o % nm;
            if (j == 0) mip = mu_inv;
            sc_mul(a, a, mip);
            sc_mul(mip, mip, mu_inv);
        }
        ws_st8(w.coef, N, t, o, a.v);
    }
}

HD void circuit_phase1(const CircuitWs& w, size_t t) {
    const size_t N = w.N;
    const CircuitDev& cd = w.cd;
    const int nm = cd.nm, nv = cd.nv, nl = cd.nl, k = cd.k;
    int32_t status = ST_OK;
    const uint8_t* pp = w.proofs + w.proof_bytes * t;
    const uint8_t* pv = w.commitments + (size_t)64 * k * t;
    apt CL, CR, CO, CS;
    bool ok = apt_from_xy64(CL, pp) & apt_from_xy64(CR, pp + 64) & apt_from_xy64(CO, pp + 128) & apt_from_xy64(CS, pp + 192);
    strobe tr;
    tio_begin(tr, status, w.tio, w.base, t);
    // a malformed instance runs on harmless values (identity points); its status forces accept = 0
#pragma nounroll
    for (int i = 0; i < k; i++) { apt V; ok &= apt_from_xy64(V, pv + 64 * i); }
    apt zero_pt;
    fe_set_u32(zero_pt.x, 0); fe_set_u32(zero_pt.y, 0);
    if (!ok) { status |= ST_BAD_ENCODING; CL = zero_pt; CR = zero_pt; CO = zero_pt; CS = zero_pt; }
    app_point(tr, "commitment_cl", CL);                                   // circuit.rs:155-159
    app_point(tr, "commitment_cr", CR);
    app_point(tr, "commitment_co", CO);
#pragma nounroll
    for (int i = 0; i < k; i++) {
        apt V;
        (void)apt_from_xy64(V, pv + 64 * i);
        if (!ok) V = zero_pt;
        app_point(tr, "commitment_v", V);
        ws_st_apt(w.pts, N, t, 4 + i, V);
    }
    sc rho, lambda, beta, delta, tau;
    bool cok = t_get_challenge(tr, "circuit_rho", rho);                   // circuit.rs:161-164
    cok &= t_get_challenge(tr, "circuit_lambda", lambda);
    cok &= t_get_challenge(tr, "circuit_beta", beta);
    cok &= t_get_challenge(tr, "circuit_delta", delta);
    app_point(tr, "commitment_cs", CS);                                   // circuit.rs:189
    cok &= t_get_challenge(tr, "circuit_tau", tau);                       // circuit.rs:191
    if (!cok) {
        status |= ST_DEGENERATE;
        sc_set_u32(rho, 1); sc_set_u32(lambda, 1); sc_set_u32(beta, 1); sc_set_u32(delta, 1); sc_set_u32(tau, 1);
    }
    ws_st_strobe(w.tstate, N, t, tr);
    ws_st_apt(w.pts, N, t, 0, CS); ws_st_apt(w.pts, N, t, 1, CO); ws_st_apt(w.pts, N, t, 2, CL); ws_st_apt(w.pts, N, t, 3, CR);
    sc mu, one, zero, t1, t2;
    sc_set_u32(one, 1);
    sc_set_u32(zero, 0);
    sc_mul(mu, rho, rho);                                                 // circuit.rs:166
    sc_to_be(w.wn_rho + 32 * t, rho);
    sc_to_be(w.wn_mu + 32 * t, mu);
    // mu^-1, tau^-1, delta^-1 from one inversion (the reference unwrap()s these: zero -> DEGENERATE)
    const bool zero_inv = sc_is_zero(mu) | sc_is_zero(tau) | sc_is_zero(delta);
    if (zero_inv) status |= ST_DEGENERATE;
    sc m_ = sc_is_zero(mu) ? one : mu, t_ = sc_is_zero(tau) ? one : tau, d_ = sc_is_zero(delta) ? one : delta;
    sc mt, mtd, inv, mu_inv, tau_inv, delta_inv;
    sc_mul(mt, m_, t_);
    sc_mul(mtd, mt, d_);
    sc_inv(inv, mtd);
    sc_mul(delta_inv, inv, mt);
    sc_mul(inv, inv, d_);              // (mu tau)^-1
    sc_mul(mu_inv, inv, t_);
    sc_mul(tau_inv, inv, m_);
    sc tau2, tau3, two_tau3, t3di;
    sc_mul(tau2, tau, tau);
    sc_mul(tau3, tau2, tau);
    sc_add(two_tau3, tau3, tau3);
    sc_mul(t3di, tau3, delta_inv);
    sc lam_nv, mu_nv;
    circuit_collect(cd, w.lamv, w.muv, w.coef, N, t, lambda, mu, mu_inv, lam_nv, mu_nv);
    sc mp;
    // pn_tau, ps_tau                                                     (circuit.rs:196-206)
    sc ps = zero;
    mp = mu;
#pragma nounroll
    for (int j = 0; j < nm; j++) {
        sc cL, cR, cO, pn;
        ws_ld8(cL.v, w.coef, N, t, j);
        ws_ld8(cR.v, w.coef, N, t, nm + j);
        ws_ld8(cO.v, w.coef, N, t, 2 * nm + j);
        sc_mul(pn, cO, t3di);
        sc_mul(t1, cL, tau2);
        sc_sub(pn, pn, t1);
        sc_mul(t1, cR, tau);
        sc_add(pn, pn, t1);
        ws_st8(w.sc0, N, t, 1 + j, pn.v);
        sc_mul(t1, pn, pn);
        sc_mul(t1, t1, mp);
        sc_add(ps, ps, t1);
        sc_mul(mp, mp, mu);
    }
    sc dl = zero, dm = zero;     // <lambda_vec, a_l>, <mu_vec, a_m>
#pragma nounroll
    for (int i = 0; i < nl; i++) { sc x, a; ws_ld8(x.v, w.lamv, N, t, i); cd_ld_sc(a, cd.a_l, i); sc_mul(x, x, a); sc_add(dl, dl, x); }
#pragma nounroll
    for (int i = 0; i < nm; i++) { sc x, a; ws_ld8(x.v, w.muv, N, t, i); cd_ld_sc(a, cd.a_m, i); sc_mul(x, x, a); sc_add(dm, dm, x); }
    sc_sub(t1, dl, dm);
    sc_mul(t1, t1, two_tau3);
    sc_add(ps, ps, t1);
    ws_st8(w.sc0, N, t, 0, ps.v);
    // variable-base scalars: tau^-1 c_s - delta c_o + tau c_l - tau^2 c_r + tau^3 * 2 sum_i coef_i v_i   (circuit.rs:176-187, 230-235)
    ws_st8(w.sc0, N, t, 1 + nm, tau_inv.v);
    sc_neg(t1, delta);
    ws_st8(w.sc0, N, t, 2 + nm, t1.v);
    ws_st8(w.sc0, N, t, 3 + nm, tau.v);
    sc_neg(t1, tau2);
    ws_st8(w.sc0, N, t, 4 + nm, t1.v);
    {
        sc lpow = one, mpow = mu;   // lambda^(nv i), mu^(nv i + 1)      (linear_comb_coef, circuit.rs:559-570)
#pragma nounroll
        for (int i = 0; i < k; i++) {
            sc cf = zero;
            if (cd.f_l) sc_add(cf, cf, lpow);
            if (cd.f_m) sc_add(cf, cf, mpow);
            sc_mul(cf, cf, two_tau3);
            ws_st8(w.sc0, N, t, 5 + nm + i, cf.v);
            sc_mul(lpow, lpow, lam_nv);
            sc_mul(mpow, mpow, mu_nv);
        }
    }
    // c = cr_tau (9) || cl_tau (nv) || zeros up to NH                    (circuit.rs:208-229)
    uint8_t* cw = w.wn_c + (size_t)t * w.NH * 32;
    sc_to_be(cw, one);
    sc_mul(t1, beta, tau_inv);
    sc_to_be(cw + 32, t1);
    sc bt = beta;
#pragma nounroll
    for (int i = 2; i < 9; i++) { sc_mul(bt, bt, tau); sc_to_be(cw + (size_t)i * 32, bt); }
    sc lp = lambda;              // lambda^(j+1)
    sc mq;
    sc_mul(mq, mu, mu);          // mu^(j+2)
#pragma nounroll
    for (int j = 0; j < nv; j++) {
        sc lL, lR, lO, cl;
        ws_ld8(lL.v, w.coef, N, t, 3 * nm + j);
        ws_ld8(lR.v, w.coef, N, t, 3 * nm + nv + j);
        ws_ld8(lO.v, w.coef, N, t, 3 * nm + 2 * nv + j);
        sc_mul(cl, lO, t3di);
        sc_mul(t1, lL, tau2);
        sc_sub(cl, cl, t1);
        sc_mul(t1, lR, tau);
        sc_add(cl, cl, t1);
        sc_add(cl, cl, cl);
        if (j < nv - 1) {        // c_l0 has nv - 1 entries (collect_cl0, circuit.rs:572-582); the ragged subtraction leaves the last one
            if (cd.f_l) sc_sub(cl, cl, lp);
            if (cd.f_m) sc_add(cl, cl, mq);
        }
        sc_to_be(cw + (size_t)(9 + j) * 32, cl);
        sc_mul(lp, lp, lambda);
        sc_mul(mq, mq, mu);
    }
#pragma nounroll
    for (int i = 9 + nv; i < w.NH; i++) sc_to_be(cw + (size_t)i * 32, zero);
    w.status[t] = status;
}
// C0 fixed-base half: ps_tau g + <g_vec, pn_tau>  (bases 0..nm of the table)
HD void circuit_c0_fixed_ranges(FbRanges& rg, const CircuitWs& w) { fb_ranges_one(rg, 0, 0, 1 + w.cd.nm); }
HD void circuit_c0_fixed_store(const CircuitWs& w, size_t t, const pt& total) { ws_st_pt(w.pfix, w.N, t, total); }
// C0 variable-base half: the 4 + k points in groups of at most 5 per shared-doubling pass
// the 4 + k points of C0's variable-base part (c_s, c_o, c_l, c_r, v_0 .. v_{k-1}: w.pts) as affine window tables 1P .. 16P, four batched inversions
HD void circuit_c0_tables(const CircuitWs& w, size_t t) {
    affine_tables_build(atab_of(w.atab, w.N, t) + w.atab_first, w.tscr, w.pts, w.N, t, 4 + w.cd.k);
}
// one chunk of at most five points of the sum on the fast path: signed 5-bit windows over the GLV halves, Jacobian accumulator, mixed additions
template <int M>
HD void circuit_c0_chunk(pt& acc, const CircuitWs& w, size_t t, int first) {
    const size_t N = w.N;
    int pslot[M];
    glv_words<M> g;
#pragma unroll
    for (int j = 0; j < M; j++) {
        pslot[j] = first + j;
        sc kk;
        ws_ld8(kk.v, w.sc0, N, t, 1 + w.cd.nm + first + j);
        glv_split sp;
        glv_decompose(sp, kk);
        glv_words_set<M>(g, j, sp);
    }
    straus_affine<M>(acc, atab_of(w.atab, N, t) + w.atab_first, pslot, g);
}
HD void circuit_c0_var(const CircuitWs& w, size_t t) {
    const size_t N = w.N;
    const int npts = 4 + w.cd.k;
    pt total;
    pt_set_identity(total);
    if (w.atab) {
        // (circuit.rs:230-235: tau^-1 c_s - delta c_o + tau c_l - tau^2 c_r + sum_i 2 tau^3 coef_i v_i, five points at a time)
#pragma nounroll
        for (int first = 0; first < npts; first += 5) {
            const int m = npts - first < 5 ? npts - first : 5;
            pt acc;
            switch (m) {
            case 5: circuit_c0_chunk<5>(acc, w, t, first); break;
            case 4: circuit_c0_chunk<4>(acc, w, t, first); break;
            case 3: circuit_c0_chunk<3>(acc, w, t, first); break;
            case 2: circuit_c0_chunk<2>(acc, w, t, first); break;
            default: circuit_c0_chunk<1>(acc, w, t, first); break;
            }
            if (first == 0) total = acc;
            else pt_add(total, total, acc);
        }
        ws_st_pt(w.acc, N, t, total);
        return;
    }
    pt_slot* tbl = w.straus + t * (5 * BPPP_STRAUS_ENTRIES);
#pragma nounroll
    for (int first = 0; first < npts; first += 5) {
        const int m = npts - first < 5 ? npts - first : 5;
        glv_split rs[5];
#pragma nounroll
        for (int j = 0; j < m; j++) {
            apt P;
            ws_ld_apt(P, w.pts, N, t, first + j);
            straus_build_table(tbl + j * BPPP_STRAUS_ENTRIES, P);
            sc kk;
            ws_ld8(kk.v, w.sc0, N, t, 1 + w.cd.nm + first + j);
            glv_decompose(rs[j], kk);
        }
        pt acc;
        straus_msm_glv(acc, tbl, rs, m);
        pt_add(total, total, acc);
    }
    ws_st_pt(w.acc, N, t, total);
}
#if defined(__HIPCC__)
// The same sum on L (a power of two >= 4 + k) lanes per instance, for calls that leave the chip empty: lane p builds the window table of
// point p by itself (a Jacobian chain and ONE inversion: affine_table_one), runs the one-point sum 2 tau^3 coef_p v_p (or c_s, c_o, c_l,
// c_r with their scalars) over it -- 130 doublings + 52 additions, where the one-lane kernel walks 130 doublings + 52 additions PER POINT
// in chunks of five -- and the L partial sums meet by shuffles.  More work in all (every lane doubles for itself), a third of the chain:
// one verify of `mixed_k2` spent 3.0 of its 4.7 ms in the one-lane kernel (round 6).  Every lane of a group must be active.
__device__ __forceinline__ void circuit_c0_var_points(const CircuitWs& w, size_t t, int p, int L) {
    const size_t N = w.N;
    const int npts = 4 + w.cd.k;
    pt acc;
    pt_set_identity(acc);
    if (p < npts) {
        const atab_ref tab = atab_of(w.atab, N, t) + w.atab_first;
        apt P;
        ws_ld_apt(P, w.pts, N, t, p);
        affine_table_one(tab + p * 16, P, 0);
        int pslot[1] = {p};
        glv_words<1> g;
        sc kk;
        ws_ld8(kk.v, w.sc0, N, t, 1 + w.cd.nm + p);
        glv_split sp;
        glv_decompose(sp, kk);
        glv_words_set<1>(g, 0, sp);
        straus_affine<1>(acc, tab, pslot, g);
    }
#pragma nounroll
    for (int m = 1; m < L; m <<= 1) {
        pt o;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            o.X.v[i] = __shfl_xor(acc.X.v[i], m, 64);
            o.Y.v[i] = __shfl_xor(acc.Y.v[i], m, 64);
            o.Z.v[i] = __shfl_xor(acc.Z.v[i], m, 64);
        }
        pt_add(acc, acc, o);
    }
    if (p == 0) ws_st_pt(w.acc, N, t, acc);
}
#endif
HD void circuit_c0_finish(const CircuitWs& w, size_t t) {
    pt a, f;
    ws_ld_pt(a, w.acc, w.N, t);
    ws_ld_pt(f, w.pfix, w.N, t);
    pt_add(a, a, f);
    apt c0;
    pt_to_affine(c0, a);
    apt_to_xy64(w.wn_commit + 64 * t, c0);
}

}  // namespace bppp
