// Generic batched `ReciprocalRangeProofProtocol::prove` (reciprocal.rs:110-146) for runtime dim_nd / dim_np.  The protocol's
// arithmetic circuit (make_circuit, reciprocal.rs:150-214) depends on the per-proof challenge e only through VALUES (-e on a
// diagonal of W_m, -1/(e + j) in dim_np columns of W_l); its sparsity pattern is fixed.  So the generic circuit prover
// (circuit_prove_core.h) runs on a shared pattern with per-instance value slots (CircuitDev::inst_*), after this stage has
// produced, per instance: e, the reciprocals r_i = 1/(d_i + e), the blinded pole commitment proof.r, the circuit witness
// v = [x, r_0, .., r_{nd-1}], s_v = s + r_blind, and the circuit's public input V + proof.r.
//
// Prover scalars `rnd` per instance, in the reference's draw order: r_blind (reciprocal.rs:121), then the circuit prover's
// 18 + (dim_nd + 1) + dim_nd.   Proof layout (the generic verifier's): c_l, c_r, c_o, c_s | r | x | reciprocal r | l | n.
#pragma once
#include "circuit_prove_core.h"

namespace bppp {

struct RecipProveWs {
    size_t N;
    int nd, np, NG, NH, n_rnd;
    const uint8_t *commitments, *x, *s, *digits, *m, *rnd;   // C-ABI layouts (device): n x 64, n x 32, n x 32, n x nd x 32, n x np x 32, n x n_rnd x 32
    int32_t* status;
    u32* tstate;
    u32* inst_vals;      // [(1 + np) * 8][N]: -e, -1/(e + j)
    u32* scr;            // [(nd + np) * 8][N] prefix products of the batched inversion
    u32* msc;            // scalar set 0 of the circuit prover (slot = table base index), zeroed by the host
    u32* pbuf;           // [30][N]
    uint8_t *cp_v, *cp_sv, *cp_wr, *cp_vpts;   // inputs of the circuit prover: n x (nd + 1) x 32, n x 32, n x nd x 32, n x 64
    uint8_t* proof_r;    // n x 64
    FbTable fb;
    FbTable fb_ct;       // "ct_prover": the sum over the reciprocals (secret: reciprocal.rs:118) in the full-scan form when ct != 0
    int ct;
    strobe base;
    TranscriptIo tio;    // caller's transcripts (reciprocal.rs:109 `t: &mut Transcript`)
    int divergent_positions;
};
HD void recip_prove_ranges(FbRanges& rg, const RecipProveWs& w) { fb_ranges_one(rg, 1 + w.NG, 1 + w.NG, 10 + w.nd); }   // h_vec[0 .. 9 + nv)

// transcript to e, the reciprocals and the circuit's per-instance matrix values, scalar set of commit_poles (reciprocal.rs:111-121)
HD void recip_prove_stage_r1(const RecipProveWs& w, size_t t) {
    const size_t N = w.N;
    const int nd = w.nd, np = w.np;
    int32_t status = ST_OK;
    apt V;
    bool ok = apt_from_xy64(V, w.commitments + 64 * t);
    if (!ok) { fe_set_u32(V.x, 0); fe_set_u32(V.y, 0); }
    sc xs, ss, rb, one, zero, e;
    sc_set_u32(one, 1);
    sc_set_u32(zero, 0);
    ok &= sc_from_be(xs, w.x + 32 * t);
    ok &= sc_from_be(ss, w.s + 32 * t);
    ok &= sc_from_be(rb, w.rnd + (size_t)t * w.n_rnd * 32);
#pragma nounroll
    for (int j = 0; j < np; j++) { sc mj; ok &= sc_from_be(mj, w.m + ((size_t)t * np + j) * 32); }
    strobe tr;
    tio_begin(tr, status, w.tio, w.base, t);
    app_point(tr, "reciprocal_commitment", V);                            // reciprocal.rs:111
    if (!t_get_challenge(tr, "reciprocal_challenge", e)) { status |= ST_DEGENERATE; e = one; }
    ws_st_strobe(w.tstate, N, t, tr);
    // one inversion for d_i + e (i < nd) and e + j (j < np): prefix products forward, peel backwards
    bool zero_inv = false;
    sc prod = one;
#pragma nounroll
    for (int i = 0; i < nd + np; i++) {
        sc a;
        if (i < nd) { sc d; ok &= sc_from_be(d, w.digits + ((size_t)t * nd + i) * 32); sc_add(a, d, e); }
        else { sc js; sc_set_u32(js, (u32)(i - nd)); sc_add(a, e, js); }
        if (sc_is_zero(a)) { zero_inv = true; a = one; }
        ws_st8(w.scr, N, t, i, prod.v);
        sc_mul(prod, prod, a);
    }
    if (zero_inv) status |= ST_DEGENERATE;                                // the reference's invert().unwrap()
    sc inv;
    sc_inv(inv, prod);
    u32* m0 = w.msc;
    sc neg;
    sc_neg(neg, e);
    ws_st8(w.inst_vals, N, t, 0, neg.v);                                  // W_m[i][i + nm] = -e   (reciprocal.rs:162)
#pragma nounroll
    for (int i = nd + np - 1; i >= 0; i--) {
        sc a, pre, ai;
        if (i < nd) { sc d; (void)sc_from_be(d, w.digits + ((size_t)t * nd + i) * 32); sc_add(a, d, e); }
        else { sc js; sc_set_u32(js, (u32)(i - nd)); sc_add(a, e, js); }
        if (sc_is_zero(a)) a = one;
        ws_ld8(pre.v, w.scr, N, t, i);
        sc_mul(ai, inv, pre);
        sc_mul(inv, inv, a);
        if (i < nd) {
            sc_to_be(w.cp_wr + ((size_t)t * nd + i) * 32, ai);            // w_r = r                (reciprocal.rs:137)
            sc_to_be(w.cp_v + ((size_t)t * (nd + 1) + 1 + i) * 32, ai);   // v = [x, r...]          (reciprocal.rs:123-124)
            ws_st8(m0, N, t, 1 + w.NG + 9 + i, ai.v);                     // commit_poles: h_vec[9 + i] r_i
        } else {
            sc_neg(ai, ai);
            ws_st8(w.inst_vals, N, t, 1 + (i - nd), ai.v);                // W_l[i + 1][j + 2 nm] = -1/(e + j)  (reciprocal.rs:179-183)
        }
    }
    ws_st8(m0, N, t, 1 + w.NG, rb.v);                                     // commit_poles: h_vec[0] r_blind (reciprocal.rs:93-95)
    sc_to_be(w.cp_v + (size_t)t * (nd + 1) * 32, xs);
    sc sv;
    sc_add(sv, ss, rb);
    sc_to_be(w.cp_sv + 32 * t, sv);                                       // s_v = s + r_blind      (reciprocal.rs:133)
    if (!ok) status |= ST_BAD_ENCODING;
    w.status[t] = status;
}
// proof.r to affine; the circuit's public input V + proof.r (reciprocal.rs:141)
HD void recip_prove_stage_r2(const RecipProveWs& w, size_t t) {
    pt R, S;
    ws_ld_pt(R, w.pbuf, w.N, t);
    apt V;
    if (!apt_from_xy64(V, w.commitments + 64 * t)) { fe_set_u32(V.x, 0); fe_set_u32(V.y, 0); }
    pt_madd(S, R, V, apt_is_identity(V));
    pt P[2] = {R, S};
    apt A[2];
    batch_to_affine<2>(A, P);
    apt_to_xy64(w.proof_r + 64 * t, A[0]);
    apt_to_xy64(w.cp_vpts + 64 * t, A[1]);
}

// Host side: the reciprocal circuit's sparsity pattern for (dim_nd, dim_np) as column-compressed data (make_circuit,
// reciprocal.rs:150-214), shared values filled in, per-instance entries mapped to value slots: 0 = -e, 1 + j = -1/(e + j).
struct RecipPattern {
    CircuitHostData hd;
    std::vector<int> inst_l, inst_m, parts;
    size_t dims[6];
};
inline void recip_pattern_build(RecipPattern& P, size_t nd, size_t np) {
    const size_t nm = nd, no = np, nv = nd + 1, nl = nv, nw = 2 * nd + np;
    P.dims[0] = nm; P.dims[1] = no; P.dims[2] = 1; P.dims[3] = nl; P.dims[4] = nv; P.dims[5] = nw;
    CircuitHostData& h = P.hd;
    auto push = [](std::vector<int>& rows, std::vector<u32>& vals, int row, const sc& v) {
        rows.push_back(row);
        vals.insert(vals.end(), v.v, v.v + 8);
    };
    sc one, minus_one, pw, base;
    sc_set_u32(one, 1);
    sc_neg(minus_one, one);
    sc_set_u32(base, (u32)np);
    h.cpl.assign(nw + 1, 0);
    h.cpm.assign(nw + 1, 0);
    pw = one;
    for (size_t col = 0; col < nw; col++) {
        h.cpl[col] = (int)h.rl.size();
        h.cpm[col] = (int)h.rm.size();
        if (col < nm) {                                   // W_l[0][i] = -(np^i)                       (reciprocal.rs:170)
            sc v;
            sc_neg(v, pw);
            push(h.rl, h.vl, 0, v);
            P.inst_l.push_back(-1);
            sc_mul(pw, pw, base);
        } else if (col < 2 * nm) {                        // W_l[i + 1][j + nm] = 1 (i != j); W_m[j][j + nm] = -e
            const size_t j = col - nm;
            for (size_t i = 0; i < nm; i++)
                if (i != j) { push(h.rl, h.vl, (int)(i + 1), one); P.inst_l.push_back(-1); }
            push(h.rm, h.vm, (int)j, minus_one);
            P.inst_m.push_back(0);
        } else {                                          // W_l[i + 1][j + 2 nm] = -1/(e + j)
            const size_t j = col - 2 * nm;
            for (size_t i = 0; i < nm; i++) { push(h.rl, h.vl, (int)(i + 1), minus_one); P.inst_l.push_back((int)(1 + j)); }
        }
    }
    h.cpl[nw] = (int)h.rl.size();
    h.cpm[nw] = (int)h.rm.size();
    h.al.assign(nl * 8, 0);                               // a_l = 0, a_m = 1                         (reciprocal.rs:155-158)
    h.am.assign(nm * 8, 0);
    for (size_t i = 0; i < nm; i++) h.am[i * 8] = 1;
    h.colmap.assign(3 * nm + 3 * nv, -1);
    for (size_t j = 0; j < nm; j++) { h.colmap[j] = (int)j; h.colmap[nm + j] = (int)(nm + j); }
    for (size_t j = 0; j < nv; j++) h.colmap[3 * nm + j] = j < no ? (int)(2 * nm + j) : -1;      // LL & index < dim_np -> Some(index) (:186-192)
    P.parts.assign(3 * nv + nm, -1);                      // LO | LL | LR | NO
    for (size_t j = 0; j < nv && j < no; j++) P.parts[nv + j] = (int)j;
    // never-empty arrays (a device pointer is taken of each)
    h.rl.push_back(0); h.rm.push_back(0); h.vl.resize(h.vl.size() + 8); h.vm.resize(h.vm.size() + 8);
    P.inst_l.push_back(-1); P.inst_m.push_back(-1);
}

}  // namespace bppp
