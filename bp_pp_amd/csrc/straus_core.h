// Variable-base sums: the GLV split, the per-proof affine window tables (1P..16P of the 13 proof points, four batched inversions) and
// the shared-doubling (Straus) sums over them -- one lane per proof, lane groups for under-filled calls, and the progress-paced wave
// priority of launches that fill the chip once.  Split out of verify_core.h in round 6.
#pragma once
#include "verify_ws.h"

namespace bppp {

// ---------------------------------------------------------------- variable-base shared-doubling MSM (Straus), signed 4-bit windows
// k = sum_{i<64} (nib_i(k') - 8) 16^i + c 16^64 with k' = k + 0x88..8 (mod 2^256), c = carry out; digits in [-8, 7].
struct straus_scalar { u32 kp[8]; u32 top; };
HD void straus_recode(straus_scalar& r, const sc& k) {
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (u64)k.v[i] + 0x88888888u; r.kp[i] = (u32)c; c >>= 32; }
    r.top = (u32)c;
}
// tbl[e] = e * P, e = 0..8 (P affine, may be the identity sentinel); entries are stored with canonical coordinates
HD void straus_build_table(pt_slot* tbl, const apt& P) {
    pt cur;
    pt_set_identity(cur);
    tbl[0].p = cur;
    pt_from_affine(cur, P);
    tbl[1].p = cur;
    const bool pid = apt_is_identity(P);
#pragma nounroll
    for (int e = 2; e <= 8; e++) {
        pt src = tbl[(e & 1) ? e - 1 : e / 2].p;
        pt d;
        if (e & 1) pt_madd(d, src, P, pid);     // loop counter: wave-uniform branch
        else pt_dbl(d, src);
        pt_normalize(d);
        tbl[e].p = d;
    }
}
// acc = sum_j k_j * P_j using tables tbl[j*9 + e]; scalars recoded in rs[0..m)
HD void straus_msm(pt& out, const pt_slot* tbl, const straus_scalar* rs, int m) {
    pt acc;
    pt_set_identity(acc);
    // top digit (0 or 1) for each scalar
#pragma nounroll
    for (int j = 0; j < m; j++) {
        pt q = tbl[j * BPPP_STRAUS_ENTRIES + (rs[j].top ? 1 : 0)].p;
        pt_add(acc, acc, q);
    }
#pragma nounroll
    for (int i = 63; i >= 0; i--) {
#pragma nounroll
        for (int d = 0; d < 4; d++) pt_dbl(acc, acc);
#pragma nounroll
        for (int j = 0; j < m; j++) {
            u32 limb = 0;
#pragma unroll
            for (int l = 0; l < 8; l++) limb = (l == (i >> 3)) ? rs[j].kp[l] : limb;
            int dg = (int)((limb >> ((i & 7) * 4)) & 15) - 8;
            int mag = dg < 0 ? -dg : dg;
            pt q = tbl[j * BPPP_STRAUS_ENTRIES + mag].p;
            fe ny;
            fe_neg_m<1>(ny, q.Y);
            fe_cmov(q.Y, dg < 0, ny);
            pt_add(acc, acc, q);
        }
    }
    out = acc;
}

// ---------------------------------------------------------------- GLV endomorphism split (secp256k1: lambda*(x, y) = (beta*x, y))
// k = k1 + k2*lambda (mod n) with |k1|, |k2| < 2^128: halves the doublings of every variable-base multiplication.
// Constants: lattice basis of (n, lambda); g1, g2 = round(2^384 * b2 / n), round(2^384 * (-b1) / n)  (derived and checked in
// tests/test_core_emul.py against big-integer arithmetic).
struct glv_split { u32 k1[5], k2[5]; bool neg1, neg2; };
HD void glv_round_shift384(sc& c, const sc& k, const u32 g[8]) {   // c = (k*g + 2^383) >> 384
    u32 t[16];
    mul256(t, k.v, g);
    u32 cy = (t[11] >> 31) & 1u;
#pragma unroll
    for (int i = 0; i < 4; i++) c.v[i] = addc(t[12 + i], 0u, cy);
#pragma unroll
    for (int i = 4; i < 8; i++) c.v[i] = 0;
}
HD bool glv_abs(u32 out[5], const sc& r) {   // r is either small (< 2^129) or n - small; returns true when negated
    bool neg = ((r.v[5] | r.v[6] | r.v[7]) != 0) | (r.v[4] > 1u);
    sc m;
    sc_neg(m, r);
#pragma unroll
    for (int i = 0; i < 5; i++) out[i] = neg ? m.v[i] : r.v[i];
    return neg;
}
HD void glv_decompose(glv_split& out, const sc& k) {
    const u32 G1[8] = {0x45DBB031u, 0xE893209Au, 0x71E8CA7Fu, 0x3DAA8A14u, 0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u};
    const u32 G2[8] = {0x8AC47F71u, 0x1571B4AEu, 0x9DF506C6u, 0x221208ACu, 0x0ABFE4C4u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u};
    const sc MB1 = {{0x0ABFE4C3u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u}};
    const sc MB2 = {{0x3DB1562Cu, 0xD765CDA8u, 0x0774346Du, 0x8A280AC5u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}};
    const sc LAM = {{0x1B23BD72u, 0xDF02967Cu, 0x20816678u, 0x122E22EAu, 0x8812645Au, 0xA5261C02u, 0xC05C30E0u, 0x5363AD4Cu}};
    sc c1, c2, r1, r2, t;
    glv_round_shift384(c1, k, G1);
    glv_round_shift384(c2, k, G2);
    sc_mul(c1, c1, MB1);
    sc_mul(c2, c2, MB2);
    sc_add(r2, c1, c2);
    sc_mul(t, r2, LAM);
    sc_sub(r1, k, t);
    out.neg1 = glv_abs(out.k1, r1);
    out.neg2 = glv_abs(out.k2, r2);
    // signed 4-bit recoding offset (33 nibbles): k' = |k| + 0x8...8; digit_i = nib_i(k') - 8 in [-8, 7]
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) out.k1[i] = addc(out.k1[i], i < 4 ? 0x88888888u : 0x8u, c);
    c = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) out.k2[i] = addc(out.k2[i], i < 4 ? 0x88888888u : 0x8u, c);
}
HD u32 limb5_at(const u32 v[5], int idx) {
    u32 r = 0;
#pragma unroll
    for (int l = 0; l < 5; l++) r = (l == idx) ? v[l] : r;
    return r;
}
// acc = sum_j k_j * P_j with tables tbl[j*9 + e] = e*P_j: 33 windows x (4 doublings + 2m additions); the lambda stream reuses
// P_j's table with X scaled by beta.
HD void straus_msm_glv(pt& out, const pt_slot* tbl, const glv_split* sp, int m) {
    const u32 BETA_W[8] = {0x719501EEu, 0xC1396C28u, 0x12F58995u, 0x9CF04975u, 0xAC3434E9u, 0x6E64479Eu, 0x657C0710u, 0x7AE96A2Bu};
    fe BETA;
    fe_from_w8(BETA, BETA_W);
    pt acc;
    pt_set_identity(acc);
#pragma nounroll
    for (int i = 32; i >= 0; i--) {
        if (i != 32) {
#pragma nounroll
            for (int d = 0; d < 4; d++) pt_dbl(acc, acc);
        }
#pragma nounroll
        for (int j = 0; j < m; j++) {
#pragma nounroll
            for (int h = 0; h < 2; h++) {
                const u32* kp = h ? sp[j].k2 : sp[j].k1;
                bool sneg = h ? sp[j].neg2 : sp[j].neg1;
                int dg = (int)((limb5_at(kp, i >> 3) >> ((i & 7) * 4)) & 15) - 8;
                int mag = dg < 0 ? -dg : dg;
                pt q = tbl[j * BPPP_STRAUS_ENTRIES + mag].p;
                fe bx, ny;
                fe_mul(bx, q.X, BETA);
                fe_cmov(q.X, h != 0, bx);
                fe_neg_m<1>(ny, q.Y);
                fe_cmov(q.Y, (dg < 0) != sneg, ny);
                pt_add(acc, acc, q);
            }
        }
    }
    out = acc;
}

// ---------------------------------------------------------------- the u64 verifier's variable-base path: affine per-proof tables
// All 13 variable-base points of a proof (c_l, c_r, c_o, c_s, r[4], x[4], V + r) are inputs, known before any challenge, so
// their window tables are built once, up front, and brought to AFFINE form with a single field inversion per proof
// (Montgomery's trick over the 91 non-trivial multiples).  The five shared-doubling sums that follow (C0 and the four WNLA
// rounds) then run on a Jacobian accumulator with mixed additions (point.h), and the GLV image tables (beta x, y) are stored
// too, so the inner loop has no beta multiplication.
#define BPPP_VPOINTS 13
#define BPPP_ATAB_PER_PROOF (BPPP_VPOINTS * 16)
// Signed 5-bit windows over the 128-bit GLV halves -- 26 windows x 2M mixed additions instead of 33 x 2M (a 4-bit recoding needs a 33rd
// window for its carry), tables 1P..16P per point (4 levels, 4 batched inversions) with the GLV image (beta x, y) formed on the fly by
// one multiplication (the table memory stays 13 x 16 entries).  (Rounds 2-5 carried a 4-bit variant behind a macro; it was never built
// again after round 2 and is gone.)
HD void ws_st_fe(u32* base, size_t N, size_t t, int slot, const fe& a) {
#pragma unroll
    for (int i = 0; i < 10; i++) base[(size_t)(slot * 10 + i) * N + t] = a.v[i];
}
HD void ws_ld_fe(fe& a, const u32* base, size_t N, size_t t, int slot, int mag) {
#pragma unroll
    for (int i = 0; i < 10; i++) a.v[i] = base[(size_t)(slot * 10 + i) * N + t];
    FE_SETMAG(a, mag);
    (void)mag;
}
// out[t] = 1 / in[t] (0 for 0, as fe_inv) for the G elements t = i, i + L, i + 2L, ... (L = ceil(N / G)) that lane i takes: their
// product is inverted once and unwound (3 multiplications per element), so a batch of N pays N / G inversions instead of N.  in and
// out are [10][N] limb arrays and may be the same one (a lane reads its G elements before it writes any; lanes share none).
template <int G>
HD void fe_batch_inv_lane(const u32* in, u32* out, size_t N, size_t i) {
    const size_t L = (N + G - 1) / G;
    fe z[G], pre[G], run, inv, one;
    bool zero[G];
    fe_set_u32(one, 1);
    run = one;
#pragma unroll
    for (int j = 0; j < G; j++) {
        const size_t t = i + (size_t)j * L;
        z[j] = one;
        zero[j] = true;
        if (t < N) {
            ws_ld_fe(z[j], in, N, t, 0, 2);
            zero[j] = fe_is_zero(z[j]);
            if (zero[j]) z[j] = one;
        }
        pre[j] = run;
        fe_mul(run, run, z[j]);
    }
    fe_inv(inv, run);
#pragma unroll
    for (int j = G - 1; j >= 0; j--) {
        const size_t t = i + (size_t)j * L;
        fe o;
        fe_mul(o, inv, pre[j]);
        fe_mul(inv, inv, z[j]);
        if (zero[j]) fe_set_u32(o, 0);
        if (t < N) ws_st_fe(out, N, t, 0, o);
    }
}
HD void glv_beta(fe& b) {
    const u32 BETA_W[8] = {0x719501EEu, 0xC1396C28u, 0x12F58995u, 0x9CF04975u, 0xAC3434E9u, 0x6E64479Eu, 0x657C0710u, 0x7AE96A2Bu};
    fe_from_w8(b, BETA_W);
}
// Layout of the per-proof window tables in HBM: entry-major, atab[i * N + t] for entry i of proof t (i = point * 16 + multiple - 1): the
// build kernel's stores coalesce across the wavefront, the sums' gathers are 64-byte records either way (measured on 2^20 proofs against
// the proof-major layout atab[t * 208 + i]: k_verify_tables 9.85 -> 8.82 ms, the five sums unchanged)
struct atab_ref {
    apt_packed* p;
    size_t s;
    HD apt_packed& operator[](int i) const { return p[(size_t)i * s]; }
    HD atab_ref operator+(int k) const { atab_ref r = {p + (size_t)k * s, s}; return r; }
};
HD atab_ref atab_of(apt_packed* atab, size_t N, size_t t, int entries_per_instance = 13 * 16) {
    (void)entries_per_instance;
    atab_ref r = {atab + t, N};
    return r;
}
HD void atab_store(atab_ref tb, int e, const apt& a, const fe& beta, bool identity) {   // e = 1..8 (1..16 with 5-bit windows)
    apt_packed k;
    fe_to_w8(k.x, a.x);
    fe_to_w8(k.y, a.y);
    if (identity) {
#pragma unroll
        for (int i = 0; i < 8; i++) k.x[i] = k.y[i] = 0;
    }
    tb[e - 1] = k;
    (void)beta;
}
// Window tables by AFFINE arithmetic, three batched inversions per proof.  The multiples of one point form three levels whose
// slopes only need earlier levels:   2P = 2.P  |  3P = 2P + P, 4P = 2.2P  |  5P = 4P + P, 6P = 2.3P, 7P = 4P + 3P, 8P = 2.4P,
// so all 13 points' level-l slope denominators (13, 26, 52 of them) are inverted together with Montgomery's trick.  An affine
// step costs 1M (prefix) + 2M (unwinding) + 1M + 2S (slope, x, y) against 12M for a complete projective step plus 6M of
// normalisation afterwards, and the only scratch is the running products (91 field elements per proof instead of 364).  The
// unwinding of level l (which produces that level's points) is fused with the forward pass of level l + 1 on the same point,
// so the passes alternate direction over the 13 points: A up, B down, C up, D down.
// No exceptional cases arise: the group has prime order n > 8, so for a point P != O none of P .. 8P is O, 2y != 0, and the
// additions jP + P (j = 2, 4) and 4P + 3P never meet equal x.  P = O (the (0, 0) sentinel, also what a malformed proof's
// points are replaced by) gives zero denominators: they are replaced by 1 and every multiple is stored as O.
#define BPPP_TSCR_FE (5 * BPPP_VPOINTS)   // running products per proof (BPPP_TSCR_PER_POINT below): 65 field elements, was 182 with one per denominator
struct aff_src { fe x, y; };
HD void aff_ld(aff_src& r, atab_ref tb, int e) {   // multiple e (1..8) of the point whose table is tb
    const apt_packed k = tb[e - 1];
    fe_from_w8(r.x, k.x);
    fe_from_w8(r.y, k.y);
}
HD void aff_den_dbl(fe& d, const aff_src& a, bool pid, const fe& one) { fe_add(d, a.y, a.y); fe_cmov(d, pid, one); }
HD void aff_den_add(fe& d, const aff_src& a, const aff_src& b, bool pid, const fe& one) { fe_sub_m<1>(d, a.x, b.x); fe_cmov(d, pid, one); }   // a + b
// 2a given 1 / (2 y_a)
HD void aff_dbl(apt& r, const aff_src& a, const fe& dinv) {
    fe num, lam, t;
    fe_sqr(num, a.x);
    fe_mul_small(num, num, 3);
    fe_mul(lam, num, dinv);
    fe_sqr(r.x, lam);
    fe_add(t, a.x, a.x);
    fe_sub_m<2>(r.x, r.x, t);            // magnitude 4
    fe_sub_m<4>(t, a.x, r.x);            // 6
    fe_mul(t, lam, t);
    fe_sub_m<1>(r.y, t, a.y);            // 3
}
// a + b given 1 / (x_a - x_b)
HD void aff_add(apt& r, const aff_src& a, const aff_src& b, const fe& dinv) {
    fe num, lam, t;
    fe_sub_m<1>(num, a.y, b.y);
    fe_mul(lam, num, dinv);
    fe_sqr(r.x, lam);
    fe_sub_m<1>(r.x, r.x, a.x);          // 3
    fe_sub_m<1>(r.x, r.x, b.x);          // 5
    fe_sub_m<5>(t, a.x, r.x);            // 7
    fe_mul(t, lam, t);
    fe_sub_m<1>(r.y, t, a.y);            // 3
}
HD void aff_take(aff_src& r, const apt& a) {   // a freshly computed point as the operand of the next level (magnitudes -> 1)
    fe_mul_small(r.x, a.x, 1);
    fe_mul_small(r.y, a.y, 1);
}
// one Montgomery-trick step forward: store the running product, multiply the denominator in
HD void aff_push(u32* tscr, size_t N, size_t t, int slot, fe& run, const fe& den) {
    ws_st_fe(tscr, N, t, slot, run);
    fe_mul(run, run, den);
}
// ... and backward: dinv = 1 / den, inv loses den
HD void aff_pop(fe& dinv, const u32* tscr, size_t N, size_t t, int slot, fe& inv, const fe& den) {
    fe pre;
    ws_ld_fe(pre, tscr, N, t, slot, 1);
    fe_mul(dinv, inv, pre);
    fe_mul(inv, inv, den);
}
// ---- blocks of four denominators (levels 3 and 4): ONE running product per block instead of one per denominator.  The running
// products are the table builder's scratch traffic (written in one pass, read back in the next, through HBM: at one lane per proof
// nothing of that size stays on chip), so a block costs a quarter of the stores and loads for three more multiplications when it is
// unwound (the block product is re-formed from the denominators, which the unwinding pass has in registers anyway).
HD void aff_push_block(u32* tscr, size_t N, size_t t, int slot, fe& run, const fe& block_product) {
    ws_st_fe(tscr, N, t, slot, run);
    fe_mul(run, run, block_product);
}
// di[k] = 1 / d[k] for the block pushed at `slot`; inv (the inverse of everything not yet unwound) loses the block
HD void aff_pop_block(fe di[4], const fe d[4], const u32* tscr, size_t N, size_t t, int slot, fe& inv) {
    fe t01, t23, B, pre, q;
    fe_mul(t01, d[0], d[1]);
    fe_mul(t23, d[2], d[3]);
    fe_mul(B, t01, t23);
    ws_ld_fe(pre, tscr, N, t, slot, 1);
    fe_mul(q, inv, pre);                 // 1 / (d0 d1 d2 d3)
    fe_mul(inv, inv, B);
    fe_mul(B, q, t23);                   // 1 / (d0 d1)
    fe_mul(q, q, t01);                   // 1 / (d2 d3)
    fe_mul(di[0], B, d[1]);
    fe_mul(di[1], B, d[0]);
    fe_mul(di[2], q, d[3]);
    fe_mul(di[3], q, d[2]);
}
// Window tables of NP points per instance (the u64 verifier's 13 proof points; the generic WNLA verifier's 2 x rounds round points):
// pts = the points in packed affine words [NP * 16][N], tscr = BPPP_TSCR_PER_POINT NP running products [.. * 10][N], tab = the
// instance's table view.
#define BPPP_TSCR_PER_POINT 5    // level 1: 1 (slots 0 .. NP, re-used by level 3: 1 block) | level 2: 2 | level 4: 2 blocks
// The build in five passes with an inversion of `run` between them: the state that crosses a boundary is `run` going in and its inverse
// coming out (tables, points and running products live in the workspace), so the passes are also kernels of their own with the
// inversions shared between proofs (k_verify_tables_pass, fe_batch_inv_lane).
HD void affine_tables_pass_a(const atab_ref tab, u32* tscr, const u32* pts, size_t N, size_t t, const int NP, fe& run) {
    const int L2 = NP, L4 = 3 * NP;          // running products: levels 1 and 3 share slots 0 .. NP, level 2 lives in NP .. 3 NP, level 4 in 3 NP .. 5 NP
    (void)L2; (void)L4;
    fe beta, one, d, dinv;
    fe_set_u32(one, 1);
    glv_beta(beta);
    (void)d; (void)dinv;
    // ---- pass A (up): entry 1 of every table; level-1 denominators 2 y_P
    fe_set_u32(run, 1);
#pragma nounroll
    for (int p = 0; p < NP; p++) {
        apt P;
        ws_ld_apt(P, pts, N, t, p);
        const bool pid = apt_is_identity(P);
        atab_store(tab + p * 16, 1, P, beta, pid);
        aff_src a = {P.x, P.y};
        aff_den_dbl(d, a, pid, one);
        aff_push(tscr, N, t, p, run, d);
    }
}
HD void affine_tables_pass_b(const atab_ref tab, u32* tscr, const u32* pts, size_t N, size_t t, const int NP, fe inv, fe& run) {
    const int L2 = NP, L4 = 3 * NP;          // running products: levels 1 and 3 share slots 0 .. NP, level 2 lives in NP .. 3 NP, level 4 in 3 NP .. 5 NP
    (void)L2; (void)L4;
    fe beta, one, d, dinv;
    fe_set_u32(one, 1);
    glv_beta(beta);
    (void)d; (void)dinv;
    // ---- pass B (down): 2P; level-2 denominators x_2P - x_P (3P = 2P + P), 2 y_2P (4P)
    fe_set_u32(run, 1);
#pragma nounroll
    for (int p = NP - 1; p >= 0; p--) {
        apt P;
        ws_ld_apt(P, pts, N, t, p);
        const bool pid = apt_is_identity(P);
        aff_src a = {P.x, P.y};
        aff_den_dbl(d, a, pid, one);
        aff_pop(dinv, tscr, N, t, p, inv, d);
        apt P2;
        aff_dbl(P2, a, dinv);
        atab_store(tab + p * 16, 2, P2, beta, pid);
        aff_src a2;
        aff_take(a2, P2);
        const int q = L2 + (NP - 1 - p) * 2;
        aff_den_add(d, a2, a, pid, one);
        aff_push(tscr, N, t, q, run, d);
        aff_den_dbl(d, a2, pid, one);
        aff_push(tscr, N, t, q + 1, run, d);
    }
}
HD void affine_tables_pass_c(const atab_ref tab, u32* tscr, const u32* pts, size_t N, size_t t, const int NP, fe inv, fe& run) {
    const int L2 = NP, L4 = 3 * NP;          // running products: levels 1 and 3 share slots 0 .. NP, level 2 lives in NP .. 3 NP, level 4 in 3 NP .. 5 NP
    (void)L2; (void)L4;
    fe beta, one, d, dinv;
    fe_set_u32(one, 1);
    glv_beta(beta);
    (void)d; (void)dinv;
    // ---- pass C (up): 4P, 3P; level-3 denominators x_4P - x_P (5P), 2 y_3P (6P), x_4P - x_3P (7P), 2 y_4P (8P): one block per point
    fe_set_u32(run, 1);
#pragma nounroll
    for (int p = 0; p < NP; p++) {
        const atab_ref tb = tab + p * 16;
        aff_src a, a2;
        aff_ld(a, tb, 1);
        aff_ld(a2, tb, 2);
        const bool pid = fe_is_zero(a.x) & fe_is_zero(a.y);
        const int q = L2 + (NP - 1 - p) * 2;
        apt P3, P4;
        aff_den_dbl(d, a2, pid, one);
        aff_pop(dinv, tscr, N, t, q + 1, inv, d);
        aff_dbl(P4, a2, dinv);
        aff_den_add(d, a2, a, pid, one);
        aff_pop(dinv, tscr, N, t, q, inv, d);
        aff_add(P3, a2, a, dinv);
        atab_store(tb, 3, P3, beta, pid);
        atab_store(tb, 4, P4, beta, pid);
        aff_src a3, a4;
        aff_take(a3, P3);
        aff_take(a4, P4);
        fe bp;
        aff_den_add(bp, a4, a, pid, one);
        aff_den_dbl(d, a3, pid, one);
        fe_mul(bp, bp, d);
        aff_den_add(d, a4, a3, pid, one);
        fe_mul(bp, bp, d);
        aff_den_dbl(d, a4, pid, one);
        fe_mul(bp, bp, d);
        aff_push_block(tscr, N, t, p, run, bp);
    }
}
HD void affine_tables_pass_d(const atab_ref tab, u32* tscr, const u32* pts, size_t N, size_t t, const int NP, fe inv, fe& run) {
    const int L2 = NP, L4 = 3 * NP;          // running products: levels 1 and 3 share slots 0 .. NP, level 2 lives in NP .. 3 NP, level 4 in 3 NP .. 5 NP
    (void)L2; (void)L4;
    fe beta, one, d, dinv;
    fe_set_u32(one, 1);
    glv_beta(beta);
    (void)d; (void)dinv;
    // ---- pass D (down): 5P, 6P, 7P, 8P  [5-bit windows: + level-4 denominators for 9P .. 16P, two blocks per point]
    fe_set_u32(run, 1);
#pragma nounroll
    for (int p = NP - 1; p >= 0; p--) {
        const atab_ref tb = tab + p * 16;
        aff_src a, a3, a4;
        aff_ld(a, tb, 1);
        aff_ld(a3, tb, 3);
        aff_ld(a4, tb, 4);
        const bool pid = fe_is_zero(a.x) & fe_is_zero(a.y);
        fe dd[4], di[4];
        aff_den_add(dd[0], a4, a, pid, one);      // 5P = 4P + P
        aff_den_dbl(dd[1], a3, pid, one);         // 6P = 2 . 3P
        aff_den_add(dd[2], a4, a3, pid, one);     // 7P = 4P + 3P
        aff_den_dbl(dd[3], a4, pid, one);         // 8P = 2 . 4P
        aff_pop_block(di, dd, tscr, N, t, p, inv);
        apt R;
        // level 4: 8P against P, 3P, 5P, 7P (9P, 11P, 13P, 15P: block "odd") and the doublings of 5P .. 8P (10P .. 16P: block "even")
        fe x8, bo, be;
        aff_src ax;
        aff_dbl(R, a4, di[3]);
        atab_store(tb, 8, R, beta, pid);
        aff_take(ax, R);
        x8 = ax.x;
        aff_den_dbl(be, ax, pid, one);                                          // 16P = 2 . 8P
        fe_sub_m<1>(bo, x8, a.x);  fe_cmov(bo, pid, one);                       //  9P = 8P + P
        fe_sub_m<1>(d, x8, a3.x);  fe_cmov(d, pid, one);  fe_mul(bo, bo, d);    // 11P = 8P + 3P
        aff_add(R, a4, a3, di[2]);
        atab_store(tb, 7, R, beta, pid);
        aff_take(ax, R);
        fe_sub_m<1>(d, x8, ax.x);  fe_cmov(d, pid, one);  fe_mul(bo, bo, d);    // 15P = 8P + 7P
        aff_den_dbl(d, ax, pid, one);                     fe_mul(be, be, d);    // 14P = 2 . 7P
        aff_dbl(R, a3, di[1]);
        atab_store(tb, 6, R, beta, pid);
        aff_take(ax, R);
        aff_den_dbl(d, ax, pid, one);                     fe_mul(be, be, d);    // 12P = 2 . 6P
        aff_add(R, a4, a, di[0]);
        atab_store(tb, 5, R, beta, pid);
        aff_take(ax, R);
        fe_sub_m<1>(d, x8, ax.x);  fe_cmov(d, pid, one);  fe_mul(bo, bo, d);    // 13P = 8P + 5P
        aff_den_dbl(d, ax, pid, one);                     fe_mul(be, be, d);    // 10P = 2 . 5P
        aff_push_block(tscr, N, t, L4 + 2 * p, run, bo);
        aff_push_block(tscr, N, t, L4 + 2 * p + 1, run, be);
    }
}
HD void affine_tables_pass_e(const atab_ref tab, u32* tscr, const u32* pts, size_t N, size_t t, const int NP, fe inv) {
    const int L2 = NP, L4 = 3 * NP;          // running products: levels 1 and 3 share slots 0 .. NP, level 2 lives in NP .. 3 NP, level 4 in 3 NP .. 5 NP
    (void)L2; (void)L4;
    fe beta, one, d, dinv;
    fe_set_u32(one, 1);
    glv_beta(beta);
    (void)d; (void)dinv;
    // ---- pass E (up): 9P .. 16P; per point the even block (pushed last) unwinds first
#pragma nounroll
    for (int p = 0; p < NP; p++) {
        const atab_ref tb = tab + p * 16;
        aff_src a5, a6, a7, a8;
        aff_ld(a5, tb, 5);
        aff_ld(a8, tb, 8);
        const bool pid = fe_is_zero(a8.x) & fe_is_zero(a8.y);       // P = O <=> every stored multiple is the (0, 0) sentinel
        fe dd[4], di[4];
        apt R;
        aff_ld(a6, tb, 6);
        aff_ld(a7, tb, 7);
        aff_den_dbl(dd[0], a5, pid, one);
        aff_den_dbl(dd[1], a6, pid, one);
        aff_den_dbl(dd[2], a7, pid, one);
        aff_den_dbl(dd[3], a8, pid, one);
        aff_pop_block(di, dd, tscr, N, t, L4 + 2 * p + 1, inv);
        aff_dbl(R, a5, di[0]);  atab_store(tb, 10, R, beta, pid);
        aff_dbl(R, a6, di[1]);  atab_store(tb, 12, R, beta, pid);
        aff_dbl(R, a7, di[2]);  atab_store(tb, 14, R, beta, pid);
        aff_dbl(R, a8, di[3]);  atab_store(tb, 16, R, beta, pid);
        aff_src a, a3;                                              // a6 is done with: its registers serve P and 3P
        aff_ld(a, tb, 1);
        aff_ld(a3, tb, 3);
        aff_den_add(dd[0], a8, a, pid, one);
        aff_den_add(dd[1], a8, a3, pid, one);
        aff_den_add(dd[2], a8, a5, pid, one);
        aff_den_add(dd[3], a8, a7, pid, one);
        aff_pop_block(di, dd, tscr, N, t, L4 + 2 * p, inv);
        aff_add(R, a8, a, di[0]);   atab_store(tb, 9, R, beta, pid);
        aff_add(R, a8, a3, di[1]);  atab_store(tb, 11, R, beta, pid);
        aff_add(R, a8, a5, di[2]);  atab_store(tb, 13, R, beta, pid);
        aff_add(R, a8, a7, di[3]);  atab_store(tb, 15, R, beta, pid);
    }
}
HD void affine_tables_build(const atab_ref tab, u32* tscr, const u32* pts, size_t N, size_t t, const int NP) {
    fe run, inv;
    affine_tables_pass_a(tab, tscr, pts, N, t, NP, run);
    fe_inv(inv, run);
    affine_tables_pass_b(tab, tscr, pts, N, t, NP, inv, run);
    fe_inv(inv, run);
    affine_tables_pass_c(tab, tscr, pts, N, t, NP, inv, run);
    fe_inv(inv, run);
    affine_tables_pass_d(tab, tscr, pts, N, t, NP, inv, run);
    fe_inv(inv, run);
    affine_tables_pass_e(tab, tscr, pts, N, t, NP, inv);
}
// pass = 0 .. 4 of the u64 verifier's 13 tables with the inversions shared: the running product goes out through ws.zinv, its inverse
// (fe_batch_inv_lane, in place) comes back through it
template <int PASS>
HD void verify_tables_pass(const VerifyWs& ws, size_t t) {
    const atab_ref tab = atab_of(ws.atab, ws.N, t);
    fe run, inv;
    if constexpr (PASS > 0) ws_ld_fe(inv, ws.zinv, ws.N, t, 0, 1);
    if constexpr (PASS == 0) affine_tables_pass_a(tab, ws.tscr, ws.pts, ws.N, t, BPPP_VPOINTS, run);
    if constexpr (PASS == 1) affine_tables_pass_b(tab, ws.tscr, ws.pts, ws.N, t, BPPP_VPOINTS, inv, run);
    if constexpr (PASS == 2) affine_tables_pass_c(tab, ws.tscr, ws.pts, ws.N, t, BPPP_VPOINTS, inv, run);
    if constexpr (PASS == 3) affine_tables_pass_d(tab, ws.tscr, ws.pts, ws.N, t, BPPP_VPOINTS, inv, run);
    if constexpr (PASS == 4) affine_tables_pass_e(tab, ws.tscr, ws.pts, ws.N, t, BPPP_VPOINTS, inv);
    if constexpr (PASS < 4) ws_st_fe(ws.zinv, ws.N, t, 0, run);
}
HD void verify_tables(const VerifyWs& ws, size_t t) {
    BPPP_STAMP(t, 16);
    affine_tables_build(atab_of(ws.atab, ws.N, t), ws.tscr, ws.pts, ws.N, t, BPPP_VPOINTS);
    BPPP_STAMP(t, 19);
}
// The same tables straight from the caller's BYTES, for batches whose one-lane kernels are a lone wavefront per SIMD (2^15, 2^16 proofs):
// the kernel then runs on the helper stream BESIDE phase 1 instead of after it, and every SIMD has two wavefronts to interleave.  It
// decodes the 13 points exactly as verify_phase1_on does -- V + proof.r to affine, ALL points zero if any field of the proof is
// malformed -- into a private copy (rows 17..42 of the final-scalar buffer, which nothing touches before k_verify_final_scalars; phase 1
// parks its reciprocals in rows 0..15), so the tables are bit for bit those of the serial order.
HD void verify_tables_own(const VerifyWs& ws, size_t t) {
    const size_t N = ws.N;
    u32* tp = ws.fsc + (size_t)(17 * 8) * N;
    const uint8_t* pv = ws.commitments + 64 * t;
    const uint8_t* pp = ws.proofs + (size_t)BPPP_U64_PROOF_BYTES * t;
    apt V, Pr;
    bool ok = apt_from_xy64(V, pv);
#pragma nounroll
    for (int i = 0; i < 12; i++) {
        apt Q;
        ok &= apt_from_xy64(Q, pp + 64 * i);
        ws_st_apt(tp, N, t, i, Q);
    }
    ok &= apt_from_xy64(Pr, pp + 64 * 12);
    sc l0;
    ok &= sc_from_be(l0, pp + 832);
    ok &= sc_from_be(l0, pp + 864);
    ok &= sc_from_be(l0, pp + 896);
    if (!ok) {
        apt zero;
        fe_set_u32(zero.x, 0);
        fe_set_u32(zero.y, 0);
        V = zero;
        Pr = zero;
#pragma nounroll
        for (int i = 0; i < 12; i++) ws_st_apt(tp, N, t, i, zero);
    }
    apt Vr;
    {
        pt s;
        pt_from_affine(s, V);
        pt_madd(s, s, Pr, apt_is_identity(Pr));
        pt_to_affine(Vr, s);
    }
    ws_st_apt(tp, N, t, 12, Vr);
    affine_tables_build(atab_of(ws.atab, ws.N, t), ws.tscr, tp, N, t, BPPP_VPOINTS);
}
// The window table of ONE point by ONE lane -- for calls so small that the chip is empty and what counts is the length of the dependent
// chain a proof has to wait for (a lane per table instead of a lane per proof: k_verify_tables_split).  pre_doublings > 0 first replaces
// P by 2^pre_doublings P: the table of a LATER part of a 26-window stream (part j starts at window split_begin(parts, j): 65, or 35 / 70 / 100, doublings),
// so that a sum can walk the parts of every stream on separate lanes (straus_affine_split).  The multiples 2P .. 16P as a Jacobian chain (one doubling, 14
// mixed additions: kP + P is never exceptional for 2 <= k <= 15 in a group of prime order), one inversion for their 15 Z's.
// A 26-window stream in `parts` parts (2 or 4): part j covers windows split_begin(parts, j) .. split_begin(parts, j + 1) - 1 over the
// table of 2^(5 split_begin(parts, j)) P -- 13 + 13 windows (tables of P, 2^65 P) or 7 + 7 + 6 + 6 (P, 2^35 P, 2^70 P, 2^100 P).
#define BPPP_SPLIT_PARTS_MAX 4
HD int split_begin(int parts, int part) {   // 26 = BPPP_STRAUS_WINDOWS (defined below)
    if (parts == 1) return part == 0 ? 0 : 26;
    if (parts == 2) return part == 0 ? 0 : part == 1 ? 13 : 26;
    return part == 0 ? 0 : part == 1 ? 7 : part == 2 ? 14 : part == 3 ? 20 : 26;
}
HD void affine_table_one(atab_ref tb, const apt& Pin, int pre_doublings) {
    fe beta;
    glv_beta(beta);
    apt P = Pin;
    const bool pid = apt_is_identity(Pin);
    if (pre_doublings) {
        ptj a;
        bool e0 = true;
        ptj_init(a);
        ptj_madd(a, e0, P, false);
#pragma nounroll
        for (int d = 0; d < pre_doublings; d++) ptj_dbl(a);
        fe zi, zi2;
        fe_inv(zi, a.Z);                          // the identity's Z is 0 and stays 0: every entry is stored as the identity below
        fe_sqr(zi2, zi);
        fe_mul(P.x, a.X, zi2);
        fe_mul(zi2, zi2, zi);
        fe_mul(P.y, a.Y, zi2);
    }
    atab_store(tb, 1, P, beta, pid);
    ptj T;
    bool empty = true;
    ptj_init(T);
    ptj_madd(T, empty, P, false);
    ptj_dbl(T);
    fe jx[15], jy[15], jz[15], pre[15];
#pragma nounroll
    for (int k = 0; k < 15; k++) {                // entry k holds (k + 2) P
        if (k) ptj_madd(T, empty, P, false);
        jx[k] = T.X; jy[k] = T.Y; jz[k] = T.Z;
        if (k) fe_mul(pre[k], pre[k - 1], T.Z);
        else fe_mul_small(pre[0], T.Z, 1);
    }
    fe inv;
    fe_inv(inv, pre[14]);
#pragma nounroll
    for (int k = 14; k >= 0; k--) {
        fe zi, zi2;
        if (k) { fe_mul(zi, inv, pre[k - 1]); fe_mul(inv, inv, jz[k]); }
        else zi = inv;
        apt R;
        fe_sqr(zi2, zi);
        fe_mul(R.x, jx[k], zi2);
        fe_mul(zi2, zi2, zi);
        fe_mul(R.y, jy[k], zi2);
        atab_store(tb, k + 2, R, beta, pid);
    }
}
// Point p of proof t's window tables straight from the caller's bytes -- what verify_phase1 parks in ws.pts: the 12 proof points it
// decodes and circuit_commitment = V + proof.r (reciprocal.rs:104), all of them the identity when anything in the proof is malformed
// -- so that the table kernel of a small call can run beside phase 1 instead of after it.
HD void verify_table_source(apt& P, const VerifyWs& ws, size_t t, int p) {
    const uint8_t* pv = ws.commitments + 64 * t;
    const uint8_t* pp = ws.proofs + (size_t)BPPP_U64_PROOF_BYTES * t;
    apt V, Pr, Q;
    fe_set_u32(P.x, 0);
    fe_set_u32(P.y, 0);
    bool ok = apt_from_xy64(V, pv);
#pragma nounroll
    for (int i = 0; i < 12; i++) {
        ok &= apt_from_xy64(Q, pp + 64 * i);
        if (i == p) P = Q;
    }
    ok &= apt_from_xy64(Pr, pp + 64 * 12);
    sc k;
    ok &= sc_from_be(k, pp + 832);
    ok &= sc_from_be(k, pp + 864);
    ok &= sc_from_be(k, pp + 896);
    if (p == 12 && ok) {
        pt s;
        pt_from_affine(s, V);
        pt_madd(s, s, Pr, apt_is_identity(Pr));
        pt_to_affine(P, s);
    }
    if (!ok) { fe_set_u32(P.x, 0); fe_set_u32(P.y, 0); }
}
// lane (point p, part h) of proof t: table slot h BPPP_VPOINTS + p
HD void verify_table_one(const VerifyWs& ws, size_t t, int p, int h, int parts, bool from_bytes = false) {
    apt P;
    if (from_bytes) verify_table_source(P, ws, t, p);
    else ws_ld_apt(P, ws.pts, ws.N, t, p);
    affine_table_one(atab_of(ws.atab, ws.N, t) + (h * BPPP_VPOINTS + p) * 16, P, 5 * split_begin(parts, h));
}
// The 2M GLV half-scalars of an M-point sum, kept in registers, recoded for signed 5-bit windows:
//   w = |k| + OFF5,  OFF5 = sum_{i < 26} 16 * 32^i   (|k| < 2^128, so w < 2^130: 26 digits),  digit_i = ((w >> 5 i) & 31) - 16 in [-16, 15].
// glv_decompose hands over |k| + 0x8...8 (the 4-bit offset of the generic path); the difference of the two offsets is added here.
template <int M>
struct glv_words {
    u32 w[2 * M][5];
    bool neg[2 * M];
};
#define BPPP_STRAUS_WINDOWS 26
HD void glv_recode5(u32 out[5], const u32 k4[5]) {
    const u32 D[5] = {0x987FB988u, 0x7FB987FBu, 0xB987FB98u, 0x87FB987Fu, 0xFFFFFFF9u};   // OFF5 - OFF4 mod 2^160
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) out[i] = addc(k4[i], D[i], c);
}
template <int M>
HD void glv_words_set(glv_words<M>& g, int j, const glv_split& sp) {
    glv_recode5(g.w[2 * j], sp.k1);
    glv_recode5(g.w[2 * j + 1], sp.k2);
    g.neg[2 * j] = sp.neg1;
    g.neg[2 * j + 1] = sp.neg2;
}
// the 2M digits of window i, 5 bits each, packed into one 64-bit word (2M <= 10): the window index is uniform over the wavefront,
// so this is a handful of selects per stream, once per window instead of once per addition
template <int M>
HD u64 glv_window_digits(const glv_words<M>& g, int i) {
    const int b = 5 * i, l = b >> 5, sh = b & 31;
    u64 pk = 0;
#pragma unroll
    for (int st = 0; st < 2 * M; st++) {
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            lo = (q == l) ? g.w[st][q] : lo;
            hi = (q == l + 1) ? g.w[st][q] : hi;
        }
        const u32 v = (u32)(((((u64)hi) << 32) | lo) >> sh) & 31u;
        pk |= (u64)v << (5 * st);
    }
    return pk;
}
template <int M>
HD void glv_digit_of(const glv_words<M>& g, u64 pk, int r, int& mag, bool& neg) {
    bool sneg = false;
#pragma unroll
    for (int st = 0; st < 2 * M; st++) sneg = (st == r) ? g.neg[st] : sneg;
    const int dg = (int)((pk >> (5 * r)) & 31u) - 16;
    mag = dg < 0 ? -dg : dg;
    neg = (dg < 0) != sneg;
}
// Progress-paced wave priority (VerifyWs::pace).  The SIMD's instruction arbiter serves the OLDER of two wavefronts first, so when a launch
// fills the chip exactly once (2^17 proofs: two wavefronts per SIMD, all started together) one wavefront of each pair runs almost
// as if alone and its partner mostly waits, then finishes alone at a lone wavefront's poor issue rate: 41 % of the SIMD-time of
// k_verify_round at 2^17 proofs has ONE wavefront resident (profiles/r06/r06_a_wave_timeline.txt).  With pacing on, a wavefront lowers
// its own priority (s_setprio 3 .. 0) as it advances through the windows of its sum, in spans that halve towards the end: whichever
// of the pair is behind is served first, the two reach the end within a few windows of each other, and the lone tail shrinks to that.
// window: 25 (first) .. 0 (last).  Wave-uniform; a handful of scalar instructions per window.
HD void straus_pace(bool pace, int window) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (!pace) return;
    if (window >= 13) __builtin_amdgcn_s_setprio(3);
    else if (window >= 6) __builtin_amdgcn_s_setprio(2);
    else if (window >= 3) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#else
    (void)pace; (void)window;
#endif
}
// sum_j k_j P_j over the affine tables; pidx[j] = table (proof point slot) of P_j.  26 windows x (5 doublings + 2M mixed
// additions); stream 2j is k1 of P_j, stream 2j + 1 its GLV partner (the entry's x times beta: the stream index is uniform over
// the wavefront, so that multiplication is behind a real branch).  The table entry of the next addition is requested before the
// current one starts.  Returns false when an exceptional addition was met (re-do with straus_affine_complete).
template <int M>
HD bool straus_affine_fast(pt& out, atab_ref tab, const int* pidx, const glv_words<M>& g, bool pace = false) {
    const int total = BPPP_STRAUS_WINDOWS * 2 * M;
    fe beta;
    glv_beta(beta);
    straus_pace(pace, BPPP_STRAUS_WINDOWS - 1);
    ptj acc;
    ptj_init(acc);
    bool empty = true;
    apt_packed cur_e, nxt_e;
    int cur_mag, nxt_mag;
    bool cur_neg, nxt_neg;
    u64 pk_cur = glv_window_digits<M>(g, BPPP_STRAUS_WINDOWS - 1), pk_nxt = glv_window_digits<M>(g, BPPP_STRAUS_WINDOWS - 2);
    glv_digit_of<M>(g, pk_cur, 0, cur_mag, cur_neg);
    cur_e = tab[pidx[0] * 16 + (cur_mag ? cur_mag - 1 : 0)];
    int r = 0, i = BPPP_STRAUS_WINDOWS - 1;
#pragma nounroll
    for (int s = 0; s < total; s++) {
        // successor step (clamped at the end: requested, never consumed)
        int rn = r + 1, in = i;
        if (rn == 2 * M) { rn = 0; in = i - 1; }
        if (in < 0) { rn = r; in = i; }
        glv_digit_of<M>(g, in == i ? pk_cur : pk_nxt, rn, nxt_mag, nxt_neg);
        int pn = 0;
#pragma unroll
        for (int j = 0; j < M; j++) pn = (j == (rn >> 1)) ? pidx[j] : pn;
        nxt_e = tab[pn * 16 + (nxt_mag ? nxt_mag - 1 : 0)];
        if (r == 0 && s != 0) {
            straus_pace(pace, i);
#pragma nounroll
            for (int d = 0; d < 5; d++) ptj_dbl(acc);
        }
        apt e;
        bool id;
        apt_unpack(e, id, cur_e);
        if (r & 1) fe_mul(e.x, e.x, beta);            // wave-uniform: the GLV image (beta x, y)
        fe ny;
        fe_neg_m<1>(ny, e.y);
        fe_cmov(e.y, cur_neg, ny);
        ptj_madd(acc, empty, e, (cur_mag == 0) | id);
        cur_e = nxt_e;
        cur_mag = nxt_mag;
        cur_neg = nxt_neg;
        if (in != i) { pk_cur = pk_nxt; pk_nxt = glv_window_digits<M>(g, in > 0 ? in - 1 : 0); }
        r = rn;
        i = in;
    }
    const bool exceptional = !empty && fe_is_zero(acc.Z);
    ptj_to_pt(out, acc, empty);
    return !exceptional;
}
template <int M>
HD void straus_affine_complete(pt& out, atab_ref tab, const int* pidx, const glv_words<M>& g) {
    fe beta;
    glv_beta(beta);
    pt acc;
    pt_set_identity(acc);
#pragma nounroll
    for (int i = BPPP_STRAUS_WINDOWS - 1; i >= 0; i--) {
        if (i != BPPP_STRAUS_WINDOWS - 1) {
#pragma nounroll
            for (int d = 0; d < 5; d++) pt_dbl(acc, acc);
        }
        const u64 pk = glv_window_digits<M>(g, i);
#pragma nounroll
        for (int r = 0; r < 2 * M; r++) {
            int mag, pn = 0;
            bool neg, id;
            glv_digit_of<M>(g, pk, r, mag, neg);
#pragma unroll
            for (int j = 0; j < M; j++) pn = (j == (r >> 1)) ? pidx[j] : pn;
            apt e;
            apt_unpack(e, id, tab[pn * 16 + (mag ? mag - 1 : 0)]);
            if (r & 1) fe_mul(e.x, e.x, beta);
            fe ny;
            fe_neg_m<1>(ny, e.y);
            fe_cmov(e.y, neg, ny);
            pt_madd(acc, acc, e, (mag == 0) | id);
        }
    }
    out = acc;
}
template <int M>
HD void straus_affine(pt& out, atab_ref tab, const int* pidx, const glv_words<M>& g, bool pace = false) {
    if (!straus_affine_fast<M>(out, tab, pidx, g, pace)) {
        // the out-of-line call takes addresses: hand it copies, so the hot loop's scalars and accumulator stay in registers
        glv_words<M> gc = g;
        int pc[M];
#pragma unroll
        for (int j = 0; j < M; j++) pc[j] = pidx[j];
        pt o;
        straus_affine_complete<M>(o, tab, pc, gc);
        out = o;
    }
}

// lane q's share of the split sum (below): part h = q / 2M (windows split_begin(parts, h) .. split_begin(parts, h + 1) - 1) of stream
// r = q % 2M over the table of 2^(5 split_begin(parts, h)) P (slot pidx + BPPP_VPOINTS h: verify_table_one); q >= 2M parts: nothing.
// False on an exceptional addition.
// (stride: table slots between the sets of consecutive parts -- BPPP_VPOINTS for the u64 verifier's 13 points, 2 x rounds for the generic
// WNLA stage's round points)
template <int M>
HD bool straus_split_lane(pt& part, atab_ref tab, const int* pidx, const glv_words<M>& g, int q, int parts, int stride = BPPP_VPOINTS) {
    fe beta;
    glv_beta(beta);
    const bool have = q < 2 * M * parts;
    const int h = have ? q / (2 * M) : 0, r = have ? q - 2 * M * h : 0;
    u32 w[5];
    bool sneg = false;
    int pn = 0;
#pragma unroll
    for (int l = 0; l < 5; l++) w[l] = 0;
#pragma unroll
    for (int st = 0; st < 2 * M; st++) {
#pragma unroll
        for (int l = 0; l < 5; l++) w[l] = (st == r) ? g.w[st][l] : w[l];
        sneg = (st == r) ? g.neg[st] : sneg;
        pn = (st == r) ? pidx[st >> 1] : pn;
    }
    const bool img = (r & 1) != 0;
    const int base = (pn + stride * h) * 16, w0 = split_begin(parts, h), nw = split_begin(parts, h + 1) - w0;   // 13, or 7 / 6, windows
    auto digit = [&](int i, int& mag, bool& neg) {
        const int b = 5 * i, l = b >> 5, sh = b & 31;
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) { lo = (k == l) ? w[k] : lo; hi = (k == l + 1) ? w[k] : hi; }
        const int dg = (int)((u32)(((((u64)hi) << 32) | lo) >> sh) & 31u) - 16;
        mag = dg < 0 ? -dg : dg;
        neg = (dg < 0) != sneg;
    };
    ptj acc;
    ptj_init(acc);
    bool empty = true;
    int cur_mag, nxt_mag;
    bool cur_neg, nxt_neg;
    apt_packed cur_e, nxt_e;
    digit(w0 + nw - 1, cur_mag, cur_neg);
    cur_e = tab[base + (cur_mag ? cur_mag - 1 : 0)];
#pragma nounroll
    for (int i = nw - 1; i >= 0; i--) {
        digit(w0 + (i > 0 ? i - 1 : 0), nxt_mag, nxt_neg);       // the next window's entry is requested before this window's doublings
        nxt_e = tab[base + (nxt_mag ? nxt_mag - 1 : 0)];
        if (i != nw - 1) {
#pragma nounroll
            for (int d = 0; d < 5; d++) ptj_dbl(acc);
        }
        apt e;
        bool id;
        apt_unpack(e, id, cur_e);
        fe bx, ny;
        fe_mul(bx, e.x, beta);
        fe_cmov(e.x, img, bx);
        fe_neg_m<1>(ny, e.y);
        fe_cmov(e.y, cur_neg, ny);
        ptj_madd(acc, empty, e, (cur_mag == 0) | id | !have);
        cur_e = nxt_e;
        cur_mag = nxt_mag;
        cur_neg = nxt_neg;
    }
    const bool exceptional = !empty && fe_is_zero(acc.Z);
    ptj_to_pt(part, acc, empty);
    return !exceptional;
}
#if defined(__HIPCC__)
// The same M-point sum spread over a GROUP OF FOUR LANES: lane q takes the GLV streams q, q + 4, q + 8 (< 2M; stream r is point
// r >> 1, its image if r & 1), i.e. 26 windows x (5 doublings + 1 .. 3 mixed additions) per lane instead of 26 x (5 + 2M), then a
// two-step shuffle tree.  For batches so small that the chip is mostly empty (one lane per proof leaves SIMDs without a wavefront)
// this shortens the dependent chain a call has to wait for; the doublings are repeated on every lane, so it is not used once one lane
// per proof fills the SIMDs.  All four lanes of a group must be active and hold the same g / pidx; every lane ends with the total.
template <int M, int G = 4>
__device__ __forceinline__ void straus_affine_g4(pt& out, atab_ref tab, const int* pidx, const glv_words<M>& g, int q) {
    // (G = 2: groups of two lanes, streams q, q + 2, ... -- for batches that fill half of the wavefront slots with one lane per proof)
    constexpr int NS = (2 * M + G - 1) / G;      // streams per lane (the last one may be missing on the last lanes of the group)
    fe beta;
    glv_beta(beta);
    u32 w[NS][5];
    bool sneg[NS], img[NS], have[NS];
    int pn[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
        const int r = q + G * j;                 // my j-th stream
        have[j] = r < 2 * M;
        sneg[j] = false;
        pn[j] = 0;
#pragma unroll
        for (int l = 0; l < 5; l++) w[j][l] = 0;
#pragma unroll
        for (int st = 0; st < 2 * M; st++) {
#pragma unroll
            for (int l = 0; l < 5; l++) w[j][l] = (st == r) ? g.w[st][l] : w[j][l];
            sneg[j] = (st == r) ? g.neg[st] : sneg[j];
            pn[j] = (st == r) ? pidx[st >> 1] : pn[j];
        }
        img[j] = (r & 1) != 0;
    }
    auto digit = [&](const u32 (&ww)[5], bool sn, int i, int& mag, bool& neg) {
        const int b = 5 * i, l = b >> 5, sh = b & 31;
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) { lo = (k == l) ? ww[k] : lo; hi = (k == l + 1) ? ww[k] : hi; }
        const int dg = (int)((u32)(((((u64)hi) << 32) | lo) >> sh) & 31u) - 16;
        mag = dg < 0 ? -dg : dg;
        neg = (dg < 0) != sn;
    };
    ptj acc;
    ptj_init(acc);
    bool empty = true;
    int cur_mag[NS], nxt_mag[NS];
    bool cur_neg[NS], nxt_neg[NS];
    apt_packed cur_e[NS], nxt_e[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
        digit(w[j], sneg[j], BPPP_STRAUS_WINDOWS - 1, cur_mag[j], cur_neg[j]);
        cur_e[j] = tab[pn[j] * 16 + (cur_mag[j] ? cur_mag[j] - 1 : 0)];
    }
#pragma nounroll
    for (int i = BPPP_STRAUS_WINDOWS - 1; i >= 0; i--) {
#pragma unroll
        for (int j = 0; j < NS; j++) {           // the next window's entries are requested before this window's doublings
            digit(w[j], sneg[j], i > 0 ? i - 1 : 0, nxt_mag[j], nxt_neg[j]);
            nxt_e[j] = tab[pn[j] * 16 + (nxt_mag[j] ? nxt_mag[j] - 1 : 0)];
        }
        if (i != BPPP_STRAUS_WINDOWS - 1) {
#pragma nounroll
            for (int d = 0; d < 5; d++) ptj_dbl(acc);
        }
#pragma unroll
        for (int j = 0; j < NS; j++) {
            apt e;
            bool id;
            apt_unpack(e, id, cur_e[j]);
            fe bx, ny;
            fe_mul(bx, e.x, beta);
            fe_cmov(e.x, img[j], bx);
            fe_neg_m<1>(ny, e.y);
            fe_cmov(e.y, cur_neg[j], ny);
            ptj_madd(acc, empty, e, (cur_mag[j] == 0) | id | !have[j]);
            cur_e[j] = nxt_e[j];
            cur_mag[j] = nxt_mag[j];
            cur_neg[j] = nxt_neg[j];
        }
    }
    int bad = (!empty && fe_is_zero(acc.Z)) ? 1 : 0;
#pragma unroll
    for (int m = 1; m < G; m <<= 1) bad |= __shfl_xor(bad, m, 64);
    if (bad) {                      // an exceptional addition somewhere in the group: every lane re-does the whole sum completely
        straus_affine_complete<M>(out, tab, pidx, g);
        return;
    }
    pt part;
    ptj_to_pt(part, acc, empty);
    lane_group_sum<G>(part);
    out = part;
}
// The same sum with every stream cut in PARTS (2 or 4): lane q < 2M PARTS of a group of G walks part h = q / 2M of stream r = q % 2M
// over the table of 2^(5 split_begin(PARTS, h)) P (slot pidx + BPPP_VPOINTS h: verify_table_one) -- 12 x 5 doublings + 13 mixed additions
// (two parts) or at most 6 x 5 + 7 (four) per lane instead of 25 x 5 + 26 ... 78, then a log2(G)-step shuffle tree.  For calls that leave the chip empty (a handful of proofs):
// the length of the chain is all that counts there.  The other lanes of the group hold no stream and add the identity.  All G lanes of
// a group must be active and hold the same g / pidx; every lane ends with the total.
template <int M, int G, int PARTS>
__device__ __forceinline__ void straus_affine_split(pt& out, atab_ref tab, const int* pidx, const glv_words<M>& g, int q, int stride = BPPP_VPOINTS) {
    static_assert(2 * M * PARTS <= G, "a lane per part of a stream");
    pt part;
    int bad = straus_split_lane<M>(part, tab, pidx, g, q, PARTS, stride) ? 0 : 1;
#pragma unroll
    for (int m = 1; m < G; m <<= 1) bad |= __shfl_xor(bad, m, 64);
    if (bad) {                      // an exceptional addition somewhere in the group: every lane re-does the whole sum completely
        straus_affine_complete<M>(out, tab, pidx, g);
        return;
    }
    lane_group_sum<G>(part);
    out = part;
}
#endif


}  // namespace bppp
