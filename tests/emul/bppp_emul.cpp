// TEST-ONLY host emulation of the device code: compiles bp_pp_amd/csrc/*.h (the exact __host__ __device__ functions
// the HIP kernels call) with g++ and runs each "thread" in a CPU loop, so the kernel logic can be checked against the
// oracle in the CPU-only test tier.  This is a development aid: the product library (libbppp_hip.so) never loads or
// falls back to it, and fails loudly without a GPU.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../bp_pp_amd/csrc/plan_core.h"      // (before the device headers: field.h pulls <stdio.h> in inside its namespace in the host build)

#define BPPP_TRACE_TABLE_READS 1      // fb_core.h: FB_TRACE reports every fixed-base table entry a sum requests (emulator only)
#include "../../bp_pp_amd/csrc/prove_core.h"
#include "../../bp_pp_amd/csrc/circuit_core.h"
#include "../../bp_pp_amd/csrc/recip_core.h"
#include "../../bp_pp_amd/csrc/bucket_core.h"
#include "../../bp_pp_amd/csrc/rlc_core.h"
#include "../../bp_pp_amd/csrc/wnla_rlc_core.h"
#include "../../bp_pp_amd/csrc/wnla_prove_core.h"
#include "../../bp_pp_amd/csrc/circuit_prove_core.h"
#include "../../bp_pp_amd/csrc/recip_prove_core.h"

using namespace bppp;

// ---- record of the table entries the provers' SECRET sums request (tests/test_ct_trace.py).  The sequence must not depend on the
// secrets in the "ct_prover" forms; in the default forms it does (that is the side channel the mode closes).
static bool g_trace_on = false;               // inside a secret sum, with recording requested
static bool g_trace_want = false;
static std::vector<uint64_t> g_trace;
void bppp::bppp_trace_table_read(const void*, size_t index) {
    if (g_trace_on) g_trace.push_back((uint64_t)index);
}
// one prover sum over secret scalars: the "ct_prover" form or the default one, recorded when asked
static void secret_sum(pt& a, const FbTable& fb, const FbTable& fb_ct, int ct, size_t t, const u32* scal, const FbRanges& rg) {
    g_trace_on = g_trace_want;
    if (ct) fb_sum_serial_ct(a, fb_ct, t, scal, rg);
    else fb_sum_serial(a, fb, t, scal, rg);
    g_trace_on = false;
}

extern "C" {
void emul_trace_begin() { g_trace.clear(); g_trace_want = true; }
// -> number of table reads recorded since emul_trace_begin; the first min(count, cap) indices are copied out
size_t emul_trace_end(uint64_t* out, size_t cap) {
    g_trace_want = false;
    const size_t n = g_trace.size();
    for (size_t i = 0; i < n && i < cap; i++) out[i] = g_trace[i];
    return n;
}

void emul_fe_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
    fe x, y, r;
    fe_from_be(x, a); fe_from_be(y, b);
    fe_mul(r, x, y);
    fe_to_be(out, r);
}
void emul_fe_addsub(const uint8_t a[32], const uint8_t b[32], uint8_t sum[32], uint8_t diff[32], uint8_t m21[32]) {
    fe x, y, r;
    fe_from_be(x, a); fe_from_be(y, b);
    fe_add(r, x, y); fe_to_be(sum, r);
    fe_sub(r, x, y); fe_to_be(diff, r);
    fe_mul_small(r, x, 21); fe_to_be(m21, r);
}
void emul_fe_inv(const uint8_t a[32], uint8_t out[32], uint8_t sqrt_out[32]) {
    fe x, r;
    fe_from_be(x, a);
    fe_inv(r, x); fe_to_be(out, r);
    fe_sqrt_candidate(r, x); fe_to_be(sqrt_out, r);
}
int emul_fe_canonical(const uint8_t a[32]) { fe x; return fe_from_be(x, a) ? 1 : 0; }
void emul_sc_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
    sc x, y, r;
    sc_from_be(x, a); sc_from_be(y, b);
    sc_mul(r, x, y);
    sc_to_be(out, r);
}
void emul_sc_addsub(const uint8_t a[32], const uint8_t b[32], uint8_t sum[32], uint8_t diff[32]) {
    sc x, y, r;
    sc_from_be(x, a); sc_from_be(y, b);
    sc_add(r, x, y); sc_to_be(sum, r);
    sc_sub(r, x, y); sc_to_be(diff, r);
}
void emul_sc_inv(const uint8_t a[32], uint8_t out[32]) {
    sc x, r;
    sc_from_be(x, a);
    sc_inv(r, x);
    sc_to_be(out, r);
}
int emul_sc_canonical(const uint8_t a[32]) { sc x; return sc_from_be(x, a) ? 1 : 0; }
// op: 0 add (complete), 1 mixed add, 2 double(A)
int emul_pt_op(int op, const uint8_t A[64], const uint8_t B[64], uint8_t out[64]) {
    apt a, b, r;
    if (!apt_from_xy64(a, A) || !apt_from_xy64(b, B)) return -1;
    pt p, q, s;
    pt_from_affine(p, a);
    pt_from_affine(q, b);
    if (op == 0) pt_add(s, p, q);
    else if (op == 1) pt_madd(s, p, b, apt_is_identity(b));
    else pt_dbl(s, p);
    pt_to_affine(r, s);
    apt_to_xy64(out, r);
    return 0;
}
// sum_j k_j P_j via the Straus path (m <= 5)
int emul_straus(int m, const uint8_t* P, const uint8_t* k, uint8_t out[64]) {
    std::vector<pt_slot> tbl(m * BPPP_STRAUS_ENTRIES);
    straus_scalar rs[5];
    for (int j = 0; j < m; j++) {
        apt a;
        sc s;
        if (!apt_from_xy64(a, P + 64 * j) || !sc_from_be(s, k + 32 * j)) return -1;
        straus_build_table(tbl.data() + j * BPPP_STRAUS_ENTRIES, a);
        straus_recode(rs[j], s);
    }
    pt acc;
    straus_msm(acc, tbl.data(), rs, m);
    apt r;
    pt_to_affine(r, acc);
    apt_to_xy64(out, r);
    return 0;
}
// sum_j k_j P_j via the GLV Straus path (m <= 5); also returns the split of k_0 (|k1|, |k2| as 20-byte LE + signs)
int emul_straus_glv(int m, const uint8_t* P, const uint8_t* k, uint8_t out[64]) {
    std::vector<pt_slot> tbl(m * BPPP_STRAUS_ENTRIES);
    glv_split rs[5];
    for (int j = 0; j < m; j++) {
        apt a;
        sc s;
        if (!apt_from_xy64(a, P + 64 * j) || !sc_from_be(s, k + 32 * j)) return -1;
        straus_build_table(tbl.data() + j * BPPP_STRAUS_ENTRIES, a);
        glv_decompose(rs[j], s);
    }
    pt acc;
    straus_msm_glv(acc, tbl.data(), rs, m);
    apt r;
    pt_to_affine(r, acc);
    apt_to_xy64(out, r);
    return 0;
}
void emul_glv_split(const uint8_t k[32], uint32_t k1p[5], uint32_t k2p[5], int* neg1, int* neg2) {
    sc s;
    sc_from_be(s, k);
    glv_split sp;
    glv_decompose(sp, s);
    for (int i = 0; i < 5; i++) { k1p[i] = sp.k1[i]; k2p[i] = sp.k2[i]; }
    *neg1 = sp.neg1; *neg2 = sp.neg2;
}
void emul_merlin_kat(const uint8_t* label, size_t label_len, const uint8_t* m1, size_t m1_len, uint8_t* out, size_t out_len) {
    strobe t;
    t_new(t, label, (u32)label_len);
    t_append(t, "some label", m1, (u32)m1_len);
    t_challenge_bytes(t, "challenge", out, (u32)out_len);
}
size_t emul_fb_table_entries(int nbases, int W) { return (size_t)nbases * fb_per_base(W); }
int emul_fb_build(const uint8_t* gens, int nbases, int W, uint8_t* table_out /* entries x 64 B, LE limbs as device */) {
    std::vector<apt> g(nbases);
    for (int i = 0; i < nbases; i++) if (!apt_from_xy64(g[i], gens + 64 * i)) return -1;
    size_t entries = emul_fb_table_entries(nbases, W);
    // in passes of two bases, as the library does for tables too large to build at once
    const size_t per_base = fb_per_base(W), group = 2, gentries = group * per_base;
    (void)entries;
    std::vector<fe> tmp(gentries * 4);
    for (int b0 = 0; b0 < nbases; b0 += (int)group) {
        const int nb = nbases - b0 < (int)group ? nbases - b0 : (int)group;
        FbBuild fb{g.data(), nbases, W, (apt_packed*)table_out, tmp.data(), tmp.data() + gentries, tmp.data() + 2 * gentries,
                   tmp.data() + 3 * gentries, b0, nb};
        size_t nthreads = (size_t)nb * fb_nwin(W) * fb_chunks_per_window(W);
        for (size_t t = 0; t < nthreads; t++) fb_build_pass1(fb, t);
        for (size_t t = 0; t < nthreads; t++) fb_build_pass2(fb, t);
    }
    return 0;
}
// geometry of a window code (fb_core.h: fb_wb): out = {windows, first bit of window w, entries of window w, entries of a generator
// before window w, entries per generator, windows a scalar below 2^bits reaches}
void emul_fb_shape(int W, int w, int bits, uint64_t out[6]) {
    out[0] = (uint64_t)fb_nwin(W); out[1] = (uint64_t)fb_pos(W, w); out[2] = fb_per_win_at(W, w); out[3] = fb_win_off(W, w);
    out[4] = fb_per_base(W); out[5] = (uint64_t)fb_windows_for(bits, W);
}
// signed / unsigned window digit of scalar k (fb_digit): magnitude index, skip and negate flags
int emul_fb_digit(int W, const uint8_t k[32], int w, uint64_t* idx, int* skip, int* neg) {
    sc s;
    if (!sc_from_be(s, k)) return -1;
    size_t i;
    bool sk, ng;
    fb_digit(s.v, W, w, i, sk, ng);
    *idx = i; *skip = sk; *neg = ng;
    return fb_nwin(W);
}
// inversions: division-step form (product) and exponentiation form (cross-check); which = 0 field, 1 scalar
int emul_inv(int which, const uint8_t a[32], uint8_t out_divsteps[32], uint8_t out_fermat[32]) {
    if (which == 0) {
        fe x, r1, r2;
        if (!fe_from_be(x, a)) return -1;
        fe_inv(r1, x);
        fe_inv_fermat(r2, x);
        fe_to_be(out_divsteps, r1);
        fe_to_be(out_fermat, r2);
    } else {
        sc x, r1, r2;
        if (!sc_from_be(x, a)) return -1;
        sc_inv(r1, x);
        sc_inv_fermat(r2, x);
        sc_to_be(out_divsteps, r1);
        sc_to_be(out_fermat, r2);
    }
    return 0;
}
// fixed-base MSM sum_j k_j G_{first+j} through the table
int emul_fb_msm(const uint8_t* table, int W, int first_base, int count, const uint8_t* k, uint8_t out[64]) {
    VerifyWs ws;
    memset(&ws, 0, sizeof ws);
    ws.N = 1;
    ws.fb_table = (const apt_packed*)table;
    ws.fb_w = W;
    std::vector<u32> scal(count * 8);
    for (int j = 0; j < count; j++) {
        sc s;
        if (!sc_from_be(s, k + 32 * j)) return -1;
        for (int i = 0; i < 8; i++) scal[(j * 8 + i)] = s.v[i];
    }
    pt acc;
    pt_set_identity(acc);
    fixed_base_msm(acc, fb_of(ws), 0, scal.data(), 0, first_base, count);
    apt r;
    pt_to_affine(r, acc);
    apt_to_xy64(out, r);
    return 0;
}
// the 8-lane form the device kernels use: XYZZ fast partial sums, complete-formula re-do when a lane reports an exceptional
// addition.  *fell_back tells the test which path produced the result.
int emul_fb_msm_lanes_nl(const uint8_t* table, int W, int first_base, int count, const uint8_t* k, uint8_t out[64], int* fell_back, int nl);
int emul_fb_msm_lanes(const uint8_t* table, int W, int first_base, int count, const uint8_t* k, uint8_t out[64], int* fell_back) {
    return emul_fb_msm_lanes_nl(table, W, first_base, count, k, out, fell_back, BPPP_FB_LANES);
}
// nl lanes per sum: 1 (the shift-register walk of the large batches), 8, 64 (a wavefront per sum: the accumulator starts empty there,
// the other forms start from the offset point)
int emul_fb_msm_lanes_nl(const uint8_t* table, int W, int first_base, int count, const uint8_t* k, uint8_t out[64], int* fell_back, int nl) {
    FbTable fbt = {};
    fbt.table = (const apt_packed*)table;
    fbt.W = W;
    fbt.N = 1;
    std::vector<u32> scal(count * 8);
    for (int j = 0; j < count; j++) {
        sc s;
        if (!sc_from_be(s, k + 32 * j)) return -1;
        for (int i = 0; i < 8; i++) scal[(j * 8 + i)] = s.v[i];
    }
    FbRanges rg;
    fb_ranges_one(rg, 0, first_base, count);
    pt acc, part;
    pt_set_identity(acc);
    bool ok = true;
    for (int lane = 0; lane < nl; lane++) {
        ok &= fb_lane_sum_fast(part, fbt, 0, lane, scal.data(), rg, nl);
        pt_add(acc, acc, part);
    }
    *fell_back = !ok;
    pt viaserial;
    fb_sum_serial(viaserial, fbt, 0, scal.data(), rg, nl);      // fast + fallback, as the kernels do
    if (ok && !pt_eq(acc, viaserial)) return -2;
    acc = viaserial;
    apt r;
    pt_to_affine(r, acc);
    apt_to_xy64(out, r);
    return 0;
}
// a sum over a table in TWO regions (FbTable::table_hi: the first hi_bases generators at W_hi bits; table_lo: the rest at W_lo bits, counted
// from 0), nl lanes, fast form + complete re-do as the kernels run it; first_base / count may straddle the boundary
int emul_fb_msm_mixed(const uint8_t* table_hi, int W_hi, int hi_bases, const uint8_t* table_lo, int W_lo, int first_base, int count,
                      const uint8_t* k, uint8_t out[64], int nl) {
    FbTable fbt = {};
    fbt.table = (const apt_packed*)table_lo; fbt.W = W_lo; fbt.N = 1;
    fbt.table_hi = (const apt_packed*)table_hi; fbt.W_hi = W_hi; fbt.hi_bases = hi_bases;
    std::vector<u32> scal(count * 8);
    for (int j = 0; j < count; j++) {
        sc s;
        if (!sc_from_be(s, k + 32 * j)) return -1;
        for (int i = 0; i < 8; i++) scal[(j * 8 + i)] = s.v[i];
    }
    FbRanges rg;
    fb_ranges_one(rg, 0, first_base, count);
    pt fast, comp, part;
    fb_sum_serial(fast, fbt, 0, scal.data(), rg, nl);
    pt_set_identity(comp);
    for (int lane = 0; lane < nl; lane++) { fb_lane_sum_complete(part, fbt, 0, lane, scal.data(), rg, nl); pt_add(comp, comp, part); }
    if (!pt_eq(fast, comp)) return -2;
    pt one;             // and term by term with the complete law (commit_value's form)
    pt_set_identity(one);
    fixed_base_msm(one, fbt, 0, scal.data(), 0, first_base, count);
    if (!pt_eq(fast, one)) return -3;
    apt r;
    pt_to_affine(r, fast);
    apt_to_xy64(out, r);
    return 0;
}
// the u64 verifier's variable-base path in isolation: affine tables of up to 13 points (verify_tables), then the Jacobian
// shared-doubling sum over the first m of them (m <= 5) with its complete-formula fallback.  Points beyond m are identities.
int emul_straus_affine(int m, const uint8_t* P, const uint8_t* k, uint8_t out[64], int* fell_back) {
    VerifyWs ws;
    memset(&ws, 0, sizeof ws);
    ws.N = 1;
    std::vector<u32> pts(208, 0);
    std::vector<apt_packed> atab(BPPP_ATAB_PER_PROOF);
    std::vector<u32> tscr((size_t)BPPP_TSCR_FE * 10);
    ws.pts = pts.data(); ws.atab = atab.data(); ws.tscr = tscr.data();
    glv_words<5> g5;
    glv_words<2> g2;
    for (int j = 0; j < 5; j++) {
        sc s;
        sc_set_u32(s, 0);
        if (j < m) {
            apt a;
            if (!apt_from_xy64(a, P + 64 * j) || !sc_from_be(s, k + 32 * j)) return -1;
            ws_st_apt(ws.pts, 1, 0, j, a);
        }
        glv_split sp;
        glv_decompose(sp, s);
        glv_words_set<5>(g5, j, sp);
        if (j < 2) glv_words_set<2>(g2, j, sp);
    }
    verify_tables(ws, 0);
    const int pidx[5] = {0, 1, 2, 3, 4};
    pt acc, viafb;
    const atab_ref tabv = atab_of(atab.data(), 1, 0);
    bool ok = (m <= 2) ? straus_affine_fast<2>(acc, tabv, pidx, g2) : straus_affine_fast<5>(acc, tabv, pidx, g5);
    *fell_back = !ok;
    if (m <= 2) straus_affine_complete<2>(viafb, tabv, pidx, g2); else straus_affine_complete<5>(viafb, tabv, pidx, g5);
    if (ok && !pt_eq(acc, viafb)) return -2;      // both laws must agree whenever the fast one claims success
    apt r;
    pt_to_affine(r, viafb);
    apt_to_xy64(out, r);
    return 0;
}
// the small-call path: a lane per window table (verify_table_one: P and 2^65 P, or P, 2^35 P, 2^70 P, 2^100 P) and a lane per half / quarter of a GLV stream (straus_split_lane);
// the lanes of a group run here one after the other and their shares are added with the complete law, as the shuffle tree does.
// tables_out (optional): the parts x 13 x 16 table entries as 64-byte affine points (identity = zeros).
int emul_straus_split(int m, int parts, const uint8_t* P, const uint8_t* k, uint8_t out[64], int* fell_back, uint8_t* tables_out) {
    VerifyWs ws;
    memset(&ws, 0, sizeof ws);
    ws.N = 1;
    std::vector<u32> pts(208, 0);
    std::vector<apt_packed> atab(BPPP_SPLIT_PARTS_MAX * BPPP_ATAB_PER_PROOF);
    ws.pts = pts.data(); ws.atab = atab.data();
    glv_words<5> g5;
    glv_words<2> g2;
    for (int j = 0; j < 5; j++) {
        sc s;
        sc_set_u32(s, 0);
        if (j < m) {
            apt a;
            if (!apt_from_xy64(a, P + 64 * j) || !sc_from_be(s, k + 32 * j)) return -1;
            ws_st_apt(ws.pts, 1, 0, j, a);
        }
        glv_split sp;
        glv_decompose(sp, s);
        glv_words_set<5>(g5, j, sp);
        if (j < 2) glv_words_set<2>(g2, j, sp);
    }
    for (int h = 0; h < parts; h++)
        for (int p = 0; p < BPPP_VPOINTS; p++) verify_table_one(ws, 0, p, h, parts);
    if (tables_out)
        for (int i = 0; i < parts * BPPP_VPOINTS * 16; i++) {
            apt e;
            bool id;
            apt_unpack(e, id, atab[i]);
            if (id) memset(tables_out + 64 * i, 0, 64);
            else apt_to_xy64(tables_out + 64 * i, e);
        }
    const int pidx[5] = {0, 1, 2, 3, 4};
    const atab_ref tabv = atab_of(atab.data(), 1, 0);
    pt total, viafb;
    pt_set_identity(total);
    bool ok = true;
    const int lanes = (m <= 2 ? 4 : 16) * parts;
    for (int q = 0; q < lanes; q++) {
        pt part;
        ok &= (m <= 2) ? straus_split_lane<2>(part, tabv, pidx, g2, q, parts) : straus_split_lane<5>(part, tabv, pidx, g5, q, parts);
        pt_add(total, total, part);
    }
    *fell_back = !ok;
    if (m <= 2) straus_affine_complete<2>(viafb, tabv, pidx, g2); else straus_affine_complete<5>(viafb, tabv, pidx, g5);
    if (ok && !pt_eq(total, viafb)) return -2;
    apt r;
    pt_to_affine(r, viafb);
    apt_to_xy64(out, r);
    return 0;
}
// full exact verify pipeline, every phase in thread order
static int emul_u64_verify_impl(const uint8_t* table, int W, const uint8_t* label, size_t label_len, size_t n, const uint8_t* V,
                                const uint8_t* proofs, uint8_t* accept, int32_t* status, uint8_t* trace, const uint8_t* states,
                                size_t n_states, uint8_t* states_out);
int emul_u64_verify_batch(const uint8_t* table, int W, const uint8_t* label, size_t label_len, size_t n, const uint8_t* V,
                          const uint8_t* proofs, uint8_t* accept, int32_t* status, uint8_t* trace) {
    return emul_u64_verify_impl(table, W, label, label_len, n, V, proofs, accept, status, trace, nullptr, 0, nullptr);
}
// the same with pre-loaded transcripts (include/bppp.h: bppp_u64_verify_batch_transcript): n_states = 1 or n serialized states
int emul_u64_verify_batch_transcript(const uint8_t* table, int W, size_t n, const uint8_t* states, size_t n_states, const uint8_t* V,
                                     const uint8_t* proofs, uint8_t* accept, int32_t* status, uint8_t* states_out) {
    return emul_u64_verify_impl(table, W, nullptr, 0, n, V, proofs, accept, status, nullptr, states, n_states, states_out);
}
// the large batches' shared inversions (plan_core.h: shared_inv; straus_core.h: fe_batch_inv_lane): G proofs per inversion, lane order
static int g_shared_inv = 0;
void emul_set_shared_inv(int g) { g_shared_inv = g; }
extern "C++" {
template <int G>
static void emul_batch_inv_g(const u32* in, u32* out, size_t n) {
    for (size_t i = 0; i < (n + G - 1) / G; i++) fe_batch_inv_lane<G>(in, out, n, i);
}
}
static void emul_batch_inv(int G, const u32* in, u32* out, size_t n) {
    switch (G) {
    case 16: emul_batch_inv_g<16>(in, out, n); break;
    case 8: emul_batch_inv_g<8>(in, out, n); break;
    case 4: emul_batch_inv_g<4>(in, out, n); break;
    default: emul_batch_inv_g<2>(in, out, n); break;
    }
}
// fe_batch_inv_lane on n field elements (32-byte big-endian each, 0 allowed), G per inversion, out = in allowed: what
// k_verify_shared_inv<G> computes, lane after lane
int emul_fe_batch_inv(int G, size_t n, const uint8_t* in, uint8_t* out, int in_place) {
    std::vector<u32> a(10 * n), b(10 * n);
    for (size_t t = 0; t < n; t++) {
        fe x;
        if (!fe_from_be(x, in + 32 * t)) return -1;
        ws_st_fe(a.data(), n, t, 0, x);
    }
    if (G != 2 && G != 4 && G != 8 && G != 16) return -2;
    emul_batch_inv(G, a.data(), in_place ? a.data() : b.data(), n);
    for (size_t t = 0; t < n; t++) {
        fe x;
        ws_ld_fe(x, in_place ? a.data() : b.data(), n, t, 0, 1);
        fe_to_be(out + 32 * t, x);
    }
    return 0;
}
static int emul_u64_verify_impl(const uint8_t* table, int W, const uint8_t* label, size_t label_len, size_t n, const uint8_t* V,
                                const uint8_t* proofs, uint8_t* accept, int32_t* status, uint8_t* trace, const uint8_t* states,
                                size_t n_states, uint8_t* states_out) {
    VerifyWs ws;
    memset(&ws, 0, sizeof ws);
    ws.N = n;
    ws.states = states; ws.n_states = n_states; ws.states_out = states_out;
    ws.commitments = V; ws.proofs = proofs; ws.accept = accept; ws.status = status; ws.trace = trace;
    std::vector<u32> strobe_(52 * n), chal(80 * n), sc0(176 * n), cvec(200 * n), pts(208 * n), lns(24 * n), acc(30 * n), pfix(30 * n),
        fsc(392 * n);
    std::vector<pt_slot> straus(n * 5 * BPPP_STRAUS_ENTRIES);
    ws.tstate = strobe_.data(); ws.chal = chal.data(); ws.sc0 = sc0.data(); ws.cvec = cvec.data(); ws.pts = pts.data();
    ws.lns = lns.data(); ws.acc = acc.data(); ws.pfix = pfix.data(); ws.fsc = fsc.data(); ws.straus = straus.data();
    std::vector<apt_packed> atab(n * BPPP_ATAB_PER_PROOF);
    std::vector<u32> tscr((size_t)BPPP_TSCR_FE * 10 * n);
    ws.atab = atab.data(); ws.tscr = tscr.data();
    ws.fb_table = (const apt_packed*)table;
    ws.fb_w = W;
    t_new(ws.base, label, (u32)label_len);
    for (size_t t = 0; t < n; t++) verify_phase1(ws, t);
    // self-check of the small-call table kernel's own decode (verify_table_source): it must reproduce what phase 1 parked in ws.pts,
    // for well-formed and malformed proofs alike
    for (size_t t = 0; t < n; t++)
        for (int p = 0; p < BPPP_VPOINTS; p++) {
            apt a, b;
            verify_table_source(a, ws, t, p);
            ws_ld_apt(b, ws.pts, n, t, p);
            uint8_t ea[64], eb[64];
            apt_to_xy64(ea, a);
            apt_to_xy64(eb, b);
            if (memcmp(ea, eb, 64) != 0) return -77;
        }
    for (size_t t = 0; t < n; t++) verify_tables(ws, t);
    {   // self-check of the table kernel that runs BESIDE phase 1 at 2^15 / 2^16 proofs (verify_tables_own: its own decode of the caller's
        // bytes, a private copy of the points): bit for bit the tables of the serial order, for well-formed and malformed proofs alike
        std::vector<apt_packed> atab2(atab.size());
        std::vector<u32> tscr2(tscr.size());
        VerifyWs w2 = ws;
        w2.atab = atab2.data(); w2.tscr = tscr2.data();
        for (size_t t = 0; t < n; t++) verify_tables_own(w2, t);
        if (memcmp(atab.data(), atab2.data(), atab.size() * sizeof(apt_packed)) != 0) return -79;
    }
    const int G = g_shared_inv;
    std::vector<u32> zinv(10 * n);
    if (G) {   // the table build as five passes with the inversions between them shared by G proofs: bit for bit the same tables
        std::vector<apt_packed> atab3(atab.size());
        std::vector<u32> tscr3(tscr.size());
        VerifyWs w3 = ws;
        w3.atab = atab3.data(); w3.tscr = tscr3.data(); w3.zinv = zinv.data();
        for (size_t t = 0; t < n; t++) verify_tables_pass<0>(w3, t);
        emul_batch_inv(G, w3.zinv, w3.zinv, n);
        for (size_t t = 0; t < n; t++) verify_tables_pass<1>(w3, t);
        emul_batch_inv(G, w3.zinv, w3.zinv, n);
        for (size_t t = 0; t < n; t++) verify_tables_pass<2>(w3, t);
        emul_batch_inv(G, w3.zinv, w3.zinv, n);
        for (size_t t = 0; t < n; t++) verify_tables_pass<3>(w3, t);
        emul_batch_inv(G, w3.zinv, w3.zinv, n);
        for (size_t t = 0; t < n; t++) verify_tables_pass<4>(w3, t);
        if (memcmp(atab.data(), atab3.data(), atab.size() * sizeof(apt_packed)) != 0) return -80;
    }
    for (size_t t = 0; t < n; t++) verify_c0_var(ws, t);
    for (size_t t = 0; t < n; t++) verify_c0_fixed(ws, t);
    if (G) {   // ... and the rounds: C0 joined first, 1 / Z of C_{k-1} from the shared inversions, no head / tail split
        ws.zinv = zinv.data();
        for (size_t t = 0; t < n; t++) verify_c0_join(ws, t);
        for (int k = 1; k <= 4; k++) {
            emul_batch_inv(G, ws.acc + 20 * n, ws.zinv, n);
            for (size_t t = 0; t < n; t++) verify_round(ws, t, k);
        }
        for (size_t t = 0; t < n; t++) verify_final_scalars(ws, t);
        for (size_t t = 0; t < n; t++) {
            FbRanges rg;
            verify_final_check_ranges(rg);
            pt rhs;
            fb_sum_serial(rhs, fb_of(ws), t, ws.fsc, rg);
            verify_final_check_store(ws, t, rhs);
        }
        for (size_t t = 0; t < n; t++) verify_accept(ws, t);
        for (size_t t = 0; t < n; t++) verify_export_state(ws, t);
        return 0;
    }
    for (int k = 1; k <= 3; k++)
        for (size_t t = 0; t < n; t++) verify_round(ws, t, k);
    // the last round as the library runs it at 2^16 proofs: head, then -- beside the tail -- the final scalars and the final sum
    for (size_t t = 0; t < n; t++) verify_round(ws, t, 4, -1, 4, 1);
    for (size_t t = 0; t < n; t++) verify_final_scalars(ws, t);
    for (size_t t = 0; t < n; t++) {
        FbRanges rg;
        verify_final_check_ranges(rg);
        pt rhs;
        fb_sum_serial(rhs, fb_of(ws), t, ws.fsc, rg);
        verify_final_check_store(ws, t, rhs);
    }
    for (size_t t = 0; t < n; t++) verify_round(ws, t, 4, -1, 4, 2);
    for (size_t t = 0; t < n; t++) verify_accept(ws, t);
    for (size_t t = 0; t < n; t++) verify_export_state(ws, t);
    return 0;
}
// the final scalars in 2^lg parts per instance (k_wnla_final_scalars_grp): the parts computed one after the other and joined
static int g_final_scalars_lg = 0;
void emul_set_final_scalars_lg(int lg) { g_final_scalars_lg = lg; }
static void emul_final_scalars(const WnlaWs& w, size_t t) {
    const int lg = wnla_final_scalars_lg(w.rounds, g_final_scalars_lg);
    if (lg == 0) { wnla_verify_final_scalars(w, t); return; }
    for (int q = (1 << lg) - 1; q >= 0; q--) wnla_final_scalars_part(w, t, q, lg);      // any order: no part reads another's tables
    wnla_final_scalars_join(w, t, lg);
}
static int g_rlc_chunk = 8;
void emul_set_rlc_chunk(int c) { g_rlc_chunk = c; }
// the RLC mode's choice of group sizes from the previous call's reject rate (plan_core.h: plan_rlc)
void emul_plan_rlc(unsigned auto_super_m, int super_is_auto, int chunk_option, double rate, unsigned out[2]) {
    const bppp_host::RlcPlan p = bppp_host::plan_rlc(auto_super_m, super_is_auto != 0, chunk_option, rate);
    out[0] = p.super_m; out[1] = p.chunk;
}
// the random-linear-combination batch mode (rlc_core.h): exact pipeline through the final scalars, weighted commitments, one
// combined check per chunk of 8 proofs, exact re-check of the chunks that fail it.  Also returns the number of re-checked
// chunks and, for tests, each proof's weight halves and weighted commitment.
int emul_u64_verify_batch_rlc(const uint8_t* table, int W, const uint8_t* label, size_t label_len, size_t n, const uint8_t* V,
                              const uint8_t* proofs, const uint8_t seed[32], uint8_t* accept, int32_t* status, int* rechecked_chunks,
                              uint64_t* weights_ab /* n x 2 or null */, uint8_t* lhs_out /* n x 64 or null */) {
    VerifyWs ws;
    memset(&ws, 0, sizeof ws);
    ws.N = n;
    ws.commitments = V; ws.proofs = proofs; ws.accept = accept; ws.status = status; ws.trace = nullptr;
    std::vector<u32> strobe_(52 * n), chal(80 * n), sc0(176 * n), cvec(200 * n), pts(208 * n), lns(24 * n), acc(30 * n), pfix(30 * n),
        fsc(392 * n), lhs(30 * n), rsc(392 * n);
    std::vector<pt_slot> straus(n * 5 * BPPP_STRAUS_ENTRIES);
    std::vector<apt_packed> atab(n * BPPP_ATAB_PER_PROOF);
    std::vector<u32> tscr((size_t)BPPP_TSCR_FE * 10 * n);
    std::vector<uint8_t> flag(n / 8 + 1);
    ws.tstate = strobe_.data(); ws.chal = chal.data(); ws.sc0 = sc0.data(); ws.cvec = cvec.data(); ws.pts = pts.data();
    ws.lns = lns.data(); ws.acc = acc.data(); ws.pfix = pfix.data(); ws.fsc = fsc.data(); ws.straus = straus.data();
    ws.atab = atab.data(); ws.tscr = tscr.data();
    ws.fb_table = (const apt_packed*)table;
    ws.fb_w = W;
    t_new(ws.base, label, (u32)label_len);
    RlcWs r;
    memset(&r, 0, sizeof r);
    for (int i = 0; i < 4; i++) {
        u64 v = 0;
        for (int k = 0; k < 8; k++) v |= (u64)seed[8 * i + k] << (8 * k);
        r.seed[i] = v;
    }
    r.lhs = lhs.data(); r.sc = rsc.data(); r.flag = flag.data();
    for (size_t t = 0; t < n; t++) verify_phase1(ws, t);
    for (size_t t = 0; t < n; t++) verify_tables(ws, t);
    for (size_t t = 0; t < n; t++) verify_c0_var(ws, t);
    for (size_t t = 0; t < n; t++) verify_c0_fixed(ws, t);
    for (int k = 1; k <= 4; k++)
        for (size_t t = 0; t < n; t++) verify_round(ws, t, k);
    for (size_t t = 0; t < n; t++) verify_final_scalars(ws, t);
    for (size_t t = 0; t < n; t++) {
        rlc_lhs(ws, r, t);
        if (weights_ab) rlc_weight(*(u64*)&weights_ab[2 * t], *(u64*)&weights_ab[2 * t + 1], r, t);
        if (lhs_out) {
            pt L;
            ws_ld_pt(L, r.lhs, n, t);
            apt a;
            pt_to_affine(a, L);
            apt_to_xy64(lhs_out + 64 * t, a);
        }
    }
    int re = 0;
    r.chunk = (u32)g_rlc_chunk;                  // 8 (round 1's chunks) or 32 (emul_set_rlc_chunk)
    const size_t C = rlc_chunk_of(r);
    const size_t nchunks = (n + C - 1) / C;
    for (size_t c = 0; c < nchunks; c++) {
        const bool ok = rlc_chunk_serial(ws, r, c);
        for (size_t t = c * C; t < n && t < (c + 1) * C; t++) {
            if (ok) accept[t] = 1;
            else verify_final_check(ws, t);        // exact: fixed-base MSM + accept
        }
        re += ok ? 0 : 1;
    }
    *rechecked_chunks = re;
    return 0;
}
// bucket stage of the RLC mode (bucket_core.h), single-thread form: exact pipeline through the final scalars, then per superchunk
// of M proofs the combined check; returns per-superchunk verdicts (1 = passed).  The proofs of failing superchunks would go on to
// the chunk-of-8 stage (emul_u64_verify_batch_rlc covers that one).
int emul_u64_bucket_stage(const uint8_t* table, int W, const uint8_t* label, size_t label_len, size_t n, const uint8_t* V,
                          const uint8_t* proofs, const uint8_t seed[32], uint32_t M, uint8_t* passed /* ceil(n / M) */, int32_t* status) {
    VerifyWs ws;
    memset(&ws, 0, sizeof ws);
    ws.N = n;
    std::vector<uint8_t> accept(n);
    ws.commitments = V; ws.proofs = proofs; ws.accept = accept.data(); ws.status = status; ws.trace = nullptr;
    std::vector<u32> strobe_(52 * n), chal(80 * n), sc0(176 * n), cvec(200 * n), pts(208 * n), lns(24 * n), acc(30 * n), pfix(30 * n), fsc(392 * n);
    std::vector<apt_packed> atab(n * BPPP_ATAB_PER_PROOF);
    std::vector<u32> tscr((size_t)BPPP_TSCR_FE * 10 * n);
    ws.tstate = strobe_.data(); ws.chal = chal.data(); ws.sc0 = sc0.data(); ws.cvec = cvec.data(); ws.pts = pts.data();
    ws.lns = lns.data(); ws.acc = acc.data(); ws.pfix = pfix.data(); ws.fsc = fsc.data();
    ws.atab = atab.data(); ws.tscr = tscr.data();
    ws.fb_table = (const apt_packed*)table;
    ws.fb_w = W;
    t_new(ws.base, label, (u32)label_len);
    for (size_t t = 0; t < n; t++) verify_phase1(ws, t);
    for (size_t t = 0; t < n; t++) verify_tables(ws, t);
    for (size_t t = 0; t < n; t++) verify_c0_var(ws, t);
    for (size_t t = 0; t < n; t++) verify_c0_fixed(ws, t);
    for (int k = 1; k <= 4; k++)
        for (size_t t = 0; t < n; t++) verify_round(ws, t, k);
    for (size_t t = 0; t < n; t++) verify_final_scalars(ws, t);
    const size_t nsuper = (n + M - 1) / M;
    BucketWs bw;
    memset(&bw, 0, sizeof bw);
    bw.N = n; bw.M = M; bw.nb = BPPP_NG;
    for (int i = 0; i < 4; i++) {
        u64 v = 0;
        for (int k = 0; k < 8; k++) v |= (u64)seed[8 * i + k] << (8 * k);
        bw.seed[i] = v;
    }
    std::vector<u64> wab(2 * n);
    std::vector<c4_packed> c4(n);
    std::vector<u32> lhs(30 * nsuper), asc((size_t)BPPP_NG * 8 * nsuper);
    bw.status = status; bw.acc = acc.data(); bw.fsc = fsc.data(); bw.wab = wab.data(); bw.c4 = c4.data(); bw.lhs = lhs.data();
    bw.asc = asc.data(); bw.accept = accept.data();
    bw.fb.table = (const apt_packed*)table; bw.fb.W = W; bw.fb.N = nsuper;
    for (size_t t = 0; t < n; t++) bkt_prepare(bw, t);
    for (size_t c = 0; c < nsuper; c++) passed[c] = bkt_superchunk_serial(bw, c) ? 1 : 0;
    return 0;
}
// full prover pipeline, every stage in thread order
static int g_prove_next_by_msm = 0;   // emul_set_prove_next_by_msm
static int g_prove_ct = 0;            // emul_set_prove_ct: the secret-scalar sums in the "ct_prover" form (needs a 4-bit table)
static int emul_u64_prove_impl(const uint8_t* table, int W, const uint8_t* label, size_t label_len, size_t n, const uint64_t* x,
                               const uint8_t* s, const uint8_t* rnd, uint8_t* proofs, uint8_t* V, int32_t* status, const uint8_t* states,
                               size_t n_states, uint8_t* states_out);
int emul_u64_prove_batch(const uint8_t* table, int W, const uint8_t* label, size_t label_len, size_t n, const uint64_t* x,
                         const uint8_t* s, const uint8_t* rnd, uint8_t* proofs, uint8_t* V, int32_t* status) {
    return emul_u64_prove_impl(table, W, label, label_len, n, x, s, rnd, proofs, V, status, nullptr, 0, nullptr);
}
// the same over pre-loaded transcripts (bppp_u64_prove_batch_transcript)
int emul_u64_prove_batch_transcript(const uint8_t* table, int W, size_t n, const uint8_t* states, size_t n_states, const uint64_t* x,
                                    const uint8_t* s, const uint8_t* rnd, uint8_t* proofs, uint8_t* V, int32_t* status, uint8_t* states_out) {
    return emul_u64_prove_impl(table, W, nullptr, 0, n, x, s, rnd, proofs, V, status, states, n_states, states_out);
}
static int emul_u64_prove_impl(const uint8_t* table, int W, const uint8_t* label, size_t label_len, size_t n, const uint64_t* x,
                               const uint8_t* s, const uint8_t* rnd, uint8_t* proofs, uint8_t* V, int32_t* status, const uint8_t* states,
                               size_t n_states, uint8_t* states_out) {
    ProveWs w;
    memset(&w, 0, sizeof w);
    w.N = n;
    w.states = states; w.n_states = n_states; w.states_out = states_out;
    w.x = x; w.s = s; w.rnd = rnd; w.proofs = proofs; w.commitments = V; w.status = status;
    std::vector<u32> tstate(52 * n), sv((size_t)SV_COUNT * 8 * n), msc((size_t)BPPP_MSC_SETS * BPPP_NG * 8 * n), pbuf((size_t)PB_COUNT * 30 * n);
    std::vector<pt_slot> straus(n * 5 * BPPP_STRAUS_ENTRIES);      // as the library sizes it: prove_round_fold carves its window tables out of it
    w.tstate = tstate.data(); w.sv = sv.data(); w.msc = msc.data(); w.pbuf = pbuf.data(); w.straus = straus.data();
    w.fb.table = (const apt_packed*)table; w.fb.W = W; w.fb.N = n;
    t_new(w.base, label, (u32)label_len);
    auto msm = [&](MsmJob job) { for (size_t t = 0; t < n; t++) prove_msm(w, job, t); };
    if (g_prove_ct) {
        if (W != 4) return -78;
        w.ct = 1; w.fb_ct = w.fb;
    }
    auto secret = [&](MsmJob job) {          // the sums over the witness and its blindings (bppp_u64.hip: PSECRET)
        for (size_t t = 0; t < n; t++) {
            g_trace_on = g_trace_want;
            if (w.ct) prove_msm_ct(w, job, t); else prove_msm(w, job, t);
            g_trace_on = false;
        }
    };
    for (size_t t = 0; t < n; t++) prove_stage_a(w, t);
    secret(job_v());
    for (size_t t = 0; t < n; t++) prove_stage_b(w, t);
    secret(job_rcom()); secret(job_co()); secret(job_cl()); secret(job_cr());
    for (size_t t = 0; t < n; t++) prove_stage_d(w, t);
    secret(job_cs());
    for (size_t t = 0; t < n; t++) prove_stage_f(w, t);
    w.next_by_msm = g_prove_next_by_msm;
    for (int k = 1; k <= 4; k++) {
        for (size_t t = 0; t < n; t++) prove_round_scalars(w, t, k);
        msm(job_x()); msm(job_r(k));
        if (w.next_by_msm || k == 1) msm(job_e(k));          // the even halves of the level's commitment, scalars left by the previous fold (the library fuses the three launches)
        for (size_t t = 0; t < n; t++) prove_round_fold(w, t, k);
        if (k < 4 && !w.next_by_msm)                            // part two: the next commitment (the library runs it under the next round's sums)
            for (size_t t = 0; t < n; t++) prove_round_next(w, t, k);
    }
    for (size_t t = 0; t < n; t++) prove_export_state(w, t);
    return 0;
}
// the small-call form of the u64 prover: next commitments as fixed-base sums (prove_core.h: ProveWs::next_by_msm)
void emul_set_prove_next_by_msm(int on) { g_prove_next_by_msm = on; }
void emul_set_prove_ct(int on) { g_prove_ct = on; }
// one window's table addition in both forms -- the digit-addressed gather and the full scan with masked select -- on the same
// accumulator: the two results (affine) and, for the scan, the number of table entries it read (always 15)
int emul_fb_lookup_both(const uint8_t* table, int base, int w, const uint8_t k32[32], const uint8_t acc_in[64], uint8_t out_fast[64], uint8_t out_ct[64]) {
    FbTable fbt = {(const apt_packed*)table, 4, 1};
    sc k;
    if (!sc_from_be(k, k32)) return -1;
    apt a0;
    if (!apt_from_xy64(a0, acc_in)) return -1;
    pt p, q;
    pt_from_affine(p, a0);
    q = p;
    { FbGeom g; fb_geom(g, fbt, false); fb_lookup_add(p, g, base, w, k.v); }
    fb_lookup_add_ct(q, fbt, base, w, k.v);
    apt r;
    pt_to_affine(r, p); apt_to_xy64(out_fast, r);
    pt_to_affine(r, q); apt_to_xy64(out_ct, r);
    return 0;
}
// wire format, the other way: the prover's 64-byte output -> SEC1 compressed, lane by lane
void emul_sec1_compress(size_t n, const uint8_t* c64, const uint8_t* p928, uint8_t* c33, uint8_t* p525) {
    for (size_t t = 0; t < n; t++)
        for (int j = 0; j < 15; j++) sec1_compress_lane(c33, p525, c64, p928, t, j);
}
// wire format: SEC1 compressed inputs -> 64-byte form, lane by lane
void emul_sec1_expand(size_t n, const uint8_t* c33, const uint8_t* p525, uint8_t* c64, uint8_t* p928) {
    for (size_t t = 0; t < n; t++)
        for (int j = 0; j < 15; j++) sec1_expand_lane(c64, p928, c33, p525, t, j);
}
// generic WNLA commit / verify, every stage in thread order (commit: out_points; verify: accept)
// Pre-loaded transcripts for the NEXT generic verify call (emul_wnla_run / emul_recip_verify / emul_circuit_verify): states in
// (n_states = 1 or n), per-instance advanced states out; consumed by that call.
// the generic verifiers' fast variable-base path (affine window tables of the round points): buffers for the emulation; switched off
// with emul_set_generic_slow_rounds(1) so that both paths are covered
static int g_generic_slow_rounds = 0;
void emul_set_generic_slow_rounds(int on) { g_generic_slow_rounds = on; }
struct FastRounds {
    std::vector<apt_packed> atab;
    std::vector<u32> tscr, rpts;
    void attach(WnlaWs& w, size_t n, int rounds, size_t extra_points = 0) {
        w.atab = nullptr; w.tscr = nullptr; w.rpts = nullptr;
        if (rounds == 0 || g_generic_slow_rounds) return;
        const size_t np = 2 * (size_t)rounds + extra_points;
        atab.assign(np * 16 * n, apt_packed());
        tscr.assign(14 * np * 10 * n, 0u);
        rpts.assign(np * 16 * n, 0u);
        w.atab = atab.data(); w.tscr = tscr.data(); w.rpts = rpts.data();
    }
};
static TranscriptIo g_tio = {nullptr, 0, nullptr, 0};
// random-linear-combination mode for the NEXT emul_recip_verify call (consumed by it); flags_out: 1 per chunk of 8 re-checked exactly
static const uint8_t* g_rlc_seed = nullptr;
static uint8_t* g_rlc_flags = nullptr;
void emul_set_rlc(const uint8_t* seed32, uint8_t* flags_out) { g_rlc_seed = seed32; g_rlc_flags = flags_out; }
// superchunk size of the bucket stage in front of the generic RLC mode (0 = chunks of 8 only); passed_out[nsuper] (optional)
// receives each superchunk's verdict.  Consumed by the next emul_recip_verify call in RLC mode.
static unsigned g_rlc_super_m = 0;
static uint8_t* g_rlc_super_passed = nullptr;
void emul_set_rlc_superchunk(unsigned m, uint8_t* passed_out) { g_rlc_super_m = m; g_rlc_super_passed = passed_out; }
void emul_set_transcripts(const uint8_t* states, size_t n_states, uint8_t* states_out) {
    g_tio.states = states; g_tio.n_states = n_states; g_tio.states_out = states_out;
}
static TranscriptIo take_tio() { TranscriptIo t = g_tio; g_tio.states = nullptr; g_tio.n_states = 0; g_tio.states_out = nullptr; return t; }
int emul_wnla_run(int commit, const uint8_t* table, int W, int ng, int nh, const uint8_t* label, size_t label_len, size_t n,
                  const uint8_t* commitments, const uint8_t* c, const uint8_t* rho, const uint8_t* mu, int rounds, const uint8_t* proof_r,
                  const uint8_t* proof_x, const uint8_t* proof_l, int nl, const uint8_t* proof_n, int nn, uint8_t* out_points,
                  uint8_t* accept, int32_t* status) {
    WnlaWs w;
    memset(&w, 0, sizeof w);
    w.N = n; w.ng = ng; w.nh = nh; w.rounds = rounds; w.nl = nl; w.nn = nn;
    w.commitments = commitments; w.c = c; w.rho = rho; w.mu = mu; w.proof_r = proof_r; w.proof_x = proof_x; w.proof_l = proof_l;
    w.proof_n = proof_n; w.out_points = out_points; w.accept = accept; w.status = status;
    w.stride_r = (size_t)rounds * 64; w.stride_x = (size_t)rounds * 64; w.stride_l = (size_t)nl * 32; w.stride_n = (size_t)nn * 32;
    const size_t T = (size_t)1 << rounds, NB = 1 + ng + nh;
    std::vector<u32> ts(52 * n), acc(30 * n), pf(30 * n), ys((rounds ? rounds : 1) * 8 * n), tab(2 * T * 8 * n), msc(NB * 8 * n);
    std::vector<pt_slot> straus(n * 2 * BPPP_STRAUS_ENTRIES);
    w.tstate = ts.data(); w.acc = acc.data(); w.pfix = pf.data(); w.ys = ys.data(); w.tab = tab.data(); w.msc = msc.data();
    w.straus = straus.data();
    w.fb.table = (const apt_packed*)table; w.fb.W = W; w.fb.N = n;
    auto msm = [&]() {
        for (size_t t = 0; t < n; t++) {
            pt a;
            FbRanges rg;
            wnla_msm_ranges(rg, w);
            fb_sum_serial(a, w.fb, t, w.msc, rg);
            wnla_verify_store(w, t, a);
        }
    };
    if (commit) {
        for (size_t t = 0; t < n; t++) wnla_commit_scalars(w, t);
        msm();
        for (size_t t = 0; t < n; t++) { pt total; ws_ld_pt(total, w.pfix, n, t); wnla_commit_store(w, t, total); }
    } else {
        t_new(w.base, label, (u32)label_len);
        w.tio = take_tio();
        w.tio.no_ops = rounds == 0;
        FastRounds fastr;
        fastr.attach(w, n, rounds);
        for (size_t t = 0; t < n; t++) wnla_verify_begin(w, t);
        if (w.atab) for (size_t t = 0; t < n; t++) wnla_verify_tables(w, t);
        for (int k = 1; k <= rounds; k++)
            for (size_t t = 0; t < n; t++) wnla_verify_round(w, t, k);
        for (size_t t = 0; t < n; t++) emul_final_scalars(w, t);
        msm();
        for (size_t t = 0; t < n; t++) wnla_verify_accept(w, t);
        for (size_t t = 0; t < n; t++) tio_export(w.tio, w.base, w.tstate, n, w.status, t);
    }
    return 0;
}
// generic reciprocal verify: circuit stage then the generic WNLA stage, thread order
int emul_recip_verify(const uint8_t* table, int W, int NG, int NH, int nd, int np, const uint8_t* label, size_t label_len, size_t n,
                      const uint8_t* commitments, const uint8_t* proofs, int rounds, int nl, int nn, uint8_t* accept, int32_t* status) {
    const size_t T = (size_t)1 << rounds, NB = 1 + NG + NH, proof_bytes = 64 * (5 + 2 * (size_t)rounds) + 32 * ((size_t)nl + nn);
    RecipWs r;
    memset(&r, 0, sizeof r);
    r.N = n; r.nd = nd; r.np = np; r.rounds = rounds; r.nl = nl; r.nn = nn; r.NG = NG; r.NH = NH; r.proof_bytes = proof_bytes;
    r.commitments = commitments; r.proofs = proofs; r.status = status;
    std::vector<u32> ts(52 * n), sc0((size_t)(nd + 6) * 8 * n), pts(80 * n), acc(30 * n), pf(30 * n), inv((size_t)np * 8 * n),
        ys((rounds ? rounds : 1) * 8 * n), tab(2 * T * 8 * n), msc(NB * 8 * n);
    std::vector<uint8_t> wc(n * 64), wcv(n * (size_t)NH * 32), wrho(n * 32), wmu(n * 32);
    std::vector<pt_slot> straus(n * 5 * BPPP_STRAUS_ENTRIES);
    r.tstate = ts.data(); r.sc0 = sc0.data(); r.pts = pts.data(); r.acc = acc.data(); r.pfix = pf.data(); r.inv = inv.data();
    r.straus = straus.data(); r.wn_commit = wc.data(); r.wn_c = wcv.data(); r.wn_rho = wrho.data(); r.wn_mu = wmu.data();
    r.fb.table = (const apt_packed*)table; r.fb.W = W; r.fb.N = n;
    t_new(r.base, label, (u32)label_len);
    r.tio = take_tio();
    WnlaWs w;
    memset(&w, 0, sizeof w);
    w.N = n; w.ng = NG; w.nh = NH; w.rounds = rounds; w.nl = nl; w.nn = nn;
    w.base = r.base; w.tio = r.tio;
    w.commitments = r.wn_commit; w.c = r.wn_c; w.rho = r.wn_rho; w.mu = r.wn_mu;
    w.proof_r = proofs + 256; w.proof_x = proofs + 256 + 64 * (size_t)rounds; w.proof_l = proofs + 320 + 128 * (size_t)rounds;
    w.proof_n = w.proof_l + 32 * (size_t)nl;
    w.stride_r = w.stride_x = w.stride_l = w.stride_n = proof_bytes;
    w.transcript_preloaded = 1;
    w.accept = accept; w.status = status; w.tstate = r.tstate; w.acc = r.acc; w.pfix = r.pfix; w.ys = ys.data(); w.tab = tab.data();
    w.msc = msc.data(); w.straus = straus.data(); w.fb = r.fb;
    for (size_t t = 0; t < n; t++) recip_phase1(r, t);
    for (size_t t = 0; t < n; t++) {
        pt a;
        FbRanges rg;
        recip_c0_fixed_ranges(rg, r);
        fb_sum_serial(a, r.fb, t, r.sc0, rg);
        recip_c0_fixed_store(r, t, a);
    }
    FastRounds fastr;
    fastr.attach(w, n, rounds, 5);                    // as recip_verify_device_impl: the five C0 points' tables behind the round points'
    r.atab = w.atab; r.tscr = w.tscr; r.atab_first = 2 * rounds * 16;
    if (r.atab) for (size_t t = 0; t < n; t++) recip_c0_tables(r, t);
    for (size_t t = 0; t < n; t++) recip_c0_var(r, t);
    for (size_t t = 0; t < n; t++) recip_c0_finish(r, t);
    for (size_t t = 0; t < n; t++) wnla_verify_begin(w, t);
    if (w.atab) for (size_t t = 0; t < n; t++) wnla_verify_tables(w, t);
    for (int k = 1; k <= rounds; k++)
        for (size_t t = 0; t < n; t++) wnla_verify_round(w, t, k);
    for (size_t t = 0; t < n; t++) emul_final_scalars(w, t);
    const uint8_t* seed = g_rlc_seed;
    uint8_t* flags_out = g_rlc_flags;
    g_rlc_seed = nullptr; g_rlc_flags = nullptr;
    const size_t nchunks = (n + BPPP_RLC_CHUNK - 1) / BPPP_RLC_CHUNK;
    std::vector<uint8_t> flag(nchunks, 1);
    std::vector<u32> rl_lhs(30 * n), rl_sc((size_t)NB * 8 * n);
    if (seed) {     // the kernels' order: weighted commitments, one combined check per chunk, exact re-check of what did not pass
        RlcWs rl;
        memset(&rl, 0, sizeof rl);
        for (int i = 0; i < 4; i++) { u64 v = 0; for (int k = 0; k < 8; k++) v |= (u64)seed[8 * i + k] << (8 * k); rl.seed[i] = v; }
        rl.lhs = rl_lhs.data(); rl.sc = rl_sc.data(); rl.flag = flag.data();
        memset(accept, 0, n);
        const unsigned SM = g_rlc_super_m;
        uint8_t* passed_out = g_rlc_super_passed;
        g_rlc_super_m = 0; g_rlc_super_passed = nullptr;
        std::vector<uint8_t> sflag;
        std::vector<u64> wab;
        std::vector<c4_packed> c4;
        std::vector<u32> blhs, basc;
        if (SM) {       // the kernels' order: bucket stage over superchunks of SM instances, then chunks of 8 for what failed it
            const size_t nsuper = (n + SM - 1) / SM;
            BucketWs bw;
            memset(&bw, 0, sizeof bw);
            bw.N = n; bw.M = SM; bw.nb = (int)NB;
            for (int i = 0; i < 4; i++) bw.seed[i] = rl.seed[i];
            sflag.assign(nsuper, 1); wab.assign(2 * n, 0); c4.resize(n); blhs.assign(30 * nsuper, 0); basc.assign(NB * 8 * nsuper, 0);
            bw.status = status; bw.acc = w.acc; bw.fsc = w.msc; bw.wab = wab.data(); bw.c4 = c4.data(); bw.lhs = blhs.data();
            bw.asc = basc.data(); bw.accept = accept; bw.sflag = sflag.data();
            bw.fb = w.fb; bw.fb.N = nsuper;
            for (size_t t = 0; t < n; t++) bkt_prepare(bw, t);
            for (size_t c = 0; c < nsuper; c++) {
                const bool ok = bkt_superchunk_serial(bw, c);
                sflag[c] = ok ? 0 : 1;
                if (passed_out) passed_out[c] = ok ? 1 : 0;
                if (ok) for (size_t t = c * SM; t < (c + 1) * (size_t)SM && t < n; t++) accept[t] = status[t] == ST_OK ? 1 : 0;
            }
            rl.sflag = sflag.data(); rl.super_m = SM;
        }
        for (size_t t = 0; t < n; t++) if (status[t] == ST_OK && !rlc_done_by_bucket_stage(rl, t)) wnla_rlc_lhs(w, rl, t);
        for (size_t ch = 0; ch < nchunks; ch++) {
            if (rlc_done_by_bucket_stage(rl, ch * BPPP_RLC_CHUNK)) { flag[ch] = 0; continue; }
            const bool ok = wnla_rlc_chunk_serial(w, rl, ch);
            flag[ch] = ok ? 0 : 1;
            if (ok) for (size_t t = ch * BPPP_RLC_CHUNK; t < (ch + 1) * BPPP_RLC_CHUNK; t++) accept[t] = 1;
        }
        if (flags_out) memcpy(flags_out, flag.data(), nchunks);
    }
    for (size_t t = 0; t < n; t++) {
        if (!flag[t / BPPP_RLC_CHUNK]) continue;
        pt a;
        FbRanges rg;
        wnla_msm_ranges(rg, w);
        fb_sum_serial(a, w.fb, t, w.msc, rg);
        wnla_verify_store(w, t, a);
        wnla_verify_accept(w, t);
    }
    for (size_t t = 0; t < n; t++) tio_export(w.tio, w.base, w.tstate, n, w.status, t);
    return 0;
}
// generic ArithmeticCircuit::verify (circuit_core.h) + the WNLA stage, every phase in thread order
int emul_circuit_verify(const uint8_t* table, int W, int NG, int NH, const size_t dims[6], int f_l, int f_m, const uint8_t* W_m,
                        const uint8_t* W_l, const uint8_t* a_m, const uint8_t* a_l, const int32_t* part_lo, const int32_t* part_ll,
                        const int32_t* part_lr, const int32_t* part_no, const uint8_t* label, size_t label_len, size_t n,
                        const uint8_t* commitments, const uint8_t* proofs, int rounds, int nl, int nn, uint8_t* accept, int32_t* status,
                        uint8_t* c0_out /* n x 64 or null */, uint8_t* c_out /* n x NH x 32 or null */) {
    CircuitHostData hd;
    if (!circuit_host_build(hd, dims, W_m, W_l, a_m, a_l, part_lo, part_ll, part_lr, part_no)) return -2;
    const size_t nm = dims[0], k = dims[2], nv = dims[4];
    if ((int)nm > NG || (int)nv + 9 > NH) return -2;
    const size_t T = (size_t)1 << rounds, NB = 1 + NG + NH, proof_bytes = 64 * (4 + 2 * (size_t)rounds) + 32 * ((size_t)nl + nn);
    CircuitWs r;
    memset(&r, 0, sizeof r);
    CircuitDev& cd = r.cd;
    cd.nm = (int)nm; cd.no = (int)dims[1]; cd.k = (int)k; cd.nl = (int)dims[3]; cd.nv = (int)nv; cd.nw = (int)dims[5]; cd.f_l = f_l; cd.f_m = f_m;
    hd.rl.push_back(0); hd.rm.push_back(0); hd.vl.resize(hd.vl.size() + 8); hd.vm.resize(hd.vm.size() + 8);   // never empty
    cd.colptr_l = hd.cpl.data(); cd.rows_l = hd.rl.data(); cd.vals_l = hd.vl.data();
    cd.colptr_m = hd.cpm.data(); cd.rows_m = hd.rm.data(); cd.vals_m = hd.vm.data();
    cd.colmap = hd.colmap.data(); cd.a_l = hd.al.data(); cd.a_m = hd.am.data();
    r.N = n; r.rounds = rounds; r.NG = NG; r.NH = NH; r.proof_bytes = proof_bytes;
    r.commitments = commitments; r.proofs = proofs; r.status = status;
    std::vector<u32> ts(52 * n), lam((size_t)cd.nl * 8 * n), muv(nm * 8 * n), coef((3 * nm + 3 * nv) * 8 * n), sc0((nm + 5 + k) * 8 * n),
        pts((4 + k) * 16 * n), acc(30 * n), pf(30 * n), ys((rounds ? rounds : 1) * 8 * n), tab(2 * T * 8 * n), msc(NB * 8 * n);
    std::vector<uint8_t> wc(n * 64), wcv(n * (size_t)NH * 32), wrho(n * 32), wmu(n * 32);
    std::vector<pt_slot> straus(n * 5 * BPPP_STRAUS_ENTRIES);
    r.tstate = ts.data(); r.lamv = lam.data(); r.muv = muv.data(); r.coef = coef.data(); r.sc0 = sc0.data(); r.pts = pts.data();
    r.acc = acc.data(); r.pfix = pf.data(); r.straus = straus.data();
    r.wn_commit = wc.data(); r.wn_c = wcv.data(); r.wn_rho = wrho.data(); r.wn_mu = wmu.data();
    r.fb.table = (const apt_packed*)table; r.fb.W = W; r.fb.N = n;
    t_new(r.base, label, (u32)label_len);
    r.tio = take_tio();
    WnlaWs w;
    memset(&w, 0, sizeof w);
    w.base = r.base; w.tio = r.tio;
    w.N = n; w.ng = NG; w.nh = NH; w.rounds = rounds; w.nl = nl; w.nn = nn;
    w.commitments = r.wn_commit; w.c = r.wn_c; w.rho = r.wn_rho; w.mu = r.wn_mu;
    w.proof_r = proofs + 256; w.proof_x = proofs + 256 + 64 * (size_t)rounds; w.proof_l = proofs + 256 + 128 * (size_t)rounds;
    w.proof_n = w.proof_l + 32 * (size_t)nl;
    w.stride_r = w.stride_x = w.stride_l = w.stride_n = proof_bytes;
    w.transcript_preloaded = 1;
    w.accept = accept; w.status = status; w.tstate = r.tstate; w.acc = r.acc; w.pfix = r.pfix; w.ys = ys.data(); w.tab = tab.data();
    w.msc = msc.data(); w.straus = straus.data(); w.fb = r.fb;
    for (size_t t = 0; t < n; t++) circuit_phase1(r, t);
    for (size_t t = 0; t < n; t++) {
        pt a;
        FbRanges rg;
        circuit_c0_fixed_ranges(rg, r);
        fb_sum_serial(a, r.fb, t, r.sc0, rg);
        circuit_c0_fixed_store(r, t, a);
    }
    FastRounds fastr;
    fastr.attach(w, n, rounds, 4 + k);                // as circuit_verify_host_impl: the 4 + k C0 points' tables behind the round points'
    r.atab = w.atab; r.tscr = w.tscr; r.atab_first = 2 * rounds * 16;
    if (r.atab) for (size_t t = 0; t < n; t++) circuit_c0_tables(r, t);
    for (size_t t = 0; t < n; t++) circuit_c0_var(r, t);
    for (size_t t = 0; t < n; t++) circuit_c0_finish(r, t);
    if (c0_out) memcpy(c0_out, wc.data(), wc.size());
    if (c_out) memcpy(c_out, wcv.data(), wcv.size());
    for (size_t t = 0; t < n; t++) wnla_verify_begin(w, t);
    if (w.atab) for (size_t t = 0; t < n; t++) wnla_verify_tables(w, t);
    for (int kk = 1; kk <= rounds; kk++)
        for (size_t t = 0; t < n; t++) wnla_verify_round(w, t, kk);
    for (size_t t = 0; t < n; t++) emul_final_scalars(w, t);
    for (size_t t = 0; t < n; t++) {
        pt a;
        FbRanges rg;
        wnla_msm_ranges(rg, w);
        fb_sum_serial(a, w.fb, t, w.msc, rg);
        wnla_verify_store(w, t, a);
    }
    for (size_t t = 0; t < n; t++) wnla_verify_accept(w, t);
    for (size_t t = 0; t < n; t++) tio_export(w.tio, w.base, w.tstate, n, w.status, t);
    return 0;
}
// generic WeightNormLinearArgument::prove (wnla_prove_core.h), every stage in thread order; returns the proof shape through
// rounds_out / nl_out / nn_out (buffers must be large enough: rounds <= 16, final vectors <= 8 entries)
int emul_wnla_prove(const uint8_t* table, int W, int ng, int nh, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                    const uint8_t* c, const uint8_t* rho, const uint8_t* mu, const uint8_t* l, int nl, const uint8_t* nv, int nn,
                    uint8_t* proof_r, uint8_t* proof_x, uint8_t* proof_l, uint8_t* proof_n, int32_t* status, int* rounds_out, int* nl_out,
                    int* nn_out) {
    size_t rounds, nl_f, nn_f;
    wnla_proof_shape((size_t)nl, (size_t)nn, rounds, nl_f, nn_f);
    *rounds_out = (int)rounds; *nl_out = (int)nl_f; *nn_out = (int)nn_f;
    const size_t NB = 1 + ng + nh;
    WnlaProveWs w;
    memset(&w, 0, sizeof w);
    w.N = n; w.ng = ng; w.nh = nh; w.nl = nl; w.nn = nn; w.rounds = (int)rounds; w.nl_f = (int)nl_f; w.nn_f = (int)nn_f;
    w.commitments = commitments; w.c = c; w.rho = rho; w.mu = mu; w.l_in = l; w.n_in = nv;
    w.proof_r = proof_r; w.proof_x = proof_x; w.proof_l = proof_l; w.proof_n = proof_n; w.status = status;
    std::vector<u32> ts(52 * n), vl((size_t)(nl + 1) * 8 * n), vn((size_t)(nn + 1) * 8 * n), vc((size_t)nh * 8 * n), ch((size_t)nh * 8 * n),
        cg((size_t)(ng + 1) * 8 * n), prm(24 * n), com(16 * n), msc(3 * NB * 8 * n), pb(90 * n);
    w.tstate = ts.data(); w.vl = vl.data(); w.vn = vn.data(); w.vc = vc.data(); w.ch = ch.data(); w.cg = cg.data(); w.prm = prm.data();
    w.com = com.data(); w.msc = msc.data(); w.pbuf = pb.data();
    std::vector<pt_slot> wstraus(n * 2 * BPPP_STRAUS_ENTRIES);
    w.straus = wstraus.data();
    w.fb.table = (const apt_packed*)table; w.fb.W = W; w.fb.N = n;
    t_new(w.base, label, (u32)label_len);
    w.tio = take_tio();
    w.tio.no_ops = rounds == 0;
    if (g_prove_ct) {          // bppp_wnla_prove_batch with "ct_prover": l and n are the caller's secrets (emulator: the 4-bit table is the table)
        if (W != 4) return -78;
        w.ct = 1; w.fb_ct = w.fb;
    }
    auto msm = [&](int set, int oddsh = -1) {
        for (size_t t = 0; t < n; t++) {
            pt a;
            FbRanges rg;
            wnla_prove_msm_ranges(rg, w, oddsh);
            secret_sum(a, w.fb, w.fb_ct, w.ct, t, w.msc + (size_t)set * wp_set_words(w), rg);
            ws_st_pt(w.pbuf + (size_t)set * 30 * n, n, t, a);
        }
    };
    for (size_t t = 0; t < n; t++) wnla_prove_init(w, t);
    for (int k = 0; k < (int)rounds; k++) {
        for (size_t t = 0; t < n; t++) wnla_prove_round_scalars(w, t, k);
        msm(0);
        msm(1, k);
        for (size_t t = 0; t < n; t++) wnla_prove_round_fold(w, t, k);
        if (k == 0 && rounds > 1) msm(2);
    }
    for (size_t t = 0; t < n; t++) wnla_prove_finish(w, t);
    for (size_t t = 0; t < n; t++) tio_export(w.tio, w.base, w.tstate, n, w.status, t);
    return 0;
}
// generic ArithmeticCircuit::prove (circuit_prove_core.h + wnla_prove_core.h), every stage in thread order
int emul_circuit_prove(const uint8_t* table, int W, int NG, int NH, const size_t dims[6], int f_l, int f_m, const uint8_t* W_m,
                       const uint8_t* W_l, const uint8_t* a_m, const uint8_t* a_l, const int32_t* part_lo, const int32_t* part_ll,
                       const int32_t* part_lr, const int32_t* part_no, const uint8_t* label, size_t label_len, size_t n,
                       const uint8_t* v_pts, const uint8_t* v, const uint8_t* s_v, const uint8_t* w_l, const uint8_t* w_r, const uint8_t* w_o,
                       const uint8_t* rnd, uint8_t* proofs, int32_t* status) {
    CircuitHostData hd;
    if (!circuit_host_build(hd, dims, W_m, W_l, a_m, a_l, part_lo, part_ll, part_lr, part_no)) return -2;
    const size_t nm = dims[0], no = dims[1], k = dims[2], nl = dims[3], nv = dims[4], NB = 1 + NG + NH, n_rnd = 18 + nv + nm;
    (void)no; (void)k;
    size_t rounds, nl_f, nn_f;
    wnla_proof_shape((size_t)NH, (size_t)NG, rounds, nl_f, nn_f);
    const size_t proof_bytes = 64 * (4 + 2 * rounds) + 32 * (nl_f + nn_f);
    CircuitProveWs p;
    memset(&p, 0, sizeof p);
    CircuitDev& cd = p.cd;
    cd.nm = (int)nm; cd.no = (int)dims[1]; cd.k = (int)dims[2]; cd.nl = (int)nl; cd.nv = (int)nv; cd.nw = (int)dims[5]; cd.f_l = f_l; cd.f_m = f_m;
    hd.rl.push_back(0); hd.rm.push_back(0); hd.vl.resize(hd.vl.size() + 8); hd.vm.resize(hd.vm.size() + 8);
    cd.colptr_l = hd.cpl.data(); cd.rows_l = hd.rl.data(); cd.vals_l = hd.vl.data();
    cd.colptr_m = hd.cpm.data(); cd.rows_m = hd.rm.data(); cd.vals_m = hd.vm.data();
    cd.colmap = hd.colmap.data(); cd.a_l = hd.al.data(); cd.a_m = hd.am.data();
    std::vector<int> parts(3 * nv + nm);
    for (size_t j = 0; j < nv; j++) { parts[j] = part_lo[j]; parts[nv + j] = part_ll[j]; parts[2 * nv + j] = part_lr[j]; }
    for (size_t j = 0; j < nm; j++) parts[3 * nv + j] = part_no[j];
    p.N = n; p.NG = NG; p.NH = NH; p.n_rnd = (int)n_rnd; p.rnd_stride = n_rnd * 32; p.part = parts.data();
    p.v_pts = v_pts; p.v = v; p.s_v = s_v; p.w_l = w_l; p.w_r = w_r; p.w_o = w_o; p.rnd = rnd; p.status = status;
    std::vector<uint8_t> head(n * 256), wc(n * 64), wcv(n * (size_t)NH * 32), wrho(n * 32), wmu(n * 32), wlv(n * (size_t)NH * 32), wnv(n * (size_t)NG * 32);
    std::vector<u32> ts(52 * n), r9(4 * 72 * n), lv(6 * nv * 8 * n), nvv(4 * nm * 8 * n), lam(nl * 8 * n), muv(nm * 8 * n),
        coef((3 * nm + 3 * nv) * 8 * n), misc(64 * n), msc(3 * NB * 8 * n, 0), pb(90 * n);
    p.proof_head = head.data(); p.tstate = ts.data();
    p.ro = r9.data(); p.rl = p.ro + 72 * n; p.rr = p.ro + 144 * n; p.rs = p.ro + 216 * n;
    p.lo = lv.data(); p.ll = p.lo + nv * 8 * n; p.lr = p.lo + 2 * nv * 8 * n; p.ls = p.lo + 3 * nv * 8 * n; p.v1 = p.lo + 4 * nv * 8 * n;
    p.cl0 = p.lo + 5 * nv * 8 * n;
    p.no = nvv.data(); p.nl = p.no + nm * 8 * n; p.nr = p.no + 2 * nm * 8 * n; p.ns = p.no + 3 * nm * 8 * n;
    p.lamv = lam.data(); p.muv = muv.data(); p.coef = coef.data(); p.misc = misc.data(); p.msc = msc.data(); p.pbuf = pb.data();
    p.wn_commit = wc.data(); p.wn_c = wcv.data(); p.wn_rho = wrho.data(); p.wn_mu = wmu.data(); p.wn_l = wlv.data(); p.wn_n = wnv.data();
    p.fb.table = (const apt_packed*)table; p.fb.W = W; p.fb.N = n;
    if (g_prove_ct) {          // "ct_prover": c_l, c_r, c_o, c_s and the prover-side WNLA commitment in the full-scan form
        if (W != 4) return -78;
        p.ct = 1; p.fb_ct = p.fb;
    }
    t_new(p.base, label, (u32)label_len);
    p.tio = take_tio();
    std::vector<uint8_t> pr(n * rounds * 64 + 1), px(n * rounds * 64 + 1), pl(n * nl_f * 32 + 1), pn(n * nn_f * 32 + 1);
    WnlaProveWs w;
    memset(&w, 0, sizeof w);
    w.N = n; w.ng = NG; w.nh = NH; w.nl = NH; w.nn = NG; w.rounds = (int)rounds; w.nl_f = (int)nl_f; w.nn_f = (int)nn_f;
    w.transcript_preloaded = 1;
    w.commitments = p.wn_commit; w.c = p.wn_c; w.rho = p.wn_rho; w.mu = p.wn_mu; w.l_in = p.wn_l; w.n_in = p.wn_n;
    w.proof_r = pr.data(); w.proof_x = px.data(); w.proof_l = pl.data(); w.proof_n = pn.data(); w.status = status;
    std::vector<u32> vl((size_t)(NH + 1) * 8 * n), vn((size_t)(NG + 1) * 8 * n), vc((size_t)NH * 8 * n), ch((size_t)NH * 8 * n),
        cg((size_t)(NG + 1) * 8 * n), prm(24 * n), com(16 * n);
    w.tstate = p.tstate; w.vl = vl.data(); w.vn = vn.data(); w.vc = vc.data(); w.ch = ch.data(); w.cg = cg.data(); w.prm = prm.data();
    w.com = com.data(); w.msc = p.msc; w.pbuf = p.pbuf; w.fb = p.fb;
    std::vector<pt_slot> wstraus(n * 2 * BPPP_STRAUS_ENTRIES);
    w.straus = wstraus.data();
    auto cmsm = [&](int set, bool with_g) {
        for (size_t t = 0; t < n; t++) {
            pt a;
            FbRanges rg;
            cp_ranges(rg, p, with_g);
            secret_sum(a, p.fb, p.fb_ct, p.ct, t, p.msc + (size_t)set * cp_set_words(p), rg);
            ws_st_pt(p.pbuf + (size_t)set * 30 * n, n, t, a);
        }
    };
    auto wmsm = [&](int set, int oddsh = -1) {
        for (size_t t = 0; t < n; t++) {
            pt a;
            FbRanges rg;
            wnla_prove_msm_ranges(rg, w, oddsh);
            fb_sum_serial(a, w.fb, t, w.msc + (size_t)set * wp_set_words(w), rg);
            ws_st_pt(w.pbuf + (size_t)set * 30 * n, n, t, a);
        }
    };
    for (size_t t = 0; t < n; t++) circuit_prove_stage_a(p, t);
    for (int set = 0; set < 3; set++) cmsm(set, false);
    for (size_t t = 0; t < n; t++) circuit_prove_stage_b(p, t);
    cmsm(0, false);
    for (size_t t = 0; t < n; t++) circuit_prove_stage_c(p, t);
    cmsm(0, true);
    for (size_t t = 0; t < n; t++) circuit_prove_stage_d(p, t);
    for (size_t t = 0; t < n; t++) wnla_prove_init(w, t);
    for (int kk = 0; kk < (int)rounds; kk++) {
        for (size_t t = 0; t < n; t++) wnla_prove_round_scalars(w, t, kk);
        wmsm(0);
        wmsm(1, kk);
        for (size_t t = 0; t < n; t++) wnla_prove_round_fold(w, t, kk);
        if (kk == 0 && rounds > 1) wmsm(2);
    }
    for (size_t t = 0; t < n; t++) wnla_prove_finish(w, t);
    for (size_t t = 0; t < n; t++) tio_export(p.tio, p.base, p.tstate, n, status, t);
    for (size_t i = 0; i < n; i++) {
        uint8_t* o = proofs + i * proof_bytes;
        if (status[i] != 0) { memset(o, 0, proof_bytes); continue; }
        memcpy(o, &head[i * 256], 256); o += 256;
        memcpy(o, &pr[i * rounds * 64], rounds * 64); o += rounds * 64;
        memcpy(o, &px[i * rounds * 64], rounds * 64); o += rounds * 64;
        memcpy(o, &pl[i * nl_f * 32], nl_f * 32); o += nl_f * 32;
        memcpy(o, &pn[i * nn_f * 32], nn_f * 32);
    }
    return (int)proof_bytes;
}
// generic ReciprocalRangeProofProtocol::prove (recip_prove_core.h -> circuit_prove_core.h -> wnla_prove_core.h), thread order
int emul_recip_prove(const uint8_t* table, int W, int NG, int NH, int nd, int np, const uint8_t* label, size_t label_len, size_t n,
                     const uint8_t* commitments, const uint8_t* x, const uint8_t* sblind, const uint8_t* digits, const uint8_t* m,
                     const uint8_t* rnd, uint8_t* proofs, int32_t* status) {
    RecipPattern P;
    recip_pattern_build(P, (size_t)nd, (size_t)np);
    const size_t nm = nd, nv = nd + 1, nl = nv, NB = 1 + NG + NH, n_rnd = 20 + 2 * (size_t)nd;
    size_t rounds, nl_f, nn_f;
    wnla_proof_shape((size_t)NH, (size_t)NG, rounds, nl_f, nn_f);
    const size_t proof_bytes = 64 * (5 + 2 * rounds) + 32 * (nl_f + nn_f);
    std::vector<u32> ts(52 * n), inst((1 + (size_t)np) * 8 * n), scr(((size_t)nd + np) * 8 * n), msc(3 * NB * 8 * n, 0), pb(90 * n);
    std::vector<uint8_t> cpv(n * nv * 32), cpsv(n * 32), cpwr(n * nm * 32), cpvp(n * 64), prr(n * 64);
    RecipProveWs r;
    memset(&r, 0, sizeof r);
    r.N = n; r.nd = nd; r.np = np; r.NG = NG; r.NH = NH; r.n_rnd = (int)n_rnd;
    r.commitments = commitments; r.x = x; r.s = sblind; r.digits = digits; r.m = m; r.rnd = rnd; r.status = status; r.tstate = ts.data();
    r.inst_vals = inst.data(); r.scr = scr.data(); r.msc = msc.data(); r.pbuf = pb.data();
    r.cp_v = cpv.data(); r.cp_sv = cpsv.data(); r.cp_wr = cpwr.data(); r.cp_vpts = cpvp.data(); r.proof_r = prr.data();
    r.fb.table = (const apt_packed*)table; r.fb.W = W; r.fb.N = n;
    t_new(r.base, label, (u32)label_len);
    r.tio = take_tio();
    CircuitProveWs p;
    memset(&p, 0, sizeof p);
    CircuitDev& cd = p.cd;
    cd.nm = (int)nm; cd.no = np; cd.k = 1; cd.nl = (int)nl; cd.nv = (int)nv; cd.nw = (int)P.dims[5]; cd.f_l = 1; cd.f_m = 0;
    cd.colptr_l = P.hd.cpl.data(); cd.rows_l = P.hd.rl.data(); cd.vals_l = P.hd.vl.data();
    cd.colptr_m = P.hd.cpm.data(); cd.rows_m = P.hd.rm.data(); cd.vals_m = P.hd.vm.data();
    cd.colmap = P.hd.colmap.data(); cd.a_l = P.hd.al.data(); cd.a_m = P.hd.am.data();
    cd.inst_l = P.inst_l.data(); cd.inst_m = P.inst_m.data(); cd.inst_vals = r.inst_vals;
    p.N = n; p.NG = NG; p.NH = NH; p.n_rnd = (int)(18 + nv + nm); p.rnd_stride = n_rnd * 32; p.part = P.parts.data(); p.transcript_preloaded = 1;
    p.v_pts = r.cp_vpts; p.v = r.cp_v; p.s_v = r.cp_sv; p.w_l = digits; p.w_r = r.cp_wr; p.w_o = m; p.rnd = rnd + 32; p.status = status;
    std::vector<uint8_t> head(n * 256), wc(n * 64), wcv(n * (size_t)NH * 32), wrho(n * 32), wmu(n * 32), wlv(n * (size_t)NH * 32), wnv(n * (size_t)NG * 32);
    std::vector<u32> r9(4 * 72 * n), lv(6 * nv * 8 * n), nvv(4 * nm * 8 * n), lam(nl * 8 * n), muv(nm * 8 * n), coef((3 * nm + 3 * nv) * 8 * n), misc(64 * n);
    p.proof_head = head.data(); p.tstate = ts.data();
    p.ro = r9.data(); p.rl = p.ro + 72 * n; p.rr = p.ro + 144 * n; p.rs = p.ro + 216 * n;
    p.lo = lv.data(); p.ll = p.lo + nv * 8 * n; p.lr = p.lo + 2 * nv * 8 * n; p.ls = p.lo + 3 * nv * 8 * n; p.v1 = p.lo + 4 * nv * 8 * n;
    p.cl0 = p.lo + 5 * nv * 8 * n;
    p.no = nvv.data(); p.nl = p.no + nm * 8 * n; p.nr = p.no + 2 * nm * 8 * n; p.ns = p.no + 3 * nm * 8 * n;
    p.lamv = lam.data(); p.muv = muv.data(); p.coef = coef.data(); p.misc = misc.data(); p.msc = msc.data(); p.pbuf = pb.data();
    p.wn_commit = wc.data(); p.wn_c = wcv.data(); p.wn_rho = wrho.data(); p.wn_mu = wmu.data(); p.wn_l = wlv.data(); p.wn_n = wnv.data();
    p.fb = r.fb;
    std::vector<uint8_t> pr(n * rounds * 64 + 1), px(n * rounds * 64 + 1), pl(n * nl_f * 32 + 1), pn(n * nn_f * 32 + 1);
    WnlaProveWs w;
    memset(&w, 0, sizeof w);
    w.N = n; w.ng = NG; w.nh = NH; w.nl = NH; w.nn = NG; w.rounds = (int)rounds; w.nl_f = (int)nl_f; w.nn_f = (int)nn_f;
    w.transcript_preloaded = 1;
    w.commitments = p.wn_commit; w.c = p.wn_c; w.rho = p.wn_rho; w.mu = p.wn_mu; w.l_in = p.wn_l; w.n_in = p.wn_n;
    w.proof_r = pr.data(); w.proof_x = px.data(); w.proof_l = pl.data(); w.proof_n = pn.data(); w.status = status;
    std::vector<u32> vl((size_t)(NH + 1) * 8 * n), vn((size_t)(NG + 1) * 8 * n), vc((size_t)NH * 8 * n), ch((size_t)NH * 8 * n),
        cg((size_t)(NG + 1) * 8 * n), prm(24 * n), com(16 * n);
    w.tstate = p.tstate; w.vl = vl.data(); w.vn = vn.data(); w.vc = vc.data(); w.ch = ch.data(); w.cg = cg.data(); w.prm = prm.data();
    w.com = com.data(); w.msc = p.msc; w.pbuf = p.pbuf; w.fb = p.fb;
    std::vector<pt_slot> wstraus(n * 2 * BPPP_STRAUS_ENTRIES);
    w.straus = wstraus.data();
    if (g_prove_ct) {          // "ct_prover": the reciprocals' commitment and the circuit stage's commitments in the full-scan form
        if (W != 4) return -78;
        r.ct = 1; r.fb_ct = r.fb; p.ct = 1; p.fb_ct = r.fb;
    }
    auto sum = [&](const FbRanges& rg, const u32* scal, u32* out) {
        for (size_t t = 0; t < n; t++) { pt a; fb_sum_serial(a, r.fb, t, scal, rg); ws_st_pt(out, n, t, a); }
    };
    auto ssum = [&](const FbRanges& rg, const u32* scal, u32* out) {          // a sum over secret scalars (k_rprove_msm, k_cprove_msm)
        for (size_t t = 0; t < n; t++) { pt a; secret_sum(a, r.fb, r.fb_ct, r.ct, t, scal, rg); ws_st_pt(out, n, t, a); }
    };
    FbRanges rg;
    for (size_t t = 0; t < n; t++) recip_prove_stage_r1(r, t);
    recip_prove_ranges(rg, r);
    ssum(rg, r.msc, r.pbuf);
    for (size_t t = 0; t < n; t++) recip_prove_stage_r2(r, t);
    std::fill(msc.begin(), msc.end(), 0u);
    for (size_t t = 0; t < n; t++) circuit_prove_stage_a(p, t);
    cp_ranges(rg, p, false);
    for (int set = 0; set < 3; set++) ssum(rg, p.msc + (size_t)set * cp_set_words(p), p.pbuf + (size_t)set * 30 * n);
    for (size_t t = 0; t < n; t++) circuit_prove_stage_b(p, t);
    ssum(rg, p.msc, p.pbuf);
    for (size_t t = 0; t < n; t++) circuit_prove_stage_c(p, t);
    cp_ranges(rg, p, true);
    ssum(rg, p.msc, p.pbuf);
    for (size_t t = 0; t < n; t++) circuit_prove_stage_d(p, t);
    for (size_t t = 0; t < n; t++) wnla_prove_init(w, t);
    wnla_prove_msm_ranges(rg, w);
    for (int kk = 0; kk < (int)rounds; kk++) {
        for (size_t t = 0; t < n; t++) wnla_prove_round_scalars(w, t, kk);
        sum(rg, w.msc, w.pbuf);
        { FbRanges rr; wnla_prove_msm_ranges(rr, w, kk); sum(rr, w.msc + wp_set_words(w), w.pbuf + 30 * n); }
        for (size_t t = 0; t < n; t++) wnla_prove_round_fold(w, t, kk);
        if (kk == 0 && rounds > 1) sum(rg, w.msc + 2 * wp_set_words(w), w.pbuf + 60 * n);
    }
    for (size_t t = 0; t < n; t++) wnla_prove_finish(w, t);
    for (size_t t = 0; t < n; t++) tio_export(r.tio, r.base, r.tstate, n, status, t);
    for (size_t i = 0; i < n; i++) {
        uint8_t* o = proofs + i * proof_bytes;
        if (status[i] != 0) { memset(o, 0, proof_bytes); continue; }
        memcpy(o, &head[i * 256], 256); o += 256;
        memcpy(o, &pr[i * rounds * 64], rounds * 64); o += rounds * 64;
        memcpy(o, &px[i * rounds * 64], rounds * 64); o += rounds * 64;
        memcpy(o, &prr[i * 64], 64); o += 64;
        memcpy(o, &pl[i * nl_f * 32], nl_f * 32); o += nl_f * 32;
        memcpy(o, &pn[i * nn_f * 32], nn_f * 32);
    }
    return (int)proof_bytes;
}

}  // extern "C"

// ---------------------------------------------------------------- an emulated device group (CPU tier of bppp_group.hip)
// The product's sharded call is csrc/group_core.h's run_sharded() over HIP streams and RCCL.  Here the SAME template runs over
// "devices" that are host threads executing the device code above on their shard, and over an in-process stand-in for
// ncclAllReduce that behaves like the real one where it matters: it does not return until every rank of the communicator has
// entered it (or the communicator is aborted), so a rank that never comes means a hang -- reported after `timeout_ms` as
// EMUL_GROUP_HANG instead of blocking the test run forever.
#include <chrono>
#include <condition_variable>
#include <mutex>

#include "../../bp_pp_amd/csrc/group_core.h"

namespace {
const int EMUL_GROUP_HANG = -100, EMUL_ERR_NOMEM = -5, EMUL_ERR_RCCL = -6;
struct FakeComm {
    std::mutex mu;
    std::condition_variable cv;
    int world, arrived = 0, sum = 0, result = 0;
    unsigned generation = 0;
    bool aborted = false;
    explicit FakeComm(int w) : world(w) {}
    // ncclAllReduce is asynchronous: the call enqueues the rank's part on its stream and returns; what blocks is the stream
    // synchronisation afterwards.  enqueue() = the call, wait() = hipStreamSynchronize on a stream that holds an all-reduce.
    int enqueue(int value, unsigned* ticket) {
        std::lock_guard<std::mutex> lk(mu);
        if (aborted) return EMUL_ERR_RCCL;
        sum += value;
        *ticket = generation;
        if (++arrived == world) { result = sum; sum = 0; arrived = 0; generation++; cv.notify_all(); }
        return 0;
    }
    // returns 0 and the global sum in *value; EMUL_GROUP_HANG when the peers did not arrive in time; EMUL_ERR_RCCL when aborted
    int wait(unsigned ticket, int* value, int timeout_ms) {
        std::unique_lock<std::mutex> lk(mu);
        if (!cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return generation != ticket || aborted; })) return EMUL_GROUP_HANG;
        if (generation == ticket) return EMUL_ERR_RCCL;     // released by abort()
        *value = result;
        return 0;
    }
    void abort() { std::lock_guard<std::mutex> lk(mu); aborted = true; cv.notify_all(); }
};
}  // namespace

extern "C" {
// One sharded verify over G emulated devices.  kind 0: u64 proofs (928 B); kind 1: reciprocal proofs of the given shape.
// fail_rank >= 0: that rank's prepare fails with NOMEM before it does anything; fail_collective_rank >= 0: that rank's all-reduce
// call fails.  use_vote = 0 reproduces round 2's control flow (a failing rank simply returns; the others go on into the
// collective) so that the test can show what the vote prevents.
// Returns the call's code; reject_out[r] = the count rank r ended with (global when the collective ran); aborted_out = 1 when the
// communicator was aborted.
int emul_group_verify(int kind, int G, int fail_rank, int fail_collective_rank, int use_vote, int timeout_ms, const uint8_t* table, int W,
                      int NG, int NH, int nd, int np, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                      const uint8_t* proofs, int rounds, int nl, int nn, uint8_t* accept, int32_t* status, int* reject_out, int* aborted_out) {
    const size_t proof_bytes = kind == 0 ? 928 : 64 * (5 + 2 * (size_t)rounds) + 32 * ((size_t)nl + nn);
    FakeComm comm(G);
    std::vector<int> rej(G, 0);
    auto range = [&](int r, size_t& lo, size_t& m) {
        const size_t q = n / G, rem = n % G;
        auto at = [&](size_t k) { return q * k + (rem * k) / G; };
        lo = at(r); m = at(r + 1) - lo;
    };
    auto prepare = [&](int r) -> int {
        if (r == fail_rank) return EMUL_ERR_NOMEM;
        if (r == fail_rank - 100) throw std::bad_alloc();      // fail_rank = 100 + r: rank r's prepare THROWS instead of returning a code
        size_t lo, m;
        range(r, lo, m);
        if (m) {
            int rc = kind == 0 ? emul_u64_verify_batch(table, W, label, label_len, m, commitments + lo * 64, proofs + lo * proof_bytes, accept + lo,
                                                       status + lo, nullptr)
                               : emul_recip_verify(table, W, NG, NH, nd, np, label, label_len, m, commitments + lo * 64, proofs + lo * proof_bytes,
                                                   rounds, nl, nn, accept + lo, status + lo);
            if (rc != 0) return rc;
        }
        int c = 0;
        for (size_t i = 0; i < m; i++) c += accept[lo + i] == 0;
        rej[r] = c;
        return 0;
    };
    std::vector<unsigned> ticket(G, 0);
    std::vector<char> enqueued(G, 0);
    auto collective = [&](int r) -> int {
        if (r == fail_collective_rank) return EMUL_ERR_RCCL;
        int rc = comm.enqueue(rej[r], &ticket[r]);
        if (rc == 0) enqueued[r] = 1;
        return rc;
    };
    // "synchronize the stream": blocks while an enqueued all-reduce is incomplete
    auto sync = [&](int r) -> int { return enqueued[r] ? comm.wait(ticket[r], &rej[r], timeout_ms) : 0; };
    int code = 0;
    if (use_vote) {
        auto res = bppp_host::run_sharded(G, [](int) {}, prepare, collective, sync, sync, [&](int) { comm.abort(); }, [](int) {},
                                          []() { return std::string(); });
        code = res.code;
    } else {
        std::vector<int> rcs(G, 0);
        std::vector<std::thread> th;
        for (int r = 0; r < G; r++)
            th.emplace_back([&, r]() {
                rcs[r] = prepare(r);
                if (rcs[r] == 0) rcs[r] = collective(r);
                if (rcs[r] == 0) rcs[r] = sync(r);
            });
        for (auto& t : th) t.join();
        for (int r = 0; r < G; r++)
            if (rcs[r] == EMUL_GROUP_HANG) code = EMUL_GROUP_HANG;
        for (int r = 0; r < G && code == 0; r++) code = rcs[r];
    }
    for (int r = 0; r < G; r++) reject_out[r] = rej[r];
    *aborted_out = comm.aborted ? 1 : 0;
    return code;
}
// A group whose rank threads from `start_fails_at` on cannot be started (std::thread throwing): the running ranks must come back from
// their vote with the failure instead of waiting for ever for parties that do not exist -- also when SEVERAL ranks are missing.
// Returns the call's code; *ran = how many ranks got as far as their first phase.
int emul_group_missing_ranks(int G, int start_fails_at, int* ran) {
    std::atomic<int> n{0};
    auto prepare = [&](int) -> int { n++; return 0; };
    auto nop = [](int) -> int { return 0; };
    auto res = bppp_host::run_sharded(G, [](int) {}, prepare, nop, nop, nop, [](int) {}, [](int) {}, []() { return std::string(); }, start_fails_at);
    *ran = n.load();
    return res.code;
}
// The sharded prover over G emulated devices (bppp_u64_prove_batch_sharded): rank r proves rows [lo, hi) of the batch; there is no
// exchange step, so the collective of run_sharded is a no-op and the vote is all the ranks share.  fail_rank as above.
int emul_group_prove(int G, int fail_rank, const uint8_t* table, int W, const uint8_t* label, size_t label_len, size_t n, const uint64_t* x,
                     const uint8_t* s, const uint8_t* rnd, uint8_t* proofs, uint8_t* V, int32_t* status) {
    auto range = [&](int r, size_t& lo, size_t& m) {
        const size_t q = n / G, rem = n % G;
        auto at = [&](size_t k) { return q * k + (rem * k) / G; };
        lo = at(r); m = at(r + 1) - lo;
    };
    auto prepare = [&](int r) -> int {
        if (r == fail_rank) return EMUL_ERR_NOMEM;
        size_t lo, m;
        range(r, lo, m);
        return m ? emul_u64_prove_batch(table, W, label, label_len, m, x + lo, s + 32 * lo, rnd + 52 * 32 * lo, proofs + 928 * lo, V + 64 * lo,
                                        status + lo)
                 : 0;
    };
    auto nop = [](int) -> int { return 0; };
    auto res = bppp_host::run_sharded(G, [](int) {}, prepare, nop, nop, nop, [](int) {}, [](int) {}, []() { return std::string(); });
    return res.code;
}
}

// ------------------------------------------------------------------ the single-proof front end (csrc/coalesce_core.h) over the emulator
// The same Coalescer template libbppp_hip.so instantiates (bppp_coalesce.hip), with malloc staging and the host build of the device
// code as the batched call: `threads` callers submit the requests i = t, t + threads, ... one after the other, exactly like the
// reference's one-proof-per-call pattern from many threads.  kind 0: verify (rows: commitment, proof, transcript -> accept, status,
// transcript out); kind 1: prove (rows: x, s, rnd, transcript -> proof, commitment, status, transcript out).
//   fail_batch >= 0         the fail_batch-th batched call (in start order) fails as a whole with NOMEM: its callers, and only they,
//                           get -5 and their outputs stay untouched
//   shutdown_after_ms >= 0  the main thread calls shutdown() that long after the callers started, with callers still submitting:
//                           every caller returns -- 0 with its proper outputs (it was drained) or -7 (it came too late)
//   run_delay_ms            extra time per batched call (stands for the GPU's latency, so that requests gather behind it)
#include "../../bp_pp_amd/csrc/coalesce_core.h"
namespace {
const int EMUL_ERR_CLOSED = -7;
struct EmulFront {
    const uint8_t* table; int W; int kind; int fail_batch; int run_delay_ms;
    std::atomic<int> batch_no{0};
    std::atomic<int> live_staging{0};
    std::mutex sizes_mu;
    std::vector<size_t> batch_sizes;
    void* alloc_staging(size_t bytes) { live_staging++; return std::malloc(bytes ? bytes : 1); }
    void free_staging(void* p) { live_staging--; std::free(p); }
    bool start_lane(int) { return true; }
    std::string last_error() { return "emulated failure"; }
    void set_last_error(const std::string&) {}
    void stop_lane(int) {}
    int run(int, size_t n, uint8_t* const in[], uint8_t* const out[]) {
        const int no = batch_no.fetch_add(1);
        { std::lock_guard<std::mutex> lk(sizes_mu); batch_sizes.push_back(n); }
        if (run_delay_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(run_delay_ms));
        if (no == fail_batch) return EMUL_ERR_NOMEM;
        if (kind == 0)
            return emul_u64_verify_batch_transcript(table, W, n, in[2], n, in[0], in[1], out[0], (int32_t*)out[1], out[2]);
        return emul_u64_prove_batch_transcript(table, W, n, in[3], n, (const uint64_t*)in[0], in[1], in[2], out[0], out[1], (int32_t*)out[2], out[3]);
    }
};
}  // namespace
extern "C" int emul_coalesce_run(int kind, const uint8_t* table, int W, int threads, size_t nreq, const uint8_t* const* in_arrays,
                                 uint8_t* const* out_arrays, int* rcs, size_t max_batch, long wait_us, int lanes, int fail_batch,
                                 int shutdown_after_ms, int run_delay_ms, uint64_t stats_out[5], size_t* batch_sizes_out, size_t batch_sizes_cap,
                                 int* live_staging_after) {
    EmulFront be{table, W, kind, fail_batch, run_delay_ms};
    bppp_host::CoalesceShape sh;
    if (kind == 0) {
        sh.n_in = 3; sh.in_stride[0] = 64; sh.in_stride[1] = 928; sh.in_stride[2] = 203;
        sh.n_out = 3; sh.out_stride[0] = 1; sh.out_stride[1] = 4; sh.out_stride[2] = 203;
    } else {
        sh.n_in = 4; sh.in_stride[0] = 8; sh.in_stride[1] = 32; sh.in_stride[2] = 52 * 32; sh.in_stride[3] = 203;
        sh.n_out = 4; sh.out_stride[0] = 928; sh.out_stride[1] = 64; sh.out_stride[2] = 4; sh.out_stride[3] = 203;
    }
    {
        bppp_host::Coalescer<EmulFront> co(&be, sh, max_batch, wait_us, lanes, EMUL_ERR_CLOSED, EMUL_ERR_NOMEM);
        int rc = co.start();
        if (rc != 0) return rc;
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++)
            th.emplace_back([&, t]() {
                for (size_t i = (size_t)t; i < nreq; i += (size_t)threads) {
                    const void* in[4];
                    void* out[4];
                    for (int k = 0; k < sh.n_in; k++) in[k] = in_arrays[k] + i * sh.in_stride[k];
                    for (int k = 0; k < sh.n_out; k++) out[k] = out_arrays[k] + i * sh.out_stride[k];
                    rcs[i] = co.submit(in, out);
                }
            });
        if (shutdown_after_ms >= 0) {
            std::this_thread::sleep_for(std::chrono::milliseconds(shutdown_after_ms));
            co.shutdown();
        }
        for (auto& t : th) t.join();
        const bppp_host::CoalesceStats s = co.stats();
        stats_out[0] = s.requests; stats_out[1] = s.batches; stats_out[2] = s.largest_batch; stats_out[3] = s.sealed_full; stats_out[4] = s.sealed_deadline;
        co.shutdown();
        // after shutdown a submission is refused, not queued
        const void* in[4];
        void* out[4];
        for (int k = 0; k < sh.n_in; k++) in[k] = in_arrays[k];
        for (int k = 0; k < sh.n_out; k++) out[k] = nullptr;
        if (co.submit(in, out) != EMUL_ERR_CLOSED) return -99;
    }
    for (size_t i = 0; i < be.batch_sizes.size() && i < batch_sizes_cap; i++) batch_sizes_out[i] = be.batch_sizes[i];
    *live_staging_after = be.live_staging.load();
    return (int)be.batch_sizes.size();
}
