/*
 * CPU oracle (plain C) for the Bulletproofs++ u64 range-proof hot path.
 *
 * TEST INFRASTRUCTURE ONLY: nothing in the product path (bp_pp_amd/, libbppp_hip.so)
 * may link, call or execute this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / timed CPU baseline.
 *
 * "Reference-shaped" restatement of distributed-lab/bp-pp 0.1.1 (citations are
 * file:line under /root/reference/src): one constant-structure scalar multiplication
 * per MSM term (util.rs:46-60), per-round generator folding (wnla.rs:96-97), dense
 * coefficient matrices (circuit.rs:584-653, reciprocal.rs:150-214), 256 recomputed
 * scalar inversions (reciprocal.rs:179-183) -- no batching, no shared-doubling, no
 * precomputed tables in verify/prove.
 *
 * The arithmetic the reference takes from k256 0.13.3 / merlin 3.0.0 (not under
 * /root/reference; Cargo.lock:411,453) is restated from the public specifications:
 * secp256k1, complete projective formulas (Renes-Costello-Batina 2016, alg. 7/9 for a=0),
 * SEC1, Keccak-f[1600], STROBE-128, Merlin.
 *
 * PARITY STATUS: parity unpinned against the reference itself (it has no golden vectors
 * and cannot be built here).  Pinned to: secp256k1 + Merlin public known answers, OpenSSL's
 * secp256k1 for the curve layer (tests/golden/openssl_secp256k1.json, tests/test_openssl_vectors.py) and,
 * byte for byte, to the independent Python big-int oracle (oracle/bppp_oracle.py) on the
 * committed fixtures under tests/golden/ (tests/test_oracle_c.py).
 *
 * Wire formats (same as include/bppp.h): point = 64 B affine big-endian x||y, identity =
 * 64 zero bytes; scalar = 32 B big-endian canonical; u64 proof = 13 points
 * (c_l, c_r, c_o, c_s, r[0..3], x[0..3], reciprocal r) + 3 scalars (l0, l1, n0) = 928 B.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe; /* mod p, canonical, little-endian limbs */
typedef struct { uint64_t v[4]; } sc; /* mod n, canonical */
typedef struct { fe X, Y, Z; } pt;    /* homogeneous projective; identity = (0:1:0) */

#define ORACLE_ERR_ENCODING (-1)
#define ORACLE_ERR_DEGENERATE (-2)

/* ------------------------------------------------------------------ 256-bit helpers */
static const uint64_t P_[4] = {0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL};
static const uint64_t N_[4] = {0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL};
static const uint64_t PC = 0x1000003D1ULL;                                                  /* 2^256 - p */
static const uint64_t ND[3] = {0x402DA1732FC9BEBFULL, 0x4551231950B75FC4ULL, 0x1ULL};      /* 2^256 - n */

static int ge256(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static uint64_t add256(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static uint64_t sub256(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - br;
        r[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}
static int is_zero256(const uint64_t a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
static void mul256(uint64_t r[8], const uint64_t a[4], const uint64_t b[4]) {
    memset(r, 0, 64);
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a[i] * b[j] + r[i + j]; r[i + j] = (uint64_t)c; c >>= 64; }
        r[i + 4] = (uint64_t)c;
    }
}
static void be32_to_limbs(uint64_t r[4], const uint8_t b[32]) {
    for (int i = 0; i < 4; i++) {
        uint64_t w = 0;
        for (int j = 0; j < 8; j++) w = (w << 8) | b[(3 - i) * 8 + j];
        r[i] = w;
    }
}
static void limbs_to_be32(uint8_t b[32], const uint64_t a[4]) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) b[(3 - i) * 8 + j] = (uint8_t)(a[i] >> (56 - 8 * j));
}

/* ------------------------------------------------------------------ Fp */
static const fe FE_ZERO = {{0, 0, 0, 0}};
static const fe FE_ONE = {{1, 0, 0, 0}};
static void fe_add(fe* r, const fe* a, const fe* b) {
    uint64_t c = add256(r->v, a->v, b->v);
    if (c || ge256(r->v, P_)) sub256(r->v, r->v, P_);
}
static void fe_sub(fe* r, const fe* a, const fe* b) {
    if (sub256(r->v, a->v, b->v)) add256(r->v, r->v, P_);
}
static void fe_neg(fe* r, const fe* a) { fe_sub(r, &FE_ZERO, a); }
/* reduce a 512-bit product t[0..7] mod p = 2^256 - PC */
static inline void fe_reduce512(fe* r, const uint64_t t[8]) {
    uint64_t s0, s1, s2, s3, s4;
    u128 c;
    c = (u128)t[4] * PC + t[0]; s0 = (uint64_t)c; c >>= 64;
    c += (u128)t[5] * PC + t[1]; s1 = (uint64_t)c; c >>= 64;
    c += (u128)t[6] * PC + t[2]; s2 = (uint64_t)c; c >>= 64;
    c += (u128)t[7] * PC + t[3]; s3 = (uint64_t)c; c >>= 64;
    s4 = (uint64_t)c; /* < 2^34 */
    c = (u128)s4 * PC + s0; uint64_t o0 = (uint64_t)c; c >>= 64;
    c += s1; uint64_t o1 = (uint64_t)c; c >>= 64;
    c += s2; uint64_t o2 = (uint64_t)c; c >>= 64;
    c += s3; uint64_t o3 = (uint64_t)c; c >>= 64;
    if (c) { /* wrapped past 2^256 once more: add PC (cannot wrap again) */
        c = (u128)o0 + PC; o0 = (uint64_t)c; c >>= 64;
        c += o1; o1 = (uint64_t)c; c >>= 64;
        c += o2; o2 = (uint64_t)c; c >>= 64;
        c += o3; o3 = (uint64_t)c;
    }
    uint64_t o[4] = {o0, o1, o2, o3};
    if (ge256(o, P_)) sub256(o, o, P_);
    memcpy(r->v, o, 32);
}
static void fe_mul(fe* r, const fe* a, const fe* b) {
    uint64_t t[8];
    const uint64_t *x = a->v, *y = b->v;
    u128 c;
    c = (u128)x[0] * y[0]; t[0] = (uint64_t)c; c >>= 64;
    c += (u128)x[0] * y[1]; t[1] = (uint64_t)c; c >>= 64;
    c += (u128)x[0] * y[2]; t[2] = (uint64_t)c; c >>= 64;
    c += (u128)x[0] * y[3]; t[3] = (uint64_t)c; t[4] = (uint64_t)(c >> 64);
    for (int i = 1; i < 4; i++) {
        c = (u128)x[i] * y[0] + t[i]; t[i] = (uint64_t)c; c >>= 64;
        c += (u128)x[i] * y[1] + t[i + 1]; t[i + 1] = (uint64_t)c; c >>= 64;
        c += (u128)x[i] * y[2] + t[i + 2]; t[i + 2] = (uint64_t)c; c >>= 64;
        c += (u128)x[i] * y[3] + t[i + 3]; t[i + 3] = (uint64_t)c; t[i + 4] = (uint64_t)(c >> 64);
    }
    fe_reduce512(r, t);
}
static void fe_sqr(fe* r, const fe* a) { fe_mul(r, a, a); }
static void fe_mul_small(fe* r, const fe* a, uint64_t k) { /* k < 2^32 */
    u128 c;
    uint64_t o[4];
    c = (u128)a->v[0] * k; o[0] = (uint64_t)c; c >>= 64;
    c += (u128)a->v[1] * k; o[1] = (uint64_t)c; c >>= 64;
    c += (u128)a->v[2] * k; o[2] = (uint64_t)c; c >>= 64;
    c += (u128)a->v[3] * k; o[3] = (uint64_t)c; c >>= 64;
    /* fold the top limb (< 2^32): top * PC < 2^66 */
    c = (u128)(uint64_t)c * PC;
    u128 d = (u128)o[0] + (uint64_t)c; o[0] = (uint64_t)d; d >>= 64;
    d += (u128)o[1] + (uint64_t)(c >> 64); o[1] = (uint64_t)d; d >>= 64;
    d += o[2]; o[2] = (uint64_t)d; d >>= 64;
    d += o[3]; o[3] = (uint64_t)d; d >>= 64;
    if (d) {
        d = (u128)o[0] + PC; o[0] = (uint64_t)d; d >>= 64;
        d += o[1]; o[1] = (uint64_t)d; d >>= 64;
        d += o[2]; o[2] = (uint64_t)d; d >>= 64;
        d += o[3]; o[3] = (uint64_t)d;
    }
    if (ge256(o, P_)) sub256(o, o, P_);
    memcpy(r->v, o, 32);
}
static void fe_pow(fe* r, const fe* a, const uint64_t e[4]) {
    fe acc = FE_ONE, base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) fe_mul(&acc, &acc, &base);
        fe_sqr(&base, &base);
    }
    *r = acc;
}
static void fe_inv(fe* r, const fe* a) { /* a^(p-2); 0 -> 0 */
    uint64_t e[4] = {P_[0] - 2, P_[1], P_[2], P_[3]};
    fe_pow(r, a, e);
}
static int fe_is_zero(const fe* a) { return is_zero256(a->v); }
static int fe_eq(const fe* a, const fe* b) { return memcmp(a->v, b->v, 32) == 0; }

/* ------------------------------------------------------------------ Fn (k256 Scalar) */
static const sc SC_ZERO = {{0, 0, 0, 0}};
static const sc SC_ONE = {{1, 0, 0, 0}};
static void sc_add(sc* r, const sc* a, const sc* b) {
    uint64_t c = add256(r->v, a->v, b->v);
    if (c || ge256(r->v, N_)) sub256(r->v, r->v, N_);
}
static void sc_sub(sc* r, const sc* a, const sc* b) {
    if (sub256(r->v, a->v, b->v)) add256(r->v, r->v, N_);
}
static void sc_reduce512(sc* r, const uint64_t t_in[8]) {
    /* x = hi*2^256 + lo == hi*ND + lo (mod n); repeat until hi == 0 */
    uint64_t t[8];
    memcpy(t, t_in, 64);
    while (t[4] | t[5] | t[6] | t[7]) {
        uint64_t m[8] = {0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 3; j++) { c += (u128)t[4 + i] * ND[j] + m[i + j]; m[i + j] = (uint64_t)c; c >>= 64; }
            int k = i + 3;
            while (c) { c += m[k]; m[k] = (uint64_t)c; c >>= 64; k++; }
        }
        u128 c = 0;
        for (int i = 0; i < 8; i++) { c += (u128)m[i] + (i < 4 ? t[i] : 0); t[i] = (uint64_t)c; c >>= 64; }
    }
    while (ge256(t, N_)) sub256(t, t, N_);
    memcpy(r->v, t, 32);
}
static void sc_mul(sc* r, const sc* a, const sc* b) {
    uint64_t t[8];
    mul256(t, a->v, b->v);
    sc_reduce512(r, t);
}
static void sc_from_u64(sc* r, uint64_t x) { r->v[0] = x; r->v[1] = r->v[2] = r->v[3] = 0; }
static int sc_is_zero(const sc* a) { return is_zero256(a->v); }
static void sc_pow_u64(sc* r, const sc* a, uint64_t e) { /* util.rs:97-99 pow_vartime([n as u64]) */
    sc acc = SC_ONE, base = *a;
    while (e) {
        if (e & 1) sc_mul(&acc, &acc, &base);
        sc_mul(&base, &base, &base);
        e >>= 1;
    }
    *r = acc;
}
static int sc_inv_fermat(sc* r, const sc* a) { /* k256 Scalar::invert (constant-time shape): a^(n-2) */
    if (sc_is_zero(a)) return 0;
    uint64_t e[4] = {N_[0] - 2, N_[1], N_[2], N_[3]};
    sc acc = SC_ONE, base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) sc_mul(&acc, &acc, &base);
        sc_mul(&base, &base, &base);
    }
    *r = acc;
    return 1;
}
static void half_mod_n(uint64_t x[4]) { /* x/2 mod n for canonical x */
    uint64_t carry = 0;
    if (x[0] & 1) carry = add256(x, x, N_);
    for (int i = 0; i < 3; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 63);
    x[3] = (x[3] >> 1) | (carry << 63);
}
static int sc_inv_vartime(sc* r, const sc* a) { /* k256 Scalar::invert_vartime: same value as invert; binary xgcd */
    if (sc_is_zero(a)) return 0;
    uint64_t u[4], v[4], x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
    memcpy(u, a->v, 32);
    memcpy(v, N_, 32);
    const uint64_t one[4] = {1, 0, 0, 0};
    while (memcmp(u, one, 32) != 0 && memcmp(v, one, 32) != 0) {
        while (!(u[0] & 1)) {
            for (int i = 0; i < 3; i++) u[i] = (u[i] >> 1) | (u[i + 1] << 63);
            u[3] >>= 1;
            half_mod_n(x1);
        }
        while (!(v[0] & 1)) {
            for (int i = 0; i < 3; i++) v[i] = (v[i] >> 1) | (v[i + 1] << 63);
            v[3] >>= 1;
            half_mod_n(x2);
        }
        if (ge256(u, v)) {
            sub256(u, u, v);
            if (sub256(x1, x1, x2)) add256(x1, x1, N_);
        } else {
            sub256(v, v, u);
            if (sub256(x2, x2, x1)) add256(x2, x2, N_);
        }
    }
    memcpy(r->v, memcmp(u, one, 32) == 0 ? x1 : x2, 32);
    return 1;
}
static void sc_minus(sc* r, const sc* v) { /* util.rs:153-155: v * (0 - 1) */
    sc m1;
    sc_sub(&m1, &SC_ZERO, &SC_ONE);
    sc_mul(r, v, &m1);
}
static int sc_from_be(sc* r, const uint8_t b[32]) { /* Scalar::from_repr: None if >= n */
    be32_to_limbs(r->v, b);
    return !ge256(r->v, N_);
}
static void sc_to_be(uint8_t b[32], const sc* a) { limbs_to_be32(b, a->v); }

/* ------------------------------------------------------------------ points (RCB16 complete formulas, a = 0, b3 = 21) */
static const pt PT_IDENTITY = {{{0, 0, 0, 0}}, {{1, 0, 0, 0}}, {{0, 0, 0, 0}}};

/* Synthetic-workload generator support ("trapdoor" prover, bench/test setup only, never timed as the baseline):
 * when the generators are k_i*G with KNOWN k_i, a group element can be carried as its discrete log w.r.t. G
 * (stored in pt.X, Y = 1, Z = 0) and every group operation becomes one Fn operation.  The prover code below
 * runs unchanged; only app_point / serialisation map a discrete log back to the real point (one fixed-base
 * multiple of G).  The resulting proofs are byte-identical to the honest prover's (tests/test_oracle_c.py). */
static __thread int DLOG_MODE = 0;
static void dl_get(sc* s, const pt* p) { memcpy(s, &p->X, 32); }
static void dl_set(pt* p, const sc* s) { memcpy(&p->X, s, 32); p->Y = FE_ONE; p->Z = FE_ZERO; }
static void dl_to_real(pt* r, const pt* p);

static void pt_add(pt* r, const pt* p, const pt* q) { /* RCB16 algorithm 7 */
    if (DLOG_MODE) { sc a, b; dl_get(&a, p); dl_get(&b, q); sc_add(&a, &a, &b); dl_set(r, &a); return; }
    fe t0, t1, t2, t3, t4, X3, Y3, Z3;
    fe_mul(&t0, &p->X, &q->X);
    fe_mul(&t1, &p->Y, &q->Y);
    fe_mul(&t2, &p->Z, &q->Z);
    fe_add(&t3, &p->X, &p->Y);
    fe_add(&t4, &q->X, &q->Y);
    fe_mul(&t3, &t3, &t4);
    fe_add(&t4, &t0, &t1);
    fe_sub(&t3, &t3, &t4);
    fe_add(&t4, &p->Y, &p->Z);
    fe_add(&X3, &q->Y, &q->Z);
    fe_mul(&t4, &t4, &X3);
    fe_add(&X3, &t1, &t2);
    fe_sub(&t4, &t4, &X3);
    fe_add(&X3, &p->X, &p->Z);
    fe_add(&Y3, &q->X, &q->Z);
    fe_mul(&X3, &X3, &Y3);
    fe_add(&Y3, &t0, &t2);
    fe_sub(&Y3, &X3, &Y3);
    fe_add(&X3, &t0, &t0);
    fe_add(&t0, &X3, &t0);
    fe_mul_small(&t2, &t2, 21);
    fe_add(&Z3, &t1, &t2);
    fe_sub(&t1, &t1, &t2);
    fe_mul_small(&Y3, &Y3, 21);
    fe_mul(&X3, &t4, &Y3);
    fe_mul(&t2, &t3, &t1);
    fe_sub(&X3, &t2, &X3);
    fe_mul(&Y3, &Y3, &t0);
    fe_mul(&t1, &t1, &Z3);
    fe_add(&Y3, &t1, &Y3);
    fe_mul(&t0, &t0, &t3);
    fe_mul(&Z3, &Z3, &t4);
    fe_add(&Z3, &Z3, &t0);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}
static void pt_dbl(pt* r, const pt* p) { /* RCB16 algorithm 9 */
    fe t0, t1, t2, X3, Y3, Z3;
    fe_sqr(&t0, &p->Y);
    fe_add(&Z3, &t0, &t0);
    fe_add(&Z3, &Z3, &Z3);
    fe_add(&Z3, &Z3, &Z3);
    fe_mul(&t1, &p->Y, &p->Z);
    fe_sqr(&t2, &p->Z);
    fe_mul_small(&t2, &t2, 21);
    fe_mul(&X3, &t2, &Z3);
    fe_add(&Y3, &t0, &t2);
    fe_mul(&Z3, &t1, &Z3);
    fe_add(&t1, &t2, &t2);
    fe_add(&t2, &t1, &t2);
    fe_sub(&t0, &t0, &t2);
    fe_mul(&Y3, &t0, &Y3);
    fe_add(&Y3, &X3, &Y3);
    fe_mul(&t1, &p->X, &p->Y);
    fe_mul(&X3, &t0, &t1);
    fe_add(&X3, &X3, &X3);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}
static void pt_neg(pt* r, const pt* p) {
    if (DLOG_MODE) { sc a; dl_get(&a, p); sc_sub(&a, &SC_ZERO, &a); dl_set(r, &a); return; }
    r->X = p->X; fe_neg(&r->Y, &p->Y); r->Z = p->Z;
}
static void pt_sub(pt* r, const pt* p, const pt* q) { pt nq; pt_neg(&nq, q); pt_add(r, p, &nq); }
static int pt_is_identity(const pt* p) { return DLOG_MODE ? fe_is_zero(&p->X) : fe_is_zero(&p->Z); }
static int pt_eq(const pt* a, const pt* b) { /* projective-class equality (k256 ProjectivePoint::eq) */
    if (DLOG_MODE) return fe_eq(&a->X, &b->X);
    fe l, r;
    fe_mul(&l, &a->X, &b->Z); fe_mul(&r, &b->X, &a->Z);
    if (!fe_eq(&l, &r)) return 0;
    fe_mul(&l, &a->Y, &b->Z); fe_mul(&r, &b->Y, &a->Z);
    return fe_eq(&l, &r);
}
/* k256 `ProjectivePoint * Scalar`: constant-structure 4-bit fixed window, 256 doublings + 64 table additions */
static void pt_mul(pt* r, const pt* p, const sc* k) {
    if (DLOG_MODE) { sc a; dl_get(&a, p); sc_mul(&a, &a, k); dl_set(r, &a); return; }
    pt tbl[16];
    tbl[0] = PT_IDENTITY;
    tbl[1] = *p;
    for (int i = 2; i < 16; i++) {
        if (i & 1) pt_add(&tbl[i], &tbl[i - 1], p);
        else pt_dbl(&tbl[i], &tbl[i / 2]);
    }
    pt acc = PT_IDENTITY;
    for (int w = 63; w >= 0; w--) {
        for (int d = 0; d < 4; d++) pt_dbl(&acc, &acc);
        unsigned digit = (unsigned)(k->v[w / 16] >> ((w % 16) * 4)) & 15;
        pt_add(&acc, &acc, &tbl[digit]);
    }
    *r = acc;
}
static void pt_to_affine(fe* x, fe* y, const pt* p) { /* identity -> (0,0) */
    fe zi;
    fe_inv(&zi, &p->Z);
    fe_mul(x, &p->X, &zi);
    fe_mul(y, &p->Y, &zi);
}
static void pt_to_xy64(uint8_t out[64], const pt* p_in) {
    pt real;
    const pt* p = p_in;
    if (DLOG_MODE) { dl_to_real(&real, p_in); p = &real; }
    if (fe_is_zero(&p->Z)) { memset(out, 0, 64); return; }
    fe x, y;
    pt_to_affine(&x, &y, p);
    limbs_to_be32(out, x.v);
    limbs_to_be32(out + 32, y.v);
}
static int pt_from_xy64(pt* r, const uint8_t in[64]) {
    int allz = 1;
    for (int i = 0; i < 64; i++) if (in[i]) { allz = 0; break; }
    if (allz) { *r = PT_IDENTITY; return 1; }
    be32_to_limbs(r->X.v, in);
    be32_to_limbs(r->Y.v, in + 32);
    if (ge256(r->X.v, P_) || ge256(r->Y.v, P_)) return 0;
    r->Z = FE_ONE;
    fe y2, x3, seven = {{7, 0, 0, 0}};
    fe_sqr(&y2, &r->Y);
    fe_sqr(&x3, &r->X);
    fe_mul(&x3, &x3, &r->X);
    fe_add(&x3, &x3, &seven);
    return fe_eq(&y2, &x3);
}
static void pt_to_sec1(uint8_t out[33], const pt* p_in) { /* k256 GroupEncoding::to_bytes; identity -> 33 zero bytes */
    pt real;
    const pt* p = p_in;
    if (DLOG_MODE) { dl_to_real(&real, p_in); p = &real; }
    if (fe_is_zero(&p->Z)) { memset(out, 0, 33); return; }
    fe x, y;
    pt_to_affine(&x, &y, p);
    out[0] = 2 + (uint8_t)(y.v[0] & 1);
    limbs_to_be32(out + 1, x.v);
}

/* fixed-base table for G (dlog mode only): GTBL[w][d-1] = d * 2^(8w) * G, d = 1..255 */
static pt GTBL[32][255];
static pthread_once_t GTBL_ONCE = PTHREAD_ONCE_INIT;
static void gtbl_build(void) {
    static const uint64_t GX[4] = {0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL, 0x79BE667EF9DCBBACULL};
    static const uint64_t GY[4] = {0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL, 0x5DA4FBFC0E1108A8ULL, 0x483ADA7726A3C465ULL};
    int saved = DLOG_MODE;
    DLOG_MODE = 0;
    pt base;
    memcpy(base.X.v, GX, 32); memcpy(base.Y.v, GY, 32); base.Z = FE_ONE;
    for (int w = 0; w < 32; w++) {
        GTBL[w][0] = base;
        for (int d = 1; d < 255; d++) pt_add(&GTBL[w][d], &GTBL[w][d - 1], &base);
        for (int i = 0; i < 8; i++) pt_dbl(&base, &base);
    }
    DLOG_MODE = saved;
}
static void dl_to_real(pt* r, const pt* p) {
    pthread_once(&GTBL_ONCE, gtbl_build);
    sc k;
    dl_get(&k, p);
    int saved = DLOG_MODE;
    DLOG_MODE = 0;
    pt acc = PT_IDENTITY;
    for (int w = 0; w < 32; w++) {
        unsigned d = (unsigned)(k.v[w / 8] >> ((w % 8) * 8)) & 255;
        if (d) pt_add(&acc, &acc, &GTBL[w][d - 1]);
    }
    DLOG_MODE = saved;
    *r = acc;
}

/* ------------------------------------------------------------------ Keccak-f[1600] / STROBE-128 / Merlin */
static const uint64_t KRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000AULL, 0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
static uint64_t rol64(uint64_t v, int r) { return r ? (v << r) | (v >> (64 - r)) : v; }
static void keccak_f1600(uint64_t a[25]) {
    for (int rnd = 0; rnd < 24; rnd++) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(a[x + 5 * y], KROT[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= KRC[rnd];
    }
}
#define STROBE_R 166
typedef struct {
    union { uint64_t q[25]; uint8_t b[200]; } st; /* little-endian host assumed (x86-64) */
    uint8_t pos, pos_begin, cur_flags;
} strobe;
static void strobe_run_f(strobe* s) {
    s->st.b[s->pos] ^= s->pos_begin;
    s->st.b[s->pos + 1] ^= 0x04;
    s->st.b[STROBE_R + 1] ^= 0x80;
    keccak_f1600(s->st.q);
    s->pos = 0;
    s->pos_begin = 0;
}
static void strobe_absorb(strobe* s, const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) {
        s->st.b[s->pos++] ^= d[i];
        if (s->pos == STROBE_R) strobe_run_f(s);
    }
}
static void strobe_squeeze(strobe* s, uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) {
        d[i] = s->st.b[s->pos];
        s->st.b[s->pos++] = 0;
        if (s->pos == STROBE_R) strobe_run_f(s);
    }
}
static void strobe_begin_op(strobe* s, uint8_t flags, int more) {
    if (more) return;
    uint8_t hdr[2] = {s->pos_begin, flags};
    s->pos_begin = s->pos + 1;
    s->cur_flags = flags;
    strobe_absorb(s, hdr, 2);
    if ((flags & (4 | 32)) && s->pos != 0) strobe_run_f(s);
}
static void strobe_meta_ad(strobe* s, const uint8_t* d, size_t n, int more) { strobe_begin_op(s, 16 | 2, more); strobe_absorb(s, d, n); }
static void strobe_ad(strobe* s, const uint8_t* d, size_t n, int more) { strobe_begin_op(s, 2, more); strobe_absorb(s, d, n); }
static void strobe_prf(strobe* s, uint8_t* d, size_t n) { strobe_begin_op(s, 1 | 2 | 4, 0); strobe_squeeze(s, d, n); }
static void strobe_init(strobe* s, const char* proto) {
    memset(s, 0, sizeof *s);
    const uint8_t hdr[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
    memcpy(s->st.b, hdr, 6);
    memcpy(s->st.b + 6, "STROBEv1.0.2", 12);
    keccak_f1600(s->st.q);
    strobe_meta_ad(s, (const uint8_t*)proto, strlen(proto), 0);
}
typedef strobe transcript;
static void t_append(transcript* t, const char* label, const uint8_t* m, uint32_t n) {
    uint8_t le[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    strobe_meta_ad(t, (const uint8_t*)label, strlen(label), 0);
    strobe_meta_ad(t, le, 4, 1);
    strobe_ad(t, m, n, 0);
}
static void t_new(transcript* t, const uint8_t* label, size_t n) {
    strobe_init(t, "Merlin v1.0");
    t_append(t, "dom-sep", label, (uint32_t)n);
}
static void t_append_u64(transcript* t, const char* label, uint64_t x) {
    uint8_t le[8];
    for (int i = 0; i < 8; i++) le[i] = (uint8_t)(x >> (8 * i));
    t_append(t, label, le, 8);
}
static void t_challenge_bytes(transcript* t, const char* label, uint8_t* out, uint32_t n) {
    uint8_t le[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    strobe_meta_ad(t, (const uint8_t*)label, strlen(label), 0);
    strobe_meta_ad(t, le, 4, 1);
    strobe_prf(t, out, n);
}
/* transcript.rs:6-8 */
static void app_point(const char* label, const pt* p, transcript* t) {
    uint8_t b[33];
    pt_to_sec1(b, p);
    t_append(t, label, b, 33);
}
/* transcript.rs:10-14; returns 0 where the reference would panic (value >= n) */
static int get_challenge(const char* label, transcript* t, sc* out) {
    uint8_t b[32];
    t_challenge_bytes(t, label, b, 32);
    return sc_from_be(out, b);
}

/* ------------------------------------------------------------------ util.rs vector helpers */
/* Lengths are explicit; `vector_extend` zero-padding (util.rs:24-26) is done by index predicate. */
#define SCV(v, n, i) ((i) < (n) ? (v)[i] : SC_ZERO)
#define PTV(v, n, i) ((i) < (n) ? (v)[i] : PT_IDENTITY)
static size_t zmax(size_t a, size_t b) { return a > b ? a : b; }

static void reduce_sc(const sc* v, size_t n, sc* even, sc* odd) { /* util.rs:7-22 */
    for (size_t i = 0; i < n; i++) (i & 1 ? odd : even)[i / 2] = v[i];
}
static void reduce_pt(const pt* v, size_t n, pt* even, pt* odd) {
    for (size_t i = 0; i < n; i++) (i & 1 ? odd : even)[i / 2] = v[i];
}
static sc weight_vector_mul(const sc* a, size_t na, const sc* b, size_t nb, const sc* w) { /* util.rs:28-44 */
    sc exp = SC_ONE, res = SC_ZERO, t;
    size_t n = zmax(na, nb);
    for (size_t i = 0; i < n; i++) {
        sc av = SCV(a, na, i), bv = SCV(b, nb, i);
        sc_mul(&exp, &exp, w);
        sc_mul(&t, &bv, &exp);
        sc_mul(&t, &av, &t);
        sc_add(&res, &res, &t);
    }
    return res;
}
static sc vector_mul_sc(const sc* a, size_t na, const sc* b, size_t nb) { /* util.rs:46-60, T = Scalar */
    sc res = SC_ZERO, t;
    size_t n = zmax(na, nb);
    for (size_t i = 0; i < n; i++) {
        sc av = SCV(a, na, i), bv = SCV(b, nb, i);
        sc_mul(&t, &av, &bv);
        sc_add(&res, &res, &t);
    }
    return res;
}
static pt vector_mul_pt(const pt* a, size_t na, const sc* b, size_t nb) { /* util.rs:46-60, T = ProjectivePoint: naive MSM */
    pt res = PT_IDENTITY, t;
    size_t n = zmax(na, nb);
    for (size_t i = 0; i < n; i++) {
        pt av = PTV(a, na, i);
        sc bv = SCV(b, nb, i);
        pt_mul(&t, &av, &bv);
        pt_add(&res, &res, &t);
    }
    return res;
}
static void vector_mul_on_scalar_sc(sc* r, const sc* a, size_t n, const sc* s) { /* util.rs:62-67 */
    for (size_t i = 0; i < n; i++) sc_mul(&r[i], &a[i], s);
}
static void vector_mul_on_scalar_pt(pt* r, const pt* a, size_t n, const sc* s) {
    for (size_t i = 0; i < n; i++) pt_mul(&r[i], &a[i], s);
}
/* util.rs:69-85; r must hold max(na, nb) */
static size_t vector_add_sc(sc* r, const sc* a, size_t na, const sc* b, size_t nb) {
    size_t n = zmax(na, nb);
    for (size_t i = 0; i < n; i++) { sc av = SCV(a, na, i), bv = SCV(b, nb, i); sc_add(&r[i], &av, &bv); }
    return n;
}
static size_t vector_sub_sc(sc* r, const sc* a, size_t na, const sc* b, size_t nb) {
    size_t n = zmax(na, nb);
    for (size_t i = 0; i < n; i++) { sc av = SCV(a, na, i), bv = SCV(b, nb, i); sc_sub(&r[i], &av, &bv); }
    return n;
}
static size_t vector_add_pt(pt* r, const pt* a, size_t na, const pt* b, size_t nb) {
    size_t n = zmax(na, nb);
    for (size_t i = 0; i < n; i++) { pt av = PTV(a, na, i), bv = PTV(b, nb, i); pt_add(&r[i], &av, &bv); }
    return n;
}
static void e_vec(sc* r, const sc* v, size_t n) { /* util.rs:87-95 */
    sc buf = SC_ONE;
    for (size_t i = 0; i < n; i++) { r[i] = buf; sc_mul(&buf, &buf, v); }
}
/* util.rs:134-142: a (len na) times dense matrix m (rows x cols, row-major); column at a time */
static void vector_mul_on_matrix(sc* r, const sc* a, size_t na, const sc* m, size_t rows, size_t cols) {
    sc* col = (sc*)malloc(sizeof(sc) * rows);
    for (size_t j = 0; j < cols; j++) {
        for (size_t i = 0; i < rows; i++) col[i] = m[i * cols + j];
        r[j] = vector_mul_sc(a, na, col, rows);
    }
    free(col);
}
/* util.rs:118-132 */
static int diag_inv(sc* m, const sc* x, size_t n) {
    sc xi, val = SC_ONE;
    if (!sc_inv_vartime(&xi, x)) return 0;
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) {
            if (i == j) { sc_mul(&val, &val, &xi); m[i * n + j] = val; }
            else m[i * n + j] = SC_ZERO;
        }
    return 1;
}

/* ------------------------------------------------------------------ wnla.rs */
typedef struct {
    pt g;
    pt* g_vec; size_t ng;
    pt* h_vec; size_t nh;
    sc* c; size_t nc;
    sc rho, mu;
} wnla_t;
typedef struct {
    pt* r; pt* x; size_t nr, nx;
    sc* l; size_t nl;
    sc* n; size_t nn;
} wnla_proof_t;

static pt wnla_commit(const wnla_t* w, const sc* l, size_t nl, const sc* n, size_t nn) { /* wnla.rs:66-72 */
    sc v = vector_mul_sc(w->c, w->nc, l, nl), t = weight_vector_mul(n, nn, n, nn, &w->mu);
    sc_add(&v, &v, &t);
    pt res, a;
    pt_mul(&res, &w->g, &v);
    a = vector_mul_pt(w->h_vec, w->nh, l, nl);
    pt_add(&res, &res, &a);
    a = vector_mul_pt(w->g_vec, w->ng, n, nn);
    pt_add(&res, &res, &a);
    return res;
}

/* trace sink for verify: challenges + computed commitments (tests compare GPU intermediates to these) */
typedef struct {
    sc chal[10];      /* e, rho, lambda, beta, delta, tau, y1..y4 */
    pt pts[6];        /* V+r, C0, C1, C2, C3, C4 */
    int n_y, n_c;
} vtrace;

/* wnla.rs:75-121, recursion written as a loop over owned copies (same operations, same order).
 * returns 1/0, or ORACLE_ERR_DEGENERATE where the reference would panic. */
static int wnla_verify(const wnla_t* w0, const pt* commitment, transcript* t, const wnla_proof_t* proof, vtrace* tr) {
    size_t ng = w0->ng, nh = w0->nh, nc = w0->nc;
    pt* gv = (pt*)malloc(sizeof(pt) * (ng + 1));
    pt* hv = (pt*)malloc(sizeof(pt) * (nh + 1));
    sc* cv = (sc*)malloc(sizeof(sc) * (nc + 1));
    pt* p0 = (pt*)malloc(sizeof(pt) * (zmax(ng, nh) + 2));
    pt* p1 = (pt*)malloc(sizeof(pt) * (zmax(ng, nh) + 2));
    sc* s0 = (sc*)malloc(sizeof(sc) * (nc + 2));
    sc* s1 = (sc*)malloc(sizeof(sc) * (nc + 2));
    memcpy(gv, w0->g_vec, sizeof(pt) * ng);
    memcpy(hv, w0->h_vec, sizeof(pt) * nh);
    memcpy(cv, w0->c, sizeof(sc) * nc);
    sc rho = w0->rho, mu = w0->mu;
    pt com = *commitment;
    size_t nx = proof->nx, nr = proof->nr;
    int result;
    for (;;) {
        if (nx != nr) { result = 0; break; }                                   /* wnla.rs:76-78 */
        if (nx == 0) {                                                          /* wnla.rs:80-82 */
            wnla_t w = {w0->g, gv, ng, hv, nh, cv, nc, rho, mu};
            pt rhs = wnla_commit(&w, proof->l, proof->nl, proof->n, proof->nn);
            if (tr && tr->n_c < 6) tr->pts[tr->n_c++] = com;
            result = pt_eq(&com, &rhs);
            break;
        }
        size_t nc0 = (nc + 1) / 2, nc1 = nc / 2, ng0 = (ng + 1) / 2, ng1 = ng / 2, nh0 = (nh + 1) / 2, nh1 = nh / 2;
        app_point("wnla_com", &com, t);                                        /* wnla.rs:88-92 */
        app_point("wnla_x", &proof->x[nx - 1], t);
        app_point("wnla_r", &proof->r[nr - 1], t);
        t_append_u64(t, "l.sz", nh);
        t_append_u64(t, "n.sz", ng);
        sc y;
        if (!get_challenge("wnla_challenge", t, &y)) { result = ORACLE_ERR_DEGENERATE; break; }
        if (tr) {
            if (tr->n_c < 6) tr->pts[tr->n_c++] = com;
            if (tr->n_y < 4) tr->chal[6 + tr->n_y++] = y;
        }
        /* h_ = h0 + y*h1   (wnla.rs:96) */
        reduce_pt(hv, nh, p0, p1);
        vector_mul_on_scalar_pt(p1, p1, nh1, &y);
        nh = vector_add_pt(hv, p0, nh0, p1, nh1);
        /* g_ = rho*g0 + y*g1   (wnla.rs:97) */
        reduce_pt(gv, ng, p0, p1);
        vector_mul_on_scalar_pt(p0, p0, ng0, &rho);
        vector_mul_on_scalar_pt(p1, p1, ng1, &y);
        ng = vector_add_pt(gv, p0, ng0, p1, ng1);
        /* c_ = c0 + y*c1   (wnla.rs:98) */
        reduce_sc(cv, nc, s0, s1);
        vector_mul_on_scalar_sc(s1, s1, nc1, &y);
        nc = vector_add_sc(cv, s0, nc0, s1, nc1);
        /* com_ = com + y*X + (y*y - 1)*R   (wnla.rs:100-102) */
        pt a;
        sc y2;
        pt_mul(&a, &proof->x[nx - 1], &y);
        pt_add(&com, &com, &a);
        sc_mul(&y2, &y, &y);
        sc_sub(&y2, &y2, &SC_ONE);
        pt_mul(&a, &proof->r[nr - 1], &y2);
        pt_add(&com, &com, &a);
        rho = mu;                                                               /* wnla.rs:109-110 */
        sc_mul(&mu, &mu, &mu);
        nx--; nr--;
    }
    free(gv); free(hv); free(cv); free(p0); free(p1); free(s0); free(s1);
    return result;
}

/* wnla.rs:125-190 (recursive).  proof arrays must have room for log2 rounds; r/x are appended innermost-first. */
static int wnla_prove(const wnla_t* w, const pt* commitment, transcript* t, const sc* l, size_t nl, const sc* n, size_t nn,
                      wnla_proof_t* out) {
    if (nl + nn < 6) {                                                          /* wnla.rs:126-133 */
        memcpy(out->l, l, sizeof(sc) * nl); out->nl = nl;
        memcpy(out->n, n, sizeof(sc) * nn); out->nn = nn;
        out->nr = out->nx = 0;
        return 1;
    }
    sc rho_inv;
    if (!sc_inv_vartime(&rho_inv, &w->rho)) return ORACLE_ERR_DEGENERATE;
    size_t nc = w->nc, ng = w->ng, nh = w->nh;
    size_t nc0 = (nc + 1) / 2, nc1 = nc / 2, nl0 = (nl + 1) / 2, nl1 = nl / 2, nn0 = (nn + 1) / 2, nn1 = nn / 2;
    size_t ng0 = (ng + 1) / 2, ng1 = ng / 2, nh0 = (nh + 1) / 2, nh1 = nh / 2;
    sc *c0 = malloc(sizeof(sc) * (nc0 + 1)), *c1 = malloc(sizeof(sc) * (nc1 + 1));
    sc *l0 = malloc(sizeof(sc) * (nl0 + 1)), *l1 = malloc(sizeof(sc) * (nl1 + 1));
    sc *n0 = malloc(sizeof(sc) * (nn0 + 1)), *n1 = malloc(sizeof(sc) * (nn1 + 1));
    pt *g0 = malloc(sizeof(pt) * (ng0 + 1)), *g1 = malloc(sizeof(pt) * (ng1 + 1));
    pt *h0 = malloc(sizeof(pt) * (nh0 + 1)), *h1 = malloc(sizeof(pt) * (nh1 + 1));
    reduce_sc(w->c, nc, c0, c1);
    reduce_sc(l, nl, l0, l1);
    reduce_sc(n, nn, n0, n1);
    reduce_pt(w->g_vec, ng, g0, g1);
    reduce_pt(w->h_vec, nh, h0, h1);
    sc mu2, two, t1, vx, vr;
    sc_mul(&mu2, &w->mu, &w->mu);
    sc_from_u64(&two, 2);
    /* vx (wnla.rs:145-148) */
    vx = weight_vector_mul(n0, nn0, n1, nn1, &mu2);
    sc_mul(&t1, &rho_inv, &two);
    sc_mul(&vx, &vx, &t1);
    t1 = vector_mul_sc(c0, nc0, l1, nl1); sc_add(&vx, &vx, &t1);
    t1 = vector_mul_sc(c1, nc1, l0, nl0); sc_add(&vx, &vx, &t1);
    /* vr (wnla.rs:150) */
    vr = weight_vector_mul(n1, nn1, n1, nn1, &mu2);
    t1 = vector_mul_sc(c1, nc1, l1, nl1); sc_add(&vr, &vr, &t1);
    /* x (wnla.rs:152-156) */
    sc* tmp = malloc(sizeof(sc) * (zmax(nn0, nn1) + 1));
    pt x, r, a;
    pt_mul(&x, &w->g, &vx);
    a = vector_mul_pt(h0, nh0, l1, nl1); pt_add(&x, &x, &a);
    a = vector_mul_pt(h1, nh1, l0, nl0); pt_add(&x, &x, &a);
    vector_mul_on_scalar_sc(tmp, n1, nn1, &w->rho);
    a = vector_mul_pt(g0, ng0, tmp, nn1); pt_add(&x, &x, &a);
    vector_mul_on_scalar_sc(tmp, n0, nn0, &rho_inv);
    a = vector_mul_pt(g1, ng1, tmp, nn0); pt_add(&x, &x, &a);
    /* r (wnla.rs:158-160) */
    pt_mul(&r, &w->g, &vr);
    a = vector_mul_pt(h1, nh1, l1, nl1); pt_add(&r, &r, &a);
    a = vector_mul_pt(g1, ng1, n1, nn1); pt_add(&r, &r, &a);
    app_point("wnla_com", commitment, t);                                       /* wnla.rs:162-166 */
    app_point("wnla_x", &x, t);
    app_point("wnla_r", &r, t);
    t_append_u64(t, "l.sz", nl);
    t_append_u64(t, "n.sz", nn);
    sc y;
    int rc = 1;
    if (!get_challenge("wnla_challenge", t, &y)) rc = ORACLE_ERR_DEGENERATE;
    if (rc == 1) {
        wnla_t w2;
        w2.g = w->g;
        w2.h_vec = malloc(sizeof(pt) * (nh0 + 1));
        w2.g_vec = malloc(sizeof(pt) * (ng0 + 1));
        w2.c = malloc(sizeof(sc) * (nc0 + 1));
        sc* l_ = malloc(sizeof(sc) * (nl0 + 1));
        sc* n_ = malloc(sizeof(sc) * (nn0 + 1));
        /* wnla.rs:170-175 */
        vector_mul_on_scalar_pt(h1, h1, nh1, &y);
        w2.nh = vector_add_pt(w2.h_vec, h0, nh0, h1, nh1);
        vector_mul_on_scalar_pt(g0, g0, ng0, &w->rho);
        vector_mul_on_scalar_pt(g1, g1, ng1, &y);
        w2.ng = vector_add_pt(w2.g_vec, g0, ng0, g1, ng1);
        vector_mul_on_scalar_sc(c1, c1, nc1, &y);
        w2.nc = vector_add_sc(w2.c, c0, nc0, c1, nc1);
        vector_mul_on_scalar_sc(l1, l1, nl1, &y);
        size_t nl_ = vector_add_sc(l_, l0, nl0, l1, nl1);
        vector_mul_on_scalar_sc(n0, n0, nn0, &rho_inv);
        vector_mul_on_scalar_sc(n1, n1, nn1, &y);
        size_t nn_ = vector_add_sc(n_, n0, nn0, n1, nn1);
        w2.rho = w->mu;
        w2.mu = mu2;
        pt com2 = wnla_commit(&w2, l_, nl_, n_, nn_);                           /* wnla.rs:186 */
        rc = wnla_prove(&w2, &com2, t, l_, nl_, n_, nn_, out);
        if (rc == 1) {
            out->r[out->nr++] = r;                                              /* wnla.rs:187-188 */
            out->x[out->nx++] = x;
        }
        free(w2.h_vec); free(w2.g_vec); free(w2.c); free(l_); free(n_);
    }
    free(c0); free(c1); free(l0); free(l1); free(n0); free(n1); free(g0); free(g1); free(h0); free(h1); free(tmp);
    return rc;
}

/* ------------------------------------------------------------------ circuit.rs */
enum { PART_LO = 0, PART_LL = 1, PART_LR = 2, PART_NO = 3 };
typedef struct {
    size_t dim_nm, dim_no, k, dim_nl, dim_nv, dim_nw;
    pt g;
    const pt* g_vec; size_t n_g_vec;      /* dim_nm */
    const pt* h_vec; size_t n_h_vec;      /* dim_nv + 9 */
    const sc* W_m;                        /* dim_nm x dim_nw, row-major */
    const sc* W_l;                        /* dim_nl x dim_nw */
    const sc* a_m;                        /* dim_nm */
    const sc* a_l;                        /* dim_nl */
    int f_l, f_m;
    const pt* g_vec_; size_t n_g_vec_;
    const pt* h_vec_; size_t n_h_vec_;
    /* partition function as index tables: part[typ][j] = index into w_o / O-block column, or -1 (None).
       LO/LL/LR tables have dim_nv entries, NO has dim_nm entries. */
    const int32_t* part[4];
} circuit_t;
typedef struct {
    pt c_l, c_r, c_o, c_s;
    pt r[64], x[64]; size_t nr, nx;
    sc l[8]; size_t nl;
    sc n[8]; size_t nn;
} circuit_proof_t;

static sc linear_comb_coef(const circuit_t* c, size_t i, const sc* lambda, const sc* mu) { /* circuit.rs:559-570 */
    sc coef = SC_ZERO, t;
    if (c->f_l) { sc_pow_u64(&t, lambda, c->dim_nv * i); sc_add(&coef, &coef, &t); }
    if (c->f_m) { sc_pow_u64(&t, mu, c->dim_nv * i + 1); sc_add(&coef, &coef, &t); }
    return coef;
}
static void collect_cl0(const circuit_t* c, const sc* lambda, const sc* mu, sc* out /* dim_nv-1 */) { /* circuit.rs:572-582 */
    size_t nv = c->dim_nv;
    sc* e = malloc(sizeof(sc) * nv);
    for (size_t i = 0; i + 1 < nv; i++) out[i] = SC_ZERO;
    if (c->f_l) { e_vec(e, lambda, nv); memcpy(out, e + 1, sizeof(sc) * (nv - 1)); }
    if (c->f_m) {
        e_vec(e, mu, nv);
        for (size_t i = 0; i + 1 < nv; i++) { sc t; sc_mul(&t, &e[i + 1], mu); sc_sub(&out[i], &out[i], &t); }
    }
    free(e);
}
static void collect_lambda(const circuit_t* c, const sc* lambda, const sc* mu, sc* out /* dim_nl */) { /* circuit.rs:601-614 */
    e_vec(out, lambda, c->dim_nl);
    if (c->f_l && c->f_m) {
        size_t nv = c->dim_nv, k = c->k;
        sc *el = malloc(sizeof(sc) * nv), *em = malloc(sizeof(sc) * nv), *ek1 = malloc(sizeof(sc) * k), *ek2 = malloc(sizeof(sc) * k);
        sc pm, pl;
        e_vec(el, lambda, nv);
        for (size_t i = 0; i < nv; i++) sc_mul(&el[i], &el[i], mu);
        sc_pow_u64(&pm, mu, nv);
        e_vec(ek1, &pm, k);
        e_vec(em, mu, nv);
        sc_pow_u64(&pl, lambda, nv);
        e_vec(ek2, &pl, k);
        /* vector_tensor_mul(a, b) = concat over b_j of a*b_j (util.rs:111-116); lengths nv*k = dim_nl */
        for (size_t j = 0; j < k; j++)
            for (size_t i = 0; i < nv; i++) {
                sc t1, t2;
                sc_mul(&t1, &el[i], &ek1[j]);
                sc_mul(&t2, &em[i], &ek2[j]);
                sc_add(&t1, &t1, &t2);
                size_t idx = j * nv + i;
                if (idx < c->dim_nl) sc_sub(&out[idx], &out[idx], &t1);
            }
        free(el); free(em); free(ek1); free(ek2);
    }
}
typedef struct { sc *c_nL, *c_nR, *c_nO, *c_lL, *c_lR, *c_lO; } ccoef; /* nm, nm, nm, nv, nv, nv */

/* circuit.rs:584-653: dense slices of W, partition remap, mat-vec, diag_inv -- as the reference does it */
static int collect_c(const circuit_t* c, const sc* lambda_vec, const sc* mu_vec, const sc* mu, ccoef* o) {
    size_t nm = c->dim_nm, nl = c->dim_nl, nv = c->dim_nv, nw = c->dim_nw;
    sc* M = malloc(sizeof(sc) * (zmax(nl, nm) * zmax(nm, nv) + 1));
    sc* t1 = malloc(sizeof(sc) * (zmax(nm, nv) + 1));
    sc* t2 = malloc(sizeof(sc) * (zmax(nm, nv) + 1));
    sc* dinv = malloc(sizeof(sc) * nm * nm);
    int ok = diag_inv(dinv, mu, nm);
    if (ok) {
        /* slice(rows, W, col0, ncols): M[i][j] = W[i][col0 + j]                                   (collect_m_rl, :616-622)
           mapf(rows, W, jsz, typ):     M[i][j] = part(typ, j) is Some(j_) ? W[i][2nm + j_] : 0    (collect_m_o,  :624-653) */
#define SLICE(rows, W, col0, ncols) \
    for (size_t i = 0; i < (rows); i++) for (size_t j = 0; j < (ncols); j++) M[i * (ncols) + j] = (W)[i * nw + (col0) + j];
#define MAPF(rows, W, jsz, typ) \
    for (size_t i = 0; i < (rows); i++) for (size_t j = 0; j < (jsz); j++) { \
        int32_t j_ = c->part[typ][j]; M[i * (jsz) + j] = j_ >= 0 ? (W)[i * nw + 2 * nm + (size_t)j_] : SC_ZERO; }
#define NTERM(out, SEL_L, SEL_M, cols) \
    SEL_L; vector_mul_on_matrix(t1, lambda_vec, nl, M, nl, cols); \
    SEL_M; vector_mul_on_matrix(t2, mu_vec, nm, M, nm, cols); \
    vector_sub_sc(t1, t1, cols, t2, cols);
        NTERM(c_nL, SLICE(nl, c->W_l, 0, nm), SLICE(nm, c->W_m, 0, nm), nm);
        vector_mul_on_matrix(o->c_nL, t1, nm, dinv, nm, nm);
        NTERM(c_nR, SLICE(nl, c->W_l, nm, nm), SLICE(nm, c->W_m, nm, nm), nm);
        vector_mul_on_matrix(o->c_nR, t1, nm, dinv, nm, nm);
        NTERM(c_nO, MAPF(nl, c->W_l, nm, PART_NO), MAPF(nm, c->W_m, nm, PART_NO), nm);
        vector_mul_on_matrix(o->c_nO, t1, nm, dinv, nm, nm);
        NTERM(c_lL, MAPF(nl, c->W_l, nv, PART_LL), MAPF(nm, c->W_m, nv, PART_LL), nv);
        memcpy(o->c_lL, t1, sizeof(sc) * nv);
        NTERM(c_lR, MAPF(nl, c->W_l, nv, PART_LR), MAPF(nm, c->W_m, nv, PART_LR), nv);
        memcpy(o->c_lR, t1, sizeof(sc) * nv);
        NTERM(c_lO, MAPF(nl, c->W_l, nv, PART_LO), MAPF(nm, c->W_m, nv, PART_LO), nv);
        memcpy(o->c_lO, t1, sizeof(sc) * nv);
#undef SLICE
#undef MAPF
#undef NTERM
    }
    free(M); free(t1); free(t2); free(dinv);
    return ok;
}
static ccoef ccoef_alloc(size_t nm, size_t nv) {
    ccoef o;
    o.c_nL = malloc(sizeof(sc) * nm); o.c_nR = malloc(sizeof(sc) * nm); o.c_nO = malloc(sizeof(sc) * nm);
    o.c_lL = malloc(sizeof(sc) * nv); o.c_lR = malloc(sizeof(sc) * nv); o.c_lO = malloc(sizeof(sc) * nv);
    return o;
}
static void ccoef_free(ccoef* o) { free(o->c_nL); free(o->c_nR); free(o->c_nO); free(o->c_lL); free(o->c_lR); free(o->c_lO); }

static pt circuit_commit(const circuit_t* c, const sc* v, size_t nv, const sc* s) { /* circuit.rs:146-151 */
    pt res, a;
    pt_mul(&res, &c->g, &v[0]);
    pt_mul(&a, &c->h_vec[0], s);
    pt_add(&res, &res, &a);
    a = vector_mul_pt(c->h_vec + 9, c->n_h_vec - 9, v + 1, nv - 1);
    pt_add(&res, &res, &a);
    return res;
}

static void cr_tau_vec(sc cr[9], const sc* tau, const sc* tau_inv, const sc* tau2, const sc* tau3, const sc* beta) { /* circuit.rs:208-218 */
    sc t;
    cr[0] = SC_ONE;
    sc_mul(&cr[1], tau_inv, beta);
    sc_mul(&cr[2], tau, beta);
    sc_mul(&cr[3], tau2, beta);
    sc_mul(&cr[4], tau3, beta);
    sc_mul(&t, tau, tau3); sc_mul(&cr[5], &t, beta);
    sc_mul(&t, tau2, tau3); sc_mul(&cr[6], &t, beta);
    sc_mul(&t, tau3, tau3); sc_mul(&cr[7], &t, beta);
    sc_mul(&t, tau3, tau3); sc_mul(&t, &t, tau); sc_mul(&cr[8], &t, beta);
}

/* circuit.rs:154-256 */
static int circuit_verify(const circuit_t* c, const pt* v, size_t nvpts, transcript* t, const circuit_proof_t* proof, vtrace* tr) {
    size_t nm = c->dim_nm, nv = c->dim_nv, nl = c->dim_nl;
    app_point("commitment_cl", &proof->c_l, t);
    app_point("commitment_cr", &proof->c_r, t);
    app_point("commitment_co", &proof->c_o, t);
    for (size_t i = 0; i < nvpts; i++) app_point("commitment_v", &v[i], t);
    sc rho, lambda, beta, delta, mu;
    if (!get_challenge("circuit_rho", t, &rho) || !get_challenge("circuit_lambda", t, &lambda) ||
        !get_challenge("circuit_beta", t, &beta) || !get_challenge("circuit_delta", t, &delta))
        return ORACLE_ERR_DEGENERATE;
    sc_mul(&mu, &rho, &rho);
    sc* lambda_vec = malloc(sizeof(sc) * nl);
    sc* mu_vec = malloc(sizeof(sc) * nm);
    collect_lambda(c, &lambda, &mu, lambda_vec);
    e_vec(mu_vec, &mu, nm);
    vector_mul_on_scalar_sc(mu_vec, mu_vec, nm, &mu);
    ccoef cc = ccoef_alloc(nm, nv);
    int rc = 1;
    sc *pn_tau = malloc(sizeof(sc) * nm), *tmp = malloc(sizeof(sc) * zmax(nm, nv)), *cl_tau = malloc(sizeof(sc) * nv);
    sc* c_l0 = malloc(sizeof(sc) * nv);
    size_t nc_full = c->n_h_vec + c->n_h_vec_;
    if (nc_full < 9 + nv) nc_full = 9 + nv;
    sc* cvec = malloc(sizeof(sc) * nc_full);
    pt* gcat = malloc(sizeof(pt) * (c->n_g_vec + c->n_g_vec_ + 1));
    pt* hcat = malloc(sizeof(pt) * (c->n_h_vec + c->n_h_vec_ + 1));
    if (!collect_c(c, lambda_vec, mu_vec, &mu, &cc)) rc = ORACLE_ERR_DEGENERATE;
    if (rc == 1) {
        sc two;
        sc_from_u64(&two, 2);
        pt v_ = PT_IDENTITY, a;                                                 /* circuit.rs:182-187 */
        for (size_t i = 0; i < c->k; i++) {
            sc coef = linear_comb_coef(c, i, &lambda, &mu);
            pt_mul(&a, &v[i], &coef);
            pt_add(&v_, &v_, &a);
        }
        pt_mul(&v_, &v_, &two);
        app_point("commitment_cs", &proof->c_s, t);
        sc tau, tau_inv, tau2, tau3, delta_inv, td;
        if (!get_challenge("circuit_tau", t, &tau) || !sc_inv_vartime(&tau_inv, &tau) || !sc_inv_vartime(&delta_inv, &delta))
            rc = ORACLE_ERR_DEGENERATE;
        if (rc == 1) {
            sc_mul(&tau2, &tau, &tau);
            sc_mul(&tau3, &tau2, &tau);
            sc_mul(&td, &tau3, &delta_inv);
            /* pn_tau (circuit.rs:198-200) */
            vector_mul_on_scalar_sc(pn_tau, cc.c_nO, nm, &td);
            vector_mul_on_scalar_sc(tmp, cc.c_nL, nm, &tau2);
            vector_sub_sc(pn_tau, pn_tau, nm, tmp, nm);
            vector_mul_on_scalar_sc(tmp, cc.c_nR, nm, &tau);
            vector_add_sc(pn_tau, pn_tau, nm, tmp, nm);
            /* ps_tau (circuit.rs:202-204) */
            sc ps_tau = weight_vector_mul(pn_tau, nm, pn_tau, nm, &mu), s1;
            s1 = vector_mul_sc(lambda_vec, nl, c->a_l, nl); sc_mul(&s1, &s1, &tau3); sc_mul(&s1, &s1, &two); sc_add(&ps_tau, &ps_tau, &s1);
            s1 = vector_mul_sc(mu_vec, nm, c->a_m, nm); sc_mul(&s1, &s1, &tau3); sc_mul(&s1, &s1, &two); sc_sub(&ps_tau, &ps_tau, &s1);
            /* pt (circuit.rs:206) */
            pt ptv;
            pt_mul(&ptv, &c->g, &ps_tau);
            a = vector_mul_pt(c->g_vec, c->n_g_vec, pn_tau, nm);
            pt_add(&ptv, &ptv, &a);
            sc cr[9];
            cr_tau_vec(cr, &tau, &tau_inv, &tau2, &tau3, &beta);
            collect_cl0(c, &lambda, &mu, c_l0);
            /* cl_tau (circuit.rs:222-226) */
            vector_mul_on_scalar_sc(cl_tau, cc.c_lO, nv, &td);
            vector_mul_on_scalar_sc(tmp, cc.c_lL, nv, &tau2);
            vector_sub_sc(cl_tau, cl_tau, nv, tmp, nv);
            vector_mul_on_scalar_sc(tmp, cc.c_lR, nv, &tau);
            vector_add_sc(cl_tau, cl_tau, nv, tmp, nv);
            vector_mul_on_scalar_sc(cl_tau, cl_tau, nv, &two);
            vector_sub_sc(cl_tau, cl_tau, nv, c_l0, nv - 1);
            size_t ncv = 9 + nv;
            memcpy(cvec, cr, sizeof(sc) * 9);
            memcpy(cvec + 9, cl_tau, sizeof(sc) * nv);
            /* commitment (circuit.rs:230-235) */
            pt com = ptv;
            pt_mul(&a, &proof->c_s, &tau_inv); pt_add(&com, &com, &a);
            pt_mul(&a, &proof->c_o, &delta); pt_sub(&com, &com, &a);
            pt_mul(&a, &proof->c_l, &tau); pt_add(&com, &com, &a);
            pt_mul(&a, &proof->c_r, &tau2); pt_sub(&com, &com, &a);
            pt_mul(&a, &v_, &tau3); pt_add(&com, &com, &a);
            while (ncv < c->n_h_vec + c->n_h_vec_) cvec[ncv++] = SC_ZERO;        /* circuit.rs:237-239 */
            memcpy(gcat, c->g_vec, sizeof(pt) * c->n_g_vec);
            memcpy(gcat + c->n_g_vec, c->g_vec_, sizeof(pt) * c->n_g_vec_);
            memcpy(hcat, c->h_vec, sizeof(pt) * c->n_h_vec);
            memcpy(hcat + c->n_h_vec, c->h_vec_, sizeof(pt) * c->n_h_vec_);
            if (tr) { tr->chal[1] = rho; tr->chal[2] = lambda; tr->chal[3] = beta; tr->chal[4] = delta; tr->chal[5] = tau; }
            wnla_t w = {c->g, gcat, c->n_g_vec + c->n_g_vec_, hcat, c->n_h_vec + c->n_h_vec_, cvec, ncv, rho, mu};
            wnla_proof_t wp = {(pt*)proof->r, (pt*)proof->x, proof->nr, proof->nx, (sc*)proof->l, proof->nl, (sc*)proof->n, proof->nn};
            rc = wnla_verify(&w, &com, t, &wp, tr);
        }
    }
    ccoef_free(&cc);
    free(lambda_vec); free(mu_vec); free(pn_tau); free(tmp); free(cl_tau); free(c_l0); free(cvec); free(gcat); free(hcat);
    return rc;
}

typedef struct { const sc* next; size_t left; } rng_t;   /* stand-in for Scalar::generate_biased(rng): caller-supplied draws */
static sc rng_draw(rng_t* r) {
    if (!r->left) return SC_ZERO;
    r->left--;
    return *r->next++;
}
typedef struct {
    const sc* v; size_t nv_each;      /* k vectors of dim_nv scalars, contiguous */
    const sc* s_v;                    /* k */
    const sc *w_l, *w_r, *w_o;        /* nm, nm, no */
} cwitness;

/* circuit.rs:260-556 */
static int circuit_prove(const circuit_t* c, const pt* v, size_t nvpts, const cwitness* wit, transcript* t, rng_t* rng,
                         circuit_proof_t* out) {
    size_t nm = c->dim_nm, nv = c->dim_nv, nl = c->dim_nl;
    sc Z = SC_ZERO;
    sc ro[9], rl[9], rr[9];
    /* circuit.rs:264-298: fixed zero slots, rng draw order ro, rl, rr */
    ro[0] = rng_draw(rng); ro[1] = rng_draw(rng); ro[2] = rng_draw(rng); ro[3] = rng_draw(rng); ro[4] = Z;
    ro[5] = rng_draw(rng); ro[6] = rng_draw(rng); ro[7] = rng_draw(rng); ro[8] = Z;
    rl[0] = rng_draw(rng); rl[1] = rng_draw(rng); rl[2] = rng_draw(rng); rl[3] = Z; rl[4] = rng_draw(rng);
    rl[5] = rng_draw(rng); rl[6] = rng_draw(rng); rl[7] = Z; rl[8] = Z;
    rr[0] = rng_draw(rng); rr[1] = rng_draw(rng); rr[2] = Z; rr[3] = rng_draw(rng); rr[4] = rng_draw(rng);
    rr[5] = rng_draw(rng); rr[6] = Z; rr[7] = Z; rr[8] = Z;
    const sc *nlv = wit->w_l, *nrv = wit->w_r;
    sc *no = malloc(sizeof(sc) * nm), *lo = malloc(sizeof(sc) * nv), *ll = malloc(sizeof(sc) * nv), *lr = malloc(sizeof(sc) * nv);
    for (size_t j = 0; j < nm; j++) { int32_t i = c->part[PART_NO][j]; no[j] = i >= 0 ? wit->w_o[i] : Z; }
    for (size_t j = 0; j < nv; j++) { int32_t i = c->part[PART_LO][j]; lo[j] = i >= 0 ? wit->w_o[i] : Z; }
    for (size_t j = 0; j < nv; j++) { int32_t i = c->part[PART_LL][j]; ll[j] = i >= 0 ? wit->w_o[i] : Z; }
    for (size_t j = 0; j < nv; j++) { int32_t i = c->part[PART_LR][j]; lr[j] = i >= 0 ? wit->w_o[i] : Z; }
    size_t nhs = 9 + nv;
    sc* cat = malloc(sizeof(sc) * nhs);
    pt co, cl, cr, a;
#define HCOMMIT(dst, r9, lvec, nvec) \
    memcpy(cat, r9, sizeof(sc) * 9); memcpy(cat + 9, lvec, sizeof(sc) * nv); \
    dst = vector_mul_pt(c->h_vec, c->n_h_vec, cat, nhs); a = vector_mul_pt(c->g_vec, c->n_g_vec, nvec, nm); pt_add(&dst, &dst, &a);
    HCOMMIT(co, ro, lo, no);                                                    /* circuit.rs:335-345 */
    HCOMMIT(cl, rl, ll, nlv);
    HCOMMIT(cr, rr, lr, nrv);
    app_point("commitment_cl", &cl, t);
    app_point("commitment_cr", &cr, t);
    app_point("commitment_co", &co, t);
    for (size_t i = 0; i < nvpts; i++) app_point("commitment_v", &v[i], t);
    sc rho, lambda, beta, delta, mu;
    int rc = 1;
    if (!get_challenge("circuit_rho", t, &rho) || !get_challenge("circuit_lambda", t, &lambda) ||
        !get_challenge("circuit_beta", t, &beta) || !get_challenge("circuit_delta", t, &delta))
        rc = ORACLE_ERR_DEGENERATE;
    sc_mul(&mu, &rho, &rho);
    sc* lambda_vec = malloc(sizeof(sc) * nl);
    sc* mu_vec = malloc(sizeof(sc) * nm);
    ccoef cc = ccoef_alloc(nm, nv);
    sc *ls = malloc(sizeof(sc) * nv), *ns = malloc(sizeof(sc) * nm), *v_1 = malloc(sizeof(sc) * nv), *c_l0 = malloc(sizeof(sc) * nv);
    sc *t1 = malloc(sizeof(sc) * (nhs + nm)), *t2 = malloc(sizeof(sc) * (nhs + nm));
    sc *lvec = malloc(sizeof(sc) * (c->n_h_vec + c->n_h_vec_ + nhs)), *nvec = malloc(sizeof(sc) * (c->n_g_vec + c->n_g_vec_ + nm));
    sc *cvec = malloc(sizeof(sc) * (c->n_h_vec + c->n_h_vec_ + nhs)), *pn_tau = malloc(sizeof(sc) * nm), *cl_tau = malloc(sizeof(sc) * nv);
    pt* gcat = malloc(sizeof(pt) * (c->n_g_vec + c->n_g_vec_ + 1));
    pt* hcat = malloc(sizeof(pt) * (c->n_h_vec + c->n_h_vec_ + 1));
    if (rc == 1) {
        collect_lambda(c, &lambda, &mu, lambda_vec);
        e_vec(mu_vec, &mu, nm);
        vector_mul_on_scalar_sc(mu_vec, mu_vec, nm, &mu);
        if (!collect_c(c, lambda_vec, mu_vec, &mu, &cc)) rc = ORACLE_ERR_DEGENERATE;
    }
    if (rc == 1) {
        for (size_t i = 0; i < nv; i++) ls[i] = rng_draw(rng);                  /* circuit.rs:371-372 */
        for (size_t i = 0; i < nm; i++) ns[i] = rng_draw(rng);
        sc two, v_0 = Z, rv[9], s1, s2;
        sc_from_u64(&two, 2);
        for (int i = 0; i < 9; i++) rv[i] = Z;
        for (size_t i = 0; i + 1 < nv; i++) v_1[i] = Z;
        for (size_t i = 0; i < c->k; i++) {                                     /* circuit.rs:376-395 */
            sc coef = linear_comb_coef(c, i, &lambda, &mu);
            sc_mul(&s1, &wit->v[i * wit->nv_each], &coef); sc_add(&v_0, &v_0, &s1);
            sc_mul(&s1, &wit->s_v[i], &coef); sc_add(&rv[0], &rv[0], &s1);
            for (size_t j = 0; j + 1 < nv; j++) { sc_mul(&s1, &wit->v[i * wit->nv_each + 1 + j], &coef); sc_add(&v_1[j], &v_1[j], &s1); }
        }
        sc_mul(&v_0, &v_0, &two);
        sc_mul(&rv[0], &rv[0], &two);
        vector_mul_on_scalar_sc(v_1, v_1, nv - 1, &two);
        collect_cl0(c, &lambda, &mu, c_l0);
        size_t ncl0 = nv - 1, nv1 = nv - 1;
        sc f_[8], delta2, delta_inv, beta_inv;
        sc_mul(&delta2, &delta, &delta);
        if (!sc_inv_vartime(&delta_inv, &delta) || !sc_inv_vartime(&beta_inv, &beta)) rc = ORACLE_ERR_DEGENERATE;
        if (rc == 1) {
#define WVM(a, na, b, nb) weight_vector_mul(a, na, b, nb, &mu)
#define VM(a, na, b, nb) vector_mul_sc(a, na, b, nb)
#define MUL(x, y) sc_mul(&(x), &(x), &(y))
            sc *nl_cnR = t1, *nr_cnL = t2;                                       /* vector_add(&nl,&c_nR), vector_add(&nr,&c_nL) */
            vector_add_sc(nl_cnR, nlv, nm, cc.c_nR, nm);
            vector_add_sc(nr_cnL, nrv, nm, cc.c_nL, nm);
            /* -2 (circuit.rs:406) */
            s1 = WVM(ns, nm, ns, nm); sc_minus(&f_[0], &s1);
            /* -1 (circuit.rs:409-410) */
            f_[1] = VM(c_l0, ncl0, ls, nv);
            s1 = WVM(ns, nm, no, nm); sc_mul(&s2, &delta, &two); MUL(s2, s1); sc_add(&f_[1], &f_[1], &s2);
            /* 0 (circuit.rs:413-416) */
            s1 = VM(cc.c_lR, nv, ls, nv); MUL(s1, two); sc_minus(&f_[2], &s1);
            s1 = VM(c_l0, ncl0, lo, nv); MUL(s1, delta); sc_sub(&f_[2], &f_[2], &s1);
            s1 = WVM(ns, nm, nl_cnR, nm); MUL(s1, two); sc_sub(&f_[2], &f_[2], &s1);
            s1 = WVM(no, nm, no, nm); MUL(s1, delta2); sc_sub(&f_[2], &f_[2], &s1);
            /* 1 (circuit.rs:419-423) */
            f_[3] = VM(cc.c_lL, nv, ls, nv); MUL(f_[3], two);
            s1 = VM(cc.c_lR, nv, lo, nv); MUL(s1, delta); MUL(s1, two); sc_add(&f_[3], &f_[3], &s1);
            s1 = VM(c_l0, ncl0, ll, nv); sc_add(&f_[3], &f_[3], &s1);
            s1 = WVM(ns, nm, nr_cnL, nm); MUL(s1, two); sc_add(&f_[3], &f_[3], &s1);
            s1 = WVM(no, nm, nl_cnR, nm); MUL(s1, two); MUL(s1, delta); sc_add(&f_[3], &f_[3], &s1);
            /* 2 (circuit.rs:426-433) */
            f_[4] = WVM(cc.c_nR, nm, cc.c_nR, nm);
            s1 = VM(cc.c_lO, nv, ls, nv); MUL(s1, delta_inv); MUL(s1, two); sc_sub(&f_[4], &f_[4], &s1);
            s1 = VM(cc.c_lL, nv, lo, nv); MUL(s1, delta); MUL(s1, two); sc_sub(&f_[4], &f_[4], &s1);
            s1 = VM(cc.c_lR, nv, ll, nv); MUL(s1, two); sc_sub(&f_[4], &f_[4], &s1);
            s1 = VM(c_l0, ncl0, lr, nv); sc_sub(&f_[4], &f_[4], &s1);
            s1 = WVM(ns, nm, cc.c_nO, nm); MUL(s1, delta_inv); MUL(s1, two); sc_sub(&f_[4], &f_[4], &s1);
            s1 = WVM(no, nm, nr_cnL, nm); MUL(s1, delta); MUL(s1, two); sc_sub(&f_[4], &f_[4], &s1);
            s1 = WVM(nl_cnR, nm, nl_cnR, nm); sc_sub(&f_[4], &f_[4], &s1);
            /* 4 (circuit.rs:438-444) */
            f_[5] = WVM(cc.c_nO, nm, cc.c_nR, nm); MUL(f_[5], delta_inv); MUL(f_[5], two);
            s1 = WVM(cc.c_nL, nm, cc.c_nL, nm); sc_add(&f_[5], &f_[5], &s1);
            s1 = VM(cc.c_lO, nv, ll, nv); MUL(s1, delta_inv); MUL(s1, two); sc_sub(&f_[5], &f_[5], &s1);
            s1 = VM(cc.c_lL, nv, lr, nv); MUL(s1, two); sc_sub(&f_[5], &f_[5], &s1);
            s1 = VM(cc.c_lR, nv, v_1, nv1); MUL(s1, two); sc_sub(&f_[5], &f_[5], &s1);
            s1 = WVM(nl_cnR, nm, cc.c_nO, nm); MUL(s1, delta_inv); MUL(s1, two); sc_sub(&f_[5], &f_[5], &s1);
            s1 = WVM(nr_cnL, nm, nr_cnL, nm); sc_sub(&f_[5], &f_[5], &s1);
            /* 5 (circuit.rs:447-450) */
            s1 = WVM(cc.c_nO, nm, cc.c_nL, nm); MUL(s1, delta_inv); MUL(s1, two); sc_minus(&f_[6], &s1);
            s1 = VM(cc.c_nO, nm, lr, nv); MUL(s1, delta_inv); MUL(s1, two); sc_add(&f_[6], &f_[6], &s1);
            s1 = VM(cc.c_lL, nv, v_1, nv1); MUL(s1, two); sc_add(&f_[6], &f_[6], &s1);
            s1 = WVM(nr_cnL, nm, cc.c_nO, nm); MUL(s1, delta_inv); MUL(s1, two); sc_add(&f_[6], &f_[6], &s1);
            /* 6 (circuit.rs:453) */
            s1 = VM(cc.c_lO, nv, v_1, nv1); MUL(s1, delta_inv); MUL(s1, two); sc_minus(&f_[7], &s1);
            /* rs (circuit.rs:457-467) */
            sc rs[9];
            s1 = ro[1]; MUL(s1, delta); MUL(s1, beta); sc_add(&rs[0], &f_[1], &s1);
            sc_mul(&rs[1], &f_[0], &beta_inv);
            s1 = ro[0]; MUL(s1, delta); sc_add(&s1, &s1, &f_[2]); MUL(s1, beta_inv); sc_sub(&rs[2], &s1, &rl[1]);
            sc_sub(&s1, &f_[3], &rl[0]); MUL(s1, beta_inv); s2 = ro[2]; MUL(s2, delta); sc_add(&s2, &s2, &rr[1]); sc_add(&rs[3], &s1, &s2);
            sc_add(&s1, &f_[4], &rr[0]); MUL(s1, beta_inv); s2 = ro[3]; MUL(s2, delta); sc_sub(&s2, &s2, &rl[2]); sc_add(&rs[4], &s1, &s2);
            sc_mul(&s1, &rv[0], &beta_inv); sc_minus(&rs[5], &s1);
            sc_mul(&s1, &f_[5], &beta_inv); s2 = ro[5]; MUL(s2, delta); sc_add(&s1, &s1, &s2); sc_add(&s1, &s1, &rr[3]); sc_sub(&rs[6], &s1, &rl[4]);
            sc_mul(&s1, &f_[6], &beta_inv); sc_add(&s1, &s1, &rr[4]); s2 = ro[6]; MUL(s2, delta); sc_add(&s1, &s1, &s2); sc_sub(&rs[7], &s1, &rl[5]);
            sc_mul(&s1, &f_[7], &beta_inv); s2 = ro[7]; MUL(s2, delta); sc_add(&s1, &s1, &s2); sc_sub(&s1, &s1, &rl[6]); sc_add(&rs[8], &s1, &rr[5]);
            pt cs;
            HCOMMIT(cs, rs, ls, ns);                                            /* circuit.rs:469-470 */
            app_point("commitment_cs", &cs, t);
            sc tau, tau_inv, tau2, tau3, td;
            if (!get_challenge("circuit_tau", t, &tau) || !sc_inv_vartime(&tau_inv, &tau)) rc = ORACLE_ERR_DEGENERATE;
            if (rc == 1) {
                sc_mul(&tau2, &tau, &tau);
                sc_mul(&tau3, &tau2, &tau);
                sc_mul(&td, &tau3, &delta_inv);
                /* l (circuit.rs:479-483) */
#define CAT(r9, lv, n_l) memcpy(cat, r9, sizeof(sc) * 9); memcpy(cat + 9, lv, sizeof(sc) * (n_l));
                size_t nlv_ = nhs;
                CAT(rs, ls, nv); vector_mul_on_scalar_sc(lvec, cat, nhs, &tau_inv);
                CAT(ro, lo, nv); vector_mul_on_scalar_sc(t1, cat, nhs, &delta); vector_sub_sc(lvec, lvec, nhs, t1, nhs);
                CAT(rl, ll, nv); vector_mul_on_scalar_sc(t1, cat, nhs, &tau); vector_add_sc(lvec, lvec, nhs, t1, nhs);
                CAT(rr, lr, nv); vector_mul_on_scalar_sc(t1, cat, nhs, &tau2); vector_sub_sc(lvec, lvec, nhs, t1, nhs);
                CAT(rv, v_1, nv1); vector_mul_on_scalar_sc(t1, cat, 9 + nv1, &tau3); vector_add_sc(lvec, lvec, nhs, t1, 9 + nv1);
                /* pn_tau, ps_tau (circuit.rs:485-491) */
                vector_mul_on_scalar_sc(pn_tau, cc.c_nO, nm, &td);
                vector_mul_on_scalar_sc(t1, cc.c_nL, nm, &tau2); vector_sub_sc(pn_tau, pn_tau, nm, t1, nm);
                vector_mul_on_scalar_sc(t1, cc.c_nR, nm, &tau); vector_add_sc(pn_tau, pn_tau, nm, t1, nm);
                sc ps_tau = weight_vector_mul(pn_tau, nm, pn_tau, nm, &mu);
                s1 = vector_mul_sc(lambda_vec, nl, c->a_l, nl); MUL(s1, tau3); MUL(s1, two); sc_add(&ps_tau, &ps_tau, &s1);
                s1 = vector_mul_sc(mu_vec, nm, c->a_m, nm); MUL(s1, tau3); MUL(s1, two); sc_sub(&ps_tau, &ps_tau, &s1);
                /* n (circuit.rs:493-498) */
                vector_mul_on_scalar_sc(nvec, ns, nm, &tau_inv);
                vector_mul_on_scalar_sc(t1, no, nm, &delta); vector_sub_sc(nvec, nvec, nm, t1, nm);
                vector_mul_on_scalar_sc(t1, nlv, nm, &tau); vector_add_sc(nvec, nvec, nm, t1, nm);
                vector_mul_on_scalar_sc(t1, nrv, nm, &tau2); vector_sub_sc(nvec, nvec, nm, t1, nm);
                vector_add_sc(nvec, pn_tau, nm, nvec, nm);
                size_t nnv = nm;
                sc crt[9];
                cr_tau_vec(crt, &tau, &tau_inv, &tau2, &tau3, &beta);
                /* cl_tau (circuit.rs:512-516) */
                vector_mul_on_scalar_sc(cl_tau, cc.c_lO, nv, &td);
                vector_mul_on_scalar_sc(t1, cc.c_lL, nv, &tau2); vector_sub_sc(cl_tau, cl_tau, nv, t1, nv);
                vector_mul_on_scalar_sc(t1, cc.c_lR, nv, &tau); vector_add_sc(cl_tau, cl_tau, nv, t1, nv);
                vector_mul_on_scalar_sc(cl_tau, cl_tau, nv, &two);
                vector_sub_sc(cl_tau, cl_tau, nv, c_l0, ncl0);
                size_t ncv = 9 + nv;
                memcpy(cvec, crt, sizeof(sc) * 9);
                memcpy(cvec + 9, cl_tau, sizeof(sc) * nv);
                /* commitment (circuit.rs:520-524) */
                sc vv;
                sc_mul(&vv, &tau3, &v_0);
                sc_add(&vv, &ps_tau, &vv);
                pt com;
                pt_mul(&com, &c->g, &vv);
                a = vector_mul_pt(c->h_vec, c->n_h_vec, lvec, nlv_); pt_add(&com, &com, &a);
                a = vector_mul_pt(c->g_vec, c->n_g_vec, nvec, nnv); pt_add(&com, &com, &a);
                while (nlv_ < c->n_h_vec + c->n_h_vec_) { lvec[nlv_++] = Z; cvec[ncv++] = Z; }   /* circuit.rs:526-529 */
                while (nnv < c->n_g_vec + c->n_g_vec_) nvec[nnv++] = Z;                          /* circuit.rs:531-533 */
                memcpy(gcat, c->g_vec, sizeof(pt) * c->n_g_vec);
                memcpy(gcat + c->n_g_vec, c->g_vec_, sizeof(pt) * c->n_g_vec_);
                memcpy(hcat, c->h_vec, sizeof(pt) * c->n_h_vec);
                memcpy(hcat + c->n_h_vec, c->h_vec_, sizeof(pt) * c->n_h_vec_);
                wnla_t w = {c->g, gcat, c->n_g_vec + c->n_g_vec_, hcat, c->n_h_vec + c->n_h_vec_, cvec, ncv, rho, mu};
                wnla_proof_t wp = {out->r, out->x, 0, 0, out->l, 0, out->n, 0};
                rc = wnla_prove(&w, &com, t, lvec, nlv_, nvec, nnv, &wp);
                out->c_l = cl; out->c_r = cr; out->c_o = co; out->c_s = cs;
                out->nr = wp.nr; out->nx = wp.nx; out->nl = wp.nl; out->nn = wp.nn;
            }
        }
    }
    ccoef_free(&cc);
    free(no); free(lo); free(ll); free(lr); free(cat); free(lambda_vec); free(mu_vec); free(ls); free(ns); free(v_1); free(c_l0);
    free(t1); free(t2); free(lvec); free(nvec); free(cvec); free(pn_tau); free(cl_tau); free(gcat); free(hcat);
    return rc;
}

/* ------------------------------------------------------------------ range_proof/reciprocal.rs */
typedef struct {
    size_t dim_nd, dim_np;
    pt g;
    const pt* g_vec; size_t n_g_vec;
    const pt* h_vec; size_t n_h_vec;
    const pt* g_vec_; size_t n_g_vec_;
    const pt* h_vec_; size_t n_h_vec_;
} reciprocal_t;
typedef struct { sc *W_m, *W_l, *a_m, *a_l; int32_t* part[4]; } circuit_store;

/* reciprocal.rs:150-214 */
static int reciprocal_make_circuit(const reciprocal_t* p, const sc* e, circuit_t* c, circuit_store* st) {
    size_t nm = p->dim_nd, no = p->dim_np, nv = p->dim_nd + 1, nl = nv, nw = p->dim_nd * 2 + p->dim_np;
    st->a_m = malloc(sizeof(sc) * nm);
    st->a_l = malloc(sizeof(sc) * nl);
    st->W_m = calloc(nm * nw, sizeof(sc));
    st->W_l = calloc(nl * nw, sizeof(sc));
    for (size_t i = 0; i < nm; i++) st->a_m[i] = SC_ONE;
    for (size_t i = 0; i < nl; i++) st->a_l[i] = SC_ZERO;
    sc me;
    sc_minus(&me, e);
    for (size_t i = 0; i < nm; i++) st->W_m[i * nw + i + nm] = me;              /* :162 */
    sc base, pw, t;
    sc_from_u64(&base, (uint32_t)p->dim_np);
    for (size_t i = 0; i < nm; i++) { sc_pow_u64(&pw, &base, i); sc_minus(&st->W_l[0 * nw + i], &pw); }   /* :170 */
    for (size_t i = 0; i < nm; i++) for (size_t j = 0; j < nm; j++) st->W_l[(i + 1) * nw + j + nm] = SC_ONE;   /* :173-175 */
    for (size_t i = 0; i < nm; i++) st->W_l[(i + 1) * nw + i + nm] = SC_ZERO;   /* :177 */
    int ok = 1;
    for (size_t i = 0; i < nm && ok; i++)
        for (size_t j = 0; j < no; j++) {                                       /* :179-183: recomputed per (i, j), as the reference */
            sc js, inv;
            sc_from_u64(&js, (uint32_t)j);
            sc_add(&t, e, &js);
            if (DLOG_MODE && i > 0) { st->W_l[(i + 1) * nw + j + 2 * nm] = st->W_l[1 * nw + j + 2 * nm]; continue; } /* setup-only shortcut */
            if (!sc_inv_vartime(&inv, &t)) { ok = 0; break; }
            sc_minus(&st->W_l[(i + 1) * nw + j + 2 * nm], &inv);
        }
    /* partition: LL & index < dim_np -> Some(index) (:186-192) */
    st->part[PART_LO] = malloc(sizeof(int32_t) * nv);
    st->part[PART_LL] = malloc(sizeof(int32_t) * nv);
    st->part[PART_LR] = malloc(sizeof(int32_t) * nv);
    st->part[PART_NO] = malloc(sizeof(int32_t) * nm);
    for (size_t j = 0; j < nv; j++) { st->part[PART_LO][j] = -1; st->part[PART_LR][j] = -1; st->part[PART_LL][j] = j < no ? (int32_t)j : -1; }
    for (size_t j = 0; j < nm; j++) st->part[PART_NO][j] = -1;
    c->dim_nm = nm; c->dim_no = no; c->k = 1; c->dim_nl = nl; c->dim_nv = nv; c->dim_nw = nw;
    c->g = p->g;
    c->g_vec = p->g_vec; c->n_g_vec = p->n_g_vec;
    c->h_vec = p->h_vec; c->n_h_vec = p->n_h_vec;
    c->W_m = st->W_m; c->W_l = st->W_l; c->a_m = st->a_m; c->a_l = st->a_l;
    c->f_l = 1; c->f_m = 0;
    c->g_vec_ = p->g_vec_; c->n_g_vec_ = p->n_g_vec_;
    c->h_vec_ = p->h_vec_; c->n_h_vec_ = p->n_h_vec_;
    for (int k = 0; k < 4; k++) c->part[k] = st->part[k];
    return ok;
}
static void circuit_store_free(circuit_store* st) {
    free(st->W_m); free(st->W_l); free(st->a_m); free(st->a_l);
    for (int k = 0; k < 4; k++) free(st->part[k]);
}
static pt reciprocal_commit_value(const reciprocal_t* p, const sc* x, const sc* s) { /* reciprocal.rs:88-90 */
    pt a, b;
    pt_mul(&a, &p->g, x);
    pt_mul(&b, &p->h_vec[0], s);
    pt_add(&a, &a, &b);
    return a;
}
static pt reciprocal_commit_poles(const reciprocal_t* p, const sc* r, size_t nr, const sc* s) { /* reciprocal.rs:93-95 */
    pt a, b;
    pt_mul(&a, &p->h_vec[0], s);
    b = vector_mul_pt(p->h_vec + 9, p->n_h_vec - 9, r, nr);
    pt_add(&a, &a, &b);
    return a;
}
/* reciprocal.rs:98-107 */
static int reciprocal_verify(const reciprocal_t* p, const pt* commitment, const circuit_proof_t* cp, const pt* proof_r, transcript* t, vtrace* tr) {
    app_point("reciprocal_commitment", commitment, t);
    sc e;
    if (!get_challenge("reciprocal_challenge", t, &e)) return ORACLE_ERR_DEGENERATE;
    circuit_t c;
    circuit_store st;
    int rc;
    if (!reciprocal_make_circuit(p, &e, &c, &st)) rc = ORACLE_ERR_DEGENERATE;
    else {
        pt cc;
        pt_add(&cc, commitment, proof_r);
        if (tr) { tr->chal[0] = e; tr->pts[0] = cc; tr->n_c = 1; tr->n_y = 0; }
        rc = circuit_verify(&c, &cc, 1, t, cp, tr);
    }
    circuit_store_free(&st);
    return rc;
}
/* reciprocal.rs:110-146 */
static int reciprocal_prove(const reciprocal_t* p, const pt* commitment, const sc* x, const sc* s, const sc* m, const sc* digits,
                            transcript* t, rng_t* rng, circuit_proof_t* cp, pt* proof_r) {
    app_point("reciprocal_commitment", commitment, t);
    sc e;
    if (!get_challenge("reciprocal_challenge", t, &e)) return ORACLE_ERR_DEGENERATE;
    size_t nd = p->dim_nd;
    sc* r = malloc(sizeof(sc) * nd);
    sc* v = malloc(sizeof(sc) * (nd + 1));
    int rc = 1;
    for (size_t i = 0; i < nd; i++) {
        sc t1;
        sc_add(&t1, &digits[i], &e);
        if (!sc_inv_fermat(&r[i], &t1)) { rc = ORACLE_ERR_DEGENERATE; break; }   /* :118 uses constant-time invert() */
    }
    if (rc == 1) {
        sc r_blind = rng_draw(rng);
        *proof_r = reciprocal_commit_poles(p, r, nd, &r_blind);
        v[0] = *x;
        memcpy(v + 1, r, sizeof(sc) * nd);
        circuit_t c;
        circuit_store st;
        if (!reciprocal_make_circuit(p, &e, &c, &st)) rc = ORACLE_ERR_DEGENERATE;
        else {
            sc sv;
            sc_add(&sv, s, &r_blind);
            cwitness w = {v, nd + 1, &sv, digits, r, m};
            pt ccom = circuit_commit(&c, v, nd + 1, &sv);
            rc = circuit_prove(&c, &ccom, 1, &w, t, rng, cp);
        }
        circuit_store_free(&st);
    }
    free(r); free(v);
    return rc;
}

/* ------------------------------------------------------------------ range_proof/u64_proof.rs */
typedef struct { pt g, g_vec[16], h_vec[32]; } u64_pub;
static int u64_pub_load(u64_pub* pub, const uint8_t gens[49 * 64]) {
    if (!pt_from_xy64(&pub->g, gens)) return 0;
    for (int i = 0; i < 16; i++) if (!pt_from_xy64(&pub->g_vec[i], gens + 64 * (1 + i))) return 0;
    for (int i = 0; i < 32; i++) if (!pt_from_xy64(&pub->h_vec[i], gens + 64 * (17 + i))) return 0;
    return 1;
}
static reciprocal_t u64_reciprocal(const u64_pub* pub) { /* u64_proof.rs:43-51 */
    reciprocal_t r = {16, 16, pub->g, pub->g_vec, 16, pub->h_vec, 26, NULL, 0, pub->h_vec + 26, 6};
    return r;
}
static int u64_proof_load(circuit_proof_t* cp, pt* proof_r, const uint8_t b[928]) {
    pt pts[13];
    for (int i = 0; i < 13; i++) if (!pt_from_xy64(&pts[i], b + 64 * i)) return 0;
    cp->c_l = pts[0]; cp->c_r = pts[1]; cp->c_o = pts[2]; cp->c_s = pts[3];
    for (int i = 0; i < 4; i++) { cp->r[i] = pts[4 + i]; cp->x[i] = pts[8 + i]; }
    cp->nr = cp->nx = 4;
    *proof_r = pts[12];
    cp->nl = 2; cp->nn = 1;
    if (!sc_from_be(&cp->l[0], b + 832) || !sc_from_be(&cp->l[1], b + 864) || !sc_from_be(&cp->n[0], b + 896)) return 0;
    return 1;
}
static int u64_proof_store(uint8_t b[928], const circuit_proof_t* cp, const pt* proof_r) {
    if (cp->nr != 4 || cp->nx != 4 || cp->nl != 2 || cp->nn != 1) return 0;
    pt_to_xy64(b, &cp->c_l); pt_to_xy64(b + 64, &cp->c_r); pt_to_xy64(b + 128, &cp->c_o); pt_to_xy64(b + 192, &cp->c_s);
    for (int i = 0; i < 4; i++) { pt_to_xy64(b + 64 * (4 + i), &cp->r[i]); pt_to_xy64(b + 64 * (8 + i), &cp->x[i]); }
    pt_to_xy64(b + 64 * 12, proof_r);
    sc_to_be(b + 832, &cp->l[0]); sc_to_be(b + 864, &cp->l[1]); sc_to_be(b + 896, &cp->n[0]);
    return 1;
}
static void trace_store(uint8_t out[704], const vtrace* tr) {
    for (int i = 0; i < 10; i++) sc_to_be(out + 32 * i, &tr->chal[i]);
    for (int i = 0; i < 6; i++) {
        if (i < tr->n_c) pt_to_xy64(out + 320 + 64 * i, &tr->pts[i]);
        else memset(out + 320 + 64 * i, 0, 64);
    }
}

/* ================================================================== exported (ctypes) API */
#define API __attribute__((visibility("default")))

/* u64_proof.rs:37-39 */
API int bppp_oracle_u64_commit_value(const uint8_t gens[49 * 64], uint64_t x, const uint8_t s[32], uint8_t out[64]) {
    u64_pub pub;
    sc xs, ss;
    if (!u64_pub_load(&pub, gens) || !sc_from_be(&ss, s)) return ORACLE_ERR_ENCODING;
    sc_from_u64(&xs, x);
    reciprocal_t r = u64_reciprocal(&pub);
    pt c = reciprocal_commit_value(&r, &xs, &ss);
    pt_to_xy64(out, &c);
    return 0;
}

/* u64_proof.rs:42-54.  Returns 1 accept, 0 reject, <0 error.  trace (704 B) optional:
 * 10 challenges (e, rho, lambda, beta, delta, tau, y1..y4) then 6 points (V+r, C0..C4) as 64-B affine. */
API int bppp_oracle_u64_verify(const uint8_t gens[49 * 64], const uint8_t* label, size_t label_len, const uint8_t V[64],
                               const uint8_t proof[928], uint8_t* trace) {
    u64_pub pub;
    circuit_proof_t cp;
    pt proof_r, v;
    if (!u64_pub_load(&pub, gens) || !pt_from_xy64(&v, V) || !u64_proof_load(&cp, &proof_r, proof)) return ORACLE_ERR_ENCODING;
    reciprocal_t r = u64_reciprocal(&pub);
    transcript t;
    t_new(&t, label, label_len);
    vtrace tr;
    memset(&tr, 0, sizeof tr);
    int rc = reciprocal_verify(&r, &v, &cp, &proof_r, &t, trace ? &tr : NULL);
    if (trace) trace_store(trace, &tr);
    return rc;
}

/* u64_proof.rs:57-102.  rnd = 52 scalars (32 B BE each) in the reference's generate_biased draw order. */
API int bppp_oracle_u64_prove(const uint8_t gens[49 * 64], const uint8_t* label, size_t label_len, uint64_t x, const uint8_t s[32],
                              const uint8_t* rnd, size_t n_rnd, uint8_t proof_out[928], uint8_t V_out[64]) {
    u64_pub pub;
    sc ss, xs, digits[16], poles[16], draws[64];
    if (!u64_pub_load(&pub, gens) || !sc_from_be(&ss, s) || n_rnd > 64) return ORACLE_ERR_ENCODING;
    for (size_t i = 0; i < n_rnd; i++) if (!sc_from_be(&draws[i], rnd + 32 * i)) return ORACLE_ERR_ENCODING;
    uint64_t xx = x;
    for (int i = 0; i < 16; i++) poles[i] = SC_ZERO;
    for (int i = 0; i < 16; i++) {                                              /* u64_proof.rs:84-102 */
        sc_from_u64(&digits[i], xx % 16);
        sc_add(&poles[xx % 16], &poles[xx % 16], &SC_ONE);
        xx /= 16;
    }
    sc_from_u64(&xs, x);
    reciprocal_t r = u64_reciprocal(&pub);
    pt com = reciprocal_commit_value(&r, &xs, &ss);
    transcript t;
    t_new(&t, label, label_len);
    rng_t rng = {draws, n_rnd};
    circuit_proof_t cp;
    pt proof_r;
    int rc = reciprocal_prove(&r, &com, &xs, &ss, poles, digits, &t, &rng, &cp, &proof_r);
    if (rc != 1) return rc;
    if (!u64_proof_store(proof_out, &cp, &proof_r)) return ORACLE_ERR_ENCODING;
    pt_to_xy64(V_out, &com);
    return 0;
}

/* Generic wnla.rs entry points over byte buffers (tests.rs:139-171 shape). */
API int bppp_oracle_wnla_commit(const uint8_t* g, const uint8_t* g_vec, size_t ng, const uint8_t* h_vec, size_t nh, const uint8_t* c,
                                size_t nc, const uint8_t rho[32], const uint8_t mu[32], const uint8_t* l, size_t nl,
                                const uint8_t* n, size_t nn, uint8_t out[64]) {
    wnla_t w;
    int ok = pt_from_xy64(&w.g, g);
    w.g_vec = malloc(sizeof(pt) * (ng + 1)); w.h_vec = malloc(sizeof(pt) * (nh + 1)); w.c = malloc(sizeof(sc) * (nc + 1));
    sc *lv = malloc(sizeof(sc) * (nl + 1)), *nv = malloc(sizeof(sc) * (nn + 1));
    w.ng = ng; w.nh = nh; w.nc = nc;
    for (size_t i = 0; i < ng; i++) ok &= pt_from_xy64(&w.g_vec[i], g_vec + 64 * i);
    for (size_t i = 0; i < nh; i++) ok &= pt_from_xy64(&w.h_vec[i], h_vec + 64 * i);
    for (size_t i = 0; i < nc; i++) ok &= sc_from_be(&w.c[i], c + 32 * i);
    for (size_t i = 0; i < nl; i++) ok &= sc_from_be(&lv[i], l + 32 * i);
    for (size_t i = 0; i < nn; i++) ok &= sc_from_be(&nv[i], n + 32 * i);
    ok &= sc_from_be(&w.rho, rho) & sc_from_be(&w.mu, mu);
    if (ok) { pt r = wnla_commit(&w, lv, nl, nv, nn); pt_to_xy64(out, &r); }
    free(w.g_vec); free(w.h_vec); free(w.c); free(lv); free(nv);
    return ok ? 0 : ORACLE_ERR_ENCODING;
}
/* proof buffers: r (n_rounds x 64), x (n_rounds x 64), l (nl_out x 32), n (nn_out x 32) */
API int bppp_oracle_wnla_prove(const uint8_t* g, const uint8_t* g_vec, size_t ng, const uint8_t* h_vec, size_t nh, const uint8_t* c,
                               size_t nc, const uint8_t rho[32], const uint8_t mu[32], const uint8_t* label, size_t label_len,
                               const uint8_t commitment[64], const uint8_t* l, size_t nl, const uint8_t* n, size_t nn,
                               uint8_t* r_out, uint8_t* x_out, size_t* n_rounds, uint8_t* l_out, size_t* nl_out, uint8_t* n_out,
                               size_t* nn_out) {
    wnla_t w;
    pt com;
    int ok = pt_from_xy64(&w.g, g) & pt_from_xy64(&com, commitment);
    w.g_vec = malloc(sizeof(pt) * (ng + 1)); w.h_vec = malloc(sizeof(pt) * (nh + 1)); w.c = malloc(sizeof(sc) * (nc + 1));
    sc *lv = malloc(sizeof(sc) * (nl + 1)), *nv = malloc(sizeof(sc) * (nn + 1));
    w.ng = ng; w.nh = nh; w.nc = nc;
    for (size_t i = 0; i < ng; i++) ok &= pt_from_xy64(&w.g_vec[i], g_vec + 64 * i);
    for (size_t i = 0; i < nh; i++) ok &= pt_from_xy64(&w.h_vec[i], h_vec + 64 * i);
    for (size_t i = 0; i < nc; i++) ok &= sc_from_be(&w.c[i], c + 32 * i);
    for (size_t i = 0; i < nl; i++) ok &= sc_from_be(&lv[i], l + 32 * i);
    for (size_t i = 0; i < nn; i++) ok &= sc_from_be(&nv[i], n + 32 * i);
    ok &= sc_from_be(&w.rho, rho) & sc_from_be(&w.mu, mu);
    int rc = ORACLE_ERR_ENCODING;
    if (ok) {
        pt rr[64], xx[64];
        sc lo[8], no[8];
        wnla_proof_t wp = {rr, xx, 0, 0, lo, 0, no, 0};
        transcript t;
        t_new(&t, label, label_len);
        rc = wnla_prove(&w, &com, &t, lv, nl, nv, nn, &wp);
        if (rc == 1) {
            for (size_t i = 0; i < wp.nr; i++) { pt_to_xy64(r_out + 64 * i, &rr[i]); pt_to_xy64(x_out + 64 * i, &xx[i]); }
            for (size_t i = 0; i < wp.nl; i++) sc_to_be(l_out + 32 * i, &lo[i]);
            for (size_t i = 0; i < wp.nn; i++) sc_to_be(n_out + 32 * i, &no[i]);
            *n_rounds = wp.nr; *nl_out = wp.nl; *nn_out = wp.nn;
            rc = 0;
        }
    }
    free(w.g_vec); free(w.h_vec); free(w.c); free(lv); free(nv);
    return rc;
}
API int bppp_oracle_wnla_verify(const uint8_t* g, const uint8_t* g_vec, size_t ng, const uint8_t* h_vec, size_t nh, const uint8_t* c,
                                size_t nc, const uint8_t rho[32], const uint8_t mu[32], const uint8_t* label, size_t label_len,
                                const uint8_t commitment[64], const uint8_t* r_in, const uint8_t* x_in, size_t n_rounds,
                                const uint8_t* l, size_t nl, const uint8_t* n, size_t nn) {
    wnla_t w;
    pt com;
    if (n_rounds > 64) return ORACLE_ERR_ENCODING;
    int ok = pt_from_xy64(&w.g, g) & pt_from_xy64(&com, commitment);
    w.g_vec = malloc(sizeof(pt) * (ng + 1)); w.h_vec = malloc(sizeof(pt) * (nh + 1)); w.c = malloc(sizeof(sc) * (nc + 1));
    sc *lv = malloc(sizeof(sc) * (nl + 1)), *nv = malloc(sizeof(sc) * (nn + 1));
    pt rr[64], xx[64];
    w.ng = ng; w.nh = nh; w.nc = nc;
    for (size_t i = 0; i < ng; i++) ok &= pt_from_xy64(&w.g_vec[i], g_vec + 64 * i);
    for (size_t i = 0; i < nh; i++) ok &= pt_from_xy64(&w.h_vec[i], h_vec + 64 * i);
    for (size_t i = 0; i < nc; i++) ok &= sc_from_be(&w.c[i], c + 32 * i);
    for (size_t i = 0; i < nl; i++) ok &= sc_from_be(&lv[i], l + 32 * i);
    for (size_t i = 0; i < nn; i++) ok &= sc_from_be(&nv[i], n + 32 * i);
    for (size_t i = 0; i < n_rounds; i++) ok &= pt_from_xy64(&rr[i], r_in + 64 * i) & pt_from_xy64(&xx[i], x_in + 64 * i);
    ok &= sc_from_be(&w.rho, rho) & sc_from_be(&w.mu, mu);
    int rc = ORACLE_ERR_ENCODING;
    if (ok) {
        wnla_proof_t wp = {rr, xx, n_rounds, n_rounds, lv, nl, nv, nn};
        transcript t;
        t_new(&t, label, label_len);
        rc = wnla_verify(&w, &com, &t, &wp, NULL);
    }
    free(w.g_vec); free(w.h_vec); free(w.c); free(lv); free(nv);
    return rc;
}

/* Merlin known-answer hook: Transcript::new(label); append_message(l1, m1); challenge_bytes(l2, out) */
API void bppp_oracle_merlin_kat(const uint8_t* label, size_t label_len, const char* l1, const uint8_t* m1, size_t m1_len,
                                const char* l2, uint8_t* out, size_t out_len) {
    transcript t;
    t_new(&t, label, label_len);
    t_append(&t, l1, m1, (uint32_t)m1_len);
    t_challenge_bytes(&t, l2, out, (uint32_t)out_len);
}
/* k*P for parity tests of the point arithmetic (P = 64 zero bytes means the secp256k1 base point G) */
API int bppp_oracle_point_mul(const uint8_t P[64], const uint8_t k[32], uint8_t out[64]) {
    static const uint8_t GXY[64] = {
        0x79, 0xBE, 0x66, 0x7E, 0xF9, 0xDC, 0xBB, 0xAC, 0x55, 0xA0, 0x62, 0x95, 0xCE, 0x87, 0x0B, 0x07, 0x02, 0x9B, 0xFC, 0xDB, 0x2D, 0xCE,
        0x28, 0xD9, 0x59, 0xF2, 0x81, 0x5B, 0x16, 0xF8, 0x17, 0x98, 0x48, 0x3A, 0xDA, 0x77, 0x26, 0xA3, 0xC4, 0x65, 0x5D, 0xA4, 0xFB, 0xFC,
        0x0E, 0x11, 0x08, 0xA8, 0xFD, 0x17, 0xB4, 0x48, 0xA6, 0x85, 0x54, 0x19, 0x9C, 0x47, 0xD0, 0x8F, 0xFB, 0x10, 0xD4, 0xB8};
    int allz = 1;
    for (int i = 0; i < 64; i++) if (P[i]) allz = 0;
    pt p, r;
    sc ks;
    if (!pt_from_xy64(&p, allz ? GXY : P) || !sc_from_be(&ks, k)) return ORACLE_ERR_ENCODING;
    pt_mul(&r, &p, &ks);
    pt_to_xy64(out, &r);
    return 0;
}
API int bppp_oracle_point_add(const uint8_t A[64], const uint8_t B[64], uint8_t out[64]) {
    pt a, b, r;
    if (!pt_from_xy64(&a, A) || !pt_from_xy64(&b, B)) return ORACLE_ERR_ENCODING;
    pt_add(&r, &a, &b);
    pt_to_xy64(out, &r);
    return 0;
}
API int bppp_oracle_scalar_inv(const uint8_t a[32], uint8_t out_fermat[32], uint8_t out_vartime[32]) {
    sc s, r1, r2;
    if (!sc_from_be(&s, a)) return ORACLE_ERR_ENCODING;
    if (!sc_inv_fermat(&r1, &s) || !sc_inv_vartime(&r2, &s)) return ORACLE_ERR_DEGENERATE;
    sc_to_be(out_fermat, &r1);
    sc_to_be(out_vartime, &r2);
    return 0;
}

/* Generic ReciprocalRangeProofProtocol (reciprocal.rs:64-146) over byte buffers: any dim_nd / dim_np.
 * h_vec has dim_nd + 10 points, g_vec dim_nd points; g_vec_ / h_vec_ are the WNLA padding generators.
 * Proof wire layout (generic): c_l, c_r, c_o, c_s | r[rounds] | x[rounds] | reciprocal r   (64 B each)   then l[nl] | n[nn] (32 B each). */
static int recip_load(reciprocal_t* r, pt** store, const uint8_t* g, const uint8_t* g_vec, size_t dim_nd, size_t dim_np, const uint8_t* h_vec,
                      const uint8_t* g_vec_, size_t ng_, const uint8_t* h_vec_, size_t nh_) {
    size_t nh = dim_nd + 10;
    pt* a = (pt*)malloc(sizeof(pt) * (dim_nd + nh + ng_ + nh_ + 1));
    *store = a;
    int ok = pt_from_xy64(&r->g, g);
    for (size_t i = 0; i < dim_nd; i++) ok &= pt_from_xy64(&a[i], g_vec + 64 * i);
    for (size_t i = 0; i < nh; i++) ok &= pt_from_xy64(&a[dim_nd + i], h_vec + 64 * i);
    for (size_t i = 0; i < ng_; i++) ok &= pt_from_xy64(&a[dim_nd + nh + i], g_vec_ + 64 * i);
    for (size_t i = 0; i < nh_; i++) ok &= pt_from_xy64(&a[dim_nd + nh + ng_ + i], h_vec_ + 64 * i);
    r->dim_nd = dim_nd; r->dim_np = dim_np;
    r->g_vec = a; r->n_g_vec = dim_nd;
    r->h_vec = a + dim_nd; r->n_h_vec = nh;
    r->g_vec_ = a + dim_nd + nh; r->n_g_vec_ = ng_;
    r->h_vec_ = a + dim_nd + nh + ng_; r->n_h_vec_ = nh_;
    return ok;
}
/* x: value as a 32-byte scalar; digits: dim_nd scalars (base dim_np digits of x); m: dim_np multiplicities; rnd: 20 + 2*dim_nd draws.
 * Outputs: commitment (64 B), proof buffer, and the proof shape. */
API int bppp_oracle_reciprocal_prove(const uint8_t* g, const uint8_t* g_vec, size_t dim_nd, size_t dim_np, const uint8_t* h_vec,
                                     const uint8_t* g_vec_, size_t ng_, const uint8_t* h_vec_, size_t nh_, const uint8_t* label,
                                     size_t label_len, const uint8_t x[32], const uint8_t s[32], const uint8_t* digits, const uint8_t* m,
                                     const uint8_t* rnd, size_t n_rnd, uint8_t commitment_out[64], uint8_t* proof_out, size_t* rounds,
                                     size_t* nl, size_t* nn) {
    reciprocal_t r;
    pt* store;
    int ok = recip_load(&r, &store, g, g_vec, dim_nd, dim_np, h_vec, g_vec_, ng_, h_vec_, nh_);
    sc xs, ss;
    sc *dg = malloc(sizeof(sc) * dim_nd), *ms = malloc(sizeof(sc) * dim_np), *draws = malloc(sizeof(sc) * (n_rnd + 1));
    ok &= sc_from_be(&xs, x) & sc_from_be(&ss, s);
    for (size_t i = 0; i < dim_nd; i++) ok &= sc_from_be(&dg[i], digits + 32 * i);
    for (size_t i = 0; i < dim_np; i++) ok &= sc_from_be(&ms[i], m + 32 * i);
    for (size_t i = 0; i < n_rnd; i++) ok &= sc_from_be(&draws[i], rnd + 32 * i);
    int rc = ORACLE_ERR_ENCODING;
    if (ok) {
        pt com = reciprocal_commit_value(&r, &xs, &ss);
        transcript t;
        t_new(&t, label, label_len);
        rng_t rng = {draws, n_rnd};
        circuit_proof_t cp;
        pt proof_r;
        rc = reciprocal_prove(&r, &com, &xs, &ss, ms, dg, &t, &rng, &cp, &proof_r);
        if (rc == 1) {
            uint8_t* o = proof_out;
            pt_to_xy64(o, &cp.c_l); pt_to_xy64(o + 64, &cp.c_r); pt_to_xy64(o + 128, &cp.c_o); pt_to_xy64(o + 192, &cp.c_s);
            o += 256;
            for (size_t i = 0; i < cp.nr; i++, o += 64) pt_to_xy64(o, &cp.r[i]);
            for (size_t i = 0; i < cp.nx; i++, o += 64) pt_to_xy64(o, &cp.x[i]);
            pt_to_xy64(o, &proof_r); o += 64;
            for (size_t i = 0; i < cp.nl; i++, o += 32) sc_to_be(o, &cp.l[i]);
            for (size_t i = 0; i < cp.nn; i++, o += 32) sc_to_be(o, &cp.n[i]);
            *rounds = cp.nr; *nl = cp.nl; *nn = cp.nn;
            pt_to_xy64(commitment_out, &com);
            rc = 0;
        }
    }
    free(store); free(dg); free(ms); free(draws);
    return rc;
}
API int bppp_oracle_reciprocal_verify(const uint8_t* g, const uint8_t* g_vec, size_t dim_nd, size_t dim_np, const uint8_t* h_vec,
                                      const uint8_t* g_vec_, size_t ng_, const uint8_t* h_vec_, size_t nh_, const uint8_t* label,
                                      size_t label_len, const uint8_t commitment[64], const uint8_t* proof, size_t rounds, size_t nl,
                                      size_t nn) {
    if (rounds > 64 || nl > 8 || nn > 8) return ORACLE_ERR_ENCODING;
    reciprocal_t r;
    pt* store;
    int ok = recip_load(&r, &store, g, g_vec, dim_nd, dim_np, h_vec, g_vec_, ng_, h_vec_, nh_);
    circuit_proof_t cp;
    pt proof_r, com;
    ok &= pt_from_xy64(&com, commitment);
    const uint8_t* o = proof;
    ok &= pt_from_xy64(&cp.c_l, o) & pt_from_xy64(&cp.c_r, o + 64) & pt_from_xy64(&cp.c_o, o + 128) & pt_from_xy64(&cp.c_s, o + 192);
    o += 256;
    for (size_t i = 0; i < rounds; i++, o += 64) ok &= pt_from_xy64(&cp.r[i], o);
    for (size_t i = 0; i < rounds; i++, o += 64) ok &= pt_from_xy64(&cp.x[i], o);
    ok &= pt_from_xy64(&proof_r, o); o += 64;
    for (size_t i = 0; i < nl; i++, o += 32) ok &= sc_from_be(&cp.l[i], o);
    for (size_t i = 0; i < nn; i++, o += 32) ok &= sc_from_be(&cp.n[i], o);
    cp.nr = cp.nx = rounds; cp.nl = nl; cp.nn = nn;
    int rc = ORACLE_ERR_ENCODING;
    if (ok) {
        transcript t;
        t_new(&t, label, label_len);
        rc = reciprocal_verify(&r, &com, &cp, &proof_r, &t, NULL);
    }
    free(store);
    return rc;
}

/* ------------------------------------------------------------------ generic ArithmeticCircuit exports (circuit.rs:95-556)
 * Circuit description as flat arrays: dims = {dim_nm, dim_no, k, dim_nl, dim_nv, dim_nw}; W_m (nm x nw), W_l (nl x nw), a_m, a_l
 * as 32-byte big-endian scalars, row-major; the partition closure as four index tables (-1 = None): LO, LL, LR (dim_nv
 * entries each) and NO (dim_nm entries).  Generators: g, g_vec (dim_nm), h_vec (9 + dim_nv), padding g_vec_ / h_vec_.
 * Proof layout: c_l, c_r, c_o, c_s | r[rounds] | x[rounds] | l[nl] | n[nn]. */
typedef struct { circuit_t c; pt* pts; sc* scs; } circ_loaded;
static int circ_load(circ_loaded* L, const uint8_t* g, const uint8_t* g_vec, const uint8_t* h_vec, const uint8_t* g_vec_, size_t ng_,
                     const uint8_t* h_vec_, size_t nh_, const size_t dims[6], int f_l, int f_m, const uint8_t* W_m, const uint8_t* W_l,
                     const uint8_t* a_m, const uint8_t* a_l, const int32_t* part_lo, const int32_t* part_ll, const int32_t* part_lr,
                     const int32_t* part_no) {
    size_t nm = dims[0], no = dims[1], k = dims[2], nl = dims[3], nv = dims[4], nw = dims[5], nh = nv + 9;
    circuit_t* c = &L->c;
    L->pts = (pt*)malloc(sizeof(pt) * (nm + nh + ng_ + nh_ + 1));
    L->scs = (sc*)malloc(sizeof(sc) * (nm * nw + nl * nw + nm + nl + 1));
    int ok = pt_from_xy64(&c->g, g);
    pt* a = L->pts;
    for (size_t i = 0; i < nm; i++) ok &= pt_from_xy64(&a[i], g_vec + 64 * i);
    for (size_t i = 0; i < nh; i++) ok &= pt_from_xy64(&a[nm + i], h_vec + 64 * i);
    for (size_t i = 0; i < ng_; i++) ok &= pt_from_xy64(&a[nm + nh + i], g_vec_ + 64 * i);
    for (size_t i = 0; i < nh_; i++) ok &= pt_from_xy64(&a[nm + nh + ng_ + i], h_vec_ + 64 * i);
    sc* q = L->scs;
    sc *Wm = q, *Wl = q + nm * nw, *am = Wl + nl * nw, *al = am + nm;
    for (size_t i = 0; i < nm * nw; i++) ok &= sc_from_be(&Wm[i], W_m + 32 * i);
    for (size_t i = 0; i < nl * nw; i++) ok &= sc_from_be(&Wl[i], W_l + 32 * i);
    for (size_t i = 0; i < nm; i++) ok &= sc_from_be(&am[i], a_m + 32 * i);
    for (size_t i = 0; i < nl; i++) ok &= sc_from_be(&al[i], a_l + 32 * i);
    for (size_t j = 0; j < nv; j++) ok &= part_lo[j] < (int32_t)no && part_ll[j] < (int32_t)no && part_lr[j] < (int32_t)no;
    for (size_t j = 0; j < nm; j++) ok &= part_no[j] < (int32_t)no;
    c->dim_nm = nm; c->dim_no = no; c->k = k; c->dim_nl = nl; c->dim_nv = nv; c->dim_nw = nw;
    c->g_vec = a; c->n_g_vec = nm;
    c->h_vec = a + nm; c->n_h_vec = nh;
    c->g_vec_ = a + nm + nh; c->n_g_vec_ = ng_;
    c->h_vec_ = a + nm + nh + ng_; c->n_h_vec_ = nh_;
    c->W_m = Wm; c->W_l = Wl; c->a_m = am; c->a_l = al;
    c->f_l = f_l; c->f_m = f_m;
    c->part[PART_LO] = part_lo; c->part[PART_LL] = part_ll; c->part[PART_LR] = part_lr; c->part[PART_NO] = part_no;
    return ok && nw == 2 * nm + no;
}
static void circ_free(circ_loaded* L) { free(L->pts); free(L->scs); }
/* witness: v = k vectors of dim_nv scalars, s_v = k blinding scalars, w_l / w_r (dim_nm), w_o (dim_no); rnd = the prover's
 * Scalar::generate_biased draws in order.  Outputs the k commitments (circuit.rs:146-151) and the proof. */
API int bppp_oracle_circuit_prove(const uint8_t* g, const uint8_t* g_vec, const uint8_t* h_vec, const uint8_t* g_vec_, size_t ng_,
                                  const uint8_t* h_vec_, size_t nh_, const size_t dims[6], int f_l, int f_m, const uint8_t* W_m,
                                  const uint8_t* W_l, const uint8_t* a_m, const uint8_t* a_l, const int32_t* part_lo,
                                  const int32_t* part_ll, const int32_t* part_lr, const int32_t* part_no, const uint8_t* label,
                                  size_t label_len, const uint8_t* v, const uint8_t* s_v, const uint8_t* w_l, const uint8_t* w_r,
                                  const uint8_t* w_o, const uint8_t* rnd, size_t n_rnd, uint8_t* commitments_out, uint8_t* proof_out,
                                  size_t* rounds, size_t* nl_out, size_t* nn_out) {
    circ_loaded L;
    int ok = circ_load(&L, g, g_vec, h_vec, g_vec_, ng_, h_vec_, nh_, dims, f_l, f_m, W_m, W_l, a_m, a_l, part_lo, part_ll, part_lr, part_no);
    size_t nm = dims[0], no = dims[1], k = dims[2], nv = dims[4];
    sc* ws = (sc*)malloc(sizeof(sc) * (k * nv + k + 2 * nm + no + n_rnd + 1));
    sc *vv = ws, *sv = vv + k * nv, *wl = sv + k, *wr = wl + nm, *wo = wr + nm, *draws = wo + no;
    for (size_t i = 0; i < k * nv; i++) ok &= sc_from_be(&vv[i], v + 32 * i);
    for (size_t i = 0; i < k; i++) ok &= sc_from_be(&sv[i], s_v + 32 * i);
    for (size_t i = 0; i < nm; i++) ok &= sc_from_be(&wl[i], w_l + 32 * i) & sc_from_be(&wr[i], w_r + 32 * i);
    for (size_t i = 0; i < no; i++) ok &= sc_from_be(&wo[i], w_o + 32 * i);
    for (size_t i = 0; i < n_rnd; i++) ok &= sc_from_be(&draws[i], rnd + 32 * i);
    int rc = ORACLE_ERR_ENCODING;
    if (ok && k <= 64) {
        pt coms[64];
        for (size_t i = 0; i < k; i++) coms[i] = circuit_commit(&L.c, vv + i * nv, nv, &sv[i]);
        cwitness w = {vv, nv, sv, wl, wr, wo};
        transcript t;
        t_new(&t, label, label_len);
        rng_t rng = {draws, n_rnd};
        circuit_proof_t cp;
        rc = circuit_prove(&L.c, coms, k, &w, &t, &rng, &cp);
        if (rc == 1) {
            uint8_t* o = proof_out;
            pt_to_xy64(o, &cp.c_l); pt_to_xy64(o + 64, &cp.c_r); pt_to_xy64(o + 128, &cp.c_o); pt_to_xy64(o + 192, &cp.c_s);
            o += 256;
            for (size_t i = 0; i < cp.nr; i++, o += 64) pt_to_xy64(o, &cp.r[i]);
            for (size_t i = 0; i < cp.nx; i++, o += 64) pt_to_xy64(o, &cp.x[i]);
            for (size_t i = 0; i < cp.nl; i++, o += 32) sc_to_be(o, &cp.l[i]);
            for (size_t i = 0; i < cp.nn; i++, o += 32) sc_to_be(o, &cp.n[i]);
            *rounds = cp.nr; *nl_out = cp.nl; *nn_out = cp.nn;
            for (size_t i = 0; i < k; i++) pt_to_xy64(commitments_out + 64 * i, &coms[i]);
            rc = 0;
        }
    }
    free(ws);
    circ_free(&L);
    return rc;
}
/* -> 1 accept, 0 reject, < 0 error */
API int bppp_oracle_circuit_verify(const uint8_t* g, const uint8_t* g_vec, const uint8_t* h_vec, const uint8_t* g_vec_, size_t ng_,
                                   const uint8_t* h_vec_, size_t nh_, const size_t dims[6], int f_l, int f_m, const uint8_t* W_m,
                                   const uint8_t* W_l, const uint8_t* a_m, const uint8_t* a_l, const int32_t* part_lo,
                                   const int32_t* part_ll, const int32_t* part_lr, const int32_t* part_no, const uint8_t* label,
                                   size_t label_len, const uint8_t* commitments, const uint8_t* proof, size_t rounds, size_t nl,
                                   size_t nn) {
    if (rounds > 64 || nl > 8 || nn > 8 || dims[2] > 64) return ORACLE_ERR_ENCODING;
    circ_loaded L;
    int ok = circ_load(&L, g, g_vec, h_vec, g_vec_, ng_, h_vec_, nh_, dims, f_l, f_m, W_m, W_l, a_m, a_l, part_lo, part_ll, part_lr, part_no);
    size_t k = dims[2];
    pt coms[64];
    for (size_t i = 0; i < k; i++) ok &= pt_from_xy64(&coms[i], commitments + 64 * i);
    circuit_proof_t cp;
    const uint8_t* o = proof;
    ok &= pt_from_xy64(&cp.c_l, o) & pt_from_xy64(&cp.c_r, o + 64) & pt_from_xy64(&cp.c_o, o + 128) & pt_from_xy64(&cp.c_s, o + 192);
    o += 256;
    for (size_t i = 0; i < rounds; i++, o += 64) ok &= pt_from_xy64(&cp.r[i], o);
    for (size_t i = 0; i < rounds; i++, o += 64) ok &= pt_from_xy64(&cp.x[i], o);
    for (size_t i = 0; i < nl; i++, o += 32) ok &= sc_from_be(&cp.l[i], o);
    for (size_t i = 0; i < nn; i++, o += 32) ok &= sc_from_be(&cp.n[i], o);
    cp.nr = cp.nx = rounds; cp.nl = nl; cp.nn = nn;
    int rc = ORACLE_ERR_ENCODING;
    if (ok) {
        transcript t;
        t_new(&t, label, label_len);
        rc = circuit_verify(&L.c, coms, k, &t, &cp, NULL);
    }
    circ_free(&L);
    return rc;
}

/* ------------------------------------------------------------------ threaded batch drivers (CPU baseline + checker) */
typedef struct {
    const uint8_t *gens, *label; size_t label_len, n, stride_v, lo, hi;
    const uint8_t *V, *proofs; uint8_t* accept; int32_t* status;
} vjob;
static void* vjob_run(void* p) {
    vjob* j = (vjob*)p;
    for (size_t i = j->lo; i < j->hi; i++) {
        int rc = bppp_oracle_u64_verify(j->gens, j->label, j->label_len, j->V + 64 * i, j->proofs + 928 * i, NULL);
        j->accept[i] = rc == 1;
        if (j->status) j->status[i] = rc < 0 ? rc : 0;
    }
    return NULL;
}
/* Reference-shaped verify of n proofs on `nthreads` host threads (the timed CPU baseline, kind "port"). */
API int bppp_oracle_u64_verify_batch(const uint8_t gens[49 * 64], const uint8_t* label, size_t label_len, size_t n, const uint8_t* V,
                                     const uint8_t* proofs, uint8_t* accept, int32_t* status, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    vjob jobs[256];
    for (int k = 0; k < nthreads; k++) {
        vjob j = {gens, label, label_len, n, 64, n * k / nthreads, n * (k + 1) / nthreads, V, proofs, accept, status};
        jobs[k] = j;
        pthread_create(&th[k], NULL, vjob_run, &jobs[k]);
    }
    for (int k = 0; k < nthreads; k++) pthread_join(th[k], NULL);
    return 0;
}

typedef struct {
    const uint8_t *gens, *label; size_t label_len, lo, hi;
    const uint64_t* x; const uint8_t *s, *rnd; uint8_t *proofs, *V; int rc;
} pjob;
static void* pjob_run(void* p) {
    pjob* j = (pjob*)p;
    j->rc = 0;
    for (size_t i = j->lo; i < j->hi; i++) {
        int rc = bppp_oracle_u64_prove(j->gens, j->label, j->label_len, j->x[i], j->s + 32 * i, j->rnd + 52 * 32 * i, 52,
                                       j->proofs + 928 * i, j->V + 64 * i);
        if (rc) j->rc = rc;
    }
    return NULL;
}
/* Reference-shaped prove of n values on `nthreads` host threads. */
API int bppp_oracle_u64_prove_batch(const uint8_t gens[49 * 64], const uint8_t* label, size_t label_len, size_t n, const uint64_t* x,
                                    const uint8_t* s, const uint8_t* rnd, uint8_t* proofs, uint8_t* V, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    pjob jobs[256];
    for (int k = 0; k < nthreads; k++) {
        pjob j = {gens, label, label_len, n * k / nthreads, n * (k + 1) / nthreads, x, s, rnd, proofs, V, 0};
        jobs[k] = j;
        pthread_create(&th[k], NULL, pjob_run, &jobs[k]);
    }
    int rc = 0;
    for (int k = 0; k < nthreads; k++) { pthread_join(th[k], NULL); if (jobs[k].rc) rc = jobs[k].rc; }
    return rc;
}

/* ------------------------------------------------------------------ synthetic-workload ("trapdoor") batch prover */
typedef struct {
    const uint8_t *dlogs, *label; size_t label_len, lo, hi;
    const uint64_t* x; const uint8_t *s, *rnd; uint8_t *proofs, *V; int rc;
} tjob;
static void* tjob_run(void* p) {
    tjob* j = (tjob*)p;
    j->rc = 0;
    DLOG_MODE = 1;
    u64_pub pub;
    sc k;
    int ok = 1;
    ok &= sc_from_be(&k, j->dlogs); dl_set(&pub.g, &k);
    for (int i = 0; i < 16; i++) { ok &= sc_from_be(&k, j->dlogs + 32 * (1 + i)); dl_set(&pub.g_vec[i], &k); }
    for (int i = 0; i < 32; i++) { ok &= sc_from_be(&k, j->dlogs + 32 * (17 + i)); dl_set(&pub.h_vec[i], &k); }
    if (!ok) { j->rc = ORACLE_ERR_ENCODING; DLOG_MODE = 0; return NULL; }
    reciprocal_t r = u64_reciprocal(&pub);
    for (size_t idx = j->lo; idx < j->hi; idx++) {
        sc ss, xs, digits[16], poles[16], draws[52];
        int good = sc_from_be(&ss, j->s + 32 * idx);
        for (int i = 0; i < 52; i++) good &= sc_from_be(&draws[i], j->rnd + 52 * 32 * idx + 32 * i);
        if (!good) { j->rc = ORACLE_ERR_ENCODING; continue; }
        uint64_t xx = j->x[idx];
        for (int i = 0; i < 16; i++) poles[i] = SC_ZERO;
        for (int i = 0; i < 16; i++) { sc_from_u64(&digits[i], xx % 16); sc_add(&poles[xx % 16], &poles[xx % 16], &SC_ONE); xx /= 16; }
        sc_from_u64(&xs, j->x[idx]);
        pt com = reciprocal_commit_value(&r, &xs, &ss);
        transcript t;
        t_new(&t, j->label, j->label_len);
        rng_t rng = {draws, 52};
        circuit_proof_t cp;
        pt proof_r;
        int rc = reciprocal_prove(&r, &com, &xs, &ss, poles, digits, &t, &rng, &cp, &proof_r);
        if (rc != 1 || !u64_proof_store(j->proofs + 928 * idx, &cp, &proof_r)) { j->rc = rc == 1 ? ORACLE_ERR_ENCODING : rc; continue; }
        pt_to_xy64(j->V + 64 * idx, &com);
    }
    DLOG_MODE = 0;
    return NULL;
}
/* gen_dlogs: 49 x 32 B big-endian k_i with generator_i = k_i*G (order g, g_vec[16], h_vec[32]). */
API int bppp_oracle_u64_prove_trapdoor_batch(const uint8_t gen_dlogs[49 * 32], const uint8_t* label, size_t label_len, size_t n,
                                             const uint64_t* x, const uint8_t* s, const uint8_t* rnd, uint8_t* proofs, uint8_t* V,
                                             int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_once(&GTBL_ONCE, gtbl_build);
    pthread_t th[256];
    tjob jobs[256];
    for (int k = 0; k < nthreads; k++) {
        tjob j = {gen_dlogs, label, label_len, n * k / nthreads, n * (k + 1) / nthreads, x, s, rnd, proofs, V, 0};
        jobs[k] = j;
        pthread_create(&th[k], NULL, tjob_run, &jobs[k]);
    }
    int rc = 0;
    for (int k = 0; k < nthreads; k++) { pthread_join(th[k], NULL); if (jobs[k].rc) rc = jobs[k].rc; }
    return rc;
}
