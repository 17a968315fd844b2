// 256-bit modular arithmetic for secp256k1 on gfx950: Fp (coordinates) and Fn (scalars).
//
// Replaces, for the batch hot path, what the reference takes from k256 0.13.3 (`Scalar`,
// `FieldElement`; every `.mul/.add/.sub/.invert*` call site in /root/reference/src, e.g.
// util.rs:28-60, wnla.rs:96-102, circuit.rs:166-235).
//
// Representation: 8 x 32-bit little-endian limbs, always canonical (< modulus).  32-bit limbs because
// the CDNA4 integer multiplier is v_mad_u64_u32 (32x32+64 -> 64): one instruction per limb product, the
// 64-bit accumulator carries the running column sum.  No MFMA: this is carry-chained integer work.
// Everything is branch-free (selects), so a 64-lane wavefront never diverges on data.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HD __host__ __device__ __forceinline__
#define HD_NOINLINE __host__ __device__ __noinline__
#else
#define HD inline
#define HD_NOINLINE inline
#endif

namespace bppp {

typedef uint32_t u32;
typedef uint64_t u64;

struct fe { u32 v[8]; };  // mod p = 2^256 - 2^32 - 977
struct sc { u32 v[8]; };  // mod n (group order)

// ---------------------------------------------------------------- generic 256-bit helpers
HD u32 add256(u32 r[8], const u32 a[8], const u32 b[8]) {
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (u64)a[i] + b[i]; r[i] = (u32)c; c >>= 32; }
    return (u32)c;
}
HD u32 sub256(u32 r[8], const u32 a[8], const u32 b[8]) {
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 d = (u64)a[i] - b[i] - borrow;
        r[i] = (u32)d;
        borrow = (u32)(d >> 32) & 1;
    }
    return borrow;
}
HD void sel256(u32 r[8], u32 take_b, const u32 a[8], const u32 b[8]) {  // r = take_b ? b : a
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = take_b ? b[i] : a[i];
}
HD bool is_zero256(const u32 a[8]) {
    u32 x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x |= a[i];
    return x == 0;
}
HD bool eq256(const u32 a[8], const u32 b[8]) {
    u32 x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x |= a[i] ^ b[i];
    return x == 0;
}
// full 256x256 -> 512 product, operand scanning: 64 limb products, each one mad into a 64-bit carry word
HD void mul256(u32 t[16], const u32 a[8], const u32 b[8]) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { c += (u64)a[0] * b[j]; t[j] = (u32)c; c >>= 32; }
    t[8] = (u32)c;
#pragma unroll
    for (int i = 1; i < 8; i++) {
        c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { c += (u64)a[i] * b[j] + t[i + j]; t[i + j] = (u32)c; c >>= 32; }
        t[i + 8] = (u32)c;
    }
}
// big-endian 32 bytes <-> limbs
HD void be32_to_limbs(u32 r[8], const uint8_t* b) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint8_t* q = b + 4 * (7 - i);
        r[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
}
HD void limbs_to_be32(uint8_t* b, const u32 a[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint8_t* q = b + 4 * (7 - i);
        q[0] = (uint8_t)(a[i] >> 24); q[1] = (uint8_t)(a[i] >> 16); q[2] = (uint8_t)(a[i] >> 8); q[3] = (uint8_t)a[i];
    }
}

// ---------------------------------------------------------------- Fp: p = 2^256 - PC, PC = 2^32 + 977
#define BPPP_PC0 0x000003D1u  // low limb of PC; limb 1 of PC is 1

// r (8 limbs) + carry*2^256, value < 2^256 + small  ->  canonical.  x >= p  <=>  x + PC carries out of 2^256.
HD void fe_final(fe& r, const u32 x[8], u32 carry) {
    u32 t[8];
    u64 c = (u64)x[0] + BPPP_PC0; t[0] = (u32)c; c >>= 32;
    c += (u64)x[1] + 1; t[1] = (u32)c; c >>= 32;
#pragma unroll
    for (int i = 2; i < 8; i++) { c += x[i]; t[i] = (u32)c; c >>= 32; }
    u32 take = carry | (u32)c;
    sel256(r.v, take, x, t);
}
HD void fe_add(fe& r, const fe& a, const fe& b) {
    u32 s[8];
    u32 k = add256(s, a.v, b.v);
    fe_final(r, s, k);
}
HD void fe_sub(fe& r, const fe& a, const fe& b) {
    u32 d[8], e[8];
    u32 borrow = sub256(d, a.v, b.v);
    // d + p = d - PC (mod 2^256)
    u64 c = (u64)d[0] - BPPP_PC0; e[0] = (u32)c; u32 bw = (u32)(c >> 32) & 1;
    c = (u64)d[1] - 1 - bw; e[1] = (u32)c; bw = (u32)(c >> 32) & 1;
#pragma unroll
    for (int i = 2; i < 8; i++) { c = (u64)d[i] - bw; e[i] = (u32)c; bw = (u32)(c >> 32) & 1; }
    sel256(r.v, borrow, d, e);
}
HD void fe_neg(fe& r, const fe& a) {
    fe z;
#pragma unroll
    for (int i = 0; i < 8; i++) z.v[i] = 0;
    fe_sub(r, z, a);
}
HD void fe_dbl(fe& r, const fe& a) { fe_add(r, a, a); }
// reduce a 512-bit value: hi*2^256 + lo == hi*PC + lo
HD void fe_reduce512(fe& r, const u32 t[16]) {
    u32 s[8];
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        c += (u64)t[8 + j] * BPPP_PC0 + t[j];
        if (j > 0) c += t[8 + j - 1];
        s[j] = (u32)c;
        c >>= 32;
    }
    c += t[15];  // overflow word(s): < 2^34
    u64 top = c;
    // fold top * PC = top*977 + (top << 32)
    c = (u64)s[0] + (top & 0xFFFFFFFFu) * BPPP_PC0 + (((top >> 32) * BPPP_PC0) << 32);
    s[0] = (u32)c; c >>= 32;
    c += (u64)s[1] + (top & 0xFFFFFFFFu); s[1] = (u32)c; c >>= 32;
    c += (u64)s[2] + (top >> 32); s[2] = (u32)c; c >>= 32;
#pragma unroll
    for (int i = 3; i < 8; i++) { c += s[i]; s[i] = (u32)c; c >>= 32; }
    // a carry here means value = 2^256 + s with s tiny; fe_final adds PC once for the wrap (cannot wrap again)
    fe_final(r, s, (u32)c);
}
HD void fe_mul(fe& r, const fe& a, const fe& b) {
    u32 t[16];
    mul256(t, a.v, b.v);
    fe_reduce512(r, t);
}
HD void fe_sqr(fe& r, const fe& a) { fe_mul(r, a, a); }
HD void fe_mul_small(fe& r, const fe& a, u32 k) {  // k < 2^16
    u32 s[8];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (u64)a.v[i] * k; s[i] = (u32)c; c >>= 32; }
    u64 top = c;  // < 2^16
    c = (u64)s[0] + top * BPPP_PC0; s[0] = (u32)c; c >>= 32;
    c += (u64)s[1] + top; s[1] = (u32)c; c >>= 32;
#pragma unroll
    for (int i = 2; i < 8; i++) { c += s[i]; s[i] = (u32)c; c >>= 32; }
    fe_final(r, s, (u32)c);
}
HD bool fe_is_zero(const fe& a) { return is_zero256(a.v); }
HD bool fe_eq(const fe& a, const fe& b) { return eq256(a.v, b.v); }
HD void fe_set_u32(fe& r, u32 x) {
    r.v[0] = x;
#pragma unroll
    for (int i = 1; i < 8; i++) r.v[i] = 0;
}
HD void fe_cmov(fe& r, bool take, const fe& b) { sel256(r.v, take ? 1u : 0u, r.v, b.v); }
HD void fe_sqr_n(fe& r, const fe& a, int n) {
    r = a;
#pragma nounroll
    for (int i = 0; i < n; i++) fe_sqr(r, r);
}
// a^(p-2) (0 -> 0).  Addition chain on the run structure of p-2: 255 squarings + 15 multiplications.
HD_NOINLINE void fe_inv(fe& r, const fe& a) {
    fe x2, x3, x6, x9, x11, x22, x44, x88, x176, x220, x223, t;
    fe_sqr(x2, a); fe_mul(x2, x2, a);
    fe_sqr(x3, x2); fe_mul(x3, x3, a);
    fe_sqr_n(x6, x3, 3); fe_mul(x6, x6, x3);
    fe_sqr_n(x9, x6, 3); fe_mul(x9, x9, x3);
    fe_sqr_n(x11, x9, 2); fe_mul(x11, x11, x2);
    fe_sqr_n(x22, x11, 11); fe_mul(x22, x22, x11);
    fe_sqr_n(x44, x22, 22); fe_mul(x44, x44, x22);
    fe_sqr_n(x88, x44, 44); fe_mul(x88, x88, x44);
    fe_sqr_n(x176, x88, 88); fe_mul(x176, x176, x88);
    fe_sqr_n(x220, x176, 44); fe_mul(x220, x220, x44);
    fe_sqr_n(x223, x220, 3); fe_mul(x223, x223, x3);
    // p - 2 = 2^256 - 2^32 - 979: 223 ones, 0, 22 ones, 0000, 1, 0, 11, 0, 1  (low bits ...1111 1100 0010 1101)
    fe_sqr_n(t, x223, 23); fe_mul(t, t, x22);
    fe_sqr_n(t, t, 5); fe_mul(t, t, a);
    fe_sqr_n(t, t, 3); fe_mul(t, t, x2);
    fe_sqr_n(t, t, 2); fe_mul(r, t, a);
}
// a^((p+1)/4): square root when a is a quadratic residue (p = 3 mod 4).  253 squarings + 13 multiplications.
HD_NOINLINE void fe_sqrt_candidate(fe& r, const fe& a) {
    fe x2, x3, x6, x9, x11, x22, x44, x88, x176, x220, x223, t;
    fe_sqr(x2, a); fe_mul(x2, x2, a);
    fe_sqr(x3, x2); fe_mul(x3, x3, a);
    fe_sqr_n(x6, x3, 3); fe_mul(x6, x6, x3);
    fe_sqr_n(x9, x6, 3); fe_mul(x9, x9, x3);
    fe_sqr_n(x11, x9, 2); fe_mul(x11, x11, x2);
    fe_sqr_n(x22, x11, 11); fe_mul(x22, x22, x11);
    fe_sqr_n(x44, x22, 22); fe_mul(x44, x44, x22);
    fe_sqr_n(x88, x44, 44); fe_mul(x88, x88, x44);
    fe_sqr_n(x176, x88, 88); fe_mul(x176, x176, x88);
    fe_sqr_n(x220, x176, 44); fe_mul(x220, x220, x44);
    fe_sqr_n(x223, x220, 3); fe_mul(x223, x223, x3);
    // (p+1)/4 = 2^254 - 2^30 - 244: 223 ones, 0, 22 ones, 0000, 11, 00
    fe_sqr_n(t, x223, 23); fe_mul(t, t, x22);
    fe_sqr_n(t, t, 6); fe_mul(t, t, x2);
    fe_sqr_n(r, t, 2);
}
// big-endian bytes -> canonical element; false if >= p
HD bool fe_from_be(fe& r, const uint8_t* b) {
    be32_to_limbs(r.v, b);
    u32 t[8];
    u64 c = (u64)r.v[0] + BPPP_PC0; t[0] = (u32)c; c >>= 32;
    c += (u64)r.v[1] + 1; t[1] = (u32)c; c >>= 32;
#pragma unroll
    for (int i = 2; i < 8; i++) { c += r.v[i]; t[i] = (u32)c; c >>= 32; }
    (void)t;
    return c == 0;
}
HD void fe_to_be(uint8_t* b, const fe& a) { limbs_to_be32(b, a.v); }

// ---------------------------------------------------------------- Fn: n = 2^256 - ND, ND = 0x1_45512319_50B75FC4_402DA173_2FC9BEBF
#define BPPP_ND0 0x2FC9BEBFu
#define BPPP_ND1 0x402DA173u
#define BPPP_ND2 0x50B75FC4u
#define BPPP_ND3 0x45512319u
#define BPPP_ND4 0x00000001u

HD void sc_final(sc& r, const u32 x[8], u32 carry) {  // x + carry*2^256 < 2n  ->  canonical
    const u32 nd[5] = {BPPP_ND0, BPPP_ND1, BPPP_ND2, BPPP_ND3, BPPP_ND4};
    u32 t[8];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (u64)x[i] + (i < 5 ? nd[i] : 0u); t[i] = (u32)c; c >>= 32; }
    u32 take = carry | (u32)c;
    sel256(r.v, take, x, t);
}
HD void sc_add(sc& r, const sc& a, const sc& b) {
    u32 s[8];
    u32 k = add256(s, a.v, b.v);
    sc_final(r, s, k);
}
HD void sc_sub(sc& r, const sc& a, const sc& b) {
    const u32 nd[5] = {BPPP_ND0, BPPP_ND1, BPPP_ND2, BPPP_ND3, BPPP_ND4};
    u32 d[8], e[8];
    u32 borrow = sub256(d, a.v, b.v);
    u32 bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 c = (u64)d[i] - (i < 5 ? nd[i] : 0u) - bw;
        e[i] = (u32)c;
        bw = (u32)(c >> 32) & 1;
    }
    sel256(r.v, borrow, d, e);
}
HD void sc_set_u32(sc& r, u32 x) {
    r.v[0] = x;
#pragma unroll
    for (int i = 1; i < 8; i++) r.v[i] = 0;
}
HD void sc_set_u64(sc& r, u64 x) {
    r.v[0] = (u32)x; r.v[1] = (u32)(x >> 32);
#pragma unroll
    for (int i = 2; i < 8; i++) r.v[i] = 0;
}
HD void sc_neg(sc& r, const sc& a) {
    sc z;
    sc_set_u32(z, 0);
    sc_sub(r, z, a);
}
// acc[0..] += hi[0..nh) * ND   (schoolbook, 5-limb ND), acc has room for nh+5 limbs (+ carry handled by caller sizes)
template <int NH, int NT>
HD void sc_fold(u32 out[NT], const u32 lo[8], const u32 hi[NH]) {
    // out = lo + hi * ND, NT >= max(9, NH + 5 + 1) limbs
    const u32 nd[5] = {BPPP_ND0, BPPP_ND1, BPPP_ND2, BPPP_ND3, BPPP_ND4};
#pragma unroll
    for (int i = 0; i < NT; i++) out[i] = i < 8 ? lo[i] : 0u;
#pragma unroll
    for (int i = 0; i < NH; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) { c += (u64)hi[i] * nd[j] + out[i + j]; out[i + j] = (u32)c; c >>= 32; }
#pragma unroll
        for (int k = i + 5; k < NT; k++) { c += out[k]; out[k] = (u32)c; c >>= 32; }
    }
}
HD void sc_reduce512(sc& r, const u32 t[16]) {
    u32 a[14];   // lo + hi*ND < 2^256 + 2^385: 13 limbs (+1 spare)
    sc_fold<8, 14>(a, t, t + 8);
    u32 b[11];   // lo + hi(6 limbs, < 2^130 in fact)*ND < 2^260: 9 limbs (+ spare)
    sc_fold<6, 11>(b, a, a + 8);
    u32 c[10];   // lo + hi(b[8..10], < 2^5)*ND < 2^256 + 2^134
    sc_fold<3, 10>(c, b, b + 8);
    // c[8] in {0,1}: one more wrap adds ND; value then < 2n
    sc_final(r, c, c[8]);
}
HD void sc_mul(sc& r, const sc& a, const sc& b) {
    u32 t[16];
    mul256(t, a.v, b.v);
    sc_reduce512(r, t);
}
HD void sc_sqr(sc& r, const sc& a) { sc_mul(r, a, a); }
HD bool sc_is_zero(const sc& a) { return is_zero256(a.v); }
HD bool sc_eq(const sc& a, const sc& b) { return eq256(a.v, b.v); }
// a^(n-2) (0 -> 0): left-to-right square-and-multiply over the public exponent
HD_NOINLINE void sc_inv(sc& r, const sc& a) {
    // n - 2 little-endian limbs
    const u32 e[8] = {0xD036413Fu, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    sc acc;
    sc_set_u32(acc, 1);
#pragma nounroll
    for (int i = 255; i >= 0; i--) {
        sc_sqr(acc, acc);
        if ((e[i >> 5] >> (i & 31)) & 1) sc_mul(acc, acc, a);  // exponent is public: wave-uniform branch
    }
    r = acc;
}
// big-endian bytes -> canonical scalar; false if >= n (k256 Scalar::from_repr returns None)
HD bool sc_from_be(sc& r, const uint8_t* b) {
    const u32 nd[5] = {BPPP_ND0, BPPP_ND1, BPPP_ND2, BPPP_ND3, BPPP_ND4};
    be32_to_limbs(r.v, b);
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (u64)r.v[i] + (i < 5 ? nd[i] : 0u); c >>= 32; }
    return c == 0;
}
HD void sc_to_be(uint8_t* b, const sc& a) { limbs_to_be32(b, a.v); }

}  // namespace bppp
