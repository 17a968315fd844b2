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

// ---------------------------------------------------------------- carry primitives
// hipcc turns __builtin_addc / __builtin_subc chains into v_add_co_u32 / v_addc_co_u32 (32-bit, full rate); the same
// arithmetic written with 64-bit temporaries compiles to half-rate v_lshl_add_u64 plus register shuffles.
HD u32 addc(u32 a, u32 b, u32& carry) {
#if defined(__clang__)
    u32 co;
    u32 r = __builtin_addc(a, b, carry, &co);
    carry = co;
    return r;
#else
    u64 s = (u64)a + b + carry;
    carry = (u32)(s >> 32);
    return (u32)s;
#endif
}
HD u32 subb(u32 a, u32 b, u32& borrow) {
#if defined(__clang__)
    u32 bo;
    u32 r = __builtin_subc(a, b, borrow, &bo);
    borrow = bo;
    return r;
#else
    u64 d = (u64)a - b - borrow;
    borrow = (u32)(d >> 32) & 1u;
    return (u32)d;
#endif
}
// acc (64-bit) += a*b with the carry out of bit 64 delivered separately: on gfx950 exactly v_mad_u64_u32 (carry-out in an
// SGPR pair) + v_addc_co_u32 (consumes it).  hipcc does not form this pair from C, hence the two one-instruction asm
// statements; being separate statements, the scheduler is free to interleave several columns between them.
#if defined(__HIP_DEVICE_COMPILE__)
HD void mad_c(u64& acc, u64& carry, u32 a, u32 b) { asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry) : "v"(a), "v"(b)); }
HD void add_c(u32& cnt, u64& carry) { asm("v_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(cnt), "+s"(carry)); }
#else
HD void mad_c(u64& acc, u64& carry, u32 a, u32 b) {
    u64 p = (u64)a * b, o = acc;
    acc = o + p;
    carry = acc < o ? 1u : 0u;
}
HD void add_c(u32& cnt, u64& carry) { cnt += (u32)carry; }
#endif

// ---------------------------------------------------------------- generic 256-bit helpers
HD u32 add256(u32 r[8], const u32 a[8], const u32 b[8]) {
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = addc(a[i], b[i], c);
    return c;
}
HD u32 sub256(u32 r[8], const u32 a[8], const u32 b[8]) {
    u32 bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = subb(a[i], b[i], bw);
    return bw;
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
// Columns K0..K0+G-1 of the 8x8 limb product, interleaved: each column is a 64-bit accumulator + carry counter.
template <int K0, int G>
HD void mul256_cols(u64* acc, u32* cnt, const u32* a, const u32* b) {
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int k = K0 + g;
        if (k > 14) continue;
        const int i0 = k < 8 ? 0 : k - 7;
        acc[k] = (u64)a[i0] * b[k - i0];   // the first product of a column cannot carry
        cnt[k] = 0;
    }
#pragma unroll
    for (int s = 1; s < 8; s++) {
        u64 cr[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int k = K0 + g;
            if (k > 14) continue;
            const int i0 = k < 8 ? 0 : k - 7, i1 = k < 8 ? k : 7;
            if (i0 + s <= i1) mad_c(acc[k], cr[g], a[i0 + s], b[k - i0 - s]);
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int k = K0 + g;
            if (k > 14) continue;
            const int i0 = k < 8 ? 0 : k - 7, i1 = k < 8 ? k : 7;
            if (i0 + s <= i1) add_c(cnt[k], cr[g]);
        }
    }
}
// full 256x256 -> 512 product, product scanning: 64 v_mad_u64_u32 + 49 v_addc (all 15 column sums independent, which
// is what lets a lone wavefront keep issuing), then one carry-propagation pass (3 full-rate adds per column).
HD void mul256(u32 t[16], const u32 a[8], const u32 b[8]) {
    u64 acc[16];
    u32 cnt[16];
    mul256_cols<0, 4>(acc, cnt, a, b);
    mul256_cols<4, 4>(acc, cnt, a, b);
    mul256_cols<8, 4>(acc, cnt, a, b);
    mul256_cols<12, 4>(acc, cnt, a, b);
    t[0] = (u32)acc[0];
    u32 c_lo = (u32)(acc[0] >> 32), c_hi = 0;
#pragma unroll
    for (int k = 1; k < 15; k++) {
        u32 cy = 0;
        t[k] = addc((u32)acc[k], c_lo, cy);
        c_lo = addc((u32)(acc[k] >> 32), c_hi, cy);
        c_hi = cnt[k] + cy;
    }
    t[15] = c_lo;
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

// x (8 limbs) + carry*2^256, value < 2^256 + small  ->  canonical.  x >= p  <=>  x + PC carries out of 2^256.
HD void fe_final(fe& r, const u32 x[8], u32 carry) {
    u32 t[8];
    u32 c = 0;
    t[0] = addc(x[0], BPPP_PC0, c);
    t[1] = addc(x[1], 1u, c);
#pragma unroll
    for (int i = 2; i < 8; i++) t[i] = addc(x[i], 0u, c);
    sel256(r.v, carry | c, x, t);
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
    u32 bw = 0;
    e[0] = subb(d[0], BPPP_PC0, bw);
    e[1] = subb(d[1], 1u, bw);
#pragma unroll
    for (int i = 2; i < 8; i++) e[i] = subb(d[i], 0u, bw);
    sel256(r.v, borrow, d, e);
}
HD void fe_neg(fe& r, const fe& a) {
    fe z;
#pragma unroll
    for (int i = 0; i < 8; i++) z.v[i] = 0;
    fe_sub(r, z, a);
}
HD void fe_dbl(fe& r, const fe& a) { fe_add(r, a, a); }
// reduce a 512-bit value: hi*2^256 + lo == lo + (hi << 32) + hi*977, then fold the (< 2^35) overflow once more
HD void fe_reduce512(fe& out, const u32 t[16]) {
    u32 r[9];
    u32 c = 0;
    r[0] = t[0];
#pragma unroll
    for (int j = 1; j < 8; j++) r[j] = addc(t[j], t[8 + j - 1], c);
    r[8] = addc(t[15], 0u, c);
    u32 r9 = c;
    u64 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = (u64)t[8 + j] * BPPP_PC0;
    c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = addc(r[j], (u32)v[j], c);
    r[8] = addc(r[8], 0u, c);
    r9 += c;
    c = 0;
#pragma unroll
    for (int j = 1; j < 8; j++) r[j] = addc(r[j], (u32)(v[j - 1] >> 32), c);
    r[8] = addc(r[8], (u32)(v[7] >> 32), c);
    r9 += c;
    // top = r[8] + r9 * 2^32 (< 2^35); top * PC = top*977 + (top << 32)
    u64 m = (u64)r[8] * BPPP_PC0 + ((u64)(r9 * BPPP_PC0) << 32);
    u32 s[8];
    c = 0;
    s[0] = addc(r[0], (u32)m, c);
    s[1] = addc(r[1], (u32)(m >> 32), c);
#pragma unroll
    for (int i = 2; i < 8; i++) s[i] = addc(r[i], 0u, c);
    u32 k1 = c;
    c = 0;
    s[1] = addc(s[1], r[8], c);
    s[2] = addc(s[2], r9, c);
#pragma unroll
    for (int i = 3; i < 8; i++) s[i] = addc(s[i], 0u, c);
    // at most one of the two chains wraps 2^256 (the folded value is < 2^256 + 2^68); fe_final adds PC for the wrap
    fe_final(out, s, k1 | c);
}
HD void fe_mul(fe& r, const fe& a, const fe& b) {
    u32 t[16];
    mul256(t, a.v, b.v);
    fe_reduce512(r, t);
}
HD void fe_sqr(fe& r, const fe& a) { fe_mul(r, a, a); }
HD void fe_mul_small(fe& r, const fe& a, u32 k) {  // k < 2^16
    u64 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (u64)a.v[i] * k;   // 8 independent mads; high words < 2^16
    u32 s[8];
    u32 c = 0;
    s[0] = (u32)v[0];
#pragma unroll
    for (int i = 1; i < 8; i++) s[i] = addc((u32)v[i], (u32)(v[i - 1] >> 32), c);
    u32 top = (u32)(v[7] >> 32) + c;   // < 2^17
    // fold top * PC = top*977 (< 2^27) + (top << 32)
    c = 0;
    s[0] = addc(s[0], top * BPPP_PC0, c);
    s[1] = addc(s[1], top, c);
#pragma unroll
    for (int i = 2; i < 8; i++) s[i] = addc(s[i], 0u, c);
    fe_final(r, s, c);
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
    u32 c = 0;
    (void)addc(r.v[0], BPPP_PC0, c);
    (void)addc(r.v[1], 1u, c);
#pragma unroll
    for (int i = 2; i < 8; i++) (void)addc(r.v[i], 0u, c);
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
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = addc(x[i], i < 5 ? nd[i] : 0u, c);
    sel256(r.v, carry | c, x, t);
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
    for (int i = 0; i < 8; i++) e[i] = subb(d[i], i < 5 ? nd[i] : 0u, bw);
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
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) (void)addc(r.v[i], i < 5 ? nd[i] : 0u, c);
    return c == 0;
}
HD void sc_to_be(uint8_t* b, const sc& a) { limbs_to_be32(b, a.v); }

}  // namespace bppp
