// 256-bit modular arithmetic for secp256k1 on gfx950: Fp (coordinates) and Fn (scalars).
//
// Replaces, for the batch hot path, what the reference takes from k256 0.13.3 (`Scalar`,
// `FieldElement`; every `.mul/.add/.sub/.invert*` call site in /root/reference/src, e.g.
// util.rs:28-60, wnla.rs:96-102, circuit.rs:166-235).
//
// The CDNA4 integer multiplier is v_mad_u64_u32 (32x32+64 -> 64, half rate): one instruction per limb product with the
// running column sum in the 64-bit accumulator.  Fp (the hot field: ~2.5e4 multiplications per verify) uses 10 x 26-bit
// unsaturated limbs; Fn (scalars: ~1.5e3 multiplications per verify) uses 8 x 32-bit canonical limbs.  No MFMA: this is
// carry-chained integer work.  Everything is branch-free (selects), so a 64-lane wavefront never diverges on data.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HD __host__ __device__ __forceinline__
#define HD_NOINLINE __host__ __device__ __noinline__ inline
#else
#define HD inline
#define HD_NOINLINE inline
#endif
#include "modinv.h"

namespace bppp {

typedef uint32_t u32;
typedef uint64_t u64;

struct sc { u32 v[8]; };  // mod n (group order), 8 x 32-bit limbs, canonical

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
// Columns K0..K0+G-1 of the NA x NB limb product a*b, interleaved: each column is a 64-bit accumulator + carry counter.
template <int NA, int NB, int K0, int G>
HD void mul_cols_group(u64* acc, u32* cnt, const u32* a, const u32* b) {
    constexpr int NC = NA + NB - 1;
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int k = K0 + g;
        if (k >= NC) continue;
        const int i0 = k < NB ? 0 : k - (NB - 1);
        acc[k] = (u64)a[i0] * b[k - i0];   // the first product of a column cannot carry
        cnt[k] = 0;
    }
#pragma unroll
    for (int s = 1; s < (NA < NB ? NA : NB); s++) {
        u64 cr[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int k = K0 + g;
            if (k >= NC) continue;
            const int i0 = k < NB ? 0 : k - (NB - 1), i1 = k < NA ? k : NA - 1;
            if (i0 + s <= i1) mad_c(acc[k], cr[g], a[i0 + s], b[k - i0 - s]);
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int k = K0 + g;
            if (k >= NC) continue;
            const int i0 = k < NB ? 0 : k - (NB - 1), i1 = k < NA ? k : NA - 1;
            if (i0 + s <= i1) add_c(cnt[k], cr[g]);
        }
    }
}
// t (NA + NB limbs) = a (NA limbs) * b (NB limbs), product scanning: all column sums are accumulated independently
// (v_mad_u64_u32 + v_addc per limb product, independent columns -> ILP for a lone wavefront), then one carry-propagation
// pass (3 full-rate adds per column).
template <int NA, int NB>
HD void mul_limbs(u32* t, const u32* a, const u32* b) {
    constexpr int NC = NA + NB - 1;
    u64 acc[NC + 3];
    u32 cnt[NC + 3];
    mul_cols_group<NA, NB, 0, 4>(acc, cnt, a, b);
    if (NC > 4) mul_cols_group<NA, NB, 4, 4>(acc, cnt, a, b);
    if (NC > 8) mul_cols_group<NA, NB, 8, 4>(acc, cnt, a, b);
    if (NC > 12) mul_cols_group<NA, NB, 12, 4>(acc, cnt, a, b);
    t[0] = (u32)acc[0];
    u32 c_lo = (u32)(acc[0] >> 32), c_hi = 0;
#pragma unroll
    for (int k = 1; k < NC; k++) {
        u32 cy = 0;
        t[k] = addc((u32)acc[k], c_lo, cy);
        c_lo = addc((u32)(acc[k] >> 32), c_hi, cy);
        c_hi = cnt[k] + cy;
    }
    t[NC] = c_lo;
}
HD void mul256(u32 t[16], const u32 a[8], const u32 b[8]) { mul_limbs<8, 8>(t, a, b); }
// big-endian 32 bytes <-> limbs
// 32 big-endian bytes <-> 8 little-endian words.  16-byte aligned addresses (every 64-byte point and 32-byte scalar slot of
// the C-ABI layouts, when the buffer itself is aligned) move as two 128-bit accesses + byte swaps; anything else goes byte by
// byte.  The alignment test is the same for every lane of a wavefront in practice (strides are multiples of 16).
struct alignas(16) u32x4 { u32 x, y, z, w; };
HD u32 bswap32(u32 v) { return __builtin_bswap32(v); }
HD void be32_to_limbs(u32 r[8], const uint8_t* b) {
    if ((((uintptr_t)b) & 15u) == 0) {
        const u32x4 hi = ((const u32x4*)b)[0], lo = ((const u32x4*)b)[1];
        r[7] = bswap32(hi.x); r[6] = bswap32(hi.y); r[5] = bswap32(hi.z); r[4] = bswap32(hi.w);
        r[3] = bswap32(lo.x); r[2] = bswap32(lo.y); r[1] = bswap32(lo.z); r[0] = bswap32(lo.w);
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint8_t* q = b + 4 * (7 - i);
        r[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
}
HD void limbs_to_be32(uint8_t* b, const u32 a[8]) {
    if ((((uintptr_t)b) & 15u) == 0) {
        u32x4 hi, lo;
        hi.x = bswap32(a[7]); hi.y = bswap32(a[6]); hi.z = bswap32(a[5]); hi.w = bswap32(a[4]);
        lo.x = bswap32(a[3]); lo.y = bswap32(a[2]); lo.z = bswap32(a[1]); lo.w = bswap32(a[0]);
        ((u32x4*)b)[0] = hi;
        ((u32x4*)b)[1] = lo;
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint8_t* q = b + 4 * (7 - i);
        q[0] = (uint8_t)(a[i] >> 24); q[1] = (uint8_t)(a[i] >> 16); q[2] = (uint8_t)(a[i] >> 8); q[3] = (uint8_t)a[i];
    }
}

// ---------------------------------------------------------------- Fp: p = 2^256 - 2^32 - 977, 10 x 26-bit limbs (unsaturated)
// Why not the saturated 8 x 32 form used for Fn below: measured on MI355X (tools/probes/intbench.hip), a lone wavefront per SIMD --
// the regime of the shared-doubling kernels at 2^16 proofs -- pays per INSTRUCTION, and the saturated product needs a carry
// counter per limb product plus canonical selects after every add/sub.  With 26-bit limbs the 100 limb products accumulate
// into 64-bit columns with no carries at all (v_mad_u64_u32 chains), add is 10 plain adds, sub is 10 add-sub pairs against a
// multiple of p, and reduction happens once per multiplication: 1.5x the point-addition rate at 1 wave/SIMD, 1.1x at 4.
//
// Value = sum v[i] 2^(26 i).  Limbs may exceed 26 bits: "magnitude m" means v[i] <= 2 m (2^26 - 1) for i < 9 and
// v[9] <= 2 m (2^22 - 1) (the convention of libsecp256k1's 10x26 field).  mul/sqr/mul_small accept magnitudes <= 8 and
// return magnitude 1; add sums magnitudes; sub/neg add a multiple of p.  The host build (tests/emul) carries the magnitude
// in the struct and asserts every bound; the device build carries nothing.
#define BPPP_M26 0x3FFFFFFu
#define BPPP_M22 0x03FFFFFu
#define BPPP_R0 0x3D10u        // 2^260 mod p = 2^36 + 0x3D10  ->  R0 at limb 0, 2^10 at limb 1
#define BPPP_PC0 0x000003D1u   // 2^256 mod p = 2^32 + 0x3D1   ->  0x3D1 at limb 0, 2^6 at limb 1

#if !defined(__HIPCC__) && !defined(BPPP_NO_FE_DEBUG)
#define BPPP_FE_DEBUG 1
#include <assert.h>
#include <execinfo.h>
#include <stdio.h>
#endif

struct fe {
    u32 v[10];
#ifdef BPPP_FE_DEBUG
    int mag;
#endif
};
#ifdef BPPP_FE_DEBUG
#define FE_SETMAG(x, m) ((x).mag = (m))
#define FE_MAG(x) ((x).mag)
inline void fe_check(const fe& a, int max_mag) {
    if (!(a.mag >= 0 && a.mag <= max_mag)) {
        fprintf(stderr, "fe_check: magnitude %d exceeds %d\n", a.mag, max_mag);
        void* bt[24];
        int n = backtrace(bt, 24);
        backtrace_symbols_fd(bt, n, 2);
    }
    assert(a.mag >= 0 && a.mag <= max_mag);
    for (int i = 0; i < 9; i++) assert((u64)a.v[i] <= 2ull * (u64)a.mag * BPPP_M26 || (a.mag == 0 && a.v[i] == 0));
    assert((u64)a.v[9] <= 2ull * (u64)a.mag * BPPP_M22 || (a.mag == 0 && a.v[9] == 0));
}
#define FE_CHECK(x, m) fe_check((x), (m))
#else
#define FE_SETMAG(x, m) ((void)0)
#define FE_MAG(x) 0
#define FE_CHECK(x, m) ((void)0)
#endif

HD void fe_set_u32(fe& r, u32 x) {   // x < 2^26
    r.v[0] = x;
#pragma unroll
    for (int i = 1; i < 10; i++) r.v[i] = 0;
    FE_SETMAG(r, x ? 1 : 0);
}
HD void fe_add(fe& r, const fe& a, const fe& b) {
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
    FE_SETMAG(r, FE_MAG(a) + FE_MAG(b));
    FE_CHECK(r, 16);
}
HD void fe_dbl(fe& r, const fe& a) { fe_add(r, a, a); }
// r = a - b for mag(b) <= M: r = a + 2(M+1) p - b, magnitude mag(a) + M + 1
template <int M>
HD void fe_sub_m(fe& r, const fe& a, const fe& b) {
    FE_CHECK(b, M);
    const u32 k = 2u * (M + 1);
    r.v[0] = a.v[0] + k * 0x3FFFC2Fu - b.v[0];
    r.v[1] = a.v[1] + k * 0x3FFFFBFu - b.v[1];
#pragma unroll
    for (int i = 2; i < 9; i++) r.v[i] = a.v[i] + k * BPPP_M26 - b.v[i];
    r.v[9] = a.v[9] + k * BPPP_M22 - b.v[9];
    FE_SETMAG(r, FE_MAG(a) + M + 1);
    FE_CHECK(r, 16);
}
HD void fe_sub(fe& r, const fe& a, const fe& b) { fe_sub_m<3>(r, a, b); }   // default: subtrahend magnitude <= 3
template <int M>
HD void fe_neg_m(fe& r, const fe& a) {
    fe z;
    fe_set_u32(z, 0);
    fe_sub_m<M>(r, z, a);
}
HD void fe_neg(fe& r, const fe& a) { fe_neg_m<3>(r, a); }
HD void fe_cmov(fe& r, bool take, const fe& b) {
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = take ? b.v[i] : r.v[i];
#ifdef BPPP_FE_DEBUG
    r.mag = r.mag > b.mag ? r.mag : b.mag;
#endif
}
// A value the compiler must treat as unknown (keeps `x * 1024` a v_mad_u64_u32 instead of a 64-bit shift plus a 64-bit add).
HD u32 opaque_u32(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(x));
#endif
    return x;
}
// 19 column sums -> magnitude-1 limbs, one interleaved pass: a "high" accumulator d walks columns 9, 10..18 emitting 26-bit
// digits u, each of which is folded straight into the "low" accumulator e walking columns 0..8 through
// 2^260 = R0 + 2^10 * 2^26 (mod p); the part above 2^256 is folded once more through 2^256 = 0x3D1 + 2^6 * 2^26.
// (Same dataflow as the 10x26 field of libsecp256k1; bounds for column sums < 2^64, i.e. input magnitudes <= 8.)
HD void fe_reduce_cols(fe& r, const u64 c[19]) {
    const u32 k1024 = opaque_u32(1024u), k64 = opaque_u32(64u);
    u64 d = c[9];
    const u32 t9 = (u32)d & BPPP_M26;
    d >>= 26;                                             // < 2^38
    u64 e = 0;
    u32 t[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        d += c[k + 10];                                   // < 2^64: c <= 9 * 2^60 for the high columns
        const u32 u = (u32)d & BPPP_M26;
        d >>= 26;
        e += c[k] + (u64)u * BPPP_R0;                     // < 2^64
        t[k] = (u32)e & BPPP_M26;
        e >>= 26;
        e += (u64)u * k1024;                              // u * 2^10 as a mad: one instruction instead of shift + add
    }
    // d < 2^38 is the digit of column 19; e the carry into column 9
    const u32 d_lo = (u32)d, d_hi = (u32)(d >> 32);      // d * R0 as two 32 x 32 products
    e += (u64)t9 + (u64)d_lo * BPPP_R0 + (((u64)d_hi * BPPP_R0) << 32);
    r.v[9] = (u32)e & BPPP_M22;
    e >>= 22;                                             // counts units of 2^256 now
    e += d << 14;                                         // d * 2^10 at column 10 = 2^260 = 2^4 * 2^256; < 2^53
    // e * 2^256 = e * (0x3D1 + 2^6 * 2^26); split e into 26-bit digits so every product is 32 x 32
    const u32 e0 = (u32)e & BPPP_M26, e1 = (u32)(e >> 26);   // e1 < 2^27
    u64 f = (u64)t[0] + (u64)e0 * BPPP_PC0;
    r.v[0] = (u32)f & BPPP_M26; f >>= 26;
    f += (u64)t[1] + (u64)e0 * k64 + (u64)e1 * BPPP_PC0;
    r.v[1] = (u32)f & BPPP_M26; f >>= 26;
    f += (u64)t[2] + (u64)e1 * k64;
    r.v[2] = (u32)f & BPPP_M26; f >>= 26;
    r.v[3] = t[3] + (u32)f;                               // f < 2^8: limb 3 stays within magnitude 1
#pragma unroll
    for (int k = 4; k < 9; k++) r.v[k] = t[k];
    FE_SETMAG(r, 1);
    FE_CHECK(r, 1);
}
// (Round 5 measured a form of this reduction with the carries riding in the multiply-adds -- a column's v_mad_u64_u32 chain starting from
// the carry of the column below, overlapping 32-bit digits in the high walk: 54 of ~ 375 issue slots per multiplication fewer on paper.
// hipcc turns `carry + a b + c d` back into products plus a 64-bit addition, so it took one asm statement per multiply-add; that build
// ran the fixed-base sums 1.3 % faster, the four rounds 1.6 % SLOWER and a 2^16-proof batch 12 % slower -- a lone wavefront per SIMD
// cannot cover a 118-deep dependent chain.  Not kept: docs/design/06-measurements.md, "Round 5".)
HD void fe_mul(fe& r, const fe& a, const fe& b) {
    FE_CHECK(a, 8);
    FE_CHECK(b, 8);
    u64 c[19];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        const int i0 = k < 10 ? 0 : k - 9, i1 = k < 10 ? k : 9;
        u64 acc = (u64)a.v[i0] * b.v[k - i0];
#pragma unroll
        for (int i = i0 + 1; i <= i1; i++) acc += (u64)a.v[i] * b.v[k - i];
        c[k] = acc;
    }
    fe_reduce_cols(r, c);
}
// r = a b + c d with ONE reduction: both products accumulate into the same 19 column sums (the reduction is 44 % of a
// multiplication's issue cycles, so a sum of two products costs ~1.55 multiplications instead of 2).  Bound: the column sums
// must stay below 2^64, i.e. mag(a) mag(b) + mag(c) mag(d) <= 64 in the magnitude convention above (a single product of
// magnitude-8 inputs is 64).  Used for y3 = r (q - x3) + (-y1) ppp in the mixed additions (point.h).
HD void fe_mul2_add(fe& r, const fe& a, const fe& b, const fe& c, const fe& d) {
#ifdef BPPP_FE_DEBUG
    assert(a.mag * b.mag + c.mag * d.mag <= 64);
#endif
    u64 col[19];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        const int i0 = k < 10 ? 0 : k - 9, i1 = k < 10 ? k : 9;
        u64 acc = (u64)a.v[i0] * b.v[k - i0];
#pragma unroll
        for (int i = i0 + 1; i <= i1; i++) acc += (u64)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = i0; i <= i1; i++) acc += (u64)c.v[i] * d.v[k - i];
        col[k] = acc;
    }
    fe_reduce_cols(r, col);
}
HD void fe_sqr(fe& r, const fe& a) {   // 55 limb products: cross terms use the doubled limb
    FE_CHECK(a, 8);
    u32 a2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) a2[i] = a.v[i] << 1;   // <= 2^31
    u64 c[19];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        const int i0 = k < 10 ? 0 : k - 9;
        u64 acc = 0;
#pragma unroll
        for (int i = i0; 2 * i < k; i++) acc += (u64)a2[i] * a.v[k - i];
        if ((k & 1) == 0) acc += (u64)a.v[k / 2] * a.v[k / 2];
        c[k] = acc;
    }
    fe_reduce_cols(r, c);
}
HD void fe_mul_small(fe& r, const fe& a, u32 k) {   // k <= 32, mag(a) <= 8
    FE_CHECK(a, 8);
    u64 c[19];
#pragma unroll
    for (int i = 0; i < 10; i++) c[i] = (u64)a.v[i] * k;
#pragma unroll
    for (int i = 10; i < 19; i++) c[i] = 0;
    // same tail as a product, specialised: only 10 columns are non-zero
    u64 d = c[0];
    u32 q[10];
#pragma unroll
    for (int i = 0; i < 9; i++) { q[i] = (u32)d & BPPP_M26; d = (d >> 26) + c[i + 1]; }
    q[9] = (u32)d & BPPP_M22;
    const u32 top = (u32)(d >> 22);                          // < 2^20
    u64 f = (u64)q[0] + (u64)top * BPPP_PC0;
    r.v[0] = (u32)f & BPPP_M26; f >>= 26;
    f += (u64)q[1] + ((u64)top << 6);
    r.v[1] = (u32)f & BPPP_M26; f >>= 26;
    r.v[2] = q[2] + (u32)f;
#pragma unroll
    for (int i = 3; i < 10; i++) r.v[i] = q[i];
    FE_SETMAG(r, 1);
    FE_CHECK(r, 1);
}
// full reduction to the canonical representative in [0, p)
HD void fe_normalize(fe& r) {
    FE_CHECK(r, 16);
    // weak pass: fold the part above 2^256 once, carry through
    u32 x = r.v[9] >> 22;
    r.v[9] &= BPPP_M22;
    u32 t0 = r.v[0] + x * BPPP_PC0, t1 = r.v[1] + (x << 6), t[10];
    t1 += t0 >> 26; t[0] = t0 & BPPP_M26;
    u32 cur = r.v[2] + (t1 >> 26); t[1] = t1 & BPPP_M26;
#pragma unroll
    for (int i = 2; i < 9; i++) { t[i] = cur & BPPP_M26; cur = r.v[i + 1] + (cur >> 26); }
    t[9] = cur;                                               // may again exceed 22 bits by one unit at most
    // now value < 2^256 + 2^232-ish; decide whether to subtract p: x2 = bit 256 or (value >= p)
    u32 m = t[2];
#pragma unroll
    for (int i = 3; i < 9; i++) m &= t[i];
    u32 ge = (t[9] >> 22) | ((t[9] == BPPP_M22) & (m == BPPP_M26) & ((t[1] + 0x40u + ((t[0] + BPPP_PC0) >> 26)) > BPPP_M26));
    // add 2^256 - p = 0x1000003D1 when ge, then drop bit 256
    u32 u0 = t[0] + (ge ? BPPP_PC0 : 0u), u1 = t[1] + (ge ? 0x40u : 0u);
    u1 += u0 >> 26; r.v[0] = u0 & BPPP_M26;
    cur = t[2] + (u1 >> 26); r.v[1] = u1 & BPPP_M26;
#pragma unroll
    for (int i = 2; i < 9; i++) { r.v[i] = cur & BPPP_M26; cur = t[i + 1] + (cur >> 26); }
    r.v[9] = cur & BPPP_M22;
    FE_SETMAG(r, 1);
}
HD bool fe_is_zero(const fe& a) {
    fe t = a;
    fe_normalize(t);
    u32 x = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) x |= t.v[i];
    return x == 0;
}
HD bool fe_eq(const fe& a, const fe& b) {   // magnitudes <= 8
    fe ta = a, tb = b;
    fe_normalize(ta);
    fe_normalize(tb);
    u32 x = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) x |= ta.v[i] ^ tb.v[i];
    return x == 0;
}
HD void fe_sqr_n(fe& r, const fe& a, int n) {
    r = a;
#pragma nounroll
    for (int i = 0; i < n; i++) fe_sqr(r, r);
}
// a^(p-2) (0 -> 0).  Addition chain on the run structure of p-2: 255 squarings + 15 multiplications.  Kept as the cross-check
// of fe_inv (tests/emul); the product path inverts with division steps (modinv.h).
HD_NOINLINE void fe_inv_fermat(fe& r, const fe& a) {
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
// packed canonical form (8 x 32-bit little-endian words: HBM tables, decoded inputs) <-> limbs
HD void fe_from_w8(fe& r, const u32 w[8]) {
    r.v[0] = w[0] & BPPP_M26;
    r.v[1] = ((w[0] >> 26) | (w[1] << 6)) & BPPP_M26;
    r.v[2] = ((w[1] >> 20) | (w[2] << 12)) & BPPP_M26;
    r.v[3] = ((w[2] >> 14) | (w[3] << 18)) & BPPP_M26;
    r.v[4] = ((w[3] >> 8) | (w[4] << 24)) & BPPP_M26;
    r.v[5] = (w[4] >> 2) & BPPP_M26;
    r.v[6] = ((w[4] >> 28) | (w[5] << 4)) & BPPP_M26;
    r.v[7] = ((w[5] >> 22) | (w[6] << 10)) & BPPP_M26;
    r.v[8] = ((w[6] >> 16) | (w[7] << 16)) & BPPP_M26;
    r.v[9] = w[7] >> 10;
    FE_SETMAG(r, 1);
}
HD void fe_to_w8(u32 w[8], const fe& a) {   // normalises
    fe t = a;
    fe_normalize(t);
    w[0] = t.v[0] | (t.v[1] << 26);
    w[1] = (t.v[1] >> 6) | (t.v[2] << 20);
    w[2] = (t.v[2] >> 12) | (t.v[3] << 14);
    w[3] = (t.v[3] >> 18) | (t.v[4] << 8);
    w[4] = (t.v[4] >> 24) | (t.v[5] << 2) | (t.v[6] << 28);
    w[5] = (t.v[6] >> 4) | (t.v[7] << 22);
    w[6] = (t.v[7] >> 10) | (t.v[8] << 16);
    w[7] = (t.v[8] >> 16) | (t.v[9] << 10);
}
// big-endian bytes -> canonical element; false if >= p
// a^-1 mod p (0 -> 0) by division steps (modinv.h)
HD_NOINLINE void fe_inv(fe& r, const fe& a) {
    u32 w[8];
    fe_to_w8(w, a);
    mi_modulus m;
    mi_modulus_p(m);
    mi_s30 x;
    mi_from_w8(x, w);
    mi_modinv(x, m);
    mi_to_w8(w, x);
    fe_from_w8(r, w);
}
HD bool fe_from_be(fe& r, const uint8_t* b) {
    u32 w[8];
    be32_to_limbs(w, b);
    fe_from_w8(r, w);
    u32 c = 0;
    (void)addc(w[0], BPPP_PC0, c);
    (void)addc(w[1], 1u, c);
#pragma unroll
    for (int i = 2; i < 8; i++) (void)addc(w[i], 0u, c);
    return c == 0;
}
HD void fe_to_be(uint8_t* b, const fe& a) {
    u32 w[8];
    fe_to_w8(w, a);
    limbs_to_be32(b, w);
}
HD bool fe_is_odd(const fe& a) {
    fe t = a;
    fe_normalize(t);
    return t.v[0] & 1u;
}

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
// out (NH + 5 limbs) = lo (8 limbs) + hi (NH limbs) * ND, with ND = nd4 (4 limbs) + 2^128: the NH x 4 product by independent
// columns (mul_limbs), then two carry chains.
template <int NH>
HD void sc_fold(u32* out /* NH + 5 */, const u32* lo /* 8 */, const u32* hi /* NH */) {
    const u32 nd[4] = {BPPP_ND0, BPPP_ND1, BPPP_ND2, BPPP_ND3};
    constexpr int NT = NH + 5;
    u32 m[NH + 4];
    mul_limbs<NH, 4>(m, hi, nd);
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < NT; i++) out[i] = addc(i < 8 ? lo[i] : 0u, i < NH + 4 ? m[i] : 0u, c);
    c = 0;
#pragma unroll
    for (int i = 4; i < NT; i++) out[i] = addc(out[i], (i - 4) < NH ? hi[i - 4] : 0u, c);
}
HD void sc_reduce512(sc& r, const u32 t[16]) {
    u32 a[13];   // lo + hi*ND < 2^256 + 2^385: 13 limbs
    sc_fold<8>(a, t, t + 8);
    u32 b[10];   // lo + a[8..13) (< 2^130) * ND < 2^260: 9 limbs (+1 spare)
    sc_fold<5>(b, a, a + 8);
    // lo + b[8] (< 2^5; b[9] = 0 by the bound above) * ND < 2^256 + 2^134: 8 limbs + carry limb
    u32 d[9];
    {
        const u32 nd5[5] = {BPPP_ND0, BPPP_ND1, BPPP_ND2, BPPP_ND3, BPPP_ND4};
        const u32 h = b[8];
        u64 p[5];
#pragma unroll
        for (int j = 0; j < 5; j++) p[j] = (u64)h * nd5[j];
        u32 cy = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) d[k] = addc(k < 8 ? b[k] : 0u, k < 5 ? (u32)p[k] : 0u, cy);
        cy = 0;
#pragma unroll
        for (int k = 1; k < 9; k++) d[k] = addc(d[k], (k - 1) < 5 ? (u32)(p[k - 1] >> 32) : 0u, cy);
    }
    // d[8] in {0,1}: one more wrap adds ND; value then < 2n
    sc_final(r, d, d[8]);
}
HD void sc_mul(sc& r, const sc& a, const sc& b) {
    u32 t[16];
    mul256(t, a.v, b.v);
    sc_reduce512(r, t);
}
HD void sc_sqr(sc& r, const sc& a) { sc_mul(r, a, a); }
HD bool sc_is_zero(const sc& a) { return is_zero256(a.v); }
HD bool sc_eq(const sc& a, const sc& b) { return eq256(a.v, b.v); }
// a^(n-2) (0 -> 0): left-to-right square-and-multiply over the public exponent (cross-check of sc_inv)
HD_NOINLINE void sc_inv_fermat(sc& r, const sc& a) {
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
// a^-1 mod n (0 -> 0) by division steps (modinv.h)
HD_NOINLINE void sc_inv(sc& r, const sc& a) {
    mi_modulus m;
    mi_modulus_n(m);
    mi_s30 x;
    mi_from_w8(x, a.v);
    mi_modinv(x, m);
    mi_to_w8(r.v, x);
}
// big-endian bytes -> canonical scalar; false if >= n (k256 Scalar::from_repr returns None)
HD bool sc_is_canonical(const sc& r) {   // r < n  <=>  r + (2^256 - n) does not carry out
    const u32 nd[5] = {BPPP_ND0, BPPP_ND1, BPPP_ND2, BPPP_ND3, BPPP_ND4};
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) (void)addc(r.v[i], i < 5 ? nd[i] : 0u, c);
    return c == 0;
}
HD bool sc_from_be(sc& r, const uint8_t* b) {
    be32_to_limbs(r.v, b);
    return sc_is_canonical(r);
}
HD void sc_to_be(uint8_t* b, const sc& a) { limbs_to_be32(b, a.v); }

}  // namespace bppp
