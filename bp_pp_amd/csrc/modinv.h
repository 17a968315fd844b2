// Modular inversion by Bernstein-Yang division steps ("safegcd", https://gcd.cr.yp.to/safegcd-20190413.pdf), the constant-time
// 32-bit formulation: values in nine signed 30-bit limbs, 20 batches of 30 division steps on the low limbs, each batch folded
// into a 2x2 transition matrix that is then applied to the full-width (f, g) and (d, e) pairs.  Branch-free, so every lane of
// a wavefront runs the same instruction stream; ~19 k instructions for a 256-bit modulus against ~46 k (field, Fermat
// addition chain) and ~250 k (scalar field, square-and-multiply) for exponentiation.
//
// Used for both secp256k1 moduli: the base field prime p (point -> affine conversions: k256 `to_affine`, which the reference
// reaches through every `to_encoded_point` / transcript append of a point) and the group order n (`Scalar::invert`,
// util.rs / circuit.rs call sites).  Zero maps to zero, as with the exponentiation it replaces.
#pragma once
#include <stdint.h>

#ifndef HD
#if defined(__HIPCC__)
#define HD __host__ __device__ __forceinline__
#else
#define HD inline
#endif
#endif

namespace bppp {

struct mi_s30 { int32_t v[9]; };                      // value = sum v[i] 2^(30 i), limbs in (-2^30, 2^30)
struct mi_modulus { mi_s30 m; uint32_t m_inv30; };     // m_inv30 = m^-1 mod 2^30
struct mi_t2x2 { int32_t u, v, q, r; };

// 30 division steps on the low limbs; zeta = -(delta + 1/2).  Returns the new zeta, t = transition matrix (scaled by 2^30).
HD int32_t mi_divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, mi_t2x2& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 5
    for (int i = 0; i < 30; i++) {
        uint32_t c1 = (uint32_t)(zeta >> 31);          // all ones when zeta < 0
        const uint32_t c2 = 0u - (g & 1u);             // all ones when g is odd
        const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
        g += x & c2;
        q += y & c2;
        r += z & c2;
        c1 &= c2;
        zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;
        f += g & c1;
        u += q & c1;
        v += r & c1;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return zeta;
}
// (d, e) <- t (d, e) / 2^30 mod m
HD void mi_update_de_30(mi_s30& d, mi_s30& e, const mi_t2x2& t, const mi_modulus& mod) {
    const int32_t M30 = (int32_t)(0xFFFFFFFFu >> 2);
    const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    int32_t di = d.v[0], ei = e.v[0];
    int64_t cd = (int64_t)u * di + (int64_t)v * ei;
    int64_t ce = (int64_t)q * di + (int64_t)r * ei;
    md -= (int32_t)((mod.m_inv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((mod.m_inv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cd += (int64_t)mod.m.v[0] * md;
    ce += (int64_t)mod.m.v[0] * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        di = d.v[i];
        ei = e.v[i];
        cd += (int64_t)u * di + (int64_t)v * ei;
        ce += (int64_t)q * di + (int64_t)r * ei;
        cd += (int64_t)mod.m.v[i] * md;
        ce += (int64_t)mod.m.v[i] * me;
        d.v[i - 1] = (int32_t)cd & M30; cd >>= 30;
        e.v[i - 1] = (int32_t)ce & M30; ce >>= 30;
    }
    d.v[8] = (int32_t)cd;
    e.v[8] = (int32_t)ce;
}
// (f, g) <- t (f, g) / 2^30
HD void mi_update_fg_30(mi_s30& f, mi_s30& g, const mi_t2x2& t) {
    const int32_t M30 = (int32_t)(0xFFFFFFFFu >> 2);
    const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
    int32_t fi = f.v[0], gi = g.v[0];
    int64_t cf = (int64_t)u * fi + (int64_t)v * gi;
    int64_t cg = (int64_t)q * fi + (int64_t)r * gi;
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        fi = f.v[i];
        gi = g.v[i];
        cf += (int64_t)u * fi + (int64_t)v * gi;
        cg += (int64_t)q * fi + (int64_t)r * gi;
        f.v[i - 1] = (int32_t)cf & M30; cf >>= 30;
        g.v[i - 1] = (int32_t)cg & M30; cg >>= 30;
    }
    f.v[8] = (int32_t)cf;
    g.v[8] = (int32_t)cg;
}
// r in (-2m, m) -> [0, m), negated first when sign < 0
HD void mi_normalize_30(mi_s30& r, int32_t sign, const mi_modulus& mod) {
    const int32_t M30 = (int32_t)(0xFFFFFFFFu >> 2);
    int32_t cond_add = r.v[8] >> 31;
    const int32_t cond_negate = sign >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.v[i] += mod.m.v[i] & cond_add;
        r.v[i] = (r.v[i] ^ cond_negate) - cond_negate;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= M30; }
    cond_add = r.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] += mod.m.v[i] & cond_add;
#pragma unroll
    for (int i = 0; i < 8; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= M30; }
}
// x <- x^-1 mod m (x in [0, m); 0 -> 0)
HD void mi_modinv(mi_s30& x, const mi_modulus& mod) {
    mi_s30 d, e, f = mod.m, g = x;
#pragma unroll
    for (int i = 0; i < 9; i++) { d.v[i] = 0; e.v[i] = 0; }
    e.v[0] = 1;
    int32_t zeta = -1;
#pragma nounroll
    for (int i = 0; i < 20; i++) {
        mi_t2x2 t;
        zeta = mi_divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        mi_update_de_30(d, e, t, mod);
        mi_update_fg_30(f, g, t);
    }
    mi_normalize_30(d, f.v[8], mod);
    x = d;
}
// 256-bit little-endian words <-> signed30
HD void mi_from_w8(mi_s30& r, const uint32_t w[8]) {
    const uint32_t M30 = 0x3FFFFFFFu;
    r.v[0] = (int32_t)(w[0] & M30);
    r.v[1] = (int32_t)(((w[0] >> 30) | (w[1] << 2)) & M30);
    r.v[2] = (int32_t)(((w[1] >> 28) | (w[2] << 4)) & M30);
    r.v[3] = (int32_t)(((w[2] >> 26) | (w[3] << 6)) & M30);
    r.v[4] = (int32_t)(((w[3] >> 24) | (w[4] << 8)) & M30);
    r.v[5] = (int32_t)(((w[4] >> 22) | (w[5] << 10)) & M30);
    r.v[6] = (int32_t)(((w[5] >> 20) | (w[6] << 12)) & M30);
    r.v[7] = (int32_t)(((w[6] >> 18) | (w[7] << 14)) & M30);
    r.v[8] = (int32_t)(w[7] >> 16);
}
HD void mi_to_w8(uint32_t w[8], const mi_s30& a) {   // a in [0, 2^256)
    const uint32_t* v = (const uint32_t*)a.v;
    w[0] = v[0] | (v[1] << 30);
    w[1] = (v[1] >> 2) | (v[2] << 28);
    w[2] = (v[2] >> 4) | (v[3] << 26);
    w[3] = (v[3] >> 6) | (v[4] << 24);
    w[4] = (v[4] >> 8) | (v[5] << 22);
    w[5] = (v[5] >> 10) | (v[6] << 20);
    w[6] = (v[6] >> 12) | (v[7] << 18);
    w[7] = (v[7] >> 14) | (v[8] << 16);
}
// secp256k1 base field prime and group order
HD void mi_modulus_p(mi_modulus& m) {
    const int32_t L[9] = {0x3ffffc2f, 0x3ffffffb, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0xffff};
#pragma unroll
    for (int i = 0; i < 9; i++) m.m.v[i] = L[i];
    m.m_inv30 = 0x2ddacacfu;
}
HD void mi_modulus_n(mi_modulus& m) {
    const int32_t L[9] = {0x10364141, 0x3f497a33, 0x348a03bb, 0x2bb739ab, 0x3ffffeba, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0xffff};
#pragma unroll
    for (int i = 0; i < 9; i++) m.m.v[i] = L[i];
    m.m_inv30 = 0x2a774ec1u;
}

}  // namespace bppp
