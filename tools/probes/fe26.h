// Prototype: Fp with 10 x 26-bit limbs (unsaturated), products accumulated in u64 columns without carry counters.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
namespace fe26ns {
typedef uint32_t u32; typedef uint64_t u64;
struct fe26 { u32 n[10]; };
#define M26 0x3FFFFFFu
#define R0_26 0x3D10u   // 2^260 mod p = R1*2^26 + R0 with R1 = 2^10
__device__ __forceinline__ void fe26_mul(fe26& r, const fe26& a, const fe26& b) {
    u64 c[19];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        const int i0 = k < 10 ? 0 : k - 9, i1 = k < 10 ? k : 9;
        u64 acc = (u64)a.n[i0] * b.n[k - i0];
#pragma unroll
        for (int i = i0 + 1; i <= i1; i++) acc += (u64)a.n[i] * b.n[k - i];
        c[k] = acc;
    }
    // fold the high columns: c[k] (k >= 10) contributes c[k] * (R0 + R1 * 2^26) at column k-10.
    // first split each high column into 26-bit digits by a carry chain over columns 10..18
    u64 d = c[10];
    u32 h[10];
#pragma unroll
    for (int k = 10; k < 18; k++) { h[k - 10] = (u32)d & M26; d = (d >> 26) + c[k + 1]; }
    h[8] = (u32)d & M26; d >>= 26;
    h[9] = (u32)d;   // up to ~2^38 >> ... fits 32 bits for magnitude-bounded inputs
    // low columns + h * R
    u64 e = c[0] + (u64)h[0] * R0_26;
    u32 t[10];
    t[0] = (u32)e & M26; e >>= 26;
#pragma unroll
    for (int k = 1; k < 10; k++) {
        e += c[k] + (u64)h[k] * R0_26 + ((u64)h[k - 1] << 10);
        t[k] = (u32)e & M26; e >>= 26;
    }
    // remaining: e (carry) + h[9] << 10 at column 10 -> fold again through R
    e += (u64)h[9] << 10;
    // e < ~2^40: e * R0 at column 0, e << 10 at column 1
    u64 f = (u64)t[0] + (e & 0xFFFFFFFFull) * R0_26;   // e fits 40 bits; keep low 32 (magnitude bounded inputs keep e < 2^32)
    r.n[0] = (u32)f & M26; f >>= 26;
    f += (u64)t[1] + ((e & 0xFFFFFFFFull) << 10);
    r.n[1] = (u32)f & M26; f >>= 26;
    f += t[2];
    r.n[2] = (u32)f & M26; f >>= 26;
    r.n[3] = t[3] + (u32)f;
#pragma unroll
    for (int k = 4; k < 10; k++) r.n[k] = t[k];
}
__device__ __forceinline__ void fe26_add(fe26& r, const fe26& a, const fe26& b) {
#pragma unroll
    for (int i = 0; i < 10; i++) r.n[i] = a.n[i] + b.n[i];
}
// r = a - b + 2p-ish bias (limbs of 4*p in 26-bit form keep every limb positive for magnitude-1 b)
__device__ __forceinline__ void fe26_sub(fe26& r, const fe26& a, const fe26& b) {
    const u32 P4[10] = {4 * 0x3FFFC2Fu, 4 * 0x3FFFFBFu, 4 * 0x3FFFFFFu, 4 * 0x3FFFFFFu, 4 * 0x3FFFFFFu, 4 * 0x3FFFFFFu, 4 * 0x3FFFFFFu, 4 * 0x3FFFFFFu, 4 * 0x3FFFFFFu, 4 * 0x03FFFFFu};
#pragma unroll
    for (int i = 0; i < 10; i++) r.n[i] = a.n[i] + P4[i] - b.n[i];
}
}
