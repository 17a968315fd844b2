// Micro-benchmark: Fp multiplication on 9 x 29-bit limbs against the shipped 10 x 26-bit form (csrc/field.h), same reduction style.
// 81 limb products instead of 100 and 8 reduction steps instead of 9, at the price of almost no headroom (9 x 2^58 = 2^61.2 per
// column: operands must be weakly normalised before every multiplication).  Prints G fe_mul/s for dependent chains at 1..4 waves
// per SIMD, and for a "multiplication + the weak normalisation a sum would need" chain.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/fe29bench tools/fe29bench.hip && ./tools/fe29bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../bp_pp_amd/csrc/field.h"
using namespace bppp;
#define M29 0x1FFFFFFFu
#define R0_29 0x7A20u   // 2^261 mod p = 2^37 + 0x7A20  ->  0x7A20 at limb 0, 2^8 at limb 1
struct fe29 { u32 v[9]; };
__device__ __forceinline__ u32 opq(u32 x) { asm("" : "+v"(x)); return x; }
__device__ __forceinline__ void fe29_reduce(fe29& r, const u64 c[17]) {
    const u32 k256 = opq(256u);
    u64 d = c[8];
    const u32 t8 = (u32)d & M29;
    d >>= 29;
    u64 e = 0;
    u32 t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        d += c[k + 9];
        const u32 u = (u32)d & M29;
        d >>= 29;
        e += c[k] + (u64)u * R0_29;
        t[k] = (u32)e & M29;
        e >>= 29;
        e += (u64)u * k256;
    }
    // d = digit of column 17 (< 2^34): contributes d * R at column 8
    const u32 d_lo = (u32)d, d_hi = (u32)(d >> 32);
    e += (u64)t8 + (u64)d_lo * R0_29 + (((u64)d_hi * R0_29) << 32);
    r.v[8] = (u32)e & M29;
    e >>= 29;                       // units of 2^261
    e += d << 8;                    // d * 2^8 at column 9 = 2^261
    const u32 e0 = (u32)e & M29, e1 = (u32)(e >> 29);
    u64 f = (u64)t[0] + (u64)e0 * R0_29;
    r.v[0] = (u32)f & M29; f >>= 29;
    f += (u64)t[1] + (u64)e0 * k256 + (u64)e1 * R0_29;
    r.v[1] = (u32)f & M29; f >>= 29;
    f += (u64)t[2] + (u64)e1 * k256;
    r.v[2] = (u32)f & M29; f >>= 29;
    r.v[3] = t[3] + (u32)f;
#pragma unroll
    for (int k = 4; k < 8; k++) r.v[k] = t[k];
}
__device__ __forceinline__ void fe29_mul(fe29& r, const fe29& a, const fe29& b) {
    u64 c[17];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        const int i0 = k < 9 ? 0 : k - 8, i1 = k < 9 ? k : 8;
        u64 acc = (u64)a.v[i0] * b.v[k - i0];
#pragma unroll
        for (int i = i0 + 1; i <= i1; i++) acc += (u64)a.v[i] * b.v[k - i];
        c[k] = acc;
    }
    fe29_reduce(r, c);
}
__device__ __forceinline__ void fe29_sqr(fe29& r, const fe29& a) {
    u32 a2[9];
#pragma unroll
    for (int i = 0; i < 9; i++) a2[i] = a.v[i] << 1;
    u64 c[17];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        const int i0 = k < 9 ? 0 : k - 8;
        u64 acc = 0;
#pragma unroll
        for (int i = i0; 2 * i < k; i++) acc += (u64)a2[i] * a.v[k - i];
        if ((k & 1) == 0) acc += (u64)a.v[k / 2] * a.v[k / 2];
        c[k] = acc;
    }
    fe29_reduce(r, c);
}
// weak normalisation of a sum / difference (limbs up to a few times 2^29): one carry pass + fold of the top carry
__device__ __forceinline__ void fe29_weak(fe29& r) {
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { const u32 v = r.v[i] + c; r.v[i] = v & M29; c = v >> 29; }
    const u32 f0 = r.v[0] + c * R0_29;        // c < 8
    r.v[0] = f0 & M29;
    r.v[1] += (c << 8) + (f0 >> 29);
}
__global__ void k_mul29(u32* out, int iters) {
    fe29 a, b;
#pragma unroll
    for (int i = 0; i < 9; i++) { a.v[i] = (0x9E3779B9u * (threadIdx.x + i + 1)) & M29; b.v[i] = (0x85EBCA6Bu * (blockIdx.x + i + 7)) & M29; }
    for (int i = 0; i < iters; i++) { fe29_mul(a, a, b); fe29_mul(b, b, a); }
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) s ^= a.v[i] ^ b.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul29_weak(u32* out, int iters) {      // every second operand is a sum that needs the weak pass first
    fe29 a, b;
#pragma unroll
    for (int i = 0; i < 9; i++) { a.v[i] = (0x9E3779B9u * (threadIdx.x + i + 1)) & M29; b.v[i] = (0x85EBCA6Bu * (blockIdx.x + i + 7)) & M29; }
    for (int i = 0; i < iters; i++) {
        fe29 s;
#pragma unroll
        for (int k = 0; k < 9; k++) s.v[k] = a.v[k] + b.v[k] + b.v[k];
        fe29_weak(s);
        fe29_mul(a, s, b);
        fe29_sqr(b, a);
    }
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) s ^= a.v[i] ^ b.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul26(u32* out, int iters) {
    fe a, b;
#pragma unroll
    for (int i = 0; i < 10; i++) { a.v[i] = (0x9E3779B9u * (threadIdx.x + i + 1)) & BPPP_M26; b.v[i] = (0x85EBCA6Bu * (blockIdx.x + i + 7)) & BPPP_M26; }
    for (int i = 0; i < iters; i++) { fe_mul(a, a, b); fe_mul(b, b, a); }
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) s ^= a.v[i] ^ b.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul26_sum(u32* out, int iters) {      // the same sum-then-multiply chain on the unsaturated 26-bit limbs: no pass needed
    fe a, b;
#pragma unroll
    for (int i = 0; i < 10; i++) { a.v[i] = (0x9E3779B9u * (threadIdx.x + i + 1)) & BPPP_M26; b.v[i] = (0x85EBCA6Bu * (blockIdx.x + i + 7)) & BPPP_M26; }
    for (int i = 0; i < iters; i++) {
        fe s;
#pragma unroll
        for (int k = 0; k < 10; k++) s.v[k] = a.v[k] + b.v[k] + b.v[k];
        fe_mul(a, s, b);
        fe_sqr(b, a);
    }
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) s ^= a.v[i] ^ b.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
static double time_ms(F&& f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int CU = prop.multiProcessorCount;
    u32* out; hipMalloc(&out, (size_t)CU * 32 * 64 * 4 * 4);
    const int it = 2048;
    for (int wps : {1, 2, 4}) {
        int blocks = CU * 4 * wps; size_t lanes = (size_t)blocks * 64;
        double m26 = time_ms([&] { k_mul26<<<blocks, 64>>>(out, it); });
        double m29 = time_ms([&] { k_mul29<<<blocks, 64>>>(out, it); });
        double s26 = time_ms([&] { k_mul26_sum<<<blocks, 64>>>(out, it); });
        double s29 = time_ms([&] { k_mul29_weak<<<blocks, 64>>>(out, it); });
        printf("wps %d: mul chain 10x26 %7.2f G/s | 9x29 %7.2f G/s (x%.3f)   sum+mul+sqr chain 10x26 %7.2f | 9x29 with weak pass %7.2f (x%.3f)\n", wps,
               lanes * (double)it * 2 / m26 / 1e6, lanes * (double)it * 2 / m29 / 1e6, m26 / m29, lanes * (double)it * 2 / s26 / 1e6,
               lanes * (double)it * 2 / s29 / 1e6, s26 / s29);
    }
    return 0;
}
