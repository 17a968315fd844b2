// Integer-throughput microbenchmarks for gfx950: what bounds 256-bit modular arithmetic on MI355X.
// Prints one line per experiment: name, waves/SIMD requested, Gops/s (chip-wide).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../bp_pp_amd/csrc/point.h"
using namespace bppp;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int ILP>
__global__ void k_mad64(u32* out, u32 a0, u32 b0, int iters) {
    u64 acc[ILP];
    u32 a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = j + threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) acc[j] = (u64)a * (u32)(b + j) + acc[j];   // v_mad_u64_u32
        a += 3;
    }
    u64 s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}
template <int ILP>
__global__ void k_mullohi(u32* out, u32 a0, u32 b0, int iters) {
    u32 lo[ILP], hi[ILP];
    u32 a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
#pragma unroll
    for (int j = 0; j < ILP; j++) { lo[j] = j; hi[j] = j; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) { lo[j] += a * (b + j + lo[j]); hi[j] += __umulhi(a, b + j + hi[j]); }
        a += 3;
    }
    u32 s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= lo[j] ^ hi[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_mul24(u32* out, u32 a0, u32 b0, int iters) {
    u32 acc[ILP];
    u32 a = (a0 + threadIdx.x) & 0xFFFFFF, b = (b0 ^ threadIdx.x) & 0xFFFF;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = j;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) acc[j] = __umul24(a, (acc[j] + b + j) & 0xFFFFFF) + acc[j];   // v_mad_u32_u24
    }
    u32 s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_dfma(u32* out, double a0, double b0, int iters) {
    double acc[ILP];
    double a = a0 + threadIdx.x * 1e-9, b = b0;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = j;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) acc[j] = __builtin_fma(a, acc[j], b);
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s;
}
template <int ILP>
__global__ void k_add32(u32* out, u32 a0, u32 b0, int iters) {
    u32 acc[ILP];
    u32 a = a0 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = j ^ b0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) acc[j] = (acc[j] + a) ^ (acc[j] >> 3);   // add + 2 more full-rate ops
    }
    u32 s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_femul(u32* out, int iters) {
    fe a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a.v[i] = 0x9E3779B9u * (threadIdx.x + i + 1); b.v[i] = 0x85EBCA6Bu * (blockIdx.x + i + 7); }
    a.v[7] &= 0x7FFFFFFF; b.v[7] &= 0x7FFFFFFF;
    for (int i = 0; i < iters; i++) { fe_mul(a, a, b); fe_mul(b, b, a); }
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s ^= a.v[i] ^ b.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_femul2(u32* out, int iters) {   // two independent chains per lane
    fe a, b, c, d;
#pragma unroll
    for (int i = 0; i < 8; i++) { a.v[i] = 0x9E3779B9u * (threadIdx.x + i + 1); b.v[i] = 0x85EBCA6Bu * (blockIdx.x + i + 7); c.v[i] = a.v[i] ^ 0x55; d.v[i] = b.v[i] ^ 0x33; }
    a.v[7] &= 0x7FFFFFFF; b.v[7] &= 0x7FFFFFFF; c.v[7] &= 0x7FFFFFFF; d.v[7] &= 0x7FFFFFFF;
    for (int i = 0; i < iters; i++) { fe_mul(a, a, b); fe_mul(c, c, d); fe_mul(b, b, a); fe_mul(d, d, c); }
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s ^= a.v[i] ^ b.v[i] ^ c.v[i] ^ d.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_ptops(u32* out, int iters, int mode) {
    pt p, q;
    pt_set_identity(p);
    fe gx, gy;
    const u32 GX[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu, 0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
    const u32 GY[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u, 0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
#pragma unroll
    for (int i = 0; i < 8; i++) { gx.v[i] = GX[i]; gy.v[i] = GY[i]; }
    q.X = gx; q.Y = gy; fe_set_u32(q.Z, 1);
    apt qa; qa.x = gx; qa.y = gy;
    p = q;
    for (int i = 0; i < threadIdx.x % 7 + 1; i++) pt_dbl(p, p);
    for (int i = 0; i < iters; i++) {
        if (mode == 0) pt_dbl(p, p);
        else if (mode == 1) pt_add(p, p, q);
        else pt_madd_nonid(p, p, qa);
    }
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s ^= p.X.v[i] ^ p.Y.v[i] ^ p.Z.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F&& f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s arch %s CUs %d clock %d MHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    const int CU = prop.multiProcessorCount;
    u32* out;
    CHECK(hipMalloc(&out, (size_t)CU * 32 * 64 * 4 * 4));
    const int iters = 4096;
    for (int wps : {1, 8}) {   // waves per SIMD -> blocks of 256 threads per CU = wps
        int blocks = CU * wps;
        size_t lanes = (size_t)blocks * 256;
        double ms;
        ms = time_ms([&] { k_mad64<8><<<blocks, 256>>>(out, 12345, 6789, iters); });
        printf("mad_u64_u32 ilp8   wps %d : %8.1f Gop/s\n", wps, lanes * (double)iters * 8 / ms / 1e6);
        ms = time_ms([&] { k_mad64<1><<<blocks, 256>>>(out, 12345, 6789, iters); });
        printf("mad_u64_u32 ilp1   wps %d : %8.1f Gop/s\n", wps, lanes * (double)iters * 1 / ms / 1e6);
        ms = time_ms([&] { k_mullohi<4><<<blocks, 256>>>(out, 12345, 6789, iters); });
        printf("mul_lo+mul_hi ilp4 wps %d : %8.1f Gmul/s (lo and hi each counted)\n", wps, lanes * (double)iters * 8 / ms / 1e6);
        ms = time_ms([&] { k_mul24<8><<<blocks, 256>>>(out, 12345, 6789, iters); });
        printf("mad_u32_u24 ilp8   wps %d : %8.1f Gop/s\n", wps, lanes * (double)iters * 8 / ms / 1e6);
        ms = time_ms([&] { k_dfma<8><<<blocks, 256>>>(out, 1.0000001, 0.5, iters); });
        printf("dfma ilp8          wps %d : %8.1f Gop/s\n", wps, lanes * (double)iters * 8 / ms / 1e6);
        ms = time_ms([&] { k_add32<8><<<blocks, 256>>>(out, 12345, 6789, iters); });
        printf("add/xor/shift ilp8 wps %d : %8.1f Gop/s (3 ops per iter counted)\n", wps, lanes * (double)iters * 8 * 3 / ms / 1e6);
    }
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        int blocks = CU * 4 * wps;   // 64-thread blocks: 4 per CU = 1 wave/SIMD
        size_t lanes = (size_t)blocks * 64;
        const int it = 2048;
        double ms = time_ms([&] { k_femul<<<blocks, 64>>>(out, it); });
        printf("fe_mul chain       wps %d : %8.2f G fe_mul/s\n", wps, lanes * (double)it * 2 / ms / 1e6);
        ms = time_ms([&] { k_femul2<<<blocks, 64>>>(out, it); });
        printf("fe_mul 2 chains    wps %d : %8.2f G fe_mul/s\n", wps, lanes * (double)it * 4 / ms / 1e6);
        for (int mode = 0; mode < 3 && wps <= 4; mode++) {
            const int itp = 512;
            ms = time_ms([&] { k_ptops<<<blocks, 64>>>(out, itp, mode); });
            printf("%-18s wps %d : %8.3f G op/s  (%.2f us per op per wave)\n", mode == 0 ? "pt_dbl" : mode == 1 ? "pt_add" : "pt_madd", wps,
                   lanes * (double)itp / ms / 1e6, ms * 1e3 / itp);
        }
    }
    hipFree(out);
    return 0;
}
