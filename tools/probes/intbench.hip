// Integer-throughput microbenchmarks for gfx950: what bounds 256-bit modular arithmetic on MI355X.
// Prints one line per experiment: name, waves/SIMD requested, Gops/s (chip-wide).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../bp_pp_amd/csrc/point.h"
#include "fe26.h"
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
    u32 wa[8], wb[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { wa[i] = 0x9E3779B9u * (threadIdx.x + i + 1); wb[i] = 0x85EBCA6Bu * (blockIdx.x + i + 7); }
    wa[7] &= 0x7FFFFFFF; wb[7] &= 0x7FFFFFFF;
    fe_from_w8(a, wa); fe_from_w8(b, wb);
    for (int i = 0; i < iters; i++) { fe_mul(a, a, b); fe_mul(b, b, a); }
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) s ^= a.v[i] ^ b.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_femul2(u32* out, int iters) {   // two independent chains per lane
    fe a, b, c, d;
    u32 wa[8], wb[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { wa[i] = 0x9E3779B9u * (threadIdx.x + i + 1); wb[i] = 0x85EBCA6Bu * (blockIdx.x + i + 7); }
    wa[7] &= 0x7FFFFFFF; wb[7] &= 0x7FFFFFFF;
    fe_from_w8(a, wa); fe_from_w8(b, wb);
    wa[0] ^= 0x55; wb[0] ^= 0x33;
    fe_from_w8(c, wa); fe_from_w8(d, wb);
    for (int i = 0; i < iters; i++) { fe_sqr(a, a); fe_sqr(c, c); fe_sqr(b, b); fe_sqr(d, d); }
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) s ^= a.v[i] ^ b.v[i] ^ c.v[i] ^ d.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_ptops(u32* out, int iters, int mode) {
    pt p, q;
    pt_set_identity(p);
    fe gx, gy;
    const u32 GX[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu, 0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
    const u32 GY[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u, 0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
    fe_from_w8(gx, GX); fe_from_w8(gy, GY);
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
    for (int i = 0; i < 10; i++) s ^= p.X.v[i] ^ p.Y.v[i] ^ p.Z.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// synthetic "point addition" op mix on the 10x26 prototype: 12 mul + 22 add/sub (values are garbage; throughput only)
__global__ void k_pt26(u32* out, int iters) {
    using namespace fe26ns;
    fe26 X, Y, Z, x2, y2, z2;
#pragma unroll
    for (int i = 0; i < 10; i++) { X.n[i] = (0x9E3779B9u * (threadIdx.x + i + 1)) & M26; Y.n[i] = (0x85EBCA6Bu * (blockIdx.x + i + 7)) & M26; Z.n[i] = (X.n[i] ^ 0x155555) & M26; x2.n[i] = (Y.n[i] ^ 0x2AAAAA) & M26; y2.n[i] = (X.n[i] + 12345) & M26; z2.n[i] = (Y.n[i] + 777) & M26; }
    for (int it = 0; it < iters; it++) {
        fe26 t0, t1, t2, t3, t4, X3, Y3, Z3;
        fe26_mul(t0, X, x2); fe26_mul(t1, Y, y2); fe26_mul(t2, Z, z2);
        fe26_add(t3, X, Y); fe26_add(t4, x2, y2); fe26_mul(t3, t3, t4);
        fe26_add(t4, t0, t1); fe26_sub(t3, t3, t4); fe26_add(t4, Y, Z); fe26_add(X3, y2, z2); fe26_mul(t4, t4, X3);
        fe26_add(X3, t1, t2); fe26_sub(t4, t4, X3); fe26_add(X3, X, Z); fe26_add(Y3, x2, z2); fe26_mul(X3, X3, Y3);
        fe26_add(Y3, t0, t2); fe26_sub(Y3, X3, Y3); fe26_add(X3, t0, t0); fe26_add(t0, X3, t0);
        fe26_add(Z3, t1, t2); fe26_sub(t1, t1, t2);
        fe26_mul(X3, t4, Y3); fe26_mul(t2, t3, t1); fe26_sub(X3, t2, X3); fe26_mul(Y3, Y3, t0); fe26_mul(t1, t1, Z3);
        fe26_add(Y3, t1, Y3); fe26_mul(t0, t0, t3); fe26_mul(Z3, Z3, t4); fe26_add(Z3, Z3, t0);
#pragma unroll
        for (int i = 0; i < 10; i++) { X.n[i] = X3.n[i] & M26; Y.n[i] = Y3.n[i] & M26; Z.n[i] = Z3.n[i] & M26; }
    }
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) s ^= X.n[i] ^ Y.n[i] ^ Z.n[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_femul26(u32* out, int iters) {
    using namespace fe26ns;
    fe26 a, b;
#pragma unroll
    for (int i = 0; i < 10; i++) { a.n[i] = (0x9E3779B9u * (threadIdx.x + i + 1)) & M26; b.n[i] = (0x85EBCA6Bu * (blockIdx.x + i + 7)) & M26; }
    for (int i = 0; i < iters; i++) { fe26_mul(a, a, b); fe26_mul(b, b, a); }
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) s ^= a.n[i] ^ b.n[i];
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
        printf("fe_sqr 4 chains    wps %d : %8.2f G fe_mul/s\n", wps, lanes * (double)it * 4 / ms / 1e6);
        ms = time_ms([&] { k_femul26<<<blocks, 64>>>(out, it); });
        printf("fe26_mul chain     wps %d : %8.2f G fe_mul/s\n", wps, lanes * (double)it * 2 / ms / 1e6);
        if (wps <= 4) {
            ms = time_ms([&] { k_pt26<<<blocks, 64>>>(out, 512); });
            printf("pt26 add-mix       wps %d : %8.3f G op/s  (%.2f us per op per wave)\n", wps, lanes * 512.0 / ms / 1e6, ms * 1e3 / 512);
        }
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
