// Micro-benchmark: 64-bit right shift (v_lshrrev_b64) vs the two-instruction 32-bit form (v_alignbit_b32 + v_lshrrev_b32) vs
// v_mad_u64_u32, dependent chains x 4 independent streams, at 1 and 4 wavefronts per SIMD.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -o tools/shiftbench tools/shiftbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned u32;
template <int MODE>
__global__ __launch_bounds__(64) void k(u64* out, int iters) {
    u64 a = threadIdx.x * 0x9E3779B97F4A7C15ULL + 12345, b = a * 3 + 1, c = a * 5 + 7, d = a * 7 + 11;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (MODE == 0) {
                asm volatile("v_lshrrev_b64 %0, 26, %0" : "+v"(a)); asm volatile("v_lshrrev_b64 %0, 26, %0" : "+v"(b));
                asm volatile("v_lshrrev_b64 %0, 26, %0" : "+v"(c)); asm volatile("v_lshrrev_b64 %0, 26, %0" : "+v"(d));
                a += 0x123456789ULL << 30; b += 0x123456789ULL << 30; c += 0x123456789ULL << 30; d += 0x123456789ULL << 30;
            } else if (MODE == 1) {
                u32 lo, hi;
#define SH(x) lo = (u32)x; hi = (u32)(x >> 32); asm volatile("v_alignbit_b32 %0, %1, %0, 26" : "+v"(lo) : "v"(hi)); asm volatile("v_lshrrev_b32 %0, 26, %0" : "+v"(hi)); x = ((u64)hi << 32) | lo;
                SH(a) SH(b) SH(c) SH(d)
#undef SH
                a += 0x123456789ULL << 30; b += 0x123456789ULL << 30; c += 0x123456789ULL << 30; d += 0x123456789ULL << 30;
            } else {
                u32 m = 0x3FFFFFF;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a) : "v"(m) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(b) : "v"(m) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(c) : "v"(m) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(d) : "v"(m) : "vcc");
                a += 0x123456789ULL << 30; b += 0x123456789ULL << 30; c += 0x123456789ULL << 30; d += 0x123456789ULL << 30;
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a ^ b ^ c ^ d;
}
template <int MODE>
static void run(const char* name, int blocks) {
    u64* d;
    hipMalloc(&d, (size_t)blocks * 64 * 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000;
    k<MODE><<<blocks, 64>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 64>>>(d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // per inner repetition: 4 "ops" (one per stream) + 4 64-bit adds (8 VALU)
    printf("%-34s blocks %5d: %7.2f ns per group of 4 ops (+ 4 64-bit adds)\n", name, blocks, ms * 1e6 / ((double)iters * 16) * 1.0);
    hipFree(d);
}
int main() {
    for (int blocks : {1024, 4096}) {
        run<0>("v_lshrrev_b64", blocks);
        run<1>("v_alignbit_b32 + v_lshrrev_b32", blocks);
        run<2>("v_mad_u64_u32", blocks);
    }
    return 0;
}
