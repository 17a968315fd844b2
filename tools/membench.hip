// Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950: streams a known byte count with the two access widths the
// verify kernels use (4 B/lane coalesced SoA words, 16 B/lane table entries).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_copy_dword(const unsigned* __restrict__ in, unsigned* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i] + 1;
}
__global__ void k_copy_dwordx4(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { uint4 v = in[i]; v.x += 1; out[i] = v; }
}
// every lane reads one random, 64-byte aligned, 64-byte record (4 x dwordx4) -- the fixed-base table access pattern
__global__ void k_gather64(const uint4* __restrict__ table, size_t nrec, unsigned* __restrict__ out, size_t nreads) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < nreads; i += stride) {
        unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const uint4* r = table + (h % nrec) * 4;
        uint4 a = r[0], b = r[1], c = r[2], d = r[3];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    const size_t bytes = (size_t)1 << 30;   // 1 GiB read + 1 GiB written per kernel (past the 256 MiB Infinity Cache)
    void *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    hipMemset(a, 1, bytes);
    hipMemset(b, 2, bytes);
    hipDeviceSynchronize();
    k_copy_dword<<<2048, 256>>>((const unsigned*)a, (unsigned*)b, bytes / 4);
    k_copy_dwordx4<<<2048, 256>>>((const uint4*)a, (uint4*)b, bytes / 16);
    {   // 2^24 random 64-byte reads (1 GiB) from a 3 GiB table
        void* t;
        const size_t tb = (size_t)3 << 30;
        if (hipMalloc(&t, tb) != hipSuccess) return 1;
        hipMemset(t, 3, tb);
        hipDeviceSynchronize();
        k_gather64<<<2048, 256>>>((const uint4*)t, tb / 64, (unsigned*)b, (size_t)1 << 24);
        hipDeviceSynchronize();
        hipFree(t);
    }
    hipDeviceSynchronize();
    printf("membench: each kernel reads %zu and writes %zu bytes\n", bytes, bytes);
    return 0;
}
