// Random 64-byte record gather rate vs table size (is a 20+ GB fixed-base table still served fast enough?)
#include <hip/hip_runtime.h>
#include <cstdio>
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
    unsigned* out;
    if (hipMalloc(&out, 4096 * 256 * 4) != hipSuccess) return 1;
    for (size_t gb : {1, 3, 8, 24, 48, 96}) {
        void* t;
        size_t bytes = gb << 30;
        if (hipMalloc(&t, bytes) != hipSuccess) { printf("%zu GB: alloc failed\n", gb); continue; }
        hipMemset(t, 1, bytes);
        hipDeviceSynchronize();
        for (int blocks : {1024, 4096}) {
            size_t nreads = (size_t)1 << 26;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            k_gather64<<<blocks, 256>>>((const uint4*)t, bytes / 64, out, nreads);
            hipDeviceSynchronize();
            hipEventRecord(a);
            k_gather64<<<blocks, 256>>>((const uint4*)t, bytes / 64, out, nreads);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("table %3zu GB, %4d blocks: %6.2f G lookups/s = %5.2f TB/s\n", gb, blocks, nreads / ms / 1e6, nreads * 64.0 / ms / 1e9);
        }
        hipFree(t);
    }
    return 0;
}
