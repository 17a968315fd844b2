// Does a wavefront whose upper 32 lanes are inactive issue its VALU instructions in half the cycles on gfx950?  (If it did, a batch of
// 2^16 proofs -- one wavefront per SIMD at one lane per proof -- could run as 2,048 half-filled wavefronts, two per SIMD, at no cost.)
// Same dependent v_mad_u64_u32 chain per active lane; (a) 1024 workgroups x 64 lanes, (b) 2048 x 32, (c) 2048 x 64 (twice the work).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void chain(unsigned long long* out, int iters) {
    unsigned long long a = threadIdx.x + 1, b = blockIdx.x * 2654435761u + 12345;
    unsigned x = (unsigned)a | 1u, y = (unsigned)b | 3u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) { a = (unsigned long long)x * (unsigned)a + b; b = (unsigned long long)y * (unsigned)b + a; }
    }
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = a ^ b;
}
int main() {
    unsigned long long* d;
    hipMalloc(&d, sizeof(*d) * 64 * 8192);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    struct { const char* name; int grid, block; } cfg[] = {{"1024 x 64", 1024, 64}, {"2048 x 32", 2048, 32}, {"2048 x 64", 2048, 64}, {"4096 x 16", 4096, 16}, {"1024 x 32", 1024, 32}};
    for (auto& c : cfg) {
        chain<<<c.grid, c.block>>>(d, 10);
        hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 3; r++) {
            hipEventRecord(e0);
            chain<<<c.grid, c.block>>>(d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-10s %8.3f ms\n", c.name, best);
    }
    return 0;
}
