// Micro-benchmark: field / scalar inversion by division steps vs exponentiation, one wavefront per SIMD (the occupancy the
// verifier's per-proof kernels run at) and 4 per SIMD.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/invbench tools/invbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../bp_pp_amd/csrc/field.h"
using namespace bppp;
template <int MODE>
__global__ __launch_bounds__(64) void k_inv(u32* out, int iters) {
    size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    u32 w[8];
    for (int i = 0; i < 8; i++) w[i] = (u32)(t * 2654435761u + i * 40503u + 12345u);
    w[7] &= 0x7FFFFFFFu;
    if (MODE < 2) {
        fe a, r;
        fe_from_w8(a, w);
        for (int i = 0; i < iters; i++) {
            if (MODE == 0) fe_inv(r, a); else fe_inv_fermat(r, a);
            fe_add(a, r, a);
            fe_mul_small(a, a, 1);
        }
        fe_to_w8(w, a);
    } else {
        sc a, r;
        for (int i = 0; i < 8; i++) a.v[i] = w[i];
        for (int i = 0; i < iters; i++) {
            if (MODE == 2) sc_inv(r, a); else sc_inv_fermat(r, a);
            sc_add(a, r, a);
        }
        for (int i = 0; i < 8; i++) w[i] = a.v[i];
    }
    u32 x = 0;
    for (int i = 0; i < 8; i++) x ^= w[i];
    out[t] = x;
}
template <int MODE>
static void run(const char* name, int blocks, int iters) {
    u32* d;
    hipMalloc(&d, (size_t)blocks * 64 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_inv<MODE><<<blocks, 64>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_inv<MODE><<<blocks, 64>>>(d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s blocks %5d: %8.2f us per inversion per wave\n", name, blocks, ms * 1000.0 / iters);
    hipFree(d);
}
int main() {
    for (int blocks : {1024, 4096}) {
        run<0>("fe_inv  divsteps", blocks, 8);
        run<1>("fe_inv  fermat", blocks, 8);
        run<2>("sc_inv  divsteps", blocks, 8);
        run<3>("sc_inv  fermat", blocks, 4);
    }
    return 0;
}
