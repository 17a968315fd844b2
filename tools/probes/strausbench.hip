// Micro-benchmark of the Jacobian shared-doubling loop (straus_affine_fast<2>, the body of k_verify_round) at one wavefront per
// SIMD: per-lane window tables in HBM (as in the product) vs. one table shared by every lane (cache resident), to separate
// table-gather latency from instruction issue.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/strausbench tools/strausbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../bp_pp_amd/csrc/verify_core.h"
using namespace bppp;
__global__ __launch_bounds__(64) void k_straus(const apt_packed* tab, int shared_table, u32* out, int reps) {
    size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    glv_words<2> g;
    for (int st = 0; st < 4; st++) {
        for (int l = 0; l < 5; l++) g.w[st][l] = (u32)((t + 1) * 2654435761u + st * 40503u + l * 97u) * 2246822519u;
        g.w[st][4] &= 0xFu;
        g.neg[st] = (t >> st) & 1;
    }
    const int pidx[2] = {3, 7};
    const apt_packed* my = tab + (shared_table ? 0 : t * BPPP_ATAB_PER_PROOF);
    u32 x = 0;
    for (int r = 0; r < reps; r++) {
        pt o;
        straus_affine_fast<2>(o, my, pidx, g);
        x ^= o.X.v[0];
        g.w[0][0] ^= x;
    }
    out[t] = x;
}
int main() {
    const size_t n = 131072;
    apt_packed* d_tab;
    u32* d_out;
    (void)hipMalloc(&d_tab, n * BPPP_ATAB_PER_PROOF * sizeof(apt_packed));
    (void)hipMalloc(&d_out, n * 4);
    std::vector<u32> h(n * BPPP_ATAB_PER_PROOF * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (u32)(i * 2654435761u + 12345u) & 0x7FFFFFFFu;
    (void)hipMemcpy(d_tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int blocks : {256, 1024, 2048}) {
        for (int shared_table = 0; shared_table < 2; shared_table++) {
            k_straus<<<blocks, 64>>>(d_tab, shared_table, d_out, 1);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            k_straus<<<blocks, 64>>>(d_tab, shared_table, d_out, 2);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms;
            (void)hipEventElapsedTime(&ms, a, b);
            printf("blocks %5d  %-22s %8.3f ms per 2-point shared-doubling sum\n", blocks, shared_table ? "one shared table" : "per-lane tables (HBM)", ms / 2);
        }
    }
    return 0;
}
