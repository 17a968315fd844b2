// Where do the workgroups of an under-filled launch land?  1024 single-wavefront workgroups of a kernel that allows two wavefronts per
// SIMD (231 VGPRs), each recording (XCC, SE, CU, SIMD, wave slot) from the hardware-ID registers and spinning on VALU work for ~1 ms.
// If the dispatcher spreads them one per SIMD, the launch takes one spin; if it packs two per SIMD on half the SIMDs, two.
// usage: wgmap [workgroups=1024] [threads per workgroup=64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

// mode 1: the spin is a chain of v_mad_u64_u32 (the half-rate instruction the 256-bit field arithmetic is made of) instead of 32-bit mul-adds
__global__ __launch_bounds__(64, 2) void k_spin64(unsigned* ids, unsigned long long* clk, int iters, unsigned* sink) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("v_mov_b32 v230, 0" ::: "v230");
    const unsigned long long t0 = wall_clock64();
    unsigned long long a0 = threadIdx.x + 1, a1 = blockIdx.x + 3, a2 = 5, a3 = 7;
    unsigned b = threadIdx.x * 2654435761u + 1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {       // four independent chains: issue-bound, not latency-bound
            a0 = (unsigned long long)(unsigned)a0 * b + a1;
            a1 = (unsigned long long)(unsigned)a1 * b + a2;
            a2 = (unsigned long long)(unsigned)a2 * b + a3;
            a3 = (unsigned long long)(unsigned)a3 * b + a0;
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x % 64 == 0) {
        const int w = blockIdx.x;
        ids[2 * w] = hw; ids[2 * w + 1] = xcc;
        clk[2 * w] = t0; clk[2 * w + 1] = t1;
    }
    if (a0 + a1 + a2 + a3 == 12345) *sink = (unsigned)a0;
}

template <int TPB>
__global__ __launch_bounds__(TPB, 2) void k_spin(unsigned* ids, unsigned long long* clk, int iters, unsigned* sink) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("v_mov_b32 v230, 0" ::: "v230");      // make the allocation 231+ VGPRs: at most two such wavefronts per SIMD
    const unsigned long long t0 = wall_clock64();
    unsigned a = threadIdx.x * 2654435761u + 1, b = blockIdx.x + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) { a = a * b + 0x9e3779b9u; b = (b ^ (a >> 7)) * 5u + k; }
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x % 64 == 0) {
        const int w = blockIdx.x * (TPB / 64) + threadIdx.x / 64;
        ids[2 * w] = hw; ids[2 * w + 1] = xcc;
        clk[2 * w] = t0; clk[2 * w + 1] = t1;
    }
    if (a == 12345 && b == 6789) *sink = a;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 1024, tpb = argc > 2 ? atoi(argv[2]) : 64, mode = argc > 3 ? atoi(argv[3]) : 0;
    const int waves = wgs * (tpb / 64);
    unsigned *d_ids, *d_sink; unsigned long long* d_clk;
    hipMalloc(&d_ids, waves * 8); hipMalloc(&d_clk, waves * 16); hipMalloc(&d_sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (mode == 1) k_spin64<<<wgs, 64>>>(d_ids, d_clk, iters, d_sink);
        else if (tpb == 64) k_spin<64><<<wgs, 64>>>(d_ids, d_clk, iters, d_sink);
        else if (tpb == 128) k_spin<128><<<wgs, 128>>>(d_ids, d_clk, iters, d_sink);
        else k_spin<256><<<wgs, 256>>>(d_ids, d_clk, iters, d_sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> ids(2 * waves); std::vector<unsigned long long> clk(2 * waves);
    hipMemcpy(ids.data(), d_ids, waves * 8, hipMemcpyDeviceToHost); hipMemcpy(clk.data(), d_clk, waves * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_simd, per_cu;
    unsigned long long tmin = ~0ull, tmax = 0, lone = 0;
    for (int w = 0; w < waves; w++) {
        const unsigned hw = ids[2 * w], xcc = ids[2 * w + 1] & 0xf;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned cu_key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        per_cu[cu_key]++; per_simd[(cu_key << 2) | simd]++;
        if (clk[2 * w] < tmin) tmin = clk[2 * w];
        if (clk[2 * w + 1] > tmax) tmax = clk[2 * w + 1];
        lone += clk[2 * w + 1] - clk[2 * w];
    }
    std::map<int, int> hist_simd, hist_cu;
    for (auto& kv : per_simd) hist_simd[kv.second]++;
    for (auto& kv : per_cu) hist_cu[kv.second]++;
    if (mode == 1) printf("[v_mad_u64_u32 chains] ");
    printf("%d workgroups x %d threads = %d wavefronts: launch %.3f ms, mean wavefront %.3f ms (100 MHz clock), span %.3f ms\n", wgs, tpb, waves, ms,
           lone / (double)waves / 1e5, (tmax - tmin) / 1e5);
    printf("  distinct CUs used %zu, distinct SIMDs used %zu\n  wavefronts per SIMD -> number of SIMDs:", per_cu.size(), per_simd.size());
    for (auto& kv : hist_simd) printf("  %d:%d", kv.first, kv.second);
    printf("\n  wavefronts per CU -> number of CUs:");
    for (auto& kv : hist_cu) printf("  %d:%d", kv.first, kv.second);
    printf("\n");
    return 0;
}
