// Session anchor for bench.py's `roofline_valu`: the issue rates this chip sustains, measured in the SAME process tree and on the
// same box as the bench line that quotes them (round 2 quoted a round-1 run from another box), with the shader clock the chip held
// while doing so (s_memtime ticks against the 100 MHz s_memrealtime, read inside the kernels).  One JSON object on stdout:
//   mad_u64_u32 / add_xor_shift / fma_f64 : lane-ops per second, chip-wide, 8 independent chains per lane, 8 waves per SIMD
//   *_clock_mhz                           : shader clock during that kernel
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o tools/ratebench tools/ratebench.hip   (built by __graft_entry__.build())
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef uint32_t u32;
typedef uint64_t u64;
struct ClockStamp { u64 shader0, real0, shader1, real1; };

template <int OP>
__global__ void k_rate(u32* out, ClockStamp* st, u32 a0, u32 b0, int iters) {
    u64 acc[8];
    u32 w[8];
    double d[8];
    u32 a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
    const double x = 1.0000001 + threadIdx.x * 1e-9;
#pragma unroll
    for (int j = 0; j < 8; j++) { acc[j] = j + threadIdx.x; w[j] = j ^ b0; d[j] = j; }
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->shader0 = clock64(); st->real0 = wall_clock64(); }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (OP == 0) acc[j] = (u64)a * (u32)(b + j) + acc[j];            // v_mad_u64_u32
            if (OP == 1) w[j] = (w[j] + a) ^ (w[j] >> 3);                    // add + shift + xor: three full-rate ops
            if (OP == 2) d[j] = __builtin_fma(x, d[j], 0.5);                 // v_fma_f64
        }
        a += 3;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->shader1 = clock64(); st->real1 = wall_clock64(); }
    u64 s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s ^= acc[j] ^ w[j] ^ (u64)d[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}

template <int OP>
static int measure(const char* name, int CU, u32* out, ClockStamp* st, double ops_per_iter, bool last) {
    const int blocks = CU * 8, iters = 1 << 14;        // 256-thread blocks: 8 per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
    double best = 0, mhz = 0;
    for (int rep = 0; rep < 3; rep++) {
        k_rate<OP><<<blocks, 256>>>(out, st, 12345, 6789, iters);
        if (hipDeviceSynchronize() != hipSuccess) return 1;
        (void)hipEventRecord(e0);
        k_rate<OP><<<blocks, 256>>>(out, st, 12345, 6789, iters);
        (void)hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) return 1;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ClockStamp h;
        if (hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        const double rate = (double)blocks * 256 * iters * ops_per_iter / (ms * 1e-3);
        if (rate > best) { best = rate; mhz = (double)(h.shader1 - h.shader0) / (double)(h.real1 - h.real0) * 100.0; }
    }
    printf("\"%s\": %.6e, \"%s_clock_mhz\": %.1f%s", name, best, name, mhz, last ? "" : ", ");
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
    const int CU = prop.multiProcessorCount;
    u32* out;
    ClockStamp* st;
    if (hipMalloc(&out, (size_t)CU * 8 * 256 * 4) != hipSuccess || hipMalloc(&st, sizeof(ClockStamp)) != hipSuccess) return 1;
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"nominal_clock_mhz\": %d, \"waves_per_simd\": 8, ", prop.name, prop.gcnArchName, CU, prop.clockRate / 1000);
    if (measure<0>("mad_u64_u32", CU, out, st, 8, false)) return 1;
    if (measure<1>("add_xor_shift", CU, out, st, 24, false)) return 1;
    if (measure<2>("fma_f64", CU, out, st, 8, true)) return 1;
    printf("}\n");
    (void)hipFree(out); (void)hipFree(st);
    return 0;
}
