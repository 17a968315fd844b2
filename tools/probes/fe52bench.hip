// The one multiplier round 2 had not measured: Fp multiplication's limb products on the FP64 FMA pipe.
//
// Scheme (Emmart, "Faster modular exponentiation using double precision floating point arithmetic on the GPU", 2018): operands as
// 5 limbs of 52 bits held exactly in doubles.  With round-toward-zero,
//     hi  = fma(a, b, 2^104)            = 2^104 + floor(ab / 2^52) * 2^52        (the sum's ulp is 2^52)
//     lo  = fma(a, b, (2^104 + 2^52) - hi) = 2^52 + (ab mod 2^52)                 (exact)
// so the mantissa bits of hi / lo are the two 52-bit halves of the 104-bit product, and column sums are INTEGER additions of the
// bit patterns (the exponent fields add up to a known constant per column).  25 limb products = 50 v_fma_f64 + 25 v_add_f64 +
// 50 64-bit integer additions, against the product's 100 v_mad_u64_u32 in the shipped 10 x 26-bit form (field.h), whose
// multiply-add accumulates for free.
//
// What is measured: ONLY the schoolbook product (columns of the 520-bit result, no reduction mod p, no conversion of the result
// back into limbs beyond a cheap feedback that keeps the chain dependent) -- the FP64 form's best case; its reduction needs the
// same split again (2^260 = 2^36 + 15632 mod p is a 37-bit factor) plus 64-bit shifts.  Both kernels are checked against
// unsigned __int128 arithmetic on the host for every lane of the first workgroup.  The shader clock is read in the kernels
// (s_memtime against the 100 MHz s_memrealtime), so cycles per product are cycles at the clock the chip actually held.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/fe52bench tools/fe52bench.hip && ./tools/fe52bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef uint32_t u32;
typedef uint64_t u64;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

static const u64 M52 = (1ull << 52) - 1;
static const u32 M26 = (1u << 26) - 1;

struct ClockStamp { u64 shader0, real0, shader1, real1; };
__device__ __forceinline__ void stamp_begin(ClockStamp* s) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { s->shader0 = clock64(); s->real0 = wall_clock64(); }
}
__device__ __forceinline__ void stamp_end(ClockStamp* s) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { s->shader1 = clock64(); s->real1 = wall_clock64(); }
}

__device__ __forceinline__ double fma_rz(double a, double b, double c) {
    double d;
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));    // MODE.fp_round (f64) = toward zero, set by the kernel
    return d;
}
__device__ __forceinline__ u64 bits(double d) { return (u64)__double_as_longlong(d); }
__device__ __forceinline__ double from_limb(u64 v) { return __longlong_as_double((long long)(v | 0x4330000000000000ull)) - 4503599627370496.0; }

// columns of a * b, 5 x 52-bit limbs: cols[k] = sum of the 52-bit halves that land in column k (k = 0..9), each < 2^56
__device__ __forceinline__ void prod52(u64 cols[10], const double a[5], const double b[5]) {
    const double C1 = 20282409603651670423947251286016.0;             // 2^104
    const double C2 = 20282409603651670423947251286016.0 + 4503599627370496.0;   // 2^104 + 2^52
#pragma unroll
    for (int k = 0; k < 10; k++) cols[k] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const double hi = fma_rz(a[i], b[j], C1);
            const double lo = fma_rz(a[i], b[j], C2 - hi);
            cols[i + j + 1] += bits(hi);
            cols[i + j] += bits(lo);
        }
    }
    // remove the exponent fields: column k received n_lo(k) lo patterns (0x433 << 52) and n_hi(k) hi patterns (0x467 << 52)
#pragma unroll
    for (int k = 0; k < 10; k++) {
        const int n_lo = k <= 4 ? k + 1 : (k <= 8 ? 9 - k : 0);
        const int kk = k - 1;
        const int n_hi = k == 0 ? 0 : (kk <= 4 ? kk + 1 : 9 - kk);
        cols[k] -= (u64)n_lo * 0x4330000000000000ull + (u64)n_hi * 0x4670000000000000ull;
    }
}
__global__ void k_prod52(u64* out, ClockStamp* st, int iters, u64 seed) {
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");              // f64 rounding: toward zero
    double a[5], b[5];
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        a[i] = from_limb((seed * (2 * t + 3) * (i + 1) * 0x9E3779B97F4A7C15ull >> 7) & M52);
        b[i] = from_limb(((seed + 77) * (2 * t + 5) * (i + 3) * 0xC2B2AE3D27D4EB4Full >> 9) & M52);
    }
    u64 cols[10];
    stamp_begin(st);
    for (int it = 0; it < iters; it++) {
        prod52(cols, a, b);
        if (it + 1 < iters) {
#pragma unroll
            for (int i = 0; i < 5; i++) a[i] = from_limb((cols[i] ^ cols[i + 5]) & M52);   // keeps the chain dependent: 3 cheap ops per limb
        }
    }
    stamp_end(st);
#pragma unroll
    for (int k = 0; k < 10; k++) out[t * 10 + k] = cols[k];
}

// the shipped form's product: 10 x 26-bit limbs, 100 v_mad_u64_u32 into 19 64-bit columns
__device__ __forceinline__ void prod26(u64 cols[19], const u32 a[10], const u32 b[10]) {
#pragma unroll
    for (int k = 0; k < 19; k++) cols[k] = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
#pragma unroll
        for (int j = 0; j < 10; j++) cols[i + j] = (u64)a[i] * b[j] + cols[i + j];
    }
}
__global__ void k_prod26(u64* out, ClockStamp* st, int iters, u64 seed) {
    u32 a[10], b[10];
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        a[i] = (u32)((seed * (2 * t + 3) * (i + 1) * 0x9E3779B97F4A7C15ull) >> 20) & M26;
        b[i] = (u32)(((seed + 77) * (2 * t + 5) * (i + 3) * 0xC2B2AE3D27D4EB4Full) >> 22) & M26;
    }
    u64 cols[19];
    stamp_begin(st);
    for (int it = 0; it < iters; it++) {
        prod26(cols, a, b);
        if (it + 1 < iters) {
#pragma unroll
            for (int i = 0; i < 10; i++) a[i] = (u32)(cols[i] ^ cols[(i + 9) % 19]) & M26;
        }
    }
    stamp_end(st);
#pragma unroll
    for (int k = 0; k < 19; k++) out[t * 19 + k] = cols[k];
}
// raw issue rates in the same session: v_fma_f64, v_mad_u64_u32, 64-bit integer add, each 8 independent chains per lane
template <int OP>
__global__ void k_rate(u64* out, ClockStamp* st, int iters, u64 seed) {
    if (OP == 0) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    double d[8];
    u64 q[8];
    const double x = from_limb((seed * (t + 1)) & M52), y = from_limb((seed * (t + 9) >> 3) & M52);
    const u32 m = (u32)(seed * (t + 3)), n = (u32)(seed >> 5) + (u32)t;
#pragma unroll
    for (int j = 0; j < 8; j++) { d[j] = (double)(j + 1); q[j] = seed + j * t; }
    stamp_begin(st);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (OP == 0) d[j] = fma_rz(x, y, d[j]);
            if (OP == 1) q[j] = (u64)m * (u32)(n + j) + q[j];
            if (OP == 2) q[j] = q[j] + ((u64)m << 32 | (n + j)) + (q[j] >> 63);       // add with a carry-dependent term: stays a 64-bit add
        }
    }
    stamp_end(st);
    u64 s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s ^= q[j] ^ bits(d[j]);
    out[t] = s;
}

typedef unsigned __int128 u128;
static bool check52(const std::vector<u64>& out, int lanes, int iters, u64 seed) {
    for (int t = 0; t < lanes; t++) {
        u64 a[5], b[5], cols[10];
        for (int i = 0; i < 5; i++) {
            a[i] = (seed * (2 * (u64)t + 3) * (i + 1) * 0x9E3779B97F4A7C15ull >> 7) & M52;
            b[i] = ((seed + 77) * (2 * (u64)t + 5) * (i + 3) * 0xC2B2AE3D27D4EB4Full >> 9) & M52;
        }
        for (int it = 0; it < iters; it++) {
            for (int k = 0; k < 10; k++) cols[k] = 0;
            for (int i = 0; i < 5; i++)
                for (int j = 0; j < 5; j++) {
                    const u128 p = (u128)a[i] * b[j];
                    cols[i + j] += (u64)p & M52;
                    cols[i + j + 1] += (u64)(p >> 52);
                }
            if (it + 1 < iters)
                for (int i = 0; i < 5; i++) a[i] = (cols[i] ^ cols[i + 5]) & M52;
        }
        for (int k = 0; k < 10; k++)
            if (out[(size_t)t * 10 + k] != cols[k]) { printf("prod52 MISMATCH lane %d col %d: %llx vs %llx\n", t, k, (unsigned long long)out[(size_t)t * 10 + k], (unsigned long long)cols[k]); return false; }
    }
    return true;
}
static bool check26(const std::vector<u64>& out, int lanes, int iters, u64 seed) {
    for (int t = 0; t < lanes; t++) {
        u32 a[10], b[10];
        u64 cols[19];
        for (int i = 0; i < 10; i++) {
            a[i] = (u32)((seed * (2 * (u64)t + 3) * (i + 1) * 0x9E3779B97F4A7C15ull) >> 20) & M26;
            b[i] = (u32)(((seed + 77) * (2 * (u64)t + 5) * (i + 3) * 0xC2B2AE3D27D4EB4Full) >> 22) & M26;
        }
        for (int it = 0; it < iters; it++) {
            for (int k = 0; k < 19; k++) cols[k] = 0;
            for (int i = 0; i < 10; i++)
                for (int j = 0; j < 10; j++) cols[i + j] += (u64)a[i] * b[j];
            if (it + 1 < iters)
                for (int i = 0; i < 10; i++) a[i] = (u32)(cols[i] ^ cols[(i + 9) % 19]) & M26;
        }
        for (int k = 0; k < 19; k++)
            if (out[(size_t)t * 19 + k] != cols[k]) { printf("prod26 MISMATCH lane %d col %d\n", t, k); return false; }
    }
    return true;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int CU = prop.multiProcessorCount;
    printf("device: %s arch %s CUs %d nominal clock %d MHz\n", prop.name, prop.gcnArchName, CU, prop.clockRate / 1000);
    u64* out;
    ClockStamp* st;
    const size_t max_lanes = (size_t)CU * 4 * 8 * 64;
    CHECK(hipMalloc(&out, max_lanes * 19 * sizeof(u64)));
    CHECK(hipMalloc(&st, sizeof(ClockStamp)));
    const u64 seed = 0x1234567ull;
    auto run = [&](const char* name, auto launch, double ops_per_lane, int wps) -> int {
        const int blocks = CU * 4 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        launch(blocks);                                   // warm-up
        CHECK(hipDeviceSynchronize());
        hipEventRecord(e0);
        launch(blocks);
        hipEventRecord(e1);
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ClockStamp h;
        CHECK(hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost));
        const double mhz = (double)(h.shader1 - h.shader0) / (double)(h.real1 - h.real0) * 100.0;
        const double lane_ops = (double)blocks * 64 * ops_per_lane;
        // per SIMD: wps waves, each issuing ops_per_lane ops in ms -> cycles per wave-op at the measured clock
        const double cyc = ms * 1e-3 * mhz * 1e6 / (ops_per_lane * wps);
        printf("%-22s wps %d : %8.2f ms  %9.1f G lane-ops/s  shader clock %6.0f MHz  %6.2f cycles per wave-op per SIMD\n", name, wps, ms,
               lane_ops / ms / 1e6, mhz, cyc);
        hipEventDestroy(e0); hipEventDestroy(e1);
        return 0;
    };
    const int iters = 512;
    // correctness first (one workgroup's lanes, short chain)
    {
        const int it = 5;
        k_prod52<<<1, 64>>>(out, st, it, seed);
        std::vector<u64> h(64 * 10);
        CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        const bool ok52 = check52(h, 64, it, seed);
        k_prod26<<<1, 64>>>(out, st, it, seed);
        std::vector<u64> g(64 * 19);
        CHECK(hipMemcpy(g.data(), out, g.size() * 8, hipMemcpyDeviceToHost));
        const bool ok26 = check26(g, 64, it, seed);
        printf("check vs unsigned __int128 on the host: prod52 (FP64 FMA, round toward zero) %s, prod26 (v_mad_u64_u32) %s\n", ok52 ? "PASS" : "FAIL",
               ok26 ? "PASS" : "FAIL");
        if (!ok52 || !ok26) return 2;
    }
    for (int wps : {1, 2, 3, 4, 8}) {
        if (run("v_fma_f64 x8", [&](int b) { k_rate<0><<<b, 64>>>(out, st, 4096, seed); }, 4096.0 * 8, wps)) return 1;
        if (run("v_mad_u64_u32 x8", [&](int b) { k_rate<1><<<b, 64>>>(out, st, 4096, seed); }, 4096.0 * 8, wps)) return 1;
        if (run("64-bit add (+shift) x8", [&](int b) { k_rate<2><<<b, 64>>>(out, st, 4096, seed); }, 4096.0 * 8, wps)) return 1;
    }
    for (int wps : {1, 2, 3, 4}) {
        if (run("product 5x52 FP64", [&](int b) { k_prod52<<<b, 64>>>(out, st, iters, seed); }, iters, wps)) return 1;
        if (run("product 10x26 mad64", [&](int b) { k_prod26<<<b, 64>>>(out, st, iters, seed); }, iters, wps)) return 1;
    }
    hipFree(out); hipFree(st);
    return 0;
}
