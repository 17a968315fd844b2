// Random 64-byte gathers from a LARGE table (160 GB, the size class of the fixed-base tables) when the lookups of any one moment fall into
// a WINDOW of it: is the drop of the gather rate beyond a few GB (gatherbench: 56 G/s at 1 GB, 19 G/s from 8 GB) a property of the live
// set -- address translation reach -- or of the allocation?  Every launch draws its 2^26 lookups from one window of `win` bytes at a
// base that moves from launch to launch; also K windows of win / K bytes spread over the table (K sub-tables live at once).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_gather64(const uint4* __restrict__ table, size_t base_rec, size_t win_rec, int k, size_t spread_rec, unsigned* __restrict__ out,
                           size_t nreads) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    const size_t per = win_rec / (size_t)k;
    for (; i < nreads; i += stride) {
        unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const size_t r0 = h % win_rec, piece = r0 / per, rec = base_rec + piece * spread_rec + (r0 - piece * per);
        const uint4* r = table + rec * 4;
        uint4 a = r[0], b = r[1], c = r[2], d = r[3];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main(int argc, char** argv) {
    const size_t gb = argc > 1 ? (size_t)atol(argv[1]) : 160;
    unsigned* out;
    if (hipMalloc(&out, 4096 * 256 * 4) != hipSuccess) return 1;
    void* t;
    const size_t bytes = gb << 30, nrec = bytes / 64;
    if (hipMalloc(&t, bytes) != hipSuccess) { printf("%zu GB: alloc failed\n", gb); return 1; }
    hipMemset(t, 1, bytes);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const size_t nreads = (size_t)1 << 26;
    const double wins_gb[] = {0.25, 0.5, 1, 2, 3, 4, 6, 8, 12, 16, 32, 64, (double)gb};
    for (int k : {1, 8, 64}) {
        for (double wg : wins_gb) {
            const size_t win_rec = (size_t)(wg * (double)(1ull << 30)) / 64;
            if (win_rec > nrec) continue;
            const size_t spread_rec = k > 1 ? nrec / (size_t)k : 0;     // piece j of the window sits at base + j * (table / k)
            if (k > 1 && win_rec / (size_t)k > spread_rec) continue;
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                const size_t room = k > 1 ? spread_rec - win_rec / (size_t)k : nrec - win_rec;
                const size_t base_rec = room ? ((size_t)rep * 0x9E3779B97F4Aull) % room : 0;
                hipEventRecord(a);
                k_gather64<<<2048, 256>>>((const uint4*)t, base_rec, win_rec, k, spread_rec, out, nreads);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("table %3zu GB, live window %6.2f GB in %2d piece(s): %6.2f G lookups/s = %5.2f TB/s\n", gb, wg, k, nreads / best / 1e6, nreads * 64.0 / best / 1e9);
            fflush(stdout);
        }
    }
    hipFree(t);
    return 0;
}
