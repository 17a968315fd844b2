// TEST-ONLY stress driver of csrc/coalesce_core.h with a trivial batched call, built with -fsanitize=thread by
// tests/test_coalesce_emul.py: the ring's hand-over of rows between caller threads and dispatcher threads (slot claim under the
// lock, row copies outside it, futex completion word, batch recycling by the last reader, shutdown with callers inside) under a
// race detector, at rates the emulated verifier cannot reach.  Exit code 0 = every request got its own row's answer.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../bp_pp_amd/csrc/coalesce_core.h"

struct Backend {
    std::atomic<long> live{0}, rows{0}, calls{0};
    int delay_us;
    void* alloc_staging(size_t b) { live++; return std::malloc(b ? b : 1); }
    void free_staging(void* p) { live--; std::free(p); }
    bool start_lane(int) { return true; }
    std::string last_error() { return "emulated failure"; }
    void set_last_error(const std::string&) {}
    void stop_lane(int) {}
    int run(int, size_t n, uint8_t* const in[], uint8_t* const out[]) {
        if (delay_us) std::this_thread::sleep_for(std::chrono::microseconds(delay_us));
        const uint64_t* a = (const uint64_t*)in[0];
        const uint32_t* b = (const uint32_t*)in[1];
        uint64_t* r = (uint64_t*)out[0];
        for (size_t i = 0; i < n; i++) r[i] = a[i] * 3 + b[i];
        rows += (long)n;
        calls++;
        return 0;
    }
};

static int phase(int threads, int per_thread, size_t max_batch, long wait_us, int lanes, int delay_us, int shutdown_after_us, int bump_delay_us = 0,
                 int think_us = 0) {
    Backend be;
    be.delay_us = delay_us;
    bppp_host::CoalesceShape sh;
    sh.n_in = 2; sh.in_stride[0] = 8; sh.in_stride[1] = 4;
    sh.n_out = 1; sh.out_stride[0] = 8;
    std::atomic<long> ok{0}, closed{0}, wrong{0};
    {
        bppp_host::Coalescer<Backend> co(&be, sh, max_batch, wait_us, lanes, -7, -5);
        co.set_test_bump_delay_us(bump_delay_us);
        if (co.start() != 0) return 2;
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++)
            th.emplace_back([&, t]() {
                for (int k = 0; k < per_thread; k++) {
                    uint64_t a = (uint64_t)t * 1000003u + (uint64_t)k;
                    uint32_t b = (uint32_t)(t ^ k);
                    uint64_t r = 0;
                    const void* in[4] = {&a, &b, nullptr, nullptr};
                    void* out[4] = {&r, nullptr, nullptr, nullptr};
                    if (think_us) std::this_thread::sleep_for(std::chrono::microseconds((unsigned)((t * 7919 + k * 104729) % (think_us + 1))));
                    const int rc = co.submit(in, out);
                    if (rc == 0) { if (r == a * 3 + b) ok++; else wrong++; }
                    else if (rc == -7) closed++;
                    else wrong++;
                }
            });
        if (shutdown_after_us >= 0) {
            std::this_thread::sleep_for(std::chrono::microseconds(shutdown_after_us));
            co.shutdown();
        }
        for (auto& t : th) t.join();
        const auto s = co.stats();
        if ((long)s.requests != ok.load() + wrong.load()) { std::fprintf(stderr, "requests %lu != answered %ld\n", (unsigned long)s.requests, ok.load() + wrong.load()); return 3; }
        if (s.largest_batch > max_batch) return 4;
    }
    const long total = (long)threads * per_thread;
    std::printf("threads %d x %d max %zu wait %ld lanes %d: ok %ld closed %ld wrong %ld batched calls %ld\n", threads, per_thread, max_batch, wait_us,
                lanes, ok.load(), closed.load(), wrong.load(), be.calls.load());
    if (wrong.load() || ok.load() + closed.load() != total || be.rows.load() != ok.load() || be.live.load() != 0) return 5;
    if (shutdown_after_us < 0 && closed.load()) return 6;
    return 0;
}

int main() {
    int rc = 0;
    if ((rc = phase(32, 400, 16, 50, 3, 20, -1))) return rc;
    if ((rc = phase(64, 100, 1024, 100, 2, 100, -1))) return rc;     // everybody fits one batch: sealed by the deadline
    if ((rc = phase(8, 300, 1, 0, 2, 0, -1))) return rc;            // degenerate: batches of one, no wait
    if ((rc = phase(64, 200, 2, 10, 1, 30, -1))) return rc;         // back-pressure: 64 callers, staging for 6 rows
    if ((rc = phase(16, 2000, 4, 20, 1, 50, 3000))) return rc;      // shut down with callers inside
    if ((rc = phase(16, 2000, 4, 20, 4, 50, 0))) return rc;         // shut down at once
    // batches of 1 .. 12 rows sealed by the deadline, their staging sets re-used at once, and a dispatcher that is slow between two of its
    // eight completion words: a caller of the NEXT use of a staging set must never be woken by the previous use's bumps (round 4: it was,
    // and returned the previous batch's row -- a stale accept)
    if ((rc = phase(12, 150, 16, 30, 1, 0, -1, 300, 400))) return rc;
    if ((rc = phase(12, 150, 16, 30, 2, 20, -1, 100, 150))) return rc;
    return 0;
}
