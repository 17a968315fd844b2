/* Native caller threads for tools/concurrent_callers.py and bench.py: T pthreads, each making `calls` SINGLE-proof calls one after the
 * other through a function pointer with the signature of bppp_u64_verify_one (include/bppp.h) -- the reference's calling pattern
 * (u64_proof.rs:42; one proof per call, many threads) without an interpreter in the way.  Plain C, no dependency on the library: the
 * entry point and the context come in as pointers.  Records every call's latency. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int (*verify_one_fn)(void* ctx, const uint8_t* label, size_t label_len, const uint8_t* commitment, const uint8_t* proof, uint8_t* accept,
                             int32_t* status);
/* bppp_u64_prove_one (u64_proof.rs:57): the prover's single-value call */
typedef int (*prove_one_fn)(void* ctx, const uint8_t* label, size_t label_len, uint64_t x, const uint8_t* s32, const uint8_t* rnd, uint8_t* proof,
                            uint8_t* commitment, int32_t* status);

struct shared {
    verify_one_fn fn;
    prove_one_fn pfn;           /* prove mode: x / s / rnd are the pool, P / V the proofs and commitments a batched call made of it */
    const uint64_t* x;
    const uint8_t *sb, *rnd;
    void* const* ctxs;          /* n_ctx contexts; thread t uses ctxs[t % n_ctx] */
    int n_ctx;
    const uint8_t* label;
    size_t label_len;
    const uint8_t *V, *P, *expect;
    size_t n_pool;
    int threads, calls;
    double* lat_us;
    pthread_barrier_t start;
    long wrong, failed;
    pthread_mutex_t mu;
};
struct arg { struct shared* s; int t; };

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static void* worker(void* p) {
    struct arg* a = (struct arg*)p;
    struct shared* s = a->s;
    void* ctx = s->ctxs[a->t % s->n_ctx];
    long wrong = 0, failed = 0;
    pthread_barrier_wait(&s->start);
    for (int k = 0; k < s->calls; k++) {
        const size_t j = ((size_t)a->t * (size_t)s->calls + (size_t)k) % s->n_pool;
        uint8_t acc = 0xFF;
        int32_t st = 0;
        if (s->pfn) {
            uint8_t proof[928], com[64];
            const double t0 = now_us();
            const int rc = s->pfn(ctx, s->label, s->label_len, s->x[j], s->sb + 32 * j, s->rnd + 52 * 32 * j, proof, com, &st);
            s->lat_us[(size_t)a->t * (size_t)s->calls + (size_t)k] = now_us() - t0;
            if (rc != 0 || st != 0) failed++;
            else if (memcmp(proof, s->P + 928 * j, 928) != 0 || memcmp(com, s->V + 64 * j, 64) != 0) wrong++;
            continue;
        }
        const double t0 = now_us();
        const int rc = s->fn(ctx, s->label, s->label_len, s->V + 64 * j, s->P + 928 * j, &acc, &st);
        s->lat_us[(size_t)a->t * (size_t)s->calls + (size_t)k] = now_us() - t0;
        if (rc != 0) failed++;
        else if (acc != s->expect[j]) wrong++;
    }
    pthread_mutex_lock(&s->mu);
    s->wrong += wrong;
    s->failed += failed;
    pthread_mutex_unlock(&s->mu);
    return 0;
}
static int run(struct shared* sp, double* elapsed_s, long* wrong, long* failed);
/* returns 0, or -1 when threads could not be created; *elapsed_s = first start to last finish */
int cc_run(void* fn, void* const* ctxs, int n_ctx, const uint8_t* label, size_t label_len, const uint8_t* V, const uint8_t* P, const uint8_t* expect,
           size_t n_pool, int threads, int calls, double* lat_us, double* elapsed_s, long* wrong, long* failed) {
    struct shared s;
    memset(&s, 0, sizeof s);
    s.fn = (verify_one_fn)fn; s.ctxs = ctxs; s.n_ctx = n_ctx; s.label = label; s.label_len = label_len; s.V = V; s.P = P; s.expect = expect;
    s.n_pool = n_pool; s.threads = threads; s.calls = calls; s.lat_us = lat_us;
    return run(&s, elapsed_s, wrong, failed);
}
/* the same with bppp_u64_prove_one: every returned proof and commitment must equal the batched call's (P, V) byte for byte */
int cc_run_prove(void* fn, void* const* ctxs, int n_ctx, const uint8_t* label, size_t label_len, const uint64_t* x, const uint8_t* sb,
                 const uint8_t* rnd, const uint8_t* V, const uint8_t* P, size_t n_pool, int threads, int calls, double* lat_us, double* elapsed_s,
                 long* wrong, long* failed) {
    struct shared s;
    memset(&s, 0, sizeof s);
    s.pfn = (prove_one_fn)fn; s.ctxs = ctxs; s.n_ctx = n_ctx; s.label = label; s.label_len = label_len; s.x = x; s.sb = sb; s.rnd = rnd; s.V = V; s.P = P;
    s.n_pool = n_pool; s.threads = threads; s.calls = calls; s.lat_us = lat_us;
    return run(&s, elapsed_s, wrong, failed);
}
static int run(struct shared* sp, double* elapsed_s, long* wrong, long* failed) {
    struct shared s = *sp;
    const int threads = s.threads;
    pthread_barrier_init(&s.start, 0, (unsigned)threads + 1);
    pthread_mutex_init(&s.mu, 0);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    struct arg* args = (struct arg*)malloc(sizeof(struct arg) * (size_t)threads);
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, 256 * 1024);
    int made = 0;
    for (; made < threads; made++) {
        args[made].s = &s; args[made].t = made;
        if (pthread_create(&th[made], &at, worker, &args[made]) != 0) break;
    }
    if (made != threads) {      /* cannot release the barrier with fewer parties: cancel what exists */
        for (int i = 0; i < made; i++) { pthread_cancel(th[i]); pthread_join(th[i], 0); }
        free(th); free(args);
        return -1;
    }
    pthread_barrier_wait(&s.start);
    const double t0 = now_us();
    for (int i = 0; i < threads; i++) pthread_join(th[i], 0);
    *elapsed_s = (now_us() - t0) * 1e-6;
    *wrong = s.wrong;
    *failed = s.failed;
    free(th); free(args);
    pthread_attr_destroy(&at);
    pthread_barrier_destroy(&s.start);
    pthread_mutex_destroy(&s.mu);
    return 0;
}
