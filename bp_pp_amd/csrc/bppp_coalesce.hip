// libbppp_hip.so, host side: the reference's own calling pattern -- ONE proof per call from many threads (u64_proof.rs:42, :57;
// benches/range_proof.rs:47-50) -- served at batch speed.  bppp_u64_verify_one / bppp_u64_prove_one do not launch anything: they hand
// their request to the context's front end (coalesce_core.h), whose dispatcher threads run what has gathered as ONE batched call of the
// device verify / prove sequences (bppp_u64.hip) on contexts that share the caller's tables, and wake each caller with its own row.
#include <memory>

#include "coalesce_core.h"
#include "host.h"

using bppp_host::CoalesceShape;
using bppp_host::Coalescer;

static const size_t SB = BPPP_TRANSCRIPT_STATE_BYTES;
static const size_t RND_BYTES = 52 * 32;

// One front end = one kind of request (verify or prove) of one context: `lanes` child contexts over the parent's tables (own streams,
// own workspaces, own device staging), pinned host staging for the rows, and the coalescer that fills and flushes them.
struct RecipShape {          // the runtime shape of a generic reciprocal verify request (bppp_reciprocal_verify_batch's arguments)
    size_t nd = 0, np = 0, rounds = 0, nl = 0, nn = 0;
    size_t proof_bytes() const { return 64 * (5 + 2 * rounds) + 32 * (nl + nn); }
    bool operator==(const RecipShape& o) const { return nd == o.nd && np == o.np && rounds == o.rounds && nl == o.nl && nn == o.nn; }
};
struct bppp_front {
    bppp_ctx* parent;
    bool prove;
    bool recip = false;          // ReciprocalRangeProofProtocol::verify at runtime dimensions (rows: commitment, proof, transcript)
    RecipShape rs;
    size_t max;
    std::vector<bppp_ctx*> lanes;
    Coalescer<bppp_front> co;

    static CoalesceShape shape_of(bool prove) {
        CoalesceShape s;
        if (!prove) {               // in: commitment, proof, transcript | out: accept, status, transcript after verify
            s.n_in = 3; s.in_stride[0] = 64; s.in_stride[1] = BPPP_U64_PROOF_BYTES; s.in_stride[2] = SB;
            s.n_out = 3; s.out_stride[0] = 1; s.out_stride[1] = sizeof(int32_t); s.out_stride[2] = SB;
        } else {                    // in: x, s, the 52 draws, transcript | out: proof, commitment, status, transcript after prove
            s.n_in = 4; s.in_stride[0] = 8; s.in_stride[1] = 32; s.in_stride[2] = RND_BYTES; s.in_stride[3] = SB;
            s.n_out = 4; s.out_stride[0] = BPPP_U64_PROOF_BYTES; s.out_stride[1] = 64; s.out_stride[2] = sizeof(int32_t); s.out_stride[3] = SB;
        }
        return s;
    }
    static CoalesceShape shape_of_recip(const RecipShape& r) {
        CoalesceShape s;
        s.n_in = 3; s.in_stride[0] = 64; s.in_stride[1] = r.proof_bytes(); s.in_stride[2] = SB;
        s.n_out = 3; s.out_stride[0] = 1; s.out_stride[1] = sizeof(int32_t); s.out_stride[2] = SB;
        return s;
    }
    CtShare ct;                  // the parent's "ct_prover" state when this front end was asked for (read under the parent's lock: fronts_of)
    bppp_front(bppp_ctx* p, bool prove_, size_t max_, long wait_us, int nlanes, const CtShare& ct_)
        : parent(p), prove(prove_), max(max_), co(this, shape_of(prove_), max_, wait_us, nlanes, BPPP_ERR_CLOSED, BPPP_ERR_NOMEM), ct(ct_) {}
    bppp_front(bppp_ctx* p, const RecipShape& r, size_t max_, long wait_us, int nlanes, const CtShare& ct_)
        : parent(p), prove(false), recip(true), rs(r), max(max_), co(this, shape_of_recip(r), max_, wait_us, nlanes, BPPP_ERR_CLOSED, BPPP_ERR_NOMEM), ct(ct_) {}

    // ---- Backend of the coalescer
    void* alloc_staging(size_t bytes) {
        void* p = nullptr;
        if (hipSetDevice(parent->device) != hipSuccess || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return p;
    }
    void free_staging(void* p) { (void)hipHostFree(p); }
    bool start_lane(int lane) { return hipSetDevice(lanes[lane]->device) == hipSuccess; }
    std::string last_error() { return g_last_error; }                         // (thread-local: the dispatcher's own)
    void set_last_error(const std::string& e) { g_last_error = e; }           // ... handed to the caller's thread for bppp_last_error()
    void stop_lane(int) {}

    // device staging of one lane: the rows' arrays, each at an offset fixed by `max` (grow-only buffer of the lane's context)
    static int device_staging(bppp_ctx* c, size_t bytes) { return ensure_io(c, bytes); }
    int run(int lane, size_t n, uint8_t* const in[], uint8_t* const out[]) {
        bppp_ctx* c = lanes[lane];
        CtxLock lock_(c);
        if (recip)      // the generic verifier's host-buffer entry point over the pinned rows (it stages, runs and waits by itself)
            return bppp_reciprocal_verify_batch_transcript(c, n, in[2], n, rs.nd, rs.np, in[0], in[1], rs.rounds, rs.nl, rs.nn, out[0], (int32_t*)out[1], out[2]);
        const int rc = run_locked(c, n, in, out);
        // a failed call must not leave copies in flight over staging rows that the next batch is about to overwrite
        if (rc != BPPP_OK) quiesce(c);
        return rc;
    }
    int run_locked(bppp_ctx* c, size_t n, uint8_t* const in[], uint8_t* const out[]) {
        HIP_TRY(hipSetDevice(c->device));
        hipStream_t st = c->stream;
        if (!prove) {
            const size_t o_c = 0, o_p = align16(o_c + max * 64), o_ti = align16(o_p + max * (size_t)BPPP_U64_PROOF_BYTES), o_a = align16(o_ti + max * SB),
                         o_s = align16(o_a + max), o_to = align16(o_s + max * sizeof(int32_t)), total = align16(o_to + max * SB);
            int rc = device_staging(c, total);
            if (rc != BPPP_OK) return rc;
            uint8_t* d = c->d_io;
            HIP_TRY(hipMemcpyAsync(d + o_c, in[0], n * 64, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(d + o_p, in[1], n * (size_t)BPPP_U64_PROOF_BYTES, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(d + o_ti, in[2], n * SB, hipMemcpyHostToDevice, st));
            VerifyTranscripts tx = {d + o_ti, n, d + o_to};
            rc = verify_device_impl(c, nullptr, 0, n, d + o_c, d + o_p, d + o_a, d + o_s, nullptr, nullptr, nullptr, &tx);
            if (rc != BPPP_OK) return rc;
            HIP_TRY(hipMemcpyAsync(out[0], d + o_a, n, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out[1], d + o_s, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out[2], d + o_to, n * SB, hipMemcpyDeviceToHost, st));
        } else {
            const size_t o_x = 0, o_s = align16(o_x + max * 8), o_r = align16(o_s + max * 32), o_ti = align16(o_r + max * RND_BYTES),
                         o_p = align16(o_ti + max * SB), o_c = align16(o_p + max * (size_t)BPPP_U64_PROOF_BYTES), o_st = align16(o_c + max * 64),
                         o_to = align16(o_st + max * sizeof(int32_t)), total = align16(o_to + max * SB);
            int rc = device_staging(c, total);
            if (rc != BPPP_OK) return rc;
            uint8_t* d = c->d_io;
            HIP_TRY(hipMemcpyAsync(d + o_x, in[0], n * 8, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(d + o_s, in[1], n * 32, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(d + o_r, in[2], n * RND_BYTES, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(d + o_ti, in[3], n * SB, hipMemcpyHostToDevice, st));
            VerifyTranscripts tx = {d + o_ti, n, d + o_to};
            rc = prove_device_impl(c, nullptr, 0, n, d + o_x, d + o_s, d + o_r, d + o_p, d + o_c, d + o_st, &tx);
            if (rc != BPPP_OK) return rc;
            HIP_TRY(hipMemcpyAsync(out[0], d + o_p, n * (size_t)BPPP_U64_PROOF_BYTES, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out[1], d + o_c, n * 64, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out[2], d + o_st, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out[3], d + o_to, n * SB, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(hipStreamSynchronize(st));
        return BPPP_OK;
    }

    // lane contexts + their workspaces for a full batch up front (no allocation on the request path), then the coalescer's threads
    int start() {
        const int nl = co.lanes();
        for (int l = 0; l < nl; l++) {
            bppp_ctx* ch = nullptr;
            int rc = ctx_create_shared_with(&ch, parent, ct);      // (the 4-bit table is shared with the parent)
            if (rc != BPPP_OK) return rc;
            lanes.push_back(ch);
            CtxLock lock_(ch);
            ch->ct_prover = ct.ct_prover;
            if (recip) continue;                     // the generic verifier sizes its own (grow-only) buffers at the first batch
            if (prove) rc = ensure_prove_capacity(ch, max);
            else {
                rc = ensure_capacity(ch, max);
                // the small-call forms keep up to four sets of window tables per proof (bppp_u64.hip: verify_device_part)
                const size_t S = (size_t)ch->n_simds;
                size_t vt = max;
                if (4 * (max < S ? max : S) > vt) vt = 4 * (max < S ? max : S);
                if (2 * (max < 4 * S ? max : 4 * S) > vt) vt = 2 * (max < 4 * S ? max : 4 * S);
                if (rc == BPPP_OK) rc = ensure_vtab_capacity(ch, vt);
            }
            if (rc != BPPP_OK) return rc;
        }
        return co.start();
    }
    ~bppp_front() {
        co.shutdown();
        for (bppp_ctx* ch : lanes) bppp_ctx_destroy(ch);
    }
};

// The context's front ends, created at the first *_one call and torn down (drained) by bppp_ctx_destroy or by a change of the
// "coalesce_*" options.  Callers hold a shared_ptr while they are inside, so a teardown never frees a front end under a caller.
struct bppp_fronts {
    std::mutex mu;
    std::shared_ptr<bppp_front> f[2];     // 0 verify, 1 prove
    std::vector<std::shared_ptr<bppp_front>> generic;     // reciprocal verify, one per shape in use (at most BPPP_MAX_GENERIC_FRONTS)
    bool closed = false;
};
static const size_t BPPP_MAX_GENERIC_FRONTS = 4;

// Lock order: the context's lock (c->mu) is never taken while a front-end lock (fs->mu) is held -- bppp_ctx_set_option holds c->mu
// while it drains the front ends (which takes fs->mu), so the other order would deadlock.  What the front ends need from the context
// (the bppp_fronts object, the coalesce_* options) is therefore read under c->mu FIRST, then fs->mu is taken.
struct FrontOptions { long max, us; int lanes; CtShare ct; };
static bppp_fronts* fronts_of(bppp_ctx* c, FrontOptions& o) {
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (!c->fronts && !c->fronts_closed.load()) c->fronts = new (std::nothrow) bppp_fronts();
    o.max = c->coalesce_max; o.us = c->coalesce_us; o.lanes = c->coalesce_lanes;
    o.ct.d_table_ct = c->d_table_ct; o.ct.ct_prover = c->ct_prover;
    return c->fronts;
}
// A thread inside a *_one entry point.  The count is the FIRST thing such a call touches and the last thing it leaves behind: it is
// what lets bppp_ctx_destroy run while callers are arriving, waiting for a batch, or on their way out (include/bppp.h), without ever
// freeing the context, its bppp_fronts or a front end under one of them.
struct OneCaller {
    bppp_ctx* c;
    explicit OneCaller(bppp_ctx* c_) : c(c_) { c->one_callers.fetch_add(1, std::memory_order_acq_rel); }
    ~OneCaller() { c->one_callers.fetch_sub(1, std::memory_order_acq_rel); }
    bool closed() const { return c->fronts_closed.load(std::memory_order_acquire); }
};
static int get_front(bppp_ctx* c, int which, std::shared_ptr<bppp_front>& out) {
    FrontOptions o;
    bppp_fronts* fs = fronts_of(c, o);
    if (!fs) return c->fronts_closed.load() ? BPPP_ERR_CLOSED : BPPP_ERR_NOMEM;
    std::lock_guard<std::mutex> lk(fs->mu);
    if (fs->closed) return BPPP_ERR_CLOSED;
    if (!fs->f[which]) {
        std::shared_ptr<bppp_front> f;
        int rc;
        try {
            f = std::make_shared<bppp_front>(c, which == 1, (size_t)o.max, o.us, o.lanes, o.ct);
            rc = f->start();
        } catch (...) { rc = BPPP_ERR_NOMEM; }      // (nothing may throw across the C ABI)
        if (rc != BPPP_OK) return rc;
        fs->f[which] = f;
    }
    out = fs->f[which];
    return BPPP_OK;
}
static int get_recip_front(bppp_ctx* c, const RecipShape& r, std::shared_ptr<bppp_front>& out) {
    FrontOptions o;
    bppp_fronts* fs = fronts_of(c, o);
    if (!fs) return c->fronts_closed.load() ? BPPP_ERR_CLOSED : BPPP_ERR_NOMEM;
    std::lock_guard<std::mutex> lk(fs->mu);
    if (fs->closed) return BPPP_ERR_CLOSED;
    for (auto& f : fs->generic)
        if (f->rs == r) { out = f; return BPPP_OK; }
    if (fs->generic.size() >= BPPP_MAX_GENERIC_FRONTS) {
        g_last_error = "too many different reciprocal shapes in single-proof use on one context";
        return BPPP_ERR_INVALID_ARG;
    }
    std::shared_ptr<bppp_front> f;
    int rc;
    try {
        f = std::make_shared<bppp_front>(c, r, (size_t)o.max, o.us, o.lanes, o.ct);
        rc = f->start();
        if (rc == BPPP_OK) fs->generic.push_back(f);
    } catch (...) { rc = BPPP_ERR_NOMEM; }
    if (rc != BPPP_OK) return rc;
    out = f;
    return BPPP_OK;
}
// drain and drop the front ends (hidden; bppp_ctx.hip calls it from bppp_ctx_destroy with final = true and from bppp_ctx_set_option)
void bppp_fronts_teardown(bppp_ctx* c, bool final) {
    bppp_fronts* fs;
    {
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        if (final) c->fronts_closed.store(true, std::memory_order_release);     // sticky: no front end is ever started on this context again
        fs = c->fronts;
    }
    if (!fs) return;
    std::shared_ptr<bppp_front> old[2 + BPPP_MAX_GENERIC_FRONTS];       // (no allocation on this path)
    size_t n_old = 0;
    {
        std::lock_guard<std::mutex> lk(fs->mu);
        if (final) fs->closed = true;
        for (auto& f : fs->f)
            if (f) old[n_old++].swap(f);
        for (auto& f : fs->generic) old[n_old++].swap(f);
        fs->generic.clear();
    }
    for (size_t i = 0; i < n_old; i++) old[i]->co.shutdown();      // drains; returns when no caller is inside
    for (size_t i = 0; i < n_old; i++) old[i].reset();              // the object itself goes with its last shared_ptr
}
// The last step of bppp_ctx_destroy.  A caller that got BPPP_ERR_CLOSED from its batch (or arrived during the teardown) is still inside
// its entry point for a few instructions -- it reads fronts_closed, may lock fs->mu once more, drops its shared_ptr: wait for the last
// of them, then free the (closed, empty) bppp_fronts.  Round 4 freed it right after the drain and re-created it for such a caller,
// which then started a new front end on a context that was being destroyed.
void bppp_fronts_delete(bppp_ctx* c) {
    while (c->one_callers.load(std::memory_order_acquire) != 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
    delete c->fronts;
    c->fronts = nullptr;
}

static bool state_ok(const uint8_t* st) { return st[200] < BPPP_STROBE_R && st[201] <= BPPP_STROBE_R; }

static int submit_retry(bppp_ctx* c, int which, const void* const in[], void* const out[]) {
    // A front end torn down by an option change while this caller was on its way in answers CLOSED: take the new one, as often as it
    // takes (a caller never sees an option change).  Only bppp_ctx_destroy ends the loop: get_front then answers CLOSED itself.
    OneCaller inside(c);
    for (;;) {
        if (inside.closed()) return BPPP_ERR_CLOSED;
        std::shared_ptr<bppp_front> f;
        int rc = get_front(c, which, f);
        if (rc != BPPP_OK) return rc;
        rc = f->co.submit(in, out);
        if (rc != BPPP_ERR_CLOSED) return rc;
    }
}

extern "C" {

int bppp_u64_verify_one_transcript(bppp_ctx* c, uint8_t state[203], const uint8_t commitment[64], const uint8_t proof[928], uint8_t* accept,
                                   int32_t* status) {
    if (!c || !state || !commitment || !proof || !accept) return BPPP_ERR_INVALID_ARG;
    OneCaller entered_(c);      // before anything else of the context is read (bppp_ctx_destroy waits for the count)
    if (c->ng != 16 || c->nh != 32 || !state_ok(state)) return BPPP_ERR_INVALID_ARG;
    const void* in[4] = {commitment, proof, state, nullptr};
    void* out[4] = {accept, status, state, nullptr};
    return submit_retry(c, 0, in, out);
}
int bppp_u64_verify_one(bppp_ctx* c, const uint8_t* label, size_t label_len, const uint8_t commitment[64], const uint8_t proof[928],
                        uint8_t* accept, int32_t* status) {
    if (!c || !label_ok(label, label_len) || !commitment || !proof || !accept) return BPPP_ERR_INVALID_ARG;
    OneCaller entered_(c);      // before anything else of the context is read (bppp_ctx_destroy waits for the count)
    if (c->ng != 16 || c->nh != 32) return BPPP_ERR_INVALID_ARG;
    uint8_t st[SB];
    int rc = bppp_transcript_new(label, label_len, st);      // Transcript::new(label) on the host: one Keccak permutation
    if (rc != BPPP_OK) return rc;
    const void* in[4] = {commitment, proof, st, nullptr};
    void* out[4] = {accept, status, nullptr, nullptr};
    return submit_retry(c, 0, in, out);
}
int bppp_u64_prove_one_transcript(bppp_ctx* c, uint8_t state[203], uint64_t x, const uint8_t s[32], const uint8_t* rnd, uint8_t proof[928],
                                  uint8_t commitment[64], int32_t* status) {
    if (!c || !state || !s || !rnd || !proof || !commitment) return BPPP_ERR_INVALID_ARG;
    OneCaller entered_(c);      // before anything else of the context is read (bppp_ctx_destroy waits for the count)
    if (c->ng != 16 || c->nh != 32 || !state_ok(state)) return BPPP_ERR_INVALID_ARG;
    const void* in[4] = {&x, s, rnd, state};
    void* out[4] = {proof, commitment, status, state};
    return submit_retry(c, 1, in, out);
}
int bppp_u64_prove_one(bppp_ctx* c, const uint8_t* label, size_t label_len, uint64_t x, const uint8_t s[32], const uint8_t* rnd,
                       uint8_t proof[928], uint8_t commitment[64], int32_t* status) {
    if (!c || !label_ok(label, label_len) || !s || !rnd || !proof || !commitment) return BPPP_ERR_INVALID_ARG;
    OneCaller entered_(c);      // before anything else of the context is read (bppp_ctx_destroy waits for the count)
    if (c->ng != 16 || c->nh != 32) return BPPP_ERR_INVALID_ARG;
    uint8_t st[SB];
    int rc = bppp_transcript_new(label, label_len, st);
    if (rc != BPPP_OK) return rc;
    const void* in[4] = {&x, s, rnd, st};
    void* out[4] = {proof, commitment, status, nullptr};
    return submit_retry(c, 1, in, out);
}
// ReciprocalRangeProofProtocol::verify (reciprocal.rs:98-107) for ONE instance at runtime dimensions, from any number of threads
static int recip_one(bppp_ctx* c, uint8_t* state_io, const uint8_t* state_in, size_t dim_nd, size_t dim_np, const uint8_t* commitment,
                     const uint8_t* proof, size_t rounds, size_t nl, size_t nn, uint8_t* accept, int32_t* status) {
    if (!c || !commitment || !proof || !accept) return BPPP_ERR_INVALID_ARG;
    OneCaller inside(c);        // before anything else of the context is read (bppp_ctx_destroy waits for the count)
    // the shape must be one the context's generators can serve (bppp_generic.hip: recip_verify_check_args) before a front end is made for it
    if (dim_nd == 0 || dim_np == 0 || dim_nd > (size_t)c->ng || dim_nd + 10 > (size_t)c->nh || dim_np > dim_nd + 1 || rounds > 12 || nl > 4096 || nn > 4096)
        return BPPP_ERR_INVALID_ARG;
    RecipShape r;
    r.nd = dim_nd; r.np = dim_np; r.rounds = rounds; r.nl = nl; r.nn = nn;
    const void* in[4] = {commitment, proof, state_in, nullptr};
    void* out[4] = {accept, status, state_io, nullptr};
    for (;;) {
        if (inside.closed()) return BPPP_ERR_CLOSED;
        std::shared_ptr<bppp_front> f;
        int rc = get_recip_front(c, r, f);
        if (rc != BPPP_OK) return rc;
        rc = f->co.submit(in, out);
        if (rc != BPPP_ERR_CLOSED) return rc;
    }
}
int bppp_reciprocal_verify_one(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t dim_nd, size_t dim_np, const uint8_t commitment[64],
                               const uint8_t* proof, size_t rounds, size_t nl, size_t nn, uint8_t* accept, int32_t* status) {
    if (!label_ok(label, label_len)) return BPPP_ERR_INVALID_ARG;
    uint8_t st[SB];
    int rc = bppp_transcript_new(label, label_len, st);
    if (rc != BPPP_OK) return rc;
    return recip_one(c, nullptr, st, dim_nd, dim_np, commitment, proof, rounds, nl, nn, accept, status);
}
int bppp_reciprocal_verify_one_transcript(bppp_ctx* c, uint8_t state[203], size_t dim_nd, size_t dim_np, const uint8_t commitment[64],
                                          const uint8_t* proof, size_t rounds, size_t nl, size_t nn, uint8_t* accept, int32_t* status) {
    if (!state || !state_ok(state)) return BPPP_ERR_INVALID_ARG;
    return recip_one(c, state, state, dim_nd, dim_np, commitment, proof, rounds, nl, nn, accept, status);
}
int bppp_ctx_get_coalesce_stats(bppp_ctx* c, int which, uint64_t out[8]) {
    if (!c || !out || which < 0 || which > 1) return BPPP_ERR_INVALID_ARG;
    for (int i = 0; i < 8; i++) out[i] = 0;
    bppp_fronts* fs;
    {
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        fs = c->fronts;
    }
    if (!fs) return BPPP_OK;
    std::shared_ptr<bppp_front> f;
    {
        std::lock_guard<std::mutex> lk(fs->mu);
        f = fs->f[which];
    }
    if (!f) return BPPP_OK;
    const bppp_host::CoalesceStats s = f->co.stats();
    out[0] = s.requests; out[1] = s.batches; out[2] = s.largest_batch; out[3] = s.sealed_full; out[4] = s.sealed_deadline;
    out[5] = s.run_us; out[6] = s.fill_wait_us;
    return BPPP_OK;
}

}  // extern "C"
