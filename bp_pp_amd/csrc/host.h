// Host side of libbppp_hip.so, shared by its translation units: bppp_ctx.hip (contexts, options, tables, setup), bppp_u64.hip
// (launch sequences of the u64 verifier / prover), bppp_generic.hip (generic wnla / reciprocal / circuit verifiers and provers) and
// bppp_group.hip (one batch over the GPUs of a node).  Nothing here is exported; the C ABI is include/bppp.h.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/bppp.h"
#include "kernels.h"

using namespace bppp;

extern thread_local std::string g_last_error;   // defined in bppp_ctx.hip

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                          \
            (void)hipGetLastError(); /* a failed allocation must not poison the next call */          \
            return e_ == hipErrorOutOfMemory ? BPPP_ERR_NOMEM : BPPP_ERR_HIP;                          \
        }                                                                                              \
    } while (0)

enum KernelId {
    K_PHASE1 = 0, K_C0_FIXED, K_C0_VAR, K_ROUND, K_FINAL_SCALARS, K_FINAL_CHECK, K_ACCEPT, K_TABLES, K_RLC_LHS, K_RLC_CHUNK, K_BKT_PREPARE, K_BKT_ACCUMULATE, K_BKT_SCALARS, K_BKT_CHECK,
    // u64 batch prover
    K_PROVE_STAGES, K_PROVE_MSM, K_PROVE_ROUND_SCALARS, K_PROVE_ROUND_FOLD, K_PROVE_ROUND_NEXT,
    // generic reciprocal / WNLA verifier
    K_RECIP_PHASE1, K_RECIP_C0_FIXED, K_RECIP_C0_VAR, K_RECIP_C0_FINISH, K_WNLA_BEGIN, K_WNLA_ROUND, K_WNLA_FINAL_SCALARS, K_WNLA_MSM, K_WNLA_ACCEPT,
    K_WNLA_RLC_LHS, K_WNLA_RLC_CHUNK, K_WNLA_RLC_CHECK, K_WNLA_TABLES,
    K_CIRCUIT_PHASE1, K_CIRCUIT_C0_FIXED, K_CIRCUIT_C0_VAR, K_CIRCUIT_C0_FINISH,      // generic ArithmeticCircuit verifier (then the K_WNLA_* stage)
    K_SHARED_INV,      // u64 verifier from 2^18 proofs: the launches that invert once for 8 / 16 proofs (and the join of C0's halves ahead of round 1)
    K_COUNT
};
static const char* const kKernelNames[K_COUNT] = {
    "k_verify_phase1", "k_verify_c0_fixed", "k_verify_c0_var", "k_verify_round", "k_verify_final_scalars", "k_verify_final_check",
    "k_verify_accept", "k_verify_tables", "k_rlc_lhs", "k_rlc_chunk", "k_bkt_prepare", "k_bkt_accumulate", "k_bkt_scalars", "k_bkt_check",
    "k_prove_stage_*", "k_prove_msm", "k_prove_round_scalars", "k_prove_round_fold", "k_prove_round_next",
    "k_recip_phase1", "k_recip_c0_fixed", "k_recip_c0_var", "k_recip_c0_finish", "k_wnla_begin", "k_wnla_round", "k_wnla_final_scalars",
    "k_wnla_msm", "k_wnla_accept", "k_wnla_rlc_lhs", "k_wnla_rlc_chunk", "k_wnla_rlc_check", "k_wnla_tables",
    "k_circuit_phase1", "k_circuit_c0_fixed", "k_circuit_c0_var", "k_circuit_c0_finish", "k_verify_shared_inv"};

static inline size_t align16(size_t x) { return (x + 15) / 16 * 16; }
// a transcript label as the ABI takes it: a null pointer only with length 0, and no longer than merlin can frame (its length prefix is a
// u32: `Transcript::append_message` asserts that; here the call returns BPPP_ERR_INVALID_ARG instead of hashing a truncated length)
static inline bool label_ok(const uint8_t* label, size_t label_len) { return (label || !label_len) && label_len <= 0xFFFFFFFFu; }

struct TimedLaunch { int id; hipEvent_t a, b; };

struct bppp_ctx {
    std::recursive_mutex mu;   // every exported call on a context holds it: overlapping calls from several host threads are serialized
    int device = 0;
    int fb_w = 16;
    int ng = 16, nh = 32, nbases = BPPP_NG;   // generator set: g, g_vec[ng], h_vec[nh]
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr;   // runs the fixed-base half of C0 concurrently with the variable-base half
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_tab = nullptr;
    hipStream_t copy_stream = nullptr;  // host-buffer entry points: uploads chunk k + 1 while chunk k is being verified (created on first use)
    hipEvent_t ev_copy = nullptr;
    size_t max_batch = (size_t)1 << 21;    // proofs verified per internal part of one call: bounds the workspace (~63 GB at 2^21)
    uint64_t fb_table_budget = 0;          // bppp_wnla_ctx_create_budget: what the automatic table layout may take (0 = no cap of the caller's)
    size_t host_chunk = (size_t)1 << 17;   // proofs per pipelined chunk (one full grid at 2 waves/SIMD); 0 = upload the whole batch first
    apt* d_gens = nullptr;       // 49
    apt_packed* d_table = nullptr;
    size_t table_bytes = 0;
    // second table region (fb_core.h: FbTable): generators 0 .. fb_hi_bases - 1 at fb_w_hi bits; d_table then holds the rest
    apt_packed* d_table_hi = nullptr;
    size_t table_hi_bytes = 0;
    int fb_w_hi = 0, fb_hi_bases = 0;
    // "ct_prover": the 4-bit table the provers' secret-scalar sums scan in full (fb_core.h: fb_lookup_add_ct), built when the option is set
    apt_packed* d_table_ct = nullptr;
    size_t table_ct_bytes = 0;
    bool ct_prover = false, borrows_table_ct = false;
    // per-proof workspace
    size_t cap = 0;
    u32* d_ws = nullptr;
    size_t ws_bytes = 0;
    size_t scap = 0;
    pt_slot* d_straus = nullptr;
    size_t straus_bytes = 0;
    // u64 verifier: affine window tables of the 13 proof points + the scratch of the kernel that builds them
    size_t vcap = 0;
    apt_packed* d_atab = nullptr;
    u32* d_tscr = nullptr;
    size_t vtab_bytes = 0;
    // random-linear-combination mode: per-proof weighted commitments, chunk scalars, chunk flags
    size_t rcap = 0;
    u32* d_rlc = nullptr;
    size_t rlc_bytes = 0;
    // bucket stage of the RLC mode (bucket_core.h): superchunk size (0 = stage off) and its workspace
    unsigned rlc_super_m = 4096;       // cap of the automatic choice, or the explicit size (bucket_superchunk_for)
    bool rlc_super_auto = true;
    int rlc_chunk_opt = 0;             // option "rlc_chunk": proofs per chunk after the bucket stage, 8 or 32; 0 = from the previous call's reject rate
    // what the previous RLC call on this context rejected (plan_core.h: plan_rlc): a device counter, copied to pinned host memory at the
    // end of the call; read by the next call if the copy has landed by then
    int* d_rlc_hist = nullptr;
    int* h_rlc_hist = nullptr;
    hipEvent_t ev_rlc_hist = nullptr;
    size_t rlc_hist_n = 0;             // proofs of the call the counter belongs to (0: no call in flight or recorded)
    double rlc_rate = -1.0;            // last known reject rate; < 0: none
    unsigned last_rlc_super_m = 0, last_rlc_chunk = 0;      // what the last RLC call used ("last_rlc_superchunk" / "last_rlc_chunk")
    size_t bcap = 0;
    uint8_t* d_bkt = nullptr;
    size_t bkt_bytes = 0;
    // prover workspace
    size_t pcap = 0;
    u32* d_pws = nullptr;
    size_t pws_bytes = 0;
    // staging for the host-pointer entry points
    uint8_t* d_stage = nullptr;
    size_t stage_bytes = 0;
    uint8_t* d_io = nullptr;     // inputs / outputs of bppp_u64_verify_batch (host buffers)
    size_t io_bytes = 0;
    uint8_t* d_blob = nullptr;   // inputs | outputs | workspace of ONE generic host-buffer call (bppp_generic.hip), grow-only
    size_t blob_bytes = 0;
    uint8_t* d_txio = nullptr;   // transcripts in / out of a generic prover call
    size_t txio_bytes = 0;
    int inject_alloc_fault = 0;  // testing aid ("inject_alloc_fault"): the k-th device allocation from now on fails
    uint8_t* d_gws = nullptr;    // workspace of bppp_reciprocal_verify_batch_device
    size_t gws_bytes = 0;
    uint8_t* d_gtab = nullptr;   // generic verifiers: affine window tables of the round points + build scratch (wnla_core.h, fast path)
    size_t gtab_bytes = 0;
    // expanded (64-byte) form of SEC1-compressed inputs
    uint8_t* d_expand = nullptr;
    size_t expand_bytes = 0;
    int* d_flags = nullptr;
    int n_simds = 1024;            // CUs x 4 (device property), decides between the small-batch and the 2-waves/SIMD lane kernels
    bool borrows_tables = false;   // d_gens / d_table belong to another context (bppp_ctx_create_shared)
    bool timing = false;
    bool generic_slow_rounds = false, no_lane_groups = false, no_small = false, no_split = false, generic_u64_shape = false;
    int recip_beside = -1;                              // diagnostic BPPP_RECIP_BESIDE = 0 | 1: the reciprocal verifier's one-lane kernels beside its fixed-base ones never / always; unset = by size
    long generic_fb_wide_max = -1;                      // diagnostic BPPP_GENERIC_FB_WIDE_MAX: the largest call (instances) whose fixed-base sums run a wavefront per instance; -1 = 8 per SIMD
    int recip_p1_group = 0;                             // diagnostic BPPP_RECIP_P1_GROUP = 1 | 2 | 4 | 8: lanes per instance in the reciprocal verifier's phase 1; 0 = by size
    int generic_lane_group = 0;                         // BPPP_GENERIC_LANE_GROUP = 2 | 4: that many lanes per instance in the generic verifiers' grouped kernels at any size (tests)
    int next_overlap = -1;   // diagnostic BPPP_NEXT_OVERLAP: the variable-base next commitment on the helper stream always (1) / never (0)
    int tail_beside = -1;     // diagnostic BPPP_TAIL_BESIDE: the last round's sum beside the final fixed-base sum always (1) / never (0)
    u32* d_zinv = nullptr;    // [10][n] of the current call inside d_ws (carve)
    int generic_stagger = 1;  // the parts' chains start out of step: part i + 1 behind part i's phase 1 (1), C0 stage (2), rounds (3); 0: together
    int generic_parts = 0;    // diagnostic BPPP_GENERIC_PARTS: parts of a generic reciprocal verify call (bppp_generic.hip: generic_parts_for); 0 = by size
    int twin_stream_kind = 1; // diagnostic BPPP_TWIN_STREAMS: the second chain's stream at 0 normal priority | 1 high priority (default) | 2 with a CU mask of all CUs
    int twin = -1, pace = -1; // diagnostics BPPP_TWIN / BPPP_PACE (plan_core.h: VerifyPlan::twin, ::pace); unset = by batch size
    hipStream_t twin_stream = nullptr, twin_aux = nullptr;      // the second half's stream pair of a twin verify call (bppp_u64.hip: ensure_twin_lanes)
    hipEvent_t ev_twin_fork = nullptr, ev_twin_join = nullptr, ev2_fork = nullptr, ev2_join = nullptr, ev2_tab = nullptr;
    int shared_inv = -1;      // diagnostic BPPP_SHARED_INV: proofs per shared field inversion in the one-lane verify kernels (0 = none, 2 4 8 16); unset = by batch size (plan_core.h)
    int tables_beside = -1;   // diagnostic BPPP_TABLES_BESIDE: the one-lane table kernel beside phase 1 always (1) / never (0); unset = where the lane kernels are a lone wavefront per SIMD
    long scal_parts_max = -1;   // diagnostic BPPP_SCAL_PARTS_MAX: largest prove call whose round scalars go out as four workgroups per 64 values
    long lane4_max = -1;   // diagnostic BPPP_LANE4_MAX: largest prove call on the four-lane stage kernels
    long lane_forms_max = -1, next_msm_max = -1;   // diagnostics BPPP_LANE_FORMS_MAX / BPPP_NEXT_MSM_MAX: largest prove call on the 16-lane stage / fold kernels, on the fixed-base next commitment (-1 = by n_simds)
    int fb_one_lane_mode = -1;   // diagnostic BPPP_FB_ONE_LANE: 1 = one lane per proof in the u64 verifier's fixed-base kernels at every size, 0 = never, unset = by size   // diagnostics, read from the environment once at context creation
    // single-proof front end (bppp_coalesce.hip): created at the first *_one call; options "coalesce_max" / "coalesce_us" / "coalesce_lanes"
    uint32_t last_verify_plan = 0, last_prove_plan = 0;      // plan_core.h: the kernels the context's last u64 verify / prove call (or part) ran
    struct bppp_fronts* fronts = nullptr;      // lives until the context itself is deleted (closure is sticky: never re-created once closed)
    std::atomic<bool> fronts_closed{false};    // set by bppp_ctx_destroy before it drains: *_one callers return BPPP_ERR_CLOSED instead of retrying
    std::atomic<int> one_callers{0};           // threads inside a *_one entry point (counted before they touch anything else of the context);
                                               // bppp_ctx_destroy frees nothing before the count is back to zero
    long coalesce_max = 1024, coalesce_us = 100;
    int coalesce_lanes = 2;
    std::vector<TimedLaunch> pending;
    std::vector<hipEvent_t> event_pool;
    double total_ms[K_COUNT] = {0};
    int64_t launches[K_COUNT] = {0};
};

struct CtxLock {
    bppp_ctx* c;
    explicit CtxLock(bppp_ctx* ctx) : c(ctx) { if (c) c->mu.lock(); }
    ~CtxLock() { if (c) c->mu.unlock(); }
    CtxLock(const CtxLock&) = delete;
    CtxLock& operator=(const CtxLock&) = delete;
};

// Every device allocation of a context's workspaces and staging goes through here: a failure is an honest BPPP_ERR_NOMEM (HIP_TRY maps
// hipErrorOutOfMemory), and the "inject_alloc_fault" option can make the k-th one fail so that the tests can walk every such path.
static inline hipError_t ctx_malloc(bppp_ctx* c, void** p, size_t bytes) {
    *p = nullptr;
    if (c->inject_alloc_fault > 0 && --c->inject_alloc_fault == 0) return hipErrorOutOfMemory;
    return hipMalloc(p, bytes);
}
// grow-only buffer: keeps what it has when it is large enough, otherwise frees and allocates `bytes` (contents are not preserved)
static inline int ensure_buffer(bppp_ctx* c, uint8_t*& d, size_t& have, size_t bytes) {
    if (bytes <= have && d) return BPPP_OK;
    if (d) { (void)hipFree(d); d = nullptr; have = 0; }
    HIP_TRY(ctx_malloc(c, (void**)&d, bytes ? bytes : 16));
    have = bytes ? bytes : 16;
    return BPPP_OK;
}
// staging of the host-buffer entry points (inputs and outputs of one call; every such call holds the context's lock and waits for its
// stream before it returns, so one buffer serves them all -- no allocator round trip and no implicit device sync per call)
static inline int ensure_io(bppp_ctx* c, size_t bytes) { return ensure_buffer(c, c->d_io, c->io_bytes, bytes); }
static inline int ensure_blob(bppp_ctx* c, size_t bytes) { return ensure_buffer(c, c->d_blob, c->blob_bytes, bytes); }

// after a failed call: nothing of it may still be running when the entry point returns (the staging is reused by the next call, and
// copies from / to the caller's memory may be queued) -- what the implicit synchronisation of a per-call hipFree used to provide
static inline void quiesce(bppp_ctx* c) {
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->twin_stream) (void)hipStreamSynchronize(c->twin_stream);
    if (c->twin_aux) (void)hipStreamSynchronize(c->twin_aux);
    (void)hipGetLastError();
}

static const size_t WS_WORDS_PER_PROOF = 52 + 80 + 176 + 200 + 208 + 24 + 30 + 30 + 392 + 10;

static inline int ensure_capacity(bppp_ctx* c, size_t n) {
    if (n <= c->cap) return BPPP_OK;
    if (c->d_ws) { (void)hipFree(c->d_ws); c->d_ws = nullptr; }
    c->cap = 0;
    c->ws_bytes = 0;
    size_t cap = (n + BPPP_BLOCK - 1) / BPPP_BLOCK * BPPP_BLOCK;
    const size_t bytes = cap * WS_WORDS_PER_PROOF * sizeof(u32);
    HIP_TRY(ctx_malloc(c, (void**)&c->d_ws, bytes));
    c->ws_bytes = bytes;
    c->cap = cap;
    return BPPP_OK;
}
// projective window tables of the generic WNLA / circuit / reciprocal paths and of the provers (5.6 KB per instance); the u64
// verifier keeps its own affine tables (ensure_vtab_capacity) and never touches these
static inline int ensure_straus_capacity(bppp_ctx* c, size_t n) {
    if (n <= c->scap) return BPPP_OK;
    if (c->d_straus) { (void)hipFree(c->d_straus); c->d_straus = nullptr; }
    c->scap = 0;
    c->straus_bytes = 0;
    size_t cap = (n + BPPP_BLOCK - 1) / BPPP_BLOCK * BPPP_BLOCK;
    const size_t bytes = cap * 5 * BPPP_STRAUS_ENTRIES * sizeof(pt_slot);
    HIP_TRY(ctx_malloc(c, (void**)&c->d_straus, bytes));
    c->straus_bytes = bytes;
    c->scap = cap;
    return BPPP_OK;
}
static inline int ensure_vtab_capacity(bppp_ctx* c, size_t n) {
    if (n <= c->vcap) return BPPP_OK;
    if (c->d_atab) { (void)hipFree(c->d_atab); c->d_atab = nullptr; }
    if (c->d_tscr) { (void)hipFree(c->d_tscr); c->d_tscr = nullptr; }
    c->vcap = 0;
    c->vtab_bytes = 0;
    size_t cap = (n + BPPP_BLOCK - 1) / BPPP_BLOCK * BPPP_BLOCK;
    const size_t atab_bytes = cap * BPPP_ATAB_PER_PROOF * sizeof(apt_packed);
    const size_t tscr_bytes = cap * (size_t)(BPPP_TSCR_FE * 10) * sizeof(u32);
    HIP_TRY(ctx_malloc(c, (void**)&c->d_atab, atab_bytes));
    HIP_TRY(ctx_malloc(c, (void**)&c->d_tscr, tscr_bytes));
    c->vtab_bytes = atab_bytes + tscr_bytes;
    c->vcap = cap;
    return BPPP_OK;
}
static inline int ensure_rlc_capacity(bppp_ctx* c, size_t n) {
    if (n <= c->rcap) return BPPP_OK;
    if (c->d_rlc) { (void)hipFree(c->d_rlc); c->d_rlc = nullptr; }
    c->rcap = 0;
    c->rlc_bytes = 0;
    size_t cap = (n + BPPP_BLOCK - 1) / BPPP_BLOCK * BPPP_BLOCK;
    const size_t rbytes = cap * (30 + (size_t)BPPP_NG * 8) * sizeof(u32) + cap + (cap / BPPP_RLC_CHUNK + 4) * sizeof(u32);   // lhs, sc | flags (cap bytes) | list, count
    HIP_TRY(ctx_malloc(c, (void**)&c->d_rlc, rbytes));
    c->rlc_bytes = rbytes;
    c->rcap = cap;
    return BPPP_OK;
}
static const size_t PWS_WORDS_PER_PROOF = 52 + (size_t)SV_COUNT * 8 + (size_t)BPPP_MSC_SETS * BPPP_NG * 8 + (size_t)PB_COUNT * 30;
static inline int ensure_prove_capacity(bppp_ctx* c, size_t n) {
    int rc = ensure_capacity(c, n);
    if (rc != BPPP_OK) return rc;
    rc = ensure_straus_capacity(c, n);
    if (rc != BPPP_OK) return rc;
    if (n <= c->pcap) return BPPP_OK;
    if (c->d_pws) { (void)hipFree(c->d_pws); c->d_pws = nullptr; }
    c->pcap = 0;
    c->pws_bytes = 0;
    size_t cap = (n + BPPP_BLOCK - 1) / BPPP_BLOCK * BPPP_BLOCK;
    const size_t pbytes = cap * PWS_WORDS_PER_PROOF * sizeof(u32);
    HIP_TRY(ctx_malloc(c, (void**)&c->d_pws, pbytes));
    c->pws_bytes = pbytes;
    c->pcap = cap;
    return BPPP_OK;
}
// the per-proof workspaces of the u64 verifier given back (a call that ran out of memory retries with smaller parts: bppp_u64.hip)
static inline void release_workspaces(bppp_ctx* c) {
    if (c->d_ws) { (void)hipFree(c->d_ws); c->d_ws = nullptr; }
    c->cap = 0; c->ws_bytes = 0; c->d_zinv = nullptr;
    if (c->d_atab) { (void)hipFree(c->d_atab); c->d_atab = nullptr; }
    if (c->d_tscr) { (void)hipFree(c->d_tscr); c->d_tscr = nullptr; }
    c->vcap = 0; c->vtab_bytes = 0;
    if (c->d_rlc) { (void)hipFree(c->d_rlc); c->d_rlc = nullptr; }
    c->rcap = 0; c->rlc_bytes = 0;
    if (c->d_straus) { (void)hipFree(c->d_straus); c->d_straus = nullptr; }
    c->scap = 0; c->straus_bytes = 0;
    if (c->d_bkt) { (void)hipFree(c->d_bkt); c->d_bkt = nullptr; }
    c->bcap = 0; c->bkt_bytes = 0;
    if (c->d_pws) { (void)hipFree(c->d_pws); c->d_pws = nullptr; }
    c->pcap = 0; c->pws_bytes = 0;
    (void)hipGetLastError();
}
static inline int ensure_stage(bppp_ctx* c, size_t bytes) {
    if (bytes <= c->stage_bytes) return BPPP_OK;
    if (c->d_stage) { (void)hipFree(c->d_stage); c->d_stage = nullptr; c->stage_bytes = 0; }
    HIP_TRY(ctx_malloc(c, (void**)&c->d_stage, bytes));
    c->stage_bytes = bytes;
    return BPPP_OK;
}
// workspace carve-up: the SoA stride is the batch size n of THIS call (so lanes stay coalesced for any n <= cap)
// (carve_from: the same carve-up at any base -- the halves of a twin call sit side by side in d_ws; returns where the half's zinv rows are)
static inline u32* carve_from(bppp_ctx* c, VerifyWs& ws, size_t n, u32* p) {
    ws.N = n;
    ws.tstate = p; p += 52 * n;
    ws.chal = p; p += 80 * n;
    ws.sc0 = p; p += 176 * n;
    ws.cvec = p; p += 200 * n;
    ws.pts = p; p += 208 * n;
    ws.lns = p; p += 24 * n;
    ws.acc = p; p += 30 * n;
    ws.pfix = p; p += 30 * n;
    ws.fsc = p; p += 392 * n;
    (void)c;
    return p;
}
static inline void carve(bppp_ctx* c, VerifyWs& ws, size_t n) {
    u32* p = c->d_ws;
    ws.N = n;
    ws.tstate = p; p += 52 * n;
    ws.chal = p; p += 80 * n;
    ws.sc0 = p; p += 176 * n;
    ws.cvec = p; p += 200 * n;
    ws.pts = p; p += 208 * n;
    ws.lns = p; p += 24 * n;
    ws.acc = p; p += 30 * n;
    ws.pfix = p; p += 30 * n;
    ws.fsc = p; p += 392 * n;
    ws.zinv = nullptr; c->d_zinv = p; p += 10 * n;     // (handed to the kernels as ws.zinv by the calls whose plan shares inversions)
    ws.straus = c->d_straus;
    ws.fb_table = c->d_table;
    ws.fb_w = c->fb_w;
    ws.fb_table_hi = c->d_table_hi; ws.fb_w_hi = c->fb_w_hi; ws.fb_hi_bases = c->fb_hi_bases;
}
// the context's fixed-base tables as the kernels take them (n = SoA stride of the scalars the sums will read)
static inline FbTable fb_table_of(const bppp_ctx* c, size_t n) {
    FbTable f = {c->d_table, c->fb_w, n, c->d_table_hi, c->fb_w_hi, c->fb_hi_bases};
    return f;
}

template <typename F>
static inline int timed(bppp_ctx* c, int id, hipStream_t st, F&& launch) {
    if (!c->timing) {
        launch();
        return BPPP_OK;
    }
    auto get_event = [&](hipEvent_t& ev) -> hipError_t {
        if (!c->event_pool.empty()) { ev = c->event_pool.back(); c->event_pool.pop_back(); return hipSuccess; }
        return hipEventCreate(&ev);
    };
    TimedLaunch tl;
    tl.id = id;
    HIP_TRY(get_event(tl.a));
    HIP_TRY(get_event(tl.b));
    HIP_TRY(hipEventRecord(tl.a, st));
    launch();
    HIP_TRY(hipEventRecord(tl.b, st));
    c->pending.push_back(tl);
    return BPPP_OK;
}
// workspace of the bucket stage: half-weights, packed commitments | per superchunk: lhs, combined scalars, flag
static inline size_t bkt_bytes_for(size_t cap, size_t nsuper, size_t nb) {
    return align16(cap * 16) + align16(cap * sizeof(c4_packed)) + align16(nsuper * 30 * 4) + align16(nsuper * nb * 32) + align16(nsuper + 16);
}
static inline int ensure_bucket_capacity(bppp_ctx* c, size_t n, size_t nb) {
    const size_t cap = (n + BPPP_BLOCK - 1) / BPPP_BLOCK * BPPP_BLOCK;
    const size_t need = bkt_bytes_for(cap, cap / 64 + 1, nb);      // enough for any superchunk size >= 64
    if (need <= c->bkt_bytes) return BPPP_OK;
    if (c->d_bkt) { (void)hipFree(c->d_bkt); c->d_bkt = nullptr; c->bkt_bytes = 0; }
    HIP_TRY(ctx_malloc(c, (void**)&c->d_bkt, need));
    c->bkt_bytes = need;
    return BPPP_OK;
}
// Superchunk size of a call.  An explicit "rlc_superchunk" option is taken as it is; the default adapts to the batch: one workgroup
// works on one superchunk for ~1-3 ms whatever their number, so a batch is cut into about as many superchunks as the chip has CUs
// (2^20 proofs: 4096, the measured optimum; 2^17, one GPU's share of the 8-GPU split: 512; never below 256: emptier buckets cost more
// than idle CUs save).
static inline unsigned bucket_superchunk_for(const bppp_ctx* c, size_t n) {
    if (!c->rlc_super_m || !c->rlc_super_auto) return c->rlc_super_m;
    size_t m = 256;
    while (m * 2 <= n / (size_t)(c->n_simds / 4 > 0 ? c->n_simds / 4 : 256) && m * 2 <= c->rlc_super_m) m *= 2;
    return (unsigned)m;
}
// The bucket stage in front of the chunk-of-8 kernels, shared by the u64 verifier (nb = 49 bases, scalars in fsc) and the generic
// verifiers (nb = 1 + |g_vec| + |h_vec|, scalars in msc): fills bw and launches prepare / accumulate / scalars / check on s.
template <typename Launch>
static inline int launch_bucket_stage(bppp_ctx* c, BucketWs& bw, size_t n, unsigned SM, const u64 seed[4], const int32_t* status, const u32* acc,
                                      const u32* scalars, int nb, uint8_t* accept, hipStream_t s, Launch&& timed_launch) {
    const size_t nsuper = (n + SM - 1) / SM;
    int rc = ensure_bucket_capacity(c, n, (size_t)nb);
    if (rc != BPPP_OK) return rc;
    std::memset(&bw, 0, sizeof bw);
    bw.N = n; bw.M = SM; bw.nb = nb;
    for (int i = 0; i < 4; i++) bw.seed[i] = seed[i];
    bw.status = status; bw.acc = acc; bw.fsc = scalars; bw.accept = accept;
    uint8_t* p = c->d_bkt;
    const size_t capn = (n + BPPP_BLOCK - 1) / BPPP_BLOCK * BPPP_BLOCK;
    bw.wab = (u64*)p; p += align16(capn * 16);
    bw.c4 = (c4_packed*)p; p += align16(capn * sizeof(c4_packed));
    bw.lhs = (u32*)p; p += align16(nsuper * 30 * 4);
    bw.asc = (u32*)p; p += align16(nsuper * (size_t)nb * 32);
    bw.sflag = p;
    bw.fb = fb_table_of(c, nsuper);
    const size_t lds_bytes = ((size_t)4 * (512 + SM) + 8 * 30) * sizeof(u32);
    (void)hipFuncSetAttribute((const void*)k_bkt_accumulate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    const unsigned blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    const dim3 sgrid((unsigned)nsuper, (unsigned)((nb + BPPP_BKT_SCALAR_GROUP - 1) / BPPP_BKT_SCALAR_GROUP));
    rc = timed_launch(K_BKT_PREPARE, [&]() { k_bkt_prepare<<<blocks, BPPP_BLOCK, 0, s>>>(bw); });
    if (rc == BPPP_OK) rc = timed_launch(K_BKT_ACCUMULATE, [&]() { k_bkt_accumulate<<<(unsigned)nsuper, 256, lds_bytes, s>>>(bw); });
    if (rc == BPPP_OK) rc = timed_launch(K_BKT_SCALARS, [&]() { k_bkt_scalars<<<sgrid, 256, 0, s>>>(bw); });
    if (rc == BPPP_OK) rc = timed_launch(K_BKT_CHECK, [&]() { k_bkt_check<<<(unsigned)nsuper, 64, 0, s>>>(bw); });
    return rc;
}
// the RLC mode's reject history: a device counter, its pinned host copy and the event that says the copy has landed
static inline int ensure_rlc_history(bppp_ctx* c) {
    if (c->d_rlc_hist) return BPPP_OK;
    HIP_TRY(hipMalloc(&c->d_rlc_hist, sizeof(int)));
    HIP_TRY(hipHostMalloc((void**)&c->h_rlc_hist, sizeof(int), hipHostMallocDefault));
    *c->h_rlc_hist = 0;
    HIP_TRY(hipEventCreateWithFlags(&c->ev_rlc_hist, hipEventDisableTiming));
    return BPPP_OK;
}
static inline int drain_timings(bppp_ctx* c) {
    if (c->pending.empty()) return BPPP_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->aux_stream));
    for (auto& tl : c->pending) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, tl.a, tl.b));
        c->total_ms[tl.id] += ms;
        c->launches[tl.id] += 1;
        c->event_pool.push_back(tl.a);
        c->event_pool.push_back(tl.b);
    }
    c->pending.clear();
    return BPPP_OK;
}

static inline int device_simds(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || prop.multiProcessorCount <= 0) return 1024;
    return prop.multiProcessorCount * 4;
}
static inline int check_device(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_last_error = std::string("no HIP device: ") + hipGetErrorString(e);
        return BPPP_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        g_last_error = "device index out of range";
        return BPPP_ERR_INVALID_ARG;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return BPPP_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_last_error = std::string("this library is built for gfx950 only, found ") + prop.gcnArchName;
        return BPPP_ERR_NO_DEVICE;
    }
    return BPPP_OK;
}

// bppp_u64.hip: the context's second stream pair (twin verify calls, parts of a generic call) and its events
int bppp_ensure_twin_lanes(bppp_ctx* c);
// bppp_ctx.hip: make sure the context has the 4-bit table of the "ct_prover" mode (built once, 3 MB for the u64 generators)
int ensure_ct_table(bppp_ctx* c);
// bppp_coalesce.hip: drain and drop the context's single-proof front ends (final: refuse later *_one calls with BPPP_ERR_CLOSED)
void bppp_fronts_teardown(bppp_ctx* c, bool final);
void bppp_fronts_delete(bppp_ctx* c);      // bppp_ctx_destroy only, after the final teardown: frees the (closed, empty) bppp_fronts object
// what a child context takes from its parent besides the immutable tables: read under the parent's lock by whoever creates the child
struct CtShare { apt_packed* d_table_ct; bool ct_prover; };
int ctx_create_shared_with(bppp_ctx** out, bppp_ctx* parent, const CtShare& ct);

// ---- launch sequences used across translation units (hidden symbols; C linkage only because their definitions sit inside the
//      extern "C" blocks of the entry points they serve)
// optional pre-loaded transcripts of a verify call (device pointers): see VerifyWs::states
struct VerifyTranscripts { const void* d_states; size_t n_states; void* d_states_out; };
extern "C" {
// bppp_u64.hip: the u64 verifier over device buffers, asynchronous on c->stream (the caller holds the context's lock)
int verify_device_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments, const void* d_proofs,
                       void* d_accept, void* d_status, void* d_trace, void* d_reject_count, const uint8_t* rlc_seed,
                       const VerifyTranscripts* tx);
// bppp_u64.hip: SEC1-compressed inputs expanded into the context's buffer, then verify_device_impl
int verify_sec1_device_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments33, const void* d_proofs525,
                            void* d_accept, void* d_status, void* d_trace, void* d_reject_count);
// bppp_u64.hip: the u64 prover over device buffers, asynchronous on c->stream (the caller holds the context's lock)
int prove_device_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_x, const void* d_s, const void* d_rnd,
                      void* d_proofs, void* d_commitments, void* d_status, const VerifyTranscripts* tx);
// bppp_generic.hip: the reciprocal verifier over device buffers (exact, or RLC when rlc_seed is given); d_reject_count (device
// int32, optional) receives the number of rejected instances
int recip_verify_device_entry(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                              const void* d_commitments, const void* d_proofs, size_t rounds, size_t nl, size_t nn, void* d_accept,
                              void* d_status, const uint8_t* rlc_seed, void* d_reject_count);
}
