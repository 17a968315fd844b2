// libbppp_hip.so: HIP kernels (gfx950) + the C ABI of include/bppp.h.
//
// Host side of the batch u64 range-proof verifier: context (generators -> fixed-base tables in HBM), workspace,
// stream, and the launch sequence of the exact per-proof pipeline.  The per-lane work is in verify_core.h.
// One lane per proof; 64-thread workgroups (one wavefront) so that a 2^16-proof batch yields 1024 workgroups
// (4 per CU) and no lane ever waits on a workgroup barrier -- there is no inter-lane communication in this pipeline.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bppp.h"
#if defined(BPPP_PHASE_TIMING)
namespace bppp { __device__ unsigned long long g_bppp_stamps[1024 * 32]; }
#endif
#include "kernels.h"

using namespace bppp;

// ---------------------------------------------------------------- host side
static thread_local std::string g_last_error;

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
    K_PROVE_STAGES, K_PROVE_MSM, K_PROVE_ROUND_SCALARS, K_PROVE_ROUND_FOLD,
    // generic reciprocal / WNLA verifier
    K_RECIP_PHASE1, K_RECIP_C0_FIXED, K_RECIP_C0_VAR, K_RECIP_C0_FINISH, K_WNLA_BEGIN, K_WNLA_ROUND, K_WNLA_FINAL_SCALARS, K_WNLA_MSM, K_WNLA_ACCEPT,
    K_WNLA_RLC_LHS, K_WNLA_RLC_CHUNK, K_WNLA_RLC_CHECK, K_WNLA_TABLES,
    K_COUNT
};
static const char* const kKernelNames[K_COUNT] = {
    "k_verify_phase1", "k_verify_c0_fixed", "k_verify_c0_var", "k_verify_round", "k_verify_final_scalars", "k_verify_final_check",
    "k_verify_accept", "k_verify_tables", "k_rlc_lhs", "k_rlc_chunk", "k_bkt_prepare", "k_bkt_accumulate", "k_bkt_scalars", "k_bkt_check",
    "k_prove_stage_*", "k_prove_msm", "k_prove_round_scalars", "k_prove_round_fold",
    "k_recip_phase1", "k_recip_c0_fixed", "k_recip_c0_var", "k_recip_c0_finish", "k_wnla_begin", "k_wnla_round", "k_wnla_final_scalars",
    "k_wnla_msm", "k_wnla_accept", "k_wnla_rlc_lhs", "k_wnla_rlc_chunk", "k_wnla_rlc_check", "k_wnla_tables"};

static size_t align16(size_t x) { return (x + 15) / 16 * 16; }

struct TimedLaunch { int id; hipEvent_t a, b; };

struct bppp_ctx {
    std::recursive_mutex mu;   // every exported call on a context holds it: overlapping calls from several host threads are serialized
    int device = 0;
    int fb_w = 16;
    int ng = 16, nh = 32, nbases = BPPP_NG;   // generator set: g, g_vec[ng], h_vec[nh]
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr;   // runs the fixed-base half of C0 concurrently with the variable-base half
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t copy_stream = nullptr;  // host-buffer entry points: uploads chunk k + 1 while chunk k is being verified (created on first use)
    hipEvent_t ev_copy = nullptr;
    size_t max_batch = (size_t)1 << 21;    // proofs verified per internal part of one call: bounds the workspace (~63 GB at 2^21)
    size_t host_chunk = (size_t)1 << 17;   // proofs per pipelined chunk (one full grid at 2 waves/SIMD); 0 = upload the whole batch first
    apt* d_gens = nullptr;       // 49
    apt_packed* d_table = nullptr;
    size_t table_bytes = 0;
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
    unsigned rlc_super_m = 4096;
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
    bool serial_c0 = false, rlc_debug = false, generic_slow_rounds = false, no_lane_groups = false, force_pairs = false, no_small = false;
    int fb_one_lane_mode = -1;   // diagnostic BPPP_FB_ONE_LANE: 1 = one lane per proof in the u64 verifier's fixed-base kernels at every size, 0 = never, unset = by size   // diagnostics, read from the environment once at context creation
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

static const size_t WS_WORDS_PER_PROOF = 52 + 80 + 176 + 200 + 208 + 24 + 30 + 30 + 392;

static int ensure_capacity(bppp_ctx* c, size_t n) {
    if (n <= c->cap) return BPPP_OK;
    if (c->d_ws) { (void)hipFree(c->d_ws); c->d_ws = nullptr; }
    c->cap = 0;
    c->ws_bytes = 0;
    size_t cap = (n + BPPP_BLOCK - 1) / BPPP_BLOCK * BPPP_BLOCK;
    const size_t bytes = cap * WS_WORDS_PER_PROOF * sizeof(u32);
    HIP_TRY(hipMalloc(&c->d_ws, bytes));
    c->ws_bytes = bytes;
    c->cap = cap;
    return BPPP_OK;
}
// projective window tables of the generic WNLA / circuit / reciprocal paths and of the provers (5.6 KB per instance); the u64
// verifier keeps its own affine tables (ensure_vtab_capacity) and never touches these
static int ensure_straus_capacity(bppp_ctx* c, size_t n) {
    if (n <= c->scap) return BPPP_OK;
    if (c->d_straus) { (void)hipFree(c->d_straus); c->d_straus = nullptr; }
    c->scap = 0;
    c->straus_bytes = 0;
    size_t cap = (n + BPPP_BLOCK - 1) / BPPP_BLOCK * BPPP_BLOCK;
    const size_t bytes = cap * 5 * BPPP_STRAUS_ENTRIES * sizeof(pt_slot);
    HIP_TRY(hipMalloc(&c->d_straus, bytes));
    c->straus_bytes = bytes;
    c->scap = cap;
    return BPPP_OK;
}
static int ensure_vtab_capacity(bppp_ctx* c, size_t n) {
    if (n <= c->vcap) return BPPP_OK;
    if (c->d_atab) { (void)hipFree(c->d_atab); c->d_atab = nullptr; }
    if (c->d_tscr) { (void)hipFree(c->d_tscr); c->d_tscr = nullptr; }
    c->vcap = 0;
    c->vtab_bytes = 0;
    size_t cap = (n + BPPP_BLOCK - 1) / BPPP_BLOCK * BPPP_BLOCK;
    const size_t atab_bytes = cap * BPPP_ATAB_PER_PROOF * sizeof(apt_packed);
    const size_t tscr_bytes = cap * (size_t)(BPPP_TSCR_FE * 10) * sizeof(u32);
    HIP_TRY(hipMalloc(&c->d_atab, atab_bytes));
    HIP_TRY(hipMalloc(&c->d_tscr, tscr_bytes));
    c->vtab_bytes = atab_bytes + tscr_bytes;
    c->vcap = cap;
    return BPPP_OK;
}
static int ensure_rlc_capacity(bppp_ctx* c, size_t n) {
    if (n <= c->rcap) return BPPP_OK;
    if (c->d_rlc) { (void)hipFree(c->d_rlc); c->d_rlc = nullptr; }
    c->rcap = 0;
    c->rlc_bytes = 0;
    size_t cap = (n + BPPP_BLOCK - 1) / BPPP_BLOCK * BPPP_BLOCK;
    const size_t rbytes = cap * (30 + (size_t)BPPP_NG * 8) * sizeof(u32) + cap + (cap / BPPP_RLC_CHUNK + 4) * sizeof(u32);   // lhs, sc | flags (cap bytes) | list, count
    HIP_TRY(hipMalloc(&c->d_rlc, rbytes));
    c->rlc_bytes = rbytes;
    c->rcap = cap;
    return BPPP_OK;
}
// workspace of the bucket stage: half-weights, packed commitments | per superchunk: lhs, combined scalars, flag
static size_t bkt_bytes_for(size_t cap, size_t nsuper) {
    return align16(cap * 16) + align16(cap * sizeof(c4_packed)) + align16(nsuper * 30 * 4) + align16(nsuper * (size_t)BPPP_NG * 32) + align16(nsuper + 16);
}
static int ensure_bucket_capacity(bppp_ctx* c, size_t n) {
    const size_t cap = (n + BPPP_BLOCK - 1) / BPPP_BLOCK * BPPP_BLOCK;
    const size_t need = bkt_bytes_for(cap, cap / 64 + 1);      // enough for any superchunk size >= 64
    if (need <= c->bkt_bytes) return BPPP_OK;
    if (c->d_bkt) { (void)hipFree(c->d_bkt); c->d_bkt = nullptr; c->bkt_bytes = 0; }
    HIP_TRY(hipMalloc(&c->d_bkt, need));
    c->bkt_bytes = need;
    return BPPP_OK;
}
static const size_t PWS_WORDS_PER_PROOF = 52 + (size_t)SV_COUNT * 8 + (size_t)BPPP_MSC_SETS * BPPP_NG * 8 + (size_t)PB_COUNT * 30;
static int ensure_prove_capacity(bppp_ctx* c, size_t n) {
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
    HIP_TRY(hipMalloc(&c->d_pws, pbytes));
    c->pws_bytes = pbytes;
    c->pcap = cap;
    return BPPP_OK;
}
static int ensure_stage(bppp_ctx* c, size_t bytes) {
    if (bytes <= c->stage_bytes) return BPPP_OK;
    if (c->d_stage) { (void)hipFree(c->d_stage); c->d_stage = nullptr; c->stage_bytes = 0; }
    HIP_TRY(hipMalloc(&c->d_stage, bytes));
    c->stage_bytes = bytes;
    return BPPP_OK;
}
// workspace carve-up: the SoA stride is the batch size n of THIS call (so lanes stay coalesced for any n <= cap)
static void carve(bppp_ctx* c, VerifyWs& ws, size_t n) {
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
    ws.straus = c->d_straus;
    ws.fb_table = c->d_table;
    ws.fb_w = c->fb_w;
}

template <typename F>
static int timed(bppp_ctx* c, int id, hipStream_t st, F&& launch) {
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
static int drain_timings(bppp_ctx* c) {
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

static int device_simds(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || prop.multiProcessorCount <= 0) return 1024;
    return prop.multiProcessorCount * 4;
}
static int check_device(int device) {
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

extern "C" {

const char* bppp_strerror(int code) {
    switch (code) {
        case BPPP_OK: return "ok";
        case BPPP_ERR_NO_DEVICE: return "no usable gfx950 HIP device (this library has no CPU fallback)";
        case BPPP_ERR_INVALID_ARG: return "invalid argument";
        case BPPP_ERR_HIP: return "HIP runtime error";
        case BPPP_ERR_ENCODING: return "generator is not a valid secp256k1 point";
        case BPPP_ERR_NOMEM: return "out of memory";
        case BPPP_ERR_RCCL: return "RCCL unavailable or an RCCL call failed";
        default: return "unknown error";
    }
}
const char* bppp_last_error(void) { return g_last_error.c_str(); }

int bppp_ctx_create(bppp_ctx** out, const uint8_t g[64], const uint8_t* g_vec, const uint8_t* h_vec, int device, int fb_window_bits) {
    return bppp_wnla_ctx_create(out, g, g_vec, 16, h_vec, 32, device, fb_window_bits);
}

int bppp_wnla_ctx_create(bppp_ctx** out, const uint8_t g[64], const uint8_t* g_vec, size_t ng, const uint8_t* h_vec, size_t nh, int device,
                         int fb_window_bits) {
    if (!out || !g || (!g_vec && ng) || (!h_vec && nh) || ng > 4096 || nh > 4096) return BPPP_ERR_INVALID_ARG;
    *out = nullptr;
    const int NB = 1 + (int)ng + (int)nh;
    int W = fb_window_bits ? fb_window_bits : 20;
    if (W != 4 && W != 8 && W != 16 && W != 10 && W != 20 && W != 22) return BPPP_ERR_INVALID_ARG;
    int rc = check_device(device);
    if (rc != BPPP_OK) return rc;
    HIP_TRY(hipSetDevice(device));
    bppp_ctx* c = new (std::nothrow) bppp_ctx();
    if (!c) return BPPP_ERR_NOMEM;
    c->device = device;
    c->fb_w = W;
    c->n_simds = device_simds(device);
    c->serial_c0 = std::getenv("BPPP_SERIAL_C0") != nullptr;
    c->force_pairs = std::getenv("BPPP_FORCE_LANE_PAIRS") != nullptr;   // diagnostic: rounds on two lanes per proof at every batch size
    c->no_lane_groups = std::getenv("BPPP_NO_LANE_GROUPS") != nullptr;   // diagnostic: one lane per proof at every batch size
    c->no_small = std::getenv("BPPP_NO_SMALL_KERNELS") != nullptr;       // diagnostic: the 256-VGPR builds (two wavefronts per SIMD) at every batch size
    if (const char* e = std::getenv("BPPP_FB_ONE_LANE")) c->fb_one_lane_mode = e[0] == '0' ? 0 : 1;
    c->generic_slow_rounds = std::getenv("BPPP_GENERIC_SLOW_ROUNDS") != nullptr;   // diagnostic: projective tables + complete additions
    c->rlc_debug = std::getenv("BPPP_RLC_DEBUG") != nullptr;
    c->ng = (int)ng; c->nh = (int)nh; c->nbases = NB;
    auto fail = [&](int code) { bppp_ctx_destroy(c); return code; };
#define HIP_TRY_C(expr)                                                             \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_);       \
            (void)hipGetLastError();                                                \
            return fail(e_ == hipErrorOutOfMemory ? BPPP_ERR_NOMEM : BPPP_ERR_HIP); \
        }                                                                           \
    } while (0)
    HIP_TRY_C(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIP_TRY_C(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    HIP_TRY_C(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY_C(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    HIP_TRY_C(hipMalloc(&c->d_gens, NB * sizeof(apt)));
    HIP_TRY_C(hipMalloc(&c->d_flags, sizeof(int)));
    HIP_TRY_C(hipMemsetAsync(c->d_flags, 0, sizeof(int), c->stream));
    // upload + decode generators
    std::vector<uint8_t> hg((size_t)NB * 64);
    std::memcpy(hg.data(), g, 64);
    if (ng) std::memcpy(hg.data() + 64, g_vec, ng * 64);
    if (nh) std::memcpy(hg.data() + (1 + ng) * 64, h_vec, nh * 64);
    uint8_t* d_raw = nullptr;
    HIP_TRY_C(hipMalloc(&d_raw, hg.size()));
    HIP_TRY_C(hipMemcpyAsync(d_raw, hg.data(), hg.size(), hipMemcpyHostToDevice, c->stream));
    k_decode_generators<<<(NB + 63) / 64, 64, 0, c->stream>>>(d_raw, c->d_gens, NB, c->d_flags);
    int flags = 0;
    HIP_TRY_C(hipMemcpyAsync(&flags, c->d_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY_C(hipStreamSynchronize(c->stream));
    (void)hipFree(d_raw);
    if (flags) return fail(BPPP_ERR_ENCODING);
    // fixed-base tables
    const int nwin = fb_nwin(W);
    const size_t per_win = fb_per_win(W);
    const size_t entries = (size_t)NB * nwin * per_win;
    c->table_bytes = entries * sizeof(apt_packed);
    HIP_TRY_C(hipMalloc(&c->d_table, c->table_bytes));
    // built in passes over groups of bases so that the scratch (x, y, z, prefix product: 160 B per entry) stays below ~32 GB
    const size_t per_base = (size_t)nwin * per_win;
    size_t group = ((size_t)32 << 30) / (per_base * 4 * sizeof(fe));
    if (group < 1) group = 1;
    if (group > (size_t)NB) group = (size_t)NB;
    fe* d_tmp = nullptr;
    const size_t gentries = group * per_base;
    HIP_TRY_C(hipMalloc(&d_tmp, gentries * 4 * sizeof(fe)));
    for (size_t b0 = 0; b0 < (size_t)NB; b0 += group) {
        const size_t nb = (size_t)NB - b0 < group ? (size_t)NB - b0 : group;
        FbBuild fb{c->d_gens, NB, W, c->d_table, d_tmp, d_tmp + gentries, d_tmp + 2 * gentries, d_tmp + 3 * gentries, (int)b0, (int)nb};
        size_t nthreads = nb * nwin * fb_chunks_per_window(W);
        unsigned blocks = (unsigned)((nthreads + BPPP_BLOCK - 1) / BPPP_BLOCK);
        k_fb_build_pass1<<<blocks, BPPP_BLOCK, 0, c->stream>>>(fb, nthreads);
        k_fb_build_pass2<<<blocks, BPPP_BLOCK, 0, c->stream>>>(fb, nthreads);
    }
    HIP_TRY_C(hipGetLastError());
    HIP_TRY_C(hipStreamSynchronize(c->stream));
    (void)hipFree(d_tmp);
#undef HIP_TRY_C
    *out = c;
    return BPPP_OK;
}

void bppp_ctx_destroy(bppp_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& tl : c->pending) { (void)hipEventDestroy(tl.a); (void)hipEventDestroy(tl.b); }
    for (auto& ev : c->event_pool) (void)hipEventDestroy(ev);
    if (c->d_gens && !c->borrows_tables) (void)hipFree(c->d_gens);
    if (c->d_table && !c->borrows_tables) (void)hipFree(c->d_table);
    if (c->d_ws) (void)hipFree(c->d_ws);
    if (c->d_straus) (void)hipFree(c->d_straus);
    if (c->d_rlc) (void)hipFree(c->d_rlc);
    if (c->d_bkt) (void)hipFree(c->d_bkt);
    if (c->d_atab) (void)hipFree(c->d_atab);
    if (c->d_tscr) (void)hipFree(c->d_tscr);
    if (c->d_pws) (void)hipFree(c->d_pws);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->d_io) (void)hipFree(c->d_io);
    if (c->d_gws) (void)hipFree(c->d_gws);
    if (c->d_gtab) (void)hipFree(c->d_gtab);
    if (c->d_expand) (void)hipFree(c->d_expand);
    if (c->d_flags) (void)hipFree(c->d_flags);
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int bppp_ctx_set_stream(bppp_ctx* c, void* hip_stream) {
    CtxLock lock_(c);
    if (!c) return BPPP_ERR_INVALID_ARG;
    int rc = drain_timings(c);
    if (rc != BPPP_OK) return rc;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return BPPP_OK;
}

int bppp_ctx_set_option(bppp_ctx* c, const char* name, long value) {
    CtxLock lock_(c);
    if (!c || !name) return BPPP_ERR_INVALID_ARG;
    if (std::strcmp(name, "rlc_superchunk") == 0) {
        if (value != 0 && (value < 64 || value > BPPP_BKT_MAX_M || (value & 7))) return BPPP_ERR_INVALID_ARG;
        c->rlc_super_m = (unsigned)value;
        return BPPP_OK;
    }
    if (std::strcmp(name, "max_batch") == 0) {
        if (value < 1024 || (value & 63)) return BPPP_ERR_INVALID_ARG;
        c->max_batch = (size_t)value;
        return BPPP_OK;
    }
    if (std::strcmp(name, "host_chunk") == 0) {
        if (value != 0 && (value < 1024 || (value & 63))) return BPPP_ERR_INVALID_ARG;
        c->host_chunk = (size_t)value;
        return BPPP_OK;
    }
    return BPPP_ERR_INVALID_ARG;
}
int bppp_ctx_synchronize(bppp_ctx* c) {
    CtxLock lock_(c);
    if (!c) return BPPP_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->aux_stream));
    return BPPP_OK;
}

size_t bppp_ctx_device_bytes(const bppp_ctx* c) {
    if (!c) return 0;
    return c->table_bytes + c->ws_bytes + c->straus_bytes + c->vtab_bytes + c->rlc_bytes + c->bkt_bytes + c->pws_bytes + c->stage_bytes + c->io_bytes + c->gws_bytes + c->gtab_bytes + (size_t)c->nbases * sizeof(apt);
}

int bppp_ctx_enable_timing(bppp_ctx* c, int enable) {
    CtxLock lock_(c);
    if (!c) return BPPP_ERR_INVALID_ARG;
    int rc = drain_timings(c);
    c->timing = enable != 0;
    return rc;
}
int bppp_ctx_get_timings(bppp_ctx* c, int max_entries, const char** names, double* total_ms, int64_t* launches, int reset) {
    CtxLock lock_(c);
    if (!c) return BPPP_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    int rc = drain_timings(c);
    if (rc != BPPP_OK) return rc;
    int n = max_entries < K_COUNT ? max_entries : K_COUNT;
    for (int i = 0; i < n; i++) {
        if (names) names[i] = kKernelNames[i];
        if (total_ms) total_ms[i] = c->total_ms[i];
        if (launches) launches[i] = c->launches[i];
    }
    if (reset)
        for (int i = 0; i < K_COUNT; i++) { c->total_ms[i] = 0; c->launches[i] = 0; }
    return n;
}

// optional pre-loaded transcripts of a verify call (device pointers): see VerifyWs::states
struct VerifyTranscripts { const void* d_states; size_t n_states; void* d_states_out; };
static int verify_device_part(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments,
                              const void* d_proofs, void* d_accept, void* d_status, void* d_trace, void* d_reject_count,
                              const uint8_t* rlc_seed, const VerifyTranscripts* tx, bool reset_reject_count);
// One call = one batch for the caller; internally a batch larger than max_batch proofs runs as consecutive parts on the same
// stream, so the per-proof workspace (~30 KB per proof) is bounded by max_batch whatever n is.  Proofs are independent, the reject
// counter accumulates across parts, and every per-proof array is simply offset.
static int verify_device_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments,
                              const void* d_proofs, void* d_accept, void* d_status, void* d_trace, void* d_reject_count,
                              const uint8_t* rlc_seed, const VerifyTranscripts* tx = nullptr) {
    if (!c) return BPPP_ERR_INVALID_ARG;
    const size_t cap = c->max_batch;
    if (n <= cap || !d_commitments || !d_proofs || !d_accept)
        return verify_device_part(c, label, label_len, n, d_commitments, d_proofs, d_accept, d_status, d_trace, d_reject_count, rlc_seed, tx, true);
    if (tx && tx->d_states && tx->n_states != 1 && tx->n_states != n) return BPPP_ERR_INVALID_ARG;
    const size_t SB = BPPP_TRANSCRIPT_STATE_BYTES;
    for (size_t lo = 0; lo < n; lo += cap) {
        const size_t m = n - lo < cap ? n - lo : cap;
        VerifyTranscripts part;
        if (tx) {
            part.d_states = tx->d_states && tx->n_states != 1 ? (const uint8_t*)tx->d_states + lo * SB : tx->d_states;
            part.n_states = tx->n_states == 1 ? 1 : m;
            part.d_states_out = tx->d_states_out ? (uint8_t*)tx->d_states_out + lo * SB : nullptr;
        }
        int rc = verify_device_part(c, label, label_len, m, (const uint8_t*)d_commitments + lo * 64,
                                    (const uint8_t*)d_proofs + lo * (size_t)BPPP_U64_PROOF_BYTES, (uint8_t*)d_accept + lo,
                                    d_status ? (int32_t*)d_status + lo : nullptr, d_trace ? (uint8_t*)d_trace + lo * (size_t)BPPP_U64_TRACE_BYTES : nullptr,
                                    d_reject_count, rlc_seed, tx ? &part : nullptr, lo == 0);
        if (rc != BPPP_OK) return rc;
    }
    return BPPP_OK;
}
static int verify_device_part(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments,
                              const void* d_proofs, void* d_accept, void* d_status, void* d_trace, void* d_reject_count,
                              const uint8_t* rlc_seed, const VerifyTranscripts* tx, bool reset_reject_count) {
    if (!c || (!label && label_len) || !d_commitments || !d_proofs || !d_accept) return BPPP_ERR_INVALID_ARG;
    if (c->ng != 16 || c->nh != 32) return BPPP_ERR_INVALID_ARG;   // u64 entry points need the u64 generator shape
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_capacity(c, n);
    if (rc != BPPP_OK) return rc;
    rc = ensure_vtab_capacity(c, n);
    if (rc != BPPP_OK) return rc;
    VerifyWs ws;
    std::memset(&ws, 0, sizeof ws);
    carve(c, ws, n);
    ws.atab = c->d_atab;
    ws.tscr = c->d_tscr;
    RlcWs rl;
    std::memset(&rl, 0, sizeof rl);
    if (rlc_seed) {
        rc = ensure_rlc_capacity(c, n);
        if (rc != BPPP_OK) return rc;
        rc = ensure_straus_capacity(c, n);   // k_rlc_lhs keeps the window table of C4 there
        if (rc != BPPP_OK) return rc;
        ws.straus = c->d_straus;
        for (int i = 0; i < 4; i++) {
            u64 v = 0;
            for (int k = 0; k < 8; k++) v |= (u64)rlc_seed[8 * i + k] << (8 * k);
            rl.seed[i] = v;
        }
        rl.lhs = c->d_rlc;
        rl.sc = c->d_rlc + 30 * n;
        rl.flag = (uint8_t*)(c->d_rlc + (30 + (size_t)BPPP_NG * 8) * n);
        rl.list = c->d_rlc + (30 + (size_t)BPPP_NG * 8) * c->rcap + c->rcap / 4;
        rl.count = (int*)(rl.list + c->rcap / BPPP_RLC_CHUNK + 1);
    }
    ws.commitments = (const uint8_t*)d_commitments;
    ws.proofs = (const uint8_t*)d_proofs;
    ws.accept = (uint8_t*)d_accept;
    ws.trace = (uint8_t*)d_trace;
    // per-proof status lives in caller memory when given, else in a spare corner of the staging buffer
    if (d_status) ws.status = (int32_t*)d_status;
    else {
        rc = ensure_stage(c, c->cap * sizeof(int32_t));
        if (rc != BPPP_OK) return rc;
        ws.status = (int32_t*)c->d_stage;
    }
    t_new(ws.base, label, (u32)label_len);   // Transcript::new(label), shared by every proof of the batch
    if (tx) {
        if (tx->d_states && tx->n_states != 1 && tx->n_states != n) return BPPP_ERR_INVALID_ARG;
        ws.states = (const uint8_t*)tx->d_states;
        ws.n_states = tx->n_states;
        ws.states_out = (uint8_t*)tx->d_states_out;
    }
    if (d_reject_count && reset_reject_count) HIP_TRY(hipMemsetAsync(d_reject_count, 0, sizeof(int), c->stream));
    const unsigned blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    hipStream_t s = c->stream;
#define LAUNCH_ON(st, id, ...)                                  \
    do {                                                        \
        rc = timed(c, id, st, [&]() { __VA_ARGS__; });          \
        if (rc != BPPP_OK) return rc;                           \
    } while (0)
#define LAUNCH(id, ...) LAUNCH_ON(s, id, __VA_ARGS__)
    // a grid that does not even fill one wavefront per SIMD gains nothing from the 256-VGPR cap: use the uncapped builds
    const bool small = !c->no_small && blocks <= (unsigned)c->n_simds;
    if (small) LAUNCH(K_PHASE1, k_verify_phase1_small<<<blocks, BPPP_BLOCK, 0, s>>>(ws));
    else LAUNCH(K_PHASE1, k_verify_phase1<<<blocks, BPPP_BLOCK, 0, s>>>(ws));
    const unsigned fb_blocks = (unsigned)((n * BPPP_FB_LANES + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    // C0 = variable-base half (window tables of the proof points, then the shared-doubling sum: one lane per proof, 1 wave
    // per SIMD) + fixed-base half (8 lanes per proof): independent, so they run concurrently on two streams and share the
    // SIMDs; round 1 adds the halves.
    hipStream_t a = c->serial_c0 ? s : c->aux_stream;   // diagnostic: un-overlapped kernel times
    LAUNCH(K_TABLES, k_verify_tables<<<blocks, BPPP_BLOCK, 0, s>>>(ws));
    HIP_TRY(hipEventRecord(c->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(a, c->ev_fork, 0));
    // the two fixed-base sums: 8 lanes per proof, or one from the size at which one lane per proof fills the SIMDs twice over
    const bool fb_one_lane = c->fb_one_lane_mode >= 0 ? c->fb_one_lane_mode == 1 : n >= (size_t)128 * (size_t)c->n_simds;
    const unsigned fb1_blocks = (unsigned)((n + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    if (fb_one_lane) LAUNCH_ON(a, K_C0_FIXED, k_verify_c0_fixed_l1<<<fb1_blocks, BPPP_FB_BLOCK, 0, a>>>(ws));
    else LAUNCH_ON(a, K_C0_FIXED, k_verify_c0_fixed<<<fb_blocks, BPPP_FB_BLOCK, 0, a>>>(ws));
    HIP_TRY(hipEventRecord(c->ev_join, a));
    // a batch whose four-lanes-per-proof grid still leaves the SIMDs under-filled runs its variable-base sums on lane groups
    const bool grouped = !c->no_lane_groups && 4 * (size_t)blocks <= (size_t)c->n_simds;
    const unsigned g4_blocks = (unsigned)((4 * n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    if (grouped) LAUNCH(K_C0_VAR, k_verify_c0_var_g4<<<g4_blocks, BPPP_BLOCK, 0, s>>>(ws));
    else if (small) LAUNCH(K_C0_VAR, k_verify_c0_var_small<<<blocks, BPPP_BLOCK, 0, s>>>(ws));
    else LAUNCH(K_C0_VAR, k_verify_c0_var<<<blocks, BPPP_BLOCK, 0, s>>>(ws));
    HIP_TRY(hipStreamWaitEvent(s, c->ev_join, 0));
    for (int k = 1; k <= 4; k++) {
        if (grouped) LAUNCH(K_ROUND, k_verify_round_g4<<<g4_blocks, BPPP_BLOCK, 0, s>>>(ws, k));
        else if (c->force_pairs || (!c->no_lane_groups && 2 * (size_t)blocks <= (size_t)c->n_simds))
            LAUNCH(K_ROUND, k_verify_round_g2<<<(unsigned)((2 * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(ws, k));
        else if (small) LAUNCH(K_ROUND, k_verify_round_small<<<blocks, BPPP_BLOCK, 0, s>>>(ws, k));
        else LAUNCH(K_ROUND, k_verify_round<<<blocks, BPPP_BLOCK, 0, s>>>(ws, k));
    }
    LAUNCH(K_FINAL_SCALARS, k_verify_final_scalars<<<blocks, BPPP_BLOCK, 0, s>>>(ws));
    if (!rlc_seed) {
        if (fb_one_lane) LAUNCH(K_FINAL_CHECK, k_verify_final_check_l1<<<fb1_blocks, BPPP_FB_BLOCK, 0, s>>>(ws));
        else LAUNCH(K_FINAL_CHECK, k_verify_final_check<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(ws));
        LAUNCH(K_ACCEPT, k_verify_accept<<<blocks, BPPP_BLOCK, 0, s>>>(ws, (int*)d_reject_count));
    } else {
        // combined check per chunk of 8 proofs; chunks that fail it (or hold a flagged proof) fall through to the exact kernels
        const size_t nchunks = (n + BPPP_RLC_CHUNK - 1) / BPPP_RLC_CHUNK;
        const unsigned chunk_blocks = (unsigned)((nchunks * BPPP_RLC_CHUNK + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
        HIP_TRY(hipMemsetAsync(d_accept, 0, n, s));
        HIP_TRY(hipMemsetAsync(rl.count, 0, sizeof(int), s));
        if (c->rlc_super_m) {
            // bucket stage first: superchunks of rlc_super_m proofs, one combined check each; the chunk-of-8 kernels below only see
            // the proofs of superchunks that failed it
            const size_t SM = c->rlc_super_m, nsuper = (n + SM - 1) / SM;
            rc = ensure_bucket_capacity(c, n);
            if (rc != BPPP_OK) return rc;
            BucketWs bw;
            std::memset(&bw, 0, sizeof bw);
            bw.N = n; bw.M = (u32)SM;
            for (int i = 0; i < 4; i++) bw.seed[i] = rl.seed[i];
            bw.status = ws.status; bw.acc = ws.acc; bw.fsc = ws.fsc; bw.accept = ws.accept;
            uint8_t* p = c->d_bkt;
            const size_t capn = (n + BPPP_BLOCK - 1) / BPPP_BLOCK * BPPP_BLOCK;
            bw.wab = (u64*)p; p += align16(capn * 16);
            bw.c4 = (c4_packed*)p; p += align16(capn * sizeof(c4_packed));
            bw.lhs = (u32*)p; p += align16(nsuper * 30 * 4);
            bw.asc = (u32*)p; p += align16(nsuper * (size_t)BPPP_NG * 32);
            bw.sflag = p;
            bw.fb.table = c->d_table; bw.fb.W = c->fb_w; bw.fb.N = nsuper;
            const size_t lds_bytes = ((size_t)4 * (512 + SM) + 8 * 30) * sizeof(u32);
            (void)hipFuncSetAttribute((const void*)k_bkt_accumulate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            LAUNCH(K_BKT_PREPARE, k_bkt_prepare<<<blocks, BPPP_BLOCK, 0, s>>>(bw));
            LAUNCH(K_BKT_ACCUMULATE, k_bkt_accumulate<<<(unsigned)nsuper, 256, lds_bytes, s>>>(bw));
            LAUNCH(K_BKT_SCALARS, k_bkt_scalars<<<(unsigned)nsuper, 256, 0, s>>>(bw));
            LAUNCH(K_BKT_CHECK, k_bkt_check<<<(unsigned)nsuper, 64, 0, s>>>(bw));
            rl.sflag = bw.sflag;
            rl.super_m = (u32)SM;
        }
        LAUNCH(K_RLC_LHS, k_rlc_lhs<<<blocks, BPPP_BLOCK, 0, s>>>(ws, rl));
        LAUNCH(K_RLC_CHUNK, k_rlc_chunk<<<chunk_blocks, BPPP_FB_BLOCK, 0, s>>>(ws, rl));
        LAUNCH(K_FINAL_CHECK, k_verify_final_check_flagged<<<1024, 64, 0, s>>>(ws, rl));
        LAUNCH(K_FINAL_CHECK, k_verify_final_check_flagged_dense<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(ws, rl));
        LAUNCH(K_ACCEPT, k_verify_accept_flagged<<<blocks, BPPP_BLOCK, 0, s>>>(ws, rl, (int*)d_reject_count));
        if (c->rlc_debug) {   // diagnostic: how many chunks went to the exact kernels
            std::vector<uint8_t> hf(nchunks);
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipMemcpy(hf.data(), rl.flag, nchunks, hipMemcpyDeviceToHost));
            size_t cnt = 0;
            for (uint8_t f : hf) cnt += f ? 1 : 0;
            std::fprintf(stderr, "bppp rlc: %zu of %zu chunks re-checked exactly\n", cnt, nchunks);
        }
    }
    if (ws.states_out) k_verify_export_states<<<blocks, BPPP_BLOCK, 0, s>>>(ws);
#undef LAUNCH
#undef LAUNCH_ON
    HIP_TRY(hipGetLastError());
    return BPPP_OK;
}
int bppp_u64_verify_batch_transcript_device(bppp_ctx* c, size_t n, const void* d_states, size_t n_states, const void* d_commitments,
                                            const void* d_proofs, void* d_accept, void* d_status, void* d_reject_count, void* d_states_out) {
    CtxLock lock_(c);
    if (!d_states) return BPPP_ERR_INVALID_ARG;
    VerifyTranscripts tx = {d_states, n_states, d_states_out};
    return verify_device_impl(c, nullptr, 0, n, d_commitments, d_proofs, d_accept, d_status, nullptr, d_reject_count, nullptr, &tx);
}
int bppp_u64_verify_batch_transcript(bppp_ctx* c, size_t n, const uint8_t* states, size_t n_states, const uint8_t* commitments,
                                     const uint8_t* proofs, uint8_t* accept, int32_t* status, uint8_t* states_out) {
    CtxLock lock_(c);
    if (!c || !states || !commitments || !proofs || !accept || (n_states != 1 && n_states != n)) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    for (size_t i = 0; i < n_states; i++)
        if (states[203 * i + 200] >= BPPP_STROBE_R || states[203 * i + 201] > BPPP_STROBE_R) return BPPP_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    uint8_t* d = nullptr;
    const size_t SB = BPPP_TRANSCRIPT_STATE_BYTES;
    const size_t o_c = 0, o_p = align16(o_c + n * 64), o_a = align16(o_p + n * (size_t)BPPP_U64_PROOF_BYTES), o_s = align16(o_a + n),
                 o_ti = align16(o_s + n * sizeof(int32_t)), o_to = align16(o_ti + n_states * SB), total = align16(o_to + n * SB);
    HIP_TRY(hipMalloc(&d, total));
    int rc = BPPP_OK;
    hipStream_t st = c->stream;
    hipError_t e = hipMemcpyAsync(d + o_c, commitments, n * 64, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_p, proofs, n * (size_t)BPPP_U64_PROOF_BYTES, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_ti, states, n_states * SB, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        rc = bppp_u64_verify_batch_transcript_device(c, n, d + o_ti, n_states, d + o_c, d + o_p, d + o_a, d + o_s, nullptr,
                                                     states_out ? d + o_to : nullptr);
        if (rc == BPPP_OK) {
            e = hipMemcpyAsync(accept, d + o_a, n, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && status) e = hipMemcpyAsync(status, d + o_s, n * sizeof(int32_t), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && states_out) e = hipMemcpyAsync(states_out, d + o_to, n * SB, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
        }
    }
    (void)hipFree(d);
    if (e != hipSuccess) { g_last_error = std::string("verify_batch_transcript: ") + hipGetErrorString(e); return BPPP_ERR_HIP; }
    return rc;
}
// merlin::Transcript as 203 serialized bytes, on the host: lets a C caller build the pre-loaded states without merlin and lets
// the tests follow the reference's `t: &mut Transcript` contract end to end (no GPU involved)
int bppp_transcript_new(const uint8_t* label, size_t label_len, uint8_t state_out[203]) {
    if ((!label && label_len) || !state_out) return BPPP_ERR_INVALID_ARG;
    strobe t;
    t_new(t, label, (u32)label_len);
    strobe_to_bytes(state_out, t, 2);       // the last operation of Transcript::new is the AD of the dom-sep message
    return BPPP_OK;
}
int bppp_transcript_append_message(uint8_t state[203], const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len) {
    if (!state || (!label && label_len) || (!msg && msg_len) || msg_len > 0xFFFFFFFFu) return BPPP_ERR_INVALID_ARG;
    strobe t;
    if (!strobe_from_bytes(t, state)) return BPPP_ERR_INVALID_ARG;
    uint8_t len4[4] = {(uint8_t)msg_len, (uint8_t)(msg_len >> 8), (uint8_t)(msg_len >> 16), (uint8_t)(msg_len >> 24)};
    strobe_meta_ad(t, label, (u32)label_len, false);
    strobe_meta_ad(t, len4, 4, true);
    strobe_ad(t, msg, (u32)msg_len, false);
    strobe_to_bytes(state, t, 2);
    return BPPP_OK;
}
int bppp_transcript_challenge_bytes(uint8_t state[203], const uint8_t* label, size_t label_len, uint8_t* out, size_t n) {
    if (!state || (!label && label_len) || (!out && n) || n > 0xFFFFFFFFu) return BPPP_ERR_INVALID_ARG;
    strobe t;
    if (!strobe_from_bytes(t, state)) return BPPP_ERR_INVALID_ARG;
    uint8_t len4[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    strobe_meta_ad(t, label, (u32)label_len, false);
    strobe_meta_ad(t, len4, 4, true);
    strobe_prf(t, out, (u32)n);
    strobe_to_bytes(state, t, 7);
    return BPPP_OK;
}
int bppp_u64_verify_batch_device(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments,
                                 const void* d_proofs, void* d_accept, void* d_status, void* d_trace, void* d_reject_count) {
    CtxLock lock_(c);
    return verify_device_impl(c, label, label_len, n, d_commitments, d_proofs, d_accept, d_status, d_trace, d_reject_count, nullptr);
}
int bppp_u64_verify_batch_rlc_device(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments,
                                     const void* d_proofs, void* d_accept, void* d_status, void* d_reject_count, const uint8_t seed[32]) {
    CtxLock lock_(c);
    if (!seed) return BPPP_ERR_INVALID_ARG;
    return verify_device_impl(c, label, label_len, n, d_commitments, d_proofs, d_accept, d_status, nullptr, d_reject_count, seed);
}

static int verify_host_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                            const uint8_t* proofs, uint8_t* accept, int32_t* status, const uint8_t* rlc_seed) {
    if (!c || !commitments || !proofs || !accept) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    // persistent I/O staging of the host-buffer entry points (grow-only; separate from d_stage, which the device call may use)
    const size_t o_c = 0, o_p = align16(o_c + n * 64), o_a = align16(o_p + n * (size_t)BPPP_U64_PROOF_BYTES), o_s = align16(o_a + n),
                 need = align16(o_s + n * sizeof(int32_t));
    if (need > c->io_bytes) {
        if (c->d_io) { (void)hipFree(c->d_io); c->d_io = nullptr; c->io_bytes = 0; }
        HIP_TRY(hipMalloc(&c->d_io, need));
        c->io_bytes = need;
    }
    uint8_t *d_c = c->d_io + o_c, *d_p = c->d_io + o_p, *d_a = c->d_io + o_a;
    int32_t* d_s = (int32_t*)(c->d_io + o_s);
    const size_t CH = c->host_chunk;
    if (CH == 0 || n <= CH + CH / 2) {
        HIP_TRY(hipMemcpyAsync(d_c, commitments, n * 64, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_p, proofs, n * (size_t)BPPP_U64_PROOF_BYTES, hipMemcpyHostToDevice, c->stream));
        int rc = verify_device_impl(c, label, label_len, n, d_c, d_p, d_a, d_s, nullptr, nullptr, rlc_seed);
        if (rc != BPPP_OK) return rc;
    } else {
        // Large batch: proofs are independent, so the batch is verified chunk by chunk while the next chunk crosses PCIe on a second
        // stream (from pageable memory the runtime stages the copy and this thread blocks in it; the GPU keeps verifying meanwhile).
        // A chunk is one full grid of the lane kernels, so the kernels run exactly as they do for a resident batch.
        if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        if (!c->ev_copy) HIP_TRY(hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming));
        // the staging buffer may still be read by kernels of an earlier call on c->stream only if that call returned early on an
        // error; order the first upload after whatever is queued there
        HIP_TRY(hipEventRecord(c->ev_copy, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->ev_copy, 0));
        for (size_t lo = 0; lo < n; lo += CH) {
            size_t m = n - lo;
            if (m > CH + CH / 2) m = CH;          // the tail joins the last chunk rather than running as a sliver
            HIP_TRY(hipMemcpyAsync(d_c + lo * 64, commitments + lo * 64, m * 64, hipMemcpyHostToDevice, c->copy_stream));
            HIP_TRY(hipMemcpyAsync(d_p + lo * (size_t)BPPP_U64_PROOF_BYTES, proofs + lo * (size_t)BPPP_U64_PROOF_BYTES,
                                   m * (size_t)BPPP_U64_PROOF_BYTES, hipMemcpyHostToDevice, c->copy_stream));
            HIP_TRY(hipEventRecord(c->ev_copy, c->copy_stream));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_copy, 0));
            int rc = verify_device_impl(c, label, label_len, m, d_c + lo * 64, d_p + lo * (size_t)BPPP_U64_PROOF_BYTES, d_a + lo, d_s + lo,
                                        nullptr, nullptr, rlc_seed);
            if (rc != BPPP_OK) return rc;
            if (m != CH) break;
        }
    }
    HIP_TRY(hipMemcpyAsync(accept, d_a, n, hipMemcpyDeviceToHost, c->stream));
    if (status) HIP_TRY(hipMemcpyAsync(status, d_s, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BPPP_OK;
}
int bppp_u64_verify_batch(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                          const uint8_t* proofs, uint8_t* accept, int32_t* status) {
    CtxLock lock_(c);
    return verify_host_impl(c, label, label_len, n, commitments, proofs, accept, status, nullptr);
}
int bppp_u64_verify_batch_rlc(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                              const uint8_t* proofs, uint8_t* accept, int32_t* status, const uint8_t seed[32]) {
    CtxLock lock_(c);
    if (!seed) return BPPP_ERR_INVALID_ARG;
    return verify_host_impl(c, label, label_len, n, commitments, proofs, accept, status, seed);
}

int bppp_u64_commit_value_batch(bppp_ctx* c, size_t n, const uint64_t* x, const uint8_t* s, uint8_t* out) {
    CtxLock lock_(c);
    if (!c || !x || !s || !out) return BPPP_ERR_INVALID_ARG;
    if (c->ng != 16 || c->nh != 32) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_capacity(c, n);
    if (rc != BPPP_OK) return rc;
    size_t need = n * 8 + n * 32 + n * 64;
    rc = ensure_stage(c, need);
    if (rc != BPPP_OK) return rc;
    uint64_t* d_x = (uint64_t*)c->d_stage;
    uint8_t* d_s = c->d_stage + n * 8;
    uint8_t* d_o = d_s + n * 32;
    VerifyWs ws;
    std::memset(&ws, 0, sizeof ws);
    carve(c, ws, n);
    HIP_TRY(hipMemsetAsync(c->d_flags, 0, sizeof(int), c->stream));
    HIP_TRY(hipMemcpyAsync(d_x, x, n * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_s, s, n * 32, hipMemcpyHostToDevice, c->stream));
    const unsigned blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    k_commit_value<<<blocks, BPPP_BLOCK, 0, c->stream>>>(ws, d_x, d_s, d_o, c->d_flags);
    HIP_TRY(hipGetLastError());
    int flags = 0;
    HIP_TRY(hipMemcpyAsync(out, d_o, n * 64, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&flags, c->d_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return flags ? BPPP_ERR_INVALID_ARG : BPPP_OK;
}

static int prove_device_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_x, const void* d_s,
                             const void* d_rnd, void* d_proofs, void* d_commitments, void* d_status, const VerifyTranscripts* tx);
int bppp_u64_prove_batch_device(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_x, const void* d_s,
                                const void* d_rnd, void* d_proofs, void* d_commitments, void* d_status) {
    CtxLock lock_(c);
    return prove_device_impl(c, label, label_len, n, d_x, d_s, d_rnd, d_proofs, d_commitments, d_status, nullptr);
}
int bppp_u64_prove_batch_transcript_device(bppp_ctx* c, size_t n, const void* d_states, size_t n_states, const void* d_x, const void* d_s,
                                           const void* d_rnd, void* d_proofs, void* d_commitments, void* d_status, void* d_states_out) {
    CtxLock lock_(c);
    if (!d_states || (n_states != 1 && n_states != n)) return BPPP_ERR_INVALID_ARG;
    VerifyTranscripts tx = {d_states, n_states, d_states_out};
    return prove_device_impl(c, nullptr, 0, n, d_x, d_s, d_rnd, d_proofs, d_commitments, d_status, &tx);
}
static int prove_device_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_x, const void* d_s,
                             const void* d_rnd, void* d_proofs, void* d_commitments, void* d_status, const VerifyTranscripts* tx) {
    if (!c || (!label && label_len) || !d_x || !d_s || !d_rnd || !d_proofs || !d_commitments) return BPPP_ERR_INVALID_ARG;
    if (c->ng != 16 || c->nh != 32) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_prove_capacity(c, n);
    if (rc != BPPP_OK) return rc;
    ProveWs w;
    std::memset(&w, 0, sizeof w);
    w.N = n;
    w.x = (const uint64_t*)d_x; w.s = (const uint8_t*)d_s; w.rnd = (const uint8_t*)d_rnd;
    w.proofs = (uint8_t*)d_proofs; w.commitments = (uint8_t*)d_commitments;
    if (d_status) w.status = (int32_t*)d_status;
    else {
        rc = ensure_stage(c, c->cap * sizeof(int32_t));
        if (rc != BPPP_OK) return rc;
        w.status = (int32_t*)c->d_stage;
    }
    u32* p = c->d_pws;
    w.tstate = p; p += 52 * n;
    w.sv = p; p += (size_t)SV_COUNT * 8 * n;
    w.msc = p; p += (size_t)BPPP_MSC_SETS * BPPP_NG * 8 * n;
    w.pbuf = p;
    w.straus = c->d_straus;
    w.fb.table = c->d_table; w.fb.W = c->fb_w; w.fb.N = n;
    t_new(w.base, label, (u32)label_len);
    if (tx) { w.states = (const uint8_t*)tx->d_states; w.n_states = tx->n_states; w.states_out = (uint8_t*)tx->d_states_out; }
    const unsigned blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    const unsigned fb_blocks = (unsigned)((n * BPPP_FB_LANES + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    hipStream_t s = c->stream;
#define PLAUNCH(id, ...)                                       \
    do {                                                       \
        rc = timed(c, id, s, [&]() { __VA_ARGS__; });          \
        if (rc != BPPP_OK) return rc;                          \
    } while (0)
#define PMSM(job) PLAUNCH(K_PROVE_MSM, k_prove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, job))
    PLAUNCH(K_PROVE_STAGES, k_prove_stage_a<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    PMSM(job_v());
    PLAUNCH(K_PROVE_STAGES, k_prove_stage_b<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    PMSM(job_rcom()); PMSM(job_co()); PMSM(job_cl()); PMSM(job_cr());
    PLAUNCH(K_PROVE_STAGES, k_prove_stage_d<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    PMSM(job_cs());
    PLAUNCH(K_PROVE_STAGES, k_prove_stage_f<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    PMSM(job_c0());
    for (int k = 1; k <= 4; k++) {
        PLAUNCH(K_PROVE_ROUND_SCALARS, k_prove_round_scalars<<<blocks, BPPP_BLOCK, 0, s>>>(w, k));
        PMSM(job_x()); PMSM(job_r());
        if (!c->no_lane_groups && 4 * (size_t)blocks <= (size_t)c->n_simds)      // small batch: lane groups (see verify_device_part)
            PLAUNCH(K_PROVE_ROUND_FOLD, k_prove_round_fold_g4<<<(unsigned)((4 * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(w, k));
        else
            PLAUNCH(K_PROVE_ROUND_FOLD, k_prove_round_fold<<<blocks, BPPP_BLOCK, 0, s>>>(w, k));
    }
    if (w.states_out) k_prove_export_states<<<blocks, BPPP_BLOCK, 0, s>>>(w);
#undef PMSM
#undef PLAUNCH
    HIP_TRY(hipGetLastError());
    return BPPP_OK;
}
// U64RangeProofProtocol::prove with the caller's transcripts (u64_proof.rs:57: `t: &mut Transcript`), host buffers
int bppp_u64_prove_batch_transcript(bppp_ctx* c, size_t n, const uint8_t* states, size_t n_states, const uint64_t* x, const uint8_t* s,
                                    const uint8_t* rnd, uint8_t* proofs, uint8_t* commitments, int32_t* status, uint8_t* states_out) {
    CtxLock lock_(c);
    if (!c || !states || !x || !s || !rnd || !proofs || !commitments || (n_states != 1 && n_states != n)) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    for (size_t i = 0; i < n_states; i++)
        if (states[203 * i + 200] >= BPPP_STROBE_R || states[203 * i + 201] > BPPP_STROBE_R) return BPPP_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    const size_t SB = BPPP_TRANSCRIPT_STATE_BYTES;
    uint8_t* d = nullptr;
    const size_t o_x = 0, o_s = align16(o_x + n * 8), o_r = align16(o_s + n * 32), o_p = align16(o_r + n * 52 * 32),
                 o_c = align16(o_p + n * (size_t)BPPP_U64_PROOF_BYTES), o_st = align16(o_c + n * 64), o_ti = align16(o_st + n * sizeof(int32_t)),
                 o_to = align16(o_ti + n_states * SB), total = align16(o_to + n * SB);
    HIP_TRY(hipMalloc(&d, total));
    int rc = BPPP_OK;
    hipStream_t st = c->stream;
    hipError_t e = hipMemcpyAsync(d + o_x, x, n * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_s, s, n * 32, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_r, rnd, n * 52 * 32, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_ti, states, n_states * SB, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        rc = bppp_u64_prove_batch_transcript_device(c, n, d + o_ti, n_states, d + o_x, d + o_s, d + o_r, d + o_p, d + o_c, d + o_st,
                                                    states_out ? d + o_to : nullptr);
        if (rc == BPPP_OK) {
            e = hipMemcpyAsync(proofs, d + o_p, n * (size_t)BPPP_U64_PROOF_BYTES, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(commitments, d + o_c, n * 64, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && status) e = hipMemcpyAsync(status, d + o_st, n * sizeof(int32_t), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && states_out) e = hipMemcpyAsync(states_out, d + o_to, n * SB, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
        }
    }
    (void)hipFree(d);
    if (e != hipSuccess) { g_last_error = std::string("prove_batch_transcript: ") + hipGetErrorString(e); return BPPP_ERR_HIP; }
    return rc;
}

int bppp_u64_prove_batch(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const uint64_t* x, const uint8_t* s,
                         const uint8_t* rnd, uint8_t* proofs, uint8_t* commitments, int32_t* status) {
    CtxLock lock_(c);
    if (!c || !x || !s || !rnd || !proofs || !commitments) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    uint8_t* d = nullptr;
    const size_t o_x = 0, o_s = o_x + n * 8, o_r = o_s + n * 32, o_p = o_r + n * 52 * 32, o_c = o_p + n * (size_t)BPPP_U64_PROOF_BYTES,
                 o_st = o_c + n * 64, total = o_st + n * sizeof(int32_t);
    HIP_TRY(hipMalloc(&d, total));
    int rc = BPPP_OK;
    hipError_t e = hipMemcpyAsync(d + o_x, x, n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_s, s, n * 32, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_r, rnd, n * 52 * 32, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        rc = bppp_u64_prove_batch_device(c, label, label_len, n, d + o_x, d + o_s, d + o_r, d + o_p, d + o_c, d + o_st);
        if (rc == BPPP_OK) {
            e = hipMemcpyAsync(proofs, d + o_p, n * (size_t)BPPP_U64_PROOF_BYTES, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(commitments, d + o_c, n * 64, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess && status) e = hipMemcpyAsync(status, d + o_st, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
    }
    (void)hipFree(d);
    if (e != hipSuccess) { g_last_error = std::string("prove_batch: ") + hipGetErrorString(e); return BPPP_ERR_HIP; }
    return rc;
}

int bppp_u64_verify_batch_sec1_device(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments33,
                                      const void* d_proofs525, void* d_accept, void* d_status, void* d_trace, void* d_reject_count) {
    CtxLock lock_(c);
    if (!c || !d_commitments33 || !d_proofs525 || !d_accept) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t need = n * (64 + (size_t)BPPP_U64_PROOF_BYTES);
    if (need > c->expand_bytes) {
        if (c->d_expand) { (void)hipFree(c->d_expand); c->d_expand = nullptr; c->expand_bytes = 0; }
        HIP_TRY(hipMalloc(&c->d_expand, need));
        c->expand_bytes = need;
    }
    uint8_t* d_c64 = c->d_expand;
    uint8_t* d_p928 = c->d_expand + n * 64;
    const unsigned blocks = (unsigned)((n * 16 + 255) / 256);
    k_sec1_expand<<<blocks, 256, 0, c->stream>>>(d_c64, d_p928, (const uint8_t*)d_commitments33, (const uint8_t*)d_proofs525, n);
    HIP_TRY(hipGetLastError());
    return bppp_u64_verify_batch_device(c, label, label_len, n, d_c64, d_p928, d_accept, d_status, d_trace, d_reject_count);
}

int bppp_u64_verify_batch_sec1(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments33,
                               const uint8_t* proofs525, uint8_t* accept, int32_t* status) {
    CtxLock lock_(c);
    if (!c || !commitments33 || !proofs525 || !accept) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    uint8_t* d = nullptr;
    const size_t o_c = 0, o_p = o_c + n * 33, o_a = o_p + n * (size_t)BPPP_U64_PROOF_SEC1_BYTES, o_s = (o_a + n + 3) / 4 * 4,
                 total = o_s + n * sizeof(int32_t);
    HIP_TRY(hipMalloc(&d, total));
    int rc = BPPP_OK;
    hipError_t e = hipMemcpyAsync(d + o_c, commitments33, n * 33, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_p, proofs525, n * (size_t)BPPP_U64_PROOF_SEC1_BYTES, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        rc = bppp_u64_verify_batch_sec1_device(c, label, label_len, n, d + o_c, d + o_p, d + o_a, d + o_s, nullptr, nullptr);
        if (rc == BPPP_OK) {
            e = hipMemcpyAsync(accept, d + o_a, n, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess && status) e = hipMemcpyAsync(status, d + o_s, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
    }
    (void)hipFree(d);
    if (e != hipSuccess) { g_last_error = std::string("verify_batch_sec1: ") + hipGetErrorString(e); return BPPP_ERR_HIP; }
    return rc;
}

// caller transcripts of the generic verifiers (host pointers): n_states x 203 in, n x 203 out (optional)
struct HostTranscripts { const uint8_t* states; size_t n_states; uint8_t* states_out; };
static int check_host_transcripts(const HostTranscripts* tx, size_t n) {
    if (!tx) return BPPP_OK;
    if (!tx->states || (tx->n_states != 1 && tx->n_states != n)) return BPPP_ERR_INVALID_ARG;
    for (size_t i = 0; i < tx->n_states; i++)
        if (tx->states[203 * i + 200] >= BPPP_STROBE_R || tx->states[203 * i + 201] > BPPP_STROBE_R) return BPPP_ERR_INVALID_ARG;
    return BPPP_OK;
}
// device copies of a prover call's transcripts (host-buffer entry points): states in, advanced states out
struct TxDev {
    uint8_t *d_in = nullptr, *d_out = nullptr;
    ~TxDev() { if (d_in) (void)hipFree(d_in); if (d_out) (void)hipFree(d_out); }
    int begin(const HostTranscripts* tx, size_t n, hipStream_t s, TranscriptIo& io, int& divergent) {
        io.states = nullptr; io.n_states = 0; io.states_out = nullptr; io.no_ops = 0;
        divergent = 0;
        if (!tx) return BPPP_OK;
        int rc = check_host_transcripts(tx, n);
        if (rc != BPPP_OK) return rc;
        HIP_TRY(hipMalloc(&d_in, tx->n_states * 203));
        HIP_TRY(hipMemcpyAsync(d_in, tx->states, tx->n_states * 203, hipMemcpyHostToDevice, s));
        if (tx->states_out) HIP_TRY(hipMalloc(&d_out, n * 203));
        io.states = d_in; io.n_states = tx->n_states; io.states_out = d_out;
        divergent = tx->n_states != 1;
        return BPPP_OK;
    }
    // after the last prover kernel: serialize every instance's transcript and queue the copy back
    int finish(const HostTranscripts* tx, const TranscriptIo& io, const strobe& base, const u32* tstate, size_t n, const int32_t* status, hipStream_t s) {
        if (!tx || !tx->states_out) return BPPP_OK;
        k_gprove_export_states<<<(unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(io, base, tstate, n, status);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(tx->states_out, d_out, n * 203, hipMemcpyDeviceToHost, s));
        return BPPP_OK;
    }
};
// fast variable-base path of the generic verifiers' rounds: per instance 2 x rounds window tables (16 entries of 64 B per point), the
// running products of their build (14 per point) and the decoded round points -- 1.2 KB + 0.55 KB + 64 B per point, grow-only
static int wnla_fast_setup(bppp_ctx* c, WnlaWs& w, size_t n, size_t rounds) {
    w.atab = nullptr; w.tscr = nullptr; w.rpts = nullptr;
    if (rounds == 0 || c->generic_slow_rounds) return BPPP_OK;
    const size_t np = 2 * rounds;
    const size_t b_tab = align16(np * 16 * sizeof(apt_packed) * n), b_scr = align16(14 * np * 10 * sizeof(u32) * n), b_pts = align16(np * 16 * sizeof(u32) * n);
    const size_t need = b_tab + b_scr + b_pts;
    if (need > c->gtab_bytes) {
        if (c->d_gtab) { (void)hipFree(c->d_gtab); c->d_gtab = nullptr; c->gtab_bytes = 0; }
        HIP_TRY(hipMalloc(&c->d_gtab, need));
        c->gtab_bytes = need;
    }
    w.atab = (apt_packed*)c->d_gtab;
    w.tscr = (u32*)(c->d_gtab + b_tab);
    w.rpts = (u32*)(c->d_gtab + b_tab + b_scr);
    return BPPP_OK;
}
// lanes per instance for the generic rounds: 4 or 2 while that still leaves wavefront slots free (and the fast path's tables exist)
static int wnla_round_group(const bppp_ctx* c, const WnlaWs& w, unsigned blocks) {
    if (!w.atab || c->no_lane_groups) return 1;
    if (4 * (size_t)blocks <= (size_t)c->n_simds) return 4;
    if (2 * (size_t)blocks <= (size_t)c->n_simds) return 2;
    return 1;
}
// ---- generic WeightNormLinearArgument entry points (host pointers; one device blob per call)
struct WnlaBlob {
    uint8_t* d = nullptr;
    ~WnlaBlob() { if (d) (void)hipFree(d); }
};

static int wnla_run(bppp_ctx* c, bool commit, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                    const uint8_t* cvec, const uint8_t* rho, const uint8_t* mu, size_t rounds, const uint8_t* proof_r,
                    const uint8_t* proof_x, const uint8_t* proof_l, size_t nl, const uint8_t* proof_n, size_t nn, uint8_t* out_points,
                    uint8_t* accept, int32_t* status, const HostTranscripts* tx = nullptr) {
    HIP_TRY(hipSetDevice(c->device));
    if (rounds > 12 || nl > 4096 || nn > 4096) return BPPP_ERR_INVALID_ARG;
    int rc = check_host_transcripts(tx, n);
    if (rc != BPPP_OK) return rc;
    rc = ensure_straus_capacity(c, n);
    if (rc != BPPP_OK) return rc;
    const size_t NB = (size_t)c->nbases, T = (size_t)1 << rounds;
    // layout of the blob: inputs | outputs | workspace
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align16(off + bytes); return o; };
    const size_t o_com = take(n * 64), o_c = take(n * (size_t)c->nh * 32), o_rho = take(n * 32), o_mu = take(n * 32),
                 o_r = take(n * rounds * 64), o_x = take(n * rounds * 64), o_l = take(n * nl * 32), o_n = take(n * nn * 32),
                 o_out = take(n * 64), o_acc = take(n), o_st = take(n * 4), o_ts = take(52 * n * 4), o_a = take(30 * n * 4),
                 o_pf = take(30 * n * 4), o_ys = take((rounds ? rounds : 1) * 8 * n * 4), o_tab = take(2 * T * 8 * n * 4),
                 o_msc = take(NB * 8 * n * 4), o_ti = take(tx ? tx->n_states * 203 : 0), o_to = take(tx && tx->states_out ? n * 203 : 0);
    WnlaBlob blob;
    HIP_TRY(hipMalloc(&blob.d, off + 16));
    uint8_t* d = blob.d;
    hipStream_t s = c->stream;
    auto up = [&](size_t o, const uint8_t* src, size_t bytes) -> hipError_t {
        return (src && bytes) ? hipMemcpyAsync(d + o, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
    };
    if (tx) HIP_TRY(up(o_ti, tx->states, tx->n_states * 203));
    HIP_TRY(up(o_com, commitments, n * 64));
    HIP_TRY(up(o_c, cvec, n * (size_t)c->nh * 32));
    HIP_TRY(up(o_rho, rho, n * 32));
    HIP_TRY(up(o_mu, mu, n * 32));
    HIP_TRY(up(o_r, proof_r, n * rounds * 64));
    HIP_TRY(up(o_x, proof_x, n * rounds * 64));
    HIP_TRY(up(o_l, proof_l, n * nl * 32));
    HIP_TRY(up(o_n, proof_n, n * nn * 32));
    WnlaWs w;
    std::memset(&w, 0, sizeof w);
    w.N = n; w.ng = c->ng; w.nh = c->nh; w.rounds = (int)rounds; w.nl = (int)nl; w.nn = (int)nn;
    w.commitments = d + o_com; w.c = d + o_c; w.rho = d + o_rho; w.mu = d + o_mu; w.proof_r = d + o_r; w.proof_x = d + o_x;
    w.proof_l = d + o_l; w.proof_n = d + o_n; w.out_points = d + o_out; w.accept = d + o_acc; w.status = (int32_t*)(d + o_st);
    w.tstate = (u32*)(d + o_ts); w.acc = (u32*)(d + o_a); w.pfix = (u32*)(d + o_pf); w.ys = (u32*)(d + o_ys);
    w.tab = (u32*)(d + o_tab); w.msc = (u32*)(d + o_msc);
    w.stride_r = rounds * 64; w.stride_x = rounds * 64; w.stride_l = nl * 32; w.stride_n = nn * 32;
    w.straus = c->d_straus;
    w.fb.table = c->d_table; w.fb.W = c->fb_w; w.fb.N = n;
    if (!commit) t_new(w.base, label, (u32)label_len);
    if (tx) {
        w.tio.states = d + o_ti; w.tio.n_states = tx->n_states; w.tio.states_out = tx->states_out ? d + o_to : nullptr;
        w.tio.no_ops = rounds == 0;
        w.divergent_positions = tx->n_states != 1;
    }
    const unsigned blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    const unsigned fb_blocks = (unsigned)((n * BPPP_FB_LANES + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    if (commit) {
        k_wnla_commit_scalars<<<blocks, BPPP_BLOCK, 0, s>>>(w);
        k_wnla_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 1);
        k_wnla_commit_store<<<blocks, BPPP_BLOCK, 0, s>>>(w);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out_points, d + o_out, n * 64, hipMemcpyDeviceToHost, s));
    } else {
        rc = wnla_fast_setup(c, w, n, rounds);
        if (rc != BPPP_OK) return rc;
        k_wnla_begin<<<blocks, BPPP_BLOCK, 0, s>>>(w);
        if (w.atab) k_wnla_tables<<<blocks, BPPP_BLOCK, 0, s>>>(w);
        {
            const int grp = wnla_round_group(c, w, blocks);
            for (int k = 1; k <= (int)rounds; k++) {
                if (grp > 1) k_wnla_round_grp<<<(unsigned)(((size_t)grp * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(w, k, grp);
                else k_wnla_round<<<blocks, BPPP_BLOCK, 0, s>>>(w, k);
            }
        }
        k_wnla_final_scalars<<<blocks, BPPP_BLOCK, 0, s>>>(w);
        k_wnla_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 0);
        k_wnla_accept<<<blocks, BPPP_BLOCK, 0, s>>>(w);
        if (w.tio.states_out) k_generic_export_states<<<blocks, BPPP_BLOCK, 0, s>>>(w);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(accept, d + o_acc, n, hipMemcpyDeviceToHost, s));
        if (w.tio.states_out) HIP_TRY(hipMemcpyAsync(tx->states_out, d + o_to, n * 203, hipMemcpyDeviceToHost, s));
    }
    if (status) HIP_TRY(hipMemcpyAsync(status, d + o_st, n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return BPPP_OK;
}

int bppp_wnla_commit_batch(bppp_ctx* c, size_t n, const uint8_t* cvec, const uint8_t* mu, const uint8_t* l, size_t nl,
                           const uint8_t* nvec, size_t nn, uint8_t* out, int32_t* status) {
    CtxLock lock_(c);
    if (!c || !cvec || !mu || (!l && nl) || (!nvec && nn) || !out) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    return wnla_run(c, true, nullptr, 0, n, nullptr, cvec, nullptr, mu, 0, nullptr, nullptr, l, nl, nvec, nn, out, nullptr, status);
}

int bppp_wnla_verify_batch(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                           const uint8_t* cvec, const uint8_t* rho, const uint8_t* mu, size_t rounds, const uint8_t* proof_r,
                           const uint8_t* proof_x, const uint8_t* proof_l, size_t nl, const uint8_t* proof_n, size_t nn,
                           uint8_t* accept, int32_t* status) {
    CtxLock lock_(c);
    if (!c || (!label && label_len) || !commitments || !cvec || !rho || !mu || (rounds && (!proof_r || !proof_x)) || (!proof_l && nl) ||
        (!proof_n && nn) || !accept)
        return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    return wnla_run(c, false, label, label_len, n, commitments, cvec, rho, mu, rounds, proof_r, proof_x, proof_l, nl, proof_n, nn, nullptr,
                    accept, status);
}

int bppp_wnla_verify_batch_transcript(bppp_ctx* c, size_t n, const uint8_t* states, size_t n_states, const uint8_t* commitments,
                                      const uint8_t* cvec, const uint8_t* rho, const uint8_t* mu, size_t rounds, const uint8_t* proof_r,
                                      const uint8_t* proof_x, const uint8_t* proof_l, size_t nl, const uint8_t* proof_n, size_t nn,
                                      uint8_t* accept, int32_t* status, uint8_t* states_out) {
    CtxLock lock_(c);
    if (!c || !states || !commitments || !cvec || !rho || !mu || (rounds && (!proof_r || !proof_x)) || (!proof_l && nl) || (!proof_n && nn) ||
        !accept)
        return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HostTranscripts tx = {states, n_states, states_out};
    return wnla_run(c, false, nullptr, 0, n, commitments, cvec, rho, mu, rounds, proof_r, proof_x, proof_l, nl, proof_n, nn, nullptr, accept,
                    status, &tx);
}

// ---- generic ReciprocalRangeProofProtocol::verify (reciprocal.rs:98-107) on a context built by bppp_wnla_ctx_create over
//      g, g_vec || g_vec_, h_vec || h_vec_
// workspace bytes (beyond the caller's commitments / proofs / accept / status) of one reciprocal verify call
static size_t recip_verify_ws_bytes(const bppp_ctx* c, size_t n, size_t dim_nd, size_t dim_np, size_t rounds, bool rlc = false) {
    const size_t NB = (size_t)c->nbases, T = (size_t)1 << rounds, NH = (size_t)c->nh;
    size_t off = 0;
    auto take = [&](size_t bytes) { off = align16(off + bytes); };
    take(52 * n * 4); take((dim_nd + 6) * 8 * n * 4); take(5 * 16 * n * 4); take(30 * n * 4); take(30 * n * 4); take(dim_np * 8 * n * 4);
    take(n * 64); take(n * NH * 32); take(n * 32); take(n * 32); take((rounds ? rounds : 1) * 8 * n * 4); take(2 * T * 8 * n * 4);
    take(NB * 8 * n * 4);
    if (rlc) { take(30 * n * 4); take(NB * 8 * n * 4); take((n + BPPP_RLC_CHUNK - 1) / BPPP_RLC_CHUNK); take(((n + BPPP_RLC_CHUNK - 1) / BPPP_RLC_CHUNK + 4) * 4); }
    return off;
}
// the launch sequence, every buffer in device memory; d_ws holds recip_verify_ws_bytes()
static int recip_verify_device_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                    const uint8_t* d_com, const uint8_t* d_proofs, size_t rounds, size_t nl, size_t nn, uint8_t* d_acc,
                                    int32_t* d_st, uint8_t* d_ws, const TranscriptIo* dtio = nullptr, const uint8_t* rlc_seed = nullptr) {
    const size_t NB = (size_t)c->nbases, T = (size_t)1 << rounds, NH = (size_t)c->nh;
    const size_t proof_bytes = 64 * (5 + 2 * rounds) + 32 * (nl + nn);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align16(off + bytes); return o; };
    const size_t o_ts = take(52 * n * 4), o_sc0 = take((dim_nd + 6) * 8 * n * 4), o_pts = take(5 * 16 * n * 4), o_a = take(30 * n * 4),
                 o_pf = take(30 * n * 4), o_inv = take(dim_np * 8 * n * 4), o_wc = take(n * 64), o_wcv = take(n * NH * 32), o_rho = take(n * 32),
                 o_mu = take(n * 32), o_ys = take((rounds ? rounds : 1) * 8 * n * 4), o_tab = take(2 * T * 8 * n * 4), o_msc = take(NB * 8 * n * 4);
    const size_t nchunks = (n + BPPP_RLC_CHUNK - 1) / BPPP_RLC_CHUNK;
    const size_t o_rl = rlc_seed ? take(30 * n * 4) : 0, o_rs = rlc_seed ? take(NB * 8 * n * 4) : 0, o_rf = rlc_seed ? take(nchunks) : 0,
                 o_rli = rlc_seed ? take((nchunks + 4) * 4) : 0;
    uint8_t* d = d_ws;
    hipStream_t s = c->stream;
    RecipWs r;
    std::memset(&r, 0, sizeof r);
    r.N = n; r.nd = (int)dim_nd; r.np = (int)dim_np; r.rounds = (int)rounds; r.nl = (int)nl; r.nn = (int)nn;
    r.NG = c->ng; r.NH = c->nh; r.proof_bytes = proof_bytes;
    r.commitments = d_com; r.proofs = d_proofs; r.status = d_st; r.tstate = (u32*)(d + o_ts);
    r.sc0 = (u32*)(d + o_sc0); r.pts = (u32*)(d + o_pts); r.acc = (u32*)(d + o_a); r.pfix = (u32*)(d + o_pf); r.inv = (u32*)(d + o_inv);
    r.straus = c->d_straus;
    r.wn_commit = d + o_wc; r.wn_c = d + o_wcv; r.wn_rho = d + o_rho; r.wn_mu = d + o_mu;
    r.fb.table = c->d_table; r.fb.W = c->fb_w; r.fb.N = n;
    t_new(r.base, label, (u32)label_len);
    if (dtio) r.tio = *dtio;
    WnlaWs w;
    std::memset(&w, 0, sizeof w);
    w.N = n; w.ng = c->ng; w.nh = c->nh; w.rounds = (int)rounds; w.nl = (int)nl; w.nn = (int)nn;
    w.base = r.base;
    if (dtio) { w.tio = *dtio; w.divergent_positions = dtio->n_states != 1; }
    w.commitments = r.wn_commit; w.c = r.wn_c; w.rho = r.wn_rho; w.mu = r.wn_mu;
    w.proof_r = r.proofs + 256; w.proof_x = r.proofs + 256 + 64 * rounds; w.proof_l = r.proofs + 320 + 128 * rounds;
    w.proof_n = w.proof_l + 32 * nl;
    w.stride_r = w.stride_x = w.stride_l = w.stride_n = proof_bytes;
    w.transcript_preloaded = 1;
    w.accept = d_acc; w.status = r.status; w.tstate = r.tstate; w.acc = r.acc; w.pfix = r.pfix;
    w.ys = (u32*)(d + o_ys); w.tab = (u32*)(d + o_tab); w.msc = (u32*)(d + o_msc);
    w.straus = c->d_straus;
    w.fb = r.fb;
    const unsigned blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    const unsigned fb_blocks = (unsigned)((n * BPPP_FB_LANES + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    int rc;
#define GLAUNCH(id, ...)                                       \
    do {                                                       \
        rc = timed(c, id, s, [&]() { __VA_ARGS__; });          \
        if (rc != BPPP_OK) return rc;                          \
    } while (0)
    GLAUNCH(K_RECIP_PHASE1, k_recip_phase1<<<blocks, BPPP_BLOCK, 0, s>>>(r));
    GLAUNCH(K_RECIP_C0_FIXED, k_recip_c0_fixed<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(r));
    GLAUNCH(K_RECIP_C0_VAR, k_recip_c0_var<<<blocks, BPPP_BLOCK, 0, s>>>(r));
    GLAUNCH(K_RECIP_C0_FINISH, k_recip_c0_finish<<<blocks, BPPP_BLOCK, 0, s>>>(r));
    rc = wnla_fast_setup(c, w, n, rounds);
    if (rc != BPPP_OK) return rc;
    GLAUNCH(K_WNLA_BEGIN, k_wnla_begin<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    if (w.atab) GLAUNCH(K_WNLA_TABLES, k_wnla_tables<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    {
        const int grp = wnla_round_group(c, w, blocks);
        for (int k = 1; k <= (int)rounds; k++) {
            if (grp > 1) GLAUNCH(K_WNLA_ROUND, k_wnla_round_grp<<<(unsigned)(((size_t)grp * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(w, k, grp));
            else GLAUNCH(K_WNLA_ROUND, k_wnla_round<<<blocks, BPPP_BLOCK, 0, s>>>(w, k));
        }
    }
    GLAUNCH(K_WNLA_FINAL_SCALARS, k_wnla_final_scalars<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    if (!rlc_seed) {
        GLAUNCH(K_WNLA_MSM, k_wnla_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 0));
        GLAUNCH(K_WNLA_ACCEPT, k_wnla_accept<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    } else {
        // one MSM per chunk of 8 instances instead of one per instance; what does not pass is re-checked exactly (wnla_rlc_core.h)
        RlcWs rl;
        std::memset(&rl, 0, sizeof rl);
        for (int i = 0; i < 4; i++) {
            u64 v = 0;
            for (int k = 0; k < 8; k++) v |= (u64)rlc_seed[8 * i + k] << (8 * k);
            rl.seed[i] = v;
        }
        rl.lhs = (u32*)(d + o_rl); rl.sc = (u32*)(d + o_rs); rl.flag = d + o_rf;
        rl.list = (u32*)(d + o_rli); rl.count = (int*)(rl.list + nchunks + 1);
        const unsigned chunk_blocks = (unsigned)((nchunks * BPPP_RLC_CHUNK + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
        const unsigned check_blocks = (unsigned)(nchunks < 16384 ? nchunks : 16384);
        HIP_TRY(hipMemsetAsync(d_acc, 0, n, s));
        HIP_TRY(hipMemsetAsync(rl.count, 0, sizeof(int), s));
        GLAUNCH(K_WNLA_RLC_LHS, k_wnla_rlc_lhs<<<blocks, BPPP_BLOCK, 0, s>>>(w, rl));
        GLAUNCH(K_WNLA_RLC_CHUNK, k_wnla_rlc_chunk<<<chunk_blocks, BPPP_FB_BLOCK, 0, s>>>(w, rl));
        GLAUNCH(K_WNLA_RLC_CHECK, k_wnla_rlc_check<<<check_blocks, 64, 0, s>>>(w, rl));
        GLAUNCH(K_WNLA_MSM, k_wnla_msm_flagged<<<1024, 64, 0, s>>>(w, rl));
        GLAUNCH(K_WNLA_MSM, k_wnla_msm_flagged_dense<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, rl));
        GLAUNCH(K_WNLA_ACCEPT, k_wnla_accept_flagged<<<blocks, BPPP_BLOCK, 0, s>>>(w, rl));
    }
    if (w.tio.states_out) k_generic_export_states<<<blocks, BPPP_BLOCK, 0, s>>>(w);
#undef GLAUNCH
    HIP_TRY(hipGetLastError());
    return BPPP_OK;
}
static int recip_verify_check_args(const bppp_ctx* c, size_t dim_nd, size_t dim_np, size_t rounds, size_t nl, size_t nn) {
    if (dim_nd == 0 || dim_np == 0 || dim_nd > (size_t)c->ng || dim_nd + 10 > (size_t)c->nh || dim_np > dim_nd + 1 || rounds > 12 ||
        nl > 4096 || nn > 4096)
        return BPPP_ERR_INVALID_ARG;
    return BPPP_OK;
}
static int recip_verify_device_entry(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                     const void* d_commitments, const void* d_proofs, size_t rounds, size_t nl, size_t nn, void* d_accept,
                                     void* d_status, const uint8_t* rlc_seed) {
    if (!c || (!label && label_len) || !d_commitments || !d_proofs || !d_accept || !d_status) return BPPP_ERR_INVALID_ARG;
    int rc = recip_verify_check_args(c, dim_nd, dim_np, rounds, nl, nn);
    if (rc != BPPP_OK) return rc;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    rc = ensure_straus_capacity(c, n);
    if (rc != BPPP_OK) return rc;
    // persistent, grow-only workspace (the host-pointer entry point allocates per call instead)
    const size_t need = recip_verify_ws_bytes(c, n, dim_nd, dim_np, rounds, rlc_seed != nullptr);
    if (need > c->gws_bytes) {
        if (c->d_gws) { (void)hipFree(c->d_gws); c->d_gws = nullptr; c->gws_bytes = 0; }
        HIP_TRY(hipMalloc(&c->d_gws, need));
        c->gws_bytes = need;
    }
    return recip_verify_device_impl(c, label, label_len, n, dim_nd, dim_np, (const uint8_t*)d_commitments, (const uint8_t*)d_proofs, rounds, nl,
                                    nn, (uint8_t*)d_accept, (int32_t*)d_status, c->d_gws, nullptr, rlc_seed);
}
int bppp_reciprocal_verify_batch_device(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                        const void* d_commitments, const void* d_proofs, size_t rounds, size_t nl, size_t nn,
                                        void* d_accept, void* d_status) {
    CtxLock lock_(c);
    return recip_verify_device_entry(c, label, label_len, n, dim_nd, dim_np, d_commitments, d_proofs, rounds, nl, nn, d_accept, d_status, nullptr);
}
int bppp_reciprocal_verify_batch_rlc_device(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                            const void* d_commitments, const void* d_proofs, size_t rounds, size_t nl, size_t nn,
                                            void* d_accept, void* d_status, const uint8_t seed[32]) {
    CtxLock lock_(c);
    if (!seed) return BPPP_ERR_INVALID_ARG;
    return recip_verify_device_entry(c, label, label_len, n, dim_nd, dim_np, d_commitments, d_proofs, rounds, nl, nn, d_accept, d_status, seed);
}
static int recip_verify_host_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                  const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl, size_t nn, uint8_t* accept,
                                  int32_t* status, const HostTranscripts* tx, const uint8_t* rlc_seed = nullptr);
int bppp_reciprocal_verify_batch_rlc(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                     const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl, size_t nn, uint8_t* accept,
                                     int32_t* status, const uint8_t seed[32]) {
    CtxLock lock_(c);
    if (!seed) return BPPP_ERR_INVALID_ARG;
    return recip_verify_host_impl(c, label, label_len, n, dim_nd, dim_np, commitments, proofs, rounds, nl, nn, accept, status, nullptr, seed);
}
int bppp_reciprocal_verify_batch(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                 const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl, size_t nn, uint8_t* accept,
                                 int32_t* status) {
    CtxLock lock_(c);
    return recip_verify_host_impl(c, label, label_len, n, dim_nd, dim_np, commitments, proofs, rounds, nl, nn, accept, status, nullptr);
}
int bppp_reciprocal_verify_batch_transcript(bppp_ctx* c, size_t n, const uint8_t* states, size_t n_states, size_t dim_nd, size_t dim_np,
                                            const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl, size_t nn,
                                            uint8_t* accept, int32_t* status, uint8_t* states_out) {
    CtxLock lock_(c);
    if (!states) return BPPP_ERR_INVALID_ARG;
    HostTranscripts tx = {states, n_states, states_out};
    return recip_verify_host_impl(c, nullptr, 0, n, dim_nd, dim_np, commitments, proofs, rounds, nl, nn, accept, status, &tx);
}
static int recip_verify_host_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                  const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl, size_t nn, uint8_t* accept,
                                  int32_t* status, const HostTranscripts* tx, const uint8_t* rlc_seed) {
    if (!c || (!label && label_len) || !commitments || !proofs || !accept) return BPPP_ERR_INVALID_ARG;
    int rc = recip_verify_check_args(c, dim_nd, dim_np, rounds, nl, nn);
    if (rc != BPPP_OK) return rc;
    if (n == 0) return BPPP_OK;
    rc = check_host_transcripts(tx, n);
    if (rc != BPPP_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    rc = ensure_straus_capacity(c, n);
    if (rc != BPPP_OK) return rc;
    const size_t proof_bytes = 64 * (5 + 2 * rounds) + 32 * (nl + nn);
    const size_t o_com = 0, o_pr = align16(n * 64), o_acc = align16(o_pr + n * proof_bytes), o_st = align16(o_acc + n),
                 o_ti = align16(o_st + n * 4), o_to = align16(o_ti + (tx ? tx->n_states * 203 : 0)),
                 o_ws = align16(o_to + (tx && tx->states_out ? n * 203 : 0)),
                 total = o_ws + recip_verify_ws_bytes(c, n, dim_nd, dim_np, rounds, rlc_seed != nullptr);
    WnlaBlob blob;
    HIP_TRY(hipMalloc(&blob.d, total));
    uint8_t* d = blob.d;
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(d + o_com, commitments, n * 64, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + o_pr, proofs, n * proof_bytes, hipMemcpyHostToDevice, s));
    TranscriptIo dtio = {nullptr, 0, nullptr, 0};
    if (tx) {
        HIP_TRY(hipMemcpyAsync(d + o_ti, tx->states, tx->n_states * 203, hipMemcpyHostToDevice, s));
        dtio.states = d + o_ti; dtio.n_states = tx->n_states; dtio.states_out = tx->states_out ? d + o_to : nullptr;
    }
    rc = recip_verify_device_impl(c, label, label_len, n, dim_nd, dim_np, d + o_com, d + o_pr, rounds, nl, nn, d + o_acc, (int32_t*)(d + o_st),
                                  d + o_ws, tx ? &dtio : nullptr, rlc_seed);
    if (rc != BPPP_OK) return rc;
    if (dtio.states_out) HIP_TRY(hipMemcpyAsync(tx->states_out, d + o_to, n * 203, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(accept, d + o_acc, n, hipMemcpyDeviceToHost, s));
    if (status) HIP_TRY(hipMemcpyAsync(status, d + o_st, n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return BPPP_OK;
}

// ---------------------------------------------------------------- generic ArithmeticCircuit (circuit.rs:95-256)
struct bppp_circuit {
    CircuitDev cd;
    const int* d_part = nullptr;     // [3 nv + nm]: LO | LL | LR | NO (the prover places w_o with it)
    uint8_t* d_blob = nullptr;
    size_t blob_bytes = 0;
};
int bppp_circuit_create(bppp_ctx* c, bppp_circuit** out, const size_t dims[6], int f_l, int f_m, const uint8_t* W_m, const uint8_t* W_l,
                        const uint8_t* a_m, const uint8_t* a_l, const int32_t* part_lo, const int32_t* part_ll, const int32_t* part_lr,
                        const int32_t* part_no) {
    CtxLock lock_(c);
    if (!c || !out || !dims || !W_m || !W_l || !a_m || !a_l || !part_lo || !part_ll || !part_lr || !part_no) return BPPP_ERR_INVALID_ARG;
    const size_t nm = dims[0], no = dims[1], k = dims[2], nl = dims[3], nv = dims[4], nw = dims[5];
    // the reference's own definitions (circuit.rs:100-106) and what the context's generators can serve
    if (nm == 0 || nv == 0 || k == 0 || nl != nv * k || nw != 2 * nm + no || nm > (size_t)c->ng || nv + 9 > (size_t)c->nh || k > 1024 ||
        nm > 65536 || nv > 65536 || no > 65536)
        return BPPP_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    CircuitHostData hd;
    if (!circuit_host_build(hd, dims, W_m, W_l, a_m, a_l, part_lo, part_ll, part_lr, part_no)) return BPPP_ERR_INVALID_ARG;
    std::vector<int>&cpl = hd.cpl, &rl = hd.rl, &cpm = hd.cpm, &rm = hd.rm, &colmap = hd.colmap;
    std::vector<u32>&vl = hd.vl, &vm = hd.vm, &al = hd.al, &am = hd.am;
    bppp_circuit* q = new (std::nothrow) bppp_circuit();
    if (!q) return BPPP_ERR_INVALID_ARG;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align16(off + bytes + 16); return o; };
    const size_t o_cpl = take(cpl.size() * 4), o_rl = take(rl.size() * 4), o_vl = take(vl.size() * 4), o_cpm = take(cpm.size() * 4),
                 o_rm = take(rm.size() * 4), o_vm = take(vm.size() * 4), o_cm = take(colmap.size() * 4), o_al = take(al.size() * 4),
                 o_am = take(am.size() * 4);
    std::vector<int> parts(3 * nv + nm);
    for (size_t j = 0; j < nv; j++) { parts[j] = part_lo[j]; parts[nv + j] = part_ll[j]; parts[2 * nv + j] = part_lr[j]; }
    for (size_t j = 0; j < nm; j++) parts[3 * nv + j] = part_no[j];
    const size_t o_part = take(parts.size() * 4);
    hipError_t e = hipMalloc(&q->d_blob, off);
    if (e != hipSuccess) { delete q; g_last_error = std::string("hipMalloc: ") + hipGetErrorString(e); return BPPP_ERR_HIP; }
    q->blob_bytes = off;
    auto up = [&](size_t o, const void* src, size_t bytes) { return bytes ? hipMemcpy(q->d_blob + o, src, bytes, hipMemcpyHostToDevice) : hipSuccess; };
    if (up(o_cpl, cpl.data(), cpl.size() * 4) != hipSuccess || up(o_rl, rl.data(), rl.size() * 4) != hipSuccess ||
        up(o_vl, vl.data(), vl.size() * 4) != hipSuccess || up(o_cpm, cpm.data(), cpm.size() * 4) != hipSuccess ||
        up(o_rm, rm.data(), rm.size() * 4) != hipSuccess || up(o_vm, vm.data(), vm.size() * 4) != hipSuccess ||
        up(o_cm, colmap.data(), colmap.size() * 4) != hipSuccess || up(o_al, al.data(), al.size() * 4) != hipSuccess ||
        up(o_am, am.data(), am.size() * 4) != hipSuccess || up(o_part, parts.data(), parts.size() * 4) != hipSuccess) {
        (void)hipFree(q->d_blob);
        delete q;
        g_last_error = "circuit upload failed";
        return BPPP_ERR_HIP;
    }
    CircuitDev& cd = q->cd;
    cd.nm = (int)nm; cd.no = (int)no; cd.k = (int)k; cd.nl = (int)nl; cd.nv = (int)nv; cd.nw = (int)nw; cd.f_l = f_l ? 1 : 0; cd.f_m = f_m ? 1 : 0;
    cd.colptr_l = (const int*)(q->d_blob + o_cpl); cd.rows_l = (const int*)(q->d_blob + o_rl); cd.vals_l = (const u32*)(q->d_blob + o_vl);
    cd.colptr_m = (const int*)(q->d_blob + o_cpm); cd.rows_m = (const int*)(q->d_blob + o_rm); cd.vals_m = (const u32*)(q->d_blob + o_vm);
    cd.colmap = (const int*)(q->d_blob + o_cm); cd.a_l = (const u32*)(q->d_blob + o_al); cd.a_m = (const u32*)(q->d_blob + o_am);
    q->d_part = (const int*)(q->d_blob + o_part);
    *out = q;
    return BPPP_OK;
}
void bppp_circuit_destroy(bppp_circuit* q) {
    if (!q) return;
    if (q->d_blob) (void)hipFree(q->d_blob);
    delete q;
}
static int circuit_verify_host_impl(bppp_ctx* c, const bppp_circuit* q, const uint8_t* label, size_t label_len, size_t n,
                                    const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl, size_t nn, uint8_t* accept,
                                    int32_t* status, const HostTranscripts* tx);
int bppp_circuit_verify_batch(bppp_ctx* c, const bppp_circuit* q, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                              const uint8_t* proofs, size_t rounds, size_t nl, size_t nn, uint8_t* accept, int32_t* status) {
    CtxLock lock_(c);
    return circuit_verify_host_impl(c, q, label, label_len, n, commitments, proofs, rounds, nl, nn, accept, status, nullptr);
}
int bppp_circuit_verify_batch_transcript(bppp_ctx* c, const bppp_circuit* q, size_t n, const uint8_t* states, size_t n_states,
                                         const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl, size_t nn,
                                         uint8_t* accept, int32_t* status, uint8_t* states_out) {
    CtxLock lock_(c);
    if (!states) return BPPP_ERR_INVALID_ARG;
    HostTranscripts tx = {states, n_states, states_out};
    return circuit_verify_host_impl(c, q, nullptr, 0, n, commitments, proofs, rounds, nl, nn, accept, status, &tx);
}
static int circuit_verify_host_impl(bppp_ctx* c, const bppp_circuit* q, const uint8_t* label, size_t label_len, size_t n,
                                    const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl, size_t nn, uint8_t* accept,
                                    int32_t* status, const HostTranscripts* tx) {
    if (!c || !q || (!label && label_len) || !commitments || !proofs || !accept) return BPPP_ERR_INVALID_ARG;
    const CircuitDev& cd = q->cd;
    if (cd.nm > c->ng || cd.nv + 9 > c->nh || rounds > 12 || nl > 4096 || nn > 4096) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    int rc = check_host_transcripts(tx, n);
    if (rc != BPPP_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    rc = ensure_straus_capacity(c, n);
    if (rc != BPPP_OK) return rc;
    const size_t NB = (size_t)c->nbases, T = (size_t)1 << rounds, NH = (size_t)c->nh, k = (size_t)cd.k, nm = (size_t)cd.nm, nv = (size_t)cd.nv;
    const size_t proof_bytes = 64 * (4 + 2 * rounds) + 32 * (nl + nn);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align16(off + bytes); return o; };
    const size_t o_com = take(n * k * 64), o_pr = take(n * proof_bytes), o_acc = take(n), o_st = take(n * 4), o_ts = take(52 * n * 4),
                 o_lam = take((size_t)cd.nl * 8 * n * 4), o_muv = take(nm * 8 * n * 4), o_coef = take((3 * nm + 3 * nv) * 8 * n * 4),
                 o_sc0 = take((nm + 5 + k) * 8 * n * 4), o_pts = take((4 + k) * 16 * n * 4), o_a = take(30 * n * 4), o_pf = take(30 * n * 4),
                 o_wc = take(n * 64), o_wcv = take(n * NH * 32), o_rho = take(n * 32), o_mu = take(n * 32),
                 o_ys = take((rounds ? rounds : 1) * 8 * n * 4), o_tab = take(2 * T * 8 * n * 4), o_msc = take(NB * 8 * n * 4),
                 o_ti = take(tx ? tx->n_states * 203 : 0), o_to = take(tx && tx->states_out ? n * 203 : 0);
    WnlaBlob blob;
    HIP_TRY(hipMalloc(&blob.d, off + 16));
    uint8_t* d = blob.d;
    hipStream_t s = c->stream;
    if (tx) HIP_TRY(hipMemcpyAsync(d + o_ti, tx->states, tx->n_states * 203, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + o_com, commitments, n * k * 64, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + o_pr, proofs, n * proof_bytes, hipMemcpyHostToDevice, s));
    CircuitWs r;
    std::memset(&r, 0, sizeof r);
    r.N = n; r.cd = cd; r.rounds = (int)rounds; r.NG = c->ng; r.NH = c->nh; r.proof_bytes = proof_bytes;
    r.commitments = d + o_com; r.proofs = d + o_pr; r.status = (int32_t*)(d + o_st); r.tstate = (u32*)(d + o_ts);
    r.lamv = (u32*)(d + o_lam); r.muv = (u32*)(d + o_muv); r.coef = (u32*)(d + o_coef); r.sc0 = (u32*)(d + o_sc0); r.pts = (u32*)(d + o_pts);
    r.acc = (u32*)(d + o_a); r.pfix = (u32*)(d + o_pf);
    r.straus = c->d_straus;
    r.wn_commit = d + o_wc; r.wn_c = d + o_wcv; r.wn_rho = d + o_rho; r.wn_mu = d + o_mu;
    r.fb.table = c->d_table; r.fb.W = c->fb_w; r.fb.N = n;
    t_new(r.base, label, (u32)label_len);
    if (tx) { r.tio.states = d + o_ti; r.tio.n_states = tx->n_states; r.tio.states_out = tx->states_out ? d + o_to : nullptr; }
    WnlaWs w;
    std::memset(&w, 0, sizeof w);
    w.N = n; w.ng = c->ng; w.nh = c->nh; w.rounds = (int)rounds; w.nl = (int)nl; w.nn = (int)nn;
    w.base = r.base; w.tio = r.tio; w.divergent_positions = tx && tx->n_states != 1;
    w.commitments = r.wn_commit; w.c = r.wn_c; w.rho = r.wn_rho; w.mu = r.wn_mu;
    w.proof_r = r.proofs + 256; w.proof_x = r.proofs + 256 + 64 * rounds; w.proof_l = r.proofs + 256 + 128 * rounds;
    w.proof_n = w.proof_l + 32 * nl;
    w.stride_r = w.stride_x = w.stride_l = w.stride_n = proof_bytes;
    w.transcript_preloaded = 1;
    w.accept = d + o_acc; w.status = r.status; w.tstate = r.tstate; w.acc = r.acc; w.pfix = r.pfix;
    w.ys = (u32*)(d + o_ys); w.tab = (u32*)(d + o_tab); w.msc = (u32*)(d + o_msc);
    w.straus = c->d_straus;
    w.fb = r.fb;
    const unsigned blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    const unsigned fb_blocks = (unsigned)((n * BPPP_FB_LANES + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    k_circuit_phase1<<<blocks, BPPP_BLOCK, 0, s>>>(r);
    k_circuit_c0_fixed<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(r);
    k_circuit_c0_var<<<blocks, BPPP_BLOCK, 0, s>>>(r);
    k_circuit_c0_finish<<<blocks, BPPP_BLOCK, 0, s>>>(r);
    {
        int rcf = wnla_fast_setup(c, w, n, rounds);
        if (rcf != BPPP_OK) return rcf;
    }
    k_wnla_begin<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    if (w.atab) k_wnla_tables<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    {
        const int grp = wnla_round_group(c, w, blocks);
        for (int kk = 1; kk <= (int)rounds; kk++) {
            if (grp > 1) k_wnla_round_grp<<<(unsigned)(((size_t)grp * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(w, kk, grp);
            else k_wnla_round<<<blocks, BPPP_BLOCK, 0, s>>>(w, kk);
        }
    }
    k_wnla_final_scalars<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    k_wnla_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 0);
    k_wnla_accept<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    if (w.tio.states_out) k_generic_export_states<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(accept, d + o_acc, n, hipMemcpyDeviceToHost, s));
    if (w.tio.states_out) HIP_TRY(hipMemcpyAsync(tx->states_out, d + o_to, n * 203, hipMemcpyDeviceToHost, s));
    if (status) HIP_TRY(hipMemcpyAsync(status, d + o_st, n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return BPPP_OK;
}

// sum_j scalars[i][j] * generator[base_index[j]] for n independent rows, through the context's fixed-base tables: the crate's
// commit functions (circuit.rs:146-151, reciprocal.rs:88-95, u64_proof.rs:37-39) are instances of this with fixed index lists.
int bppp_msm_batch(bppp_ctx* c, size_t n, size_t nterms, const int32_t* base_index, const uint8_t* scalars, uint8_t* out, int32_t* status) {
    CtxLock lock_(c);
    if (!c || !base_index || !scalars || !out || nterms == 0 || nterms > 65536) return BPPP_ERR_INVALID_ARG;
    for (size_t j = 0; j < nterms; j++) {
        if (base_index[j] < 0 || base_index[j] >= c->nbases) return BPPP_ERR_INVALID_ARG;
        if (j && base_index[j] <= base_index[j - 1]) return BPPP_ERR_INVALID_ARG;      // strictly increasing
    }
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<int> runs;
    for (size_t j = 0; j < nterms;) {
        size_t e = j + 1;
        while (e < nterms && base_index[e] == base_index[e - 1] + 1) e++;
        runs.push_back((int)j); runs.push_back(base_index[j]); runs.push_back((int)(e - j));
        j = e;
    }
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align16(off + bytes); return o; };
    const size_t o_sc = take(n * nterms * 32), o_runs = take(runs.size() * 4), o_msc = take(nterms * 8 * n * 4), o_pf = take(30 * n * 4),
                 o_st = take(n * 4), o_out = take(n * 64);
    WnlaBlob blob;
    HIP_TRY(hipMalloc(&blob.d, off));
    uint8_t* d = blob.d;
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(d + o_sc, scalars, n * nterms * 32, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + o_runs, runs.data(), runs.size() * 4, hipMemcpyHostToDevice, s));
    MsmWs w;
    std::memset(&w, 0, sizeof w);
    w.N = n; w.nterms = (int)nterms; w.nruns = (int)(runs.size() / 3);
    w.scalars = d + o_sc; w.runs = (const int*)(d + o_runs); w.msc = (u32*)(d + o_msc); w.pfix = (u32*)(d + o_pf);
    w.status = (int32_t*)(d + o_st); w.out = d + o_out;
    w.fb.table = c->d_table; w.fb.W = c->fb_w; w.fb.N = n;
    const unsigned blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    const unsigned fb_blocks = (unsigned)((n * BPPP_FB_LANES + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    k_msm_scalars<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    k_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w);
    k_msm_store<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d + o_out, n * 64, hipMemcpyDeviceToHost, s));
    if (status) HIP_TRY(hipMemcpyAsync(status, d + o_st, n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return BPPP_OK;
}

// WeightNormLinearArgument::prove (wnla.rs:125-190) for n instances sharing the context's generators.
void bppp_wnla_proof_shape(size_t nl, size_t nn, size_t* rounds, size_t* nl_out, size_t* nn_out) {
    size_t r, a, b;
    wnla_proof_shape(nl, nn, r, a, b);
    if (rounds) *rounds = r;
    if (nl_out) *nl_out = a;
    if (nn_out) *nn_out = b;
}
static int wnla_prove_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, const HostTranscripts* tx, size_t n, const uint8_t* commitments,
                           const uint8_t* cvec, const uint8_t* rho, const uint8_t* mu, const uint8_t* l, size_t nl, const uint8_t* nvec, size_t nn,
                           uint8_t* proof_r, uint8_t* proof_x, uint8_t* proof_l, uint8_t* proof_n, int32_t* status) {
    if (!c || (!label && label_len) || !commitments || !cvec || !rho || !mu || (!l && nl) || (!nvec && nn) || nl > 65536 || nn > 65536)
        return BPPP_ERR_INVALID_ARG;
    size_t rounds, nl_f, nn_f;
    wnla_proof_shape(nl, nn, rounds, nl_f, nn_f);
    if ((rounds && (!proof_r || !proof_x)) || (nl_f && !proof_l) || (nn_f && !proof_n)) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t NB = (size_t)c->nbases, ng = (size_t)c->ng, nh = (size_t)c->nh;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align16(off + bytes); return o; };
    const size_t o_com = take(n * 64), o_c = take(n * nh * 32), o_rho = take(n * 32), o_mu = take(n * 32), o_l = take(n * nl * 32 + 16),
                 o_n = take(n * nn * 32 + 16), o_pr = take(n * rounds * 64 + 16), o_px = take(n * rounds * 64 + 16),
                 o_pl = take(n * nl_f * 32 + 16), o_pn = take(n * nn_f * 32 + 16), o_st = take(n * 4), o_ts = take(52 * n * 4),
                 o_vl = take((nl + 1) * 8 * n * 4), o_vn = take((nn + 1) * 8 * n * 4), o_vc = take(nh * 8 * n * 4), o_ch = take(nh * 8 * n * 4),
                 o_cg = take((ng + 1) * 8 * n * 4), o_prm = take(3 * 8 * n * 4), o_cm = take(16 * n * 4), o_msc = take(3 * NB * 8 * n * 4),
                 o_pb = take(3 * 30 * n * 4);
    WnlaBlob blob;
    HIP_TRY(hipMalloc(&blob.d, off));
    uint8_t* d = blob.d;
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(d + o_com, commitments, n * 64, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + o_c, cvec, n * nh * 32, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + o_rho, rho, n * 32, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + o_mu, mu, n * 32, hipMemcpyHostToDevice, s));
    if (nl) HIP_TRY(hipMemcpyAsync(d + o_l, l, n * nl * 32, hipMemcpyHostToDevice, s));
    if (nn) HIP_TRY(hipMemcpyAsync(d + o_n, nvec, n * nn * 32, hipMemcpyHostToDevice, s));
    WnlaProveWs w;
    std::memset(&w, 0, sizeof w);
    w.N = n; w.ng = c->ng; w.nh = c->nh; w.nl = (int)nl; w.nn = (int)nn; w.rounds = (int)rounds; w.nl_f = (int)nl_f; w.nn_f = (int)nn_f;
    w.commitments = d + o_com; w.c = d + o_c; w.rho = d + o_rho; w.mu = d + o_mu; w.l_in = d + o_l; w.n_in = d + o_n;
    w.proof_r = d + o_pr; w.proof_x = d + o_px; w.proof_l = d + o_pl; w.proof_n = d + o_pn;
    w.status = (int32_t*)(d + o_st); w.tstate = (u32*)(d + o_ts); w.vl = (u32*)(d + o_vl); w.vn = (u32*)(d + o_vn); w.vc = (u32*)(d + o_vc);
    w.ch = (u32*)(d + o_ch); w.cg = (u32*)(d + o_cg); w.prm = (u32*)(d + o_prm); w.com = (u32*)(d + o_cm); w.msc = (u32*)(d + o_msc);
    w.pbuf = (u32*)(d + o_pb);
    w.fb.table = c->d_table; w.fb.W = c->fb_w; w.fb.N = n;
    t_new(w.base, label, (u32)label_len);
    TxDev txd;
    int rc = txd.begin(tx, n, s, w.tio, w.divergent_positions);
    if (rc != BPPP_OK) return rc;
    w.tio.no_ops = rounds == 0;
    const unsigned blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    const unsigned fb_blocks = (unsigned)((n * BPPP_FB_LANES + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    k_wprove_init<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    for (int k = 0; k < (int)rounds; k++) {
        k_wprove_round_scalars<<<blocks, BPPP_BLOCK, 0, s>>>(w, k);
        k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 0);
        k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 1);
        k_wprove_round_fold<<<blocks, BPPP_BLOCK, 0, s>>>(w, k);
        if (k + 1 < (int)rounds) k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 2);
    }
    k_wprove_finish<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    HIP_TRY(hipGetLastError());
    if (rounds) {
        HIP_TRY(hipMemcpyAsync(proof_r, d + o_pr, n * rounds * 64, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(proof_x, d + o_px, n * rounds * 64, hipMemcpyDeviceToHost, s));
    }
    if (nl_f) HIP_TRY(hipMemcpyAsync(proof_l, d + o_pl, n * nl_f * 32, hipMemcpyDeviceToHost, s));
    if (nn_f) HIP_TRY(hipMemcpyAsync(proof_n, d + o_pn, n * nn_f * 32, hipMemcpyDeviceToHost, s));
    if (status) HIP_TRY(hipMemcpyAsync(status, d + o_st, n * 4, hipMemcpyDeviceToHost, s));
    rc = txd.finish(tx, w.tio, w.base, w.tstate, n, w.status, s);
    if (rc != BPPP_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    return BPPP_OK;
}
int bppp_wnla_prove_batch(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments, const uint8_t* cvec,
                          const uint8_t* rho, const uint8_t* mu, const uint8_t* l, size_t nl, const uint8_t* nvec, size_t nn,
                          uint8_t* proof_r, uint8_t* proof_x, uint8_t* proof_l, uint8_t* proof_n, int32_t* status) {
    CtxLock lock_(c);
    return wnla_prove_impl(c, label, label_len, nullptr, n, commitments, cvec, rho, mu, l, nl, nvec, nn, proof_r, proof_x, proof_l, proof_n, status);
}
int bppp_wnla_prove_batch_transcript(bppp_ctx* c, size_t n, const uint8_t* states, size_t n_states, const uint8_t* commitments,
                                     const uint8_t* cvec, const uint8_t* rho, const uint8_t* mu, const uint8_t* l, size_t nl, const uint8_t* nvec,
                                     size_t nn, uint8_t* proof_r, uint8_t* proof_x, uint8_t* proof_l, uint8_t* proof_n, int32_t* status,
                                     uint8_t* states_out) {
    CtxLock lock_(c);
    if (!states) return BPPP_ERR_INVALID_ARG;
    HostTranscripts tx = {states, n_states, states_out};
    return wnla_prove_impl(c, nullptr, 0, &tx, n, commitments, cvec, rho, mu, l, nl, nvec, nn, proof_r, proof_x, proof_l, proof_n, status);
}

// ArithmeticCircuit::prove (circuit.rs:260-556) for n instances of a shared circuit.
static int circuit_prove_impl(bppp_ctx* c, const bppp_circuit* q, const uint8_t* label, size_t label_len, const HostTranscripts* tx, size_t n,
                              const uint8_t* v_commitments, const uint8_t* v, const uint8_t* s_v, const uint8_t* w_l, const uint8_t* w_r,
                              const uint8_t* w_o, const uint8_t* rnd, uint8_t* proofs, int32_t* status) {
    if (!c || !q || (!label && label_len) || !v_commitments || !v || !s_v || !w_l || !w_r || !rnd || !proofs) return BPPP_ERR_INVALID_ARG;
    const CircuitDev& cd = q->cd;
    if ((cd.no && !w_o) || cd.nm > c->ng || cd.nv + 9 > c->nh) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t NB = (size_t)c->nbases, NG = (size_t)c->ng, NH = (size_t)c->nh, k = (size_t)cd.k, nm = (size_t)cd.nm, nv = (size_t)cd.nv,
                 no = (size_t)cd.no, nl = (size_t)cd.nl, n_rnd = 18 + nv + nm;
    size_t rounds, nl_f, nn_f;
    wnla_proof_shape(NH, NG, rounds, nl_f, nn_f);
    const size_t proof_bytes = 64 * (4 + 2 * rounds) + 32 * (nl_f + nn_f);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align16(off + bytes + 16); return o; };
    const size_t o_vp = take(n * k * 64), o_v = take(n * k * nv * 32), o_sv = take(n * k * 32), o_wl = take(n * nm * 32), o_wr = take(n * nm * 32),
                 o_wo = take(n * no * 32), o_rnd = take(n * n_rnd * 32), o_head = take(n * 256), o_st = take(n * 4), o_ts = take(52 * n * 4),
                 o_r9 = take(4 * 9 * 8 * n * 4), o_lv = take(6 * nv * 8 * n * 4), o_nv = take(4 * nm * 8 * n * 4), o_lam = take(nl * 8 * n * 4),
                 o_muv = take(nm * 8 * n * 4), o_coef = take((3 * nm + 3 * nv) * 8 * n * 4), o_misc = take(8 * 8 * n * 4),
                 o_msc = take(3 * NB * 8 * n * 4), o_pb = take(3 * 30 * n * 4), o_wc = take(n * 64), o_wcv = take(n * NH * 32),
                 o_rho = take(n * 32), o_mu = take(n * 32), o_wlv = take(n * NH * 32), o_wnv = take(n * NG * 32),
                 // WNLA prover state
                 o_pr = take(n * rounds * 64), o_px = take(n * rounds * 64), o_pl = take(n * nl_f * 32), o_pn = take(n * nn_f * 32),
                 o_vl = take((NH + 1) * 8 * n * 4), o_vn = take((NG + 1) * 8 * n * 4), o_vc = take(NH * 8 * n * 4), o_ch = take(NH * 8 * n * 4),
                 o_cg = take((NG + 1) * 8 * n * 4), o_prm = take(3 * 8 * n * 4), o_cm = take(16 * n * 4), o_proofs = take(n * proof_bytes);
    WnlaBlob blob;
    HIP_TRY(hipMalloc(&blob.d, off));
    uint8_t* d = blob.d;
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(d + o_vp, v_commitments, n * k * 64, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + o_v, v, n * k * nv * 32, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + o_sv, s_v, n * k * 32, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + o_wl, w_l, n * nm * 32, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + o_wr, w_r, n * nm * 32, hipMemcpyHostToDevice, s));
    if (no) HIP_TRY(hipMemcpyAsync(d + o_wo, w_o, n * no * 32, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + o_rnd, rnd, n * n_rnd * 32, hipMemcpyHostToDevice, s));
    CircuitProveWs p;
    std::memset(&p, 0, sizeof p);
    p.N = n; p.cd = cd; p.NG = c->ng; p.NH = c->nh; p.n_rnd = (int)n_rnd; p.rnd_stride = n_rnd * 32; p.part = q->d_part;
    p.v_pts = d + o_vp; p.v = d + o_v; p.s_v = d + o_sv; p.w_l = d + o_wl; p.w_r = d + o_wr; p.w_o = d + o_wo; p.rnd = d + o_rnd;
    p.proof_head = d + o_head; p.status = (int32_t*)(d + o_st); p.tstate = (u32*)(d + o_ts);
    u32* r9 = (u32*)(d + o_r9);
    p.ro = r9; p.rl = r9 + 72 * n; p.rr = r9 + 144 * n; p.rs = r9 + 216 * n;
    u32* lv = (u32*)(d + o_lv);
    p.lo = lv; p.ll = lv + nv * 8 * n; p.lr = lv + 2 * nv * 8 * n; p.ls = lv + 3 * nv * 8 * n; p.v1 = lv + 4 * nv * 8 * n; p.cl0 = lv + 5 * nv * 8 * n;
    u32* nvv = (u32*)(d + o_nv);
    p.no = nvv; p.nl = nvv + nm * 8 * n; p.nr = nvv + 2 * nm * 8 * n; p.ns = nvv + 3 * nm * 8 * n;
    p.lamv = (u32*)(d + o_lam); p.muv = (u32*)(d + o_muv); p.coef = (u32*)(d + o_coef); p.misc = (u32*)(d + o_misc);
    p.msc = (u32*)(d + o_msc); p.pbuf = (u32*)(d + o_pb);
    p.wn_commit = d + o_wc; p.wn_c = d + o_wcv; p.wn_rho = d + o_rho; p.wn_mu = d + o_mu; p.wn_l = d + o_wlv; p.wn_n = d + o_wnv;
    p.fb.table = c->d_table; p.fb.W = c->fb_w; p.fb.N = n;
    t_new(p.base, label, (u32)label_len);
    TxDev txd;
    int rc = txd.begin(tx, n, s, p.tio, p.divergent_positions);
    if (rc != BPPP_OK) return rc;
    WnlaProveWs w;
    std::memset(&w, 0, sizeof w);
    w.N = n; w.ng = c->ng; w.nh = c->nh; w.nl = (int)NH; w.nn = (int)NG; w.rounds = (int)rounds; w.nl_f = (int)nl_f; w.nn_f = (int)nn_f;
    w.transcript_preloaded = 1;
    w.base = p.base; w.tio = p.tio; w.divergent_positions = p.divergent_positions;
    w.commitments = p.wn_commit; w.c = p.wn_c; w.rho = p.wn_rho; w.mu = p.wn_mu; w.l_in = p.wn_l; w.n_in = p.wn_n;
    w.proof_r = d + o_pr; w.proof_x = d + o_px; w.proof_l = d + o_pl; w.proof_n = d + o_pn;
    w.status = p.status; w.tstate = p.tstate; w.vl = (u32*)(d + o_vl); w.vn = (u32*)(d + o_vn); w.vc = (u32*)(d + o_vc);
    w.ch = (u32*)(d + o_ch); w.cg = (u32*)(d + o_cg); w.prm = (u32*)(d + o_prm); w.com = (u32*)(d + o_cm); w.msc = p.msc; w.pbuf = p.pbuf;
    w.fb = p.fb;
    const unsigned blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    const unsigned fb_blocks = (unsigned)((n * BPPP_FB_LANES + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    HIP_TRY(hipMemsetAsync(p.msc, 0, 3 * NB * 8 * n * 4, s));     // the sets are written sparsely (slot = base index)
    k_cprove_stage_a<<<blocks, BPPP_BLOCK, 0, s>>>(p);
    for (int set = 0; set < 3; set++) k_cprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(p, set, 0);
    k_cprove_stage_b<<<blocks, BPPP_BLOCK, 0, s>>>(p);
    k_cprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(p, 0, 0);
    k_cprove_stage_c<<<blocks, BPPP_BLOCK, 0, s>>>(p);
    k_cprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(p, 0, 1);
    k_cprove_stage_d<<<blocks, BPPP_BLOCK, 0, s>>>(p);
    k_wprove_init<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    for (int kk = 0; kk < (int)rounds; kk++) {
        k_wprove_round_scalars<<<blocks, BPPP_BLOCK, 0, s>>>(w, kk);
        k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 0);
        k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 1);
        k_wprove_round_fold<<<blocks, BPPP_BLOCK, 0, s>>>(w, kk);
        if (kk + 1 < (int)rounds) k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 2);
    }
    k_wprove_finish<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    HIP_TRY(hipGetLastError());
    // assemble the proofs on the host side of the copy: head | r | x | l | n per instance
    std::vector<uint8_t> head(n * 256), pr(n * rounds * 64), px(n * rounds * 64), pl(n * nl_f * 32), pn(n * nn_f * 32);
    std::vector<int32_t> st(n);
    HIP_TRY(hipMemcpyAsync(head.data(), d + o_head, head.size(), hipMemcpyDeviceToHost, s));
    if (rounds) {
        HIP_TRY(hipMemcpyAsync(pr.data(), d + o_pr, pr.size(), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(px.data(), d + o_px, px.size(), hipMemcpyDeviceToHost, s));
    }
    if (nl_f) HIP_TRY(hipMemcpyAsync(pl.data(), d + o_pl, pl.size(), hipMemcpyDeviceToHost, s));
    if (nn_f) HIP_TRY(hipMemcpyAsync(pn.data(), d + o_pn, pn.size(), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(st.data(), d + o_st, n * 4, hipMemcpyDeviceToHost, s));
    rc = txd.finish(tx, p.tio, p.base, p.tstate, n, p.status, s);
    if (rc != BPPP_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    for (size_t i = 0; i < n; i++) {
        uint8_t* o = proofs + i * proof_bytes;
        if (st[i] != 0) { std::memset(o, 0, proof_bytes); continue; }
        std::memcpy(o, &head[i * 256], 256);
        o += 256;
        std::memcpy(o, &pr[i * rounds * 64], rounds * 64); o += rounds * 64;
        std::memcpy(o, &px[i * rounds * 64], rounds * 64); o += rounds * 64;
        std::memcpy(o, &pl[i * nl_f * 32], nl_f * 32); o += nl_f * 32;
        std::memcpy(o, &pn[i * nn_f * 32], nn_f * 32);
    }
    if (status) std::memcpy(status, st.data(), n * 4);
    (void)o_proofs;
    return BPPP_OK;
}
int bppp_circuit_prove_batch(bppp_ctx* c, const bppp_circuit* q, const uint8_t* label, size_t label_len, size_t n, const uint8_t* v_commitments,
                             const uint8_t* v, const uint8_t* s_v, const uint8_t* w_l, const uint8_t* w_r, const uint8_t* w_o,
                             const uint8_t* rnd, uint8_t* proofs, int32_t* status) {
    CtxLock lock_(c);
    return circuit_prove_impl(c, q, label, label_len, nullptr, n, v_commitments, v, s_v, w_l, w_r, w_o, rnd, proofs, status);
}
int bppp_circuit_prove_batch_transcript(bppp_ctx* c, const bppp_circuit* q, size_t n, const uint8_t* states, size_t n_states,
                                        const uint8_t* v_commitments, const uint8_t* v, const uint8_t* s_v, const uint8_t* w_l, const uint8_t* w_r,
                                        const uint8_t* w_o, const uint8_t* rnd, uint8_t* proofs, int32_t* status, uint8_t* states_out) {
    CtxLock lock_(c);
    if (!states) return BPPP_ERR_INVALID_ARG;
    HostTranscripts tx = {states, n_states, states_out};
    return circuit_prove_impl(c, q, nullptr, 0, &tx, n, v_commitments, v, s_v, w_l, w_r, w_o, rnd, proofs, status);
}

// ReciprocalRangeProofProtocol::prove (reciprocal.rs:110-146) for runtime dim_nd / dim_np.
static int recip_prove_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, const HostTranscripts* tx, size_t n, size_t dim_nd, size_t dim_np,
                            const uint8_t* commitments, const uint8_t* x, const uint8_t* sblind, const uint8_t* digits, const uint8_t* m,
                            const uint8_t* rnd, uint8_t* proofs, int32_t* status) {
    if (!c || (!label && label_len) || !commitments || !x || !sblind || !digits || !m || !rnd || !proofs) return BPPP_ERR_INVALID_ARG;
    if (dim_nd == 0 || dim_np == 0 || dim_nd > (size_t)c->ng || dim_nd + 10 > (size_t)c->nh || dim_np > dim_nd + 1 || dim_nd > 4096)
        return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    RecipPattern P;
    recip_pattern_build(P, dim_nd, dim_np);
    const size_t NB = (size_t)c->nbases, NG = (size_t)c->ng, NH = (size_t)c->nh, nd = dim_nd, np = dim_np, nm = nd, nv = nd + 1, nl = nv,
                 n_rnd = 20 + 2 * nd;
    size_t rounds, nl_f, nn_f;
    wnla_proof_shape(NH, NG, rounds, nl_f, nn_f);
    const size_t proof_bytes = 64 * (5 + 2 * rounds) + 32 * (nl_f + nn_f);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align16(off + bytes + 16); return o; };
    const CircuitHostData& hd = P.hd;
    // circuit pattern
    const size_t o_cpl = take(hd.cpl.size() * 4), o_rl = take(hd.rl.size() * 4), o_vl = take(hd.vl.size() * 4), o_cpm = take(hd.cpm.size() * 4),
                 o_rm = take(hd.rm.size() * 4), o_vm = take(hd.vm.size() * 4), o_cmp = take(hd.colmap.size() * 4), o_al = take(hd.al.size() * 4),
                 o_am = take(hd.am.size() * 4), o_il = take(P.inst_l.size() * 4), o_im = take(P.inst_m.size() * 4), o_part = take(P.parts.size() * 4);
    // inputs
    const size_t o_com = take(n * 64), o_x = take(n * 32), o_s = take(n * 32), o_dig = take(n * nd * 32), o_m = take(n * np * 32),
                 o_rnd = take(n * n_rnd * 32);
    // reciprocal stage
    const size_t o_st = take(n * 4), o_ts = take(52 * n * 4), o_inst = take((1 + np) * 8 * n * 4), o_scr = take((nd + np) * 8 * n * 4),
                 o_cpv = take(n * nv * 32), o_cpsv = take(n * 32), o_cpwr = take(n * nm * 32), o_cpvp = take(n * 64), o_prr = take(n * 64);
    // circuit prover
    const size_t o_head = take(n * 256), o_r9 = take(4 * 9 * 8 * n * 4), o_lv = take(6 * nv * 8 * n * 4), o_nv = take(4 * nm * 8 * n * 4),
                 o_lam = take(nl * 8 * n * 4), o_muv = take(nm * 8 * n * 4), o_coef = take((3 * nm + 3 * nv) * 8 * n * 4), o_misc = take(8 * 8 * n * 4),
                 o_msc = take(3 * NB * 8 * n * 4), o_pb = take(3 * 30 * n * 4), o_wc = take(n * 64), o_wcv = take(n * NH * 32),
                 o_rho = take(n * 32), o_mu = take(n * 32), o_wlv = take(n * NH * 32), o_wnv = take(n * NG * 32);
    // WNLA prover
    const size_t o_pr = take(n * rounds * 64), o_px = take(n * rounds * 64), o_pl = take(n * nl_f * 32), o_pn = take(n * nn_f * 32),
                 o_vl2 = take((NH + 1) * 8 * n * 4), o_vn2 = take((NG + 1) * 8 * n * 4), o_vc = take(NH * 8 * n * 4), o_ch = take(NH * 8 * n * 4),
                 o_cg = take((NG + 1) * 8 * n * 4), o_prm = take(3 * 8 * n * 4), o_cm = take(16 * n * 4);
    WnlaBlob blob;
    HIP_TRY(hipMalloc(&blob.d, off));
    uint8_t* d = blob.d;
    hipStream_t s = c->stream;
    auto up = [&](size_t o, const void* src, size_t bytes) { return bytes ? hipMemcpyAsync(d + o, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess; };
    HIP_TRY(up(o_cpl, hd.cpl.data(), hd.cpl.size() * 4)); HIP_TRY(up(o_rl, hd.rl.data(), hd.rl.size() * 4)); HIP_TRY(up(o_vl, hd.vl.data(), hd.vl.size() * 4));
    HIP_TRY(up(o_cpm, hd.cpm.data(), hd.cpm.size() * 4)); HIP_TRY(up(o_rm, hd.rm.data(), hd.rm.size() * 4)); HIP_TRY(up(o_vm, hd.vm.data(), hd.vm.size() * 4));
    HIP_TRY(up(o_cmp, hd.colmap.data(), hd.colmap.size() * 4)); HIP_TRY(up(o_al, hd.al.data(), hd.al.size() * 4)); HIP_TRY(up(o_am, hd.am.data(), hd.am.size() * 4));
    HIP_TRY(up(o_il, P.inst_l.data(), P.inst_l.size() * 4)); HIP_TRY(up(o_im, P.inst_m.data(), P.inst_m.size() * 4));
    HIP_TRY(up(o_part, P.parts.data(), P.parts.size() * 4));
    HIP_TRY(up(o_com, commitments, n * 64)); HIP_TRY(up(o_x, x, n * 32)); HIP_TRY(up(o_s, sblind, n * 32));
    HIP_TRY(up(o_dig, digits, n * nd * 32)); HIP_TRY(up(o_m, m, n * np * 32)); HIP_TRY(up(o_rnd, rnd, n * n_rnd * 32));
    HIP_TRY(hipStreamSynchronize(s));                      // the pattern vectors live on this stack frame
    RecipProveWs r;
    std::memset(&r, 0, sizeof r);
    r.N = n; r.nd = (int)nd; r.np = (int)np; r.NG = c->ng; r.NH = c->nh; r.n_rnd = (int)n_rnd;
    r.commitments = d + o_com; r.x = d + o_x; r.s = d + o_s; r.digits = d + o_dig; r.m = d + o_m; r.rnd = d + o_rnd;
    r.status = (int32_t*)(d + o_st); r.tstate = (u32*)(d + o_ts); r.inst_vals = (u32*)(d + o_inst); r.scr = (u32*)(d + o_scr);
    r.msc = (u32*)(d + o_msc); r.pbuf = (u32*)(d + o_pb);
    r.cp_v = d + o_cpv; r.cp_sv = d + o_cpsv; r.cp_wr = d + o_cpwr; r.cp_vpts = d + o_cpvp; r.proof_r = d + o_prr;
    r.fb.table = c->d_table; r.fb.W = c->fb_w; r.fb.N = n;
    t_new(r.base, label, (u32)label_len);
    TxDev txd;
    int rc = txd.begin(tx, n, s, r.tio, r.divergent_positions);
    if (rc != BPPP_OK) return rc;
    CircuitProveWs p;
    std::memset(&p, 0, sizeof p);
    p.base = r.base; p.tio = r.tio; p.divergent_positions = r.divergent_positions;
    CircuitDev& cd = p.cd;
    cd.nm = (int)nm; cd.no = (int)np; cd.k = 1; cd.nl = (int)nl; cd.nv = (int)nv; cd.nw = (int)P.dims[5]; cd.f_l = 1; cd.f_m = 0;
    cd.colptr_l = (const int*)(d + o_cpl); cd.rows_l = (const int*)(d + o_rl); cd.vals_l = (const u32*)(d + o_vl);
    cd.colptr_m = (const int*)(d + o_cpm); cd.rows_m = (const int*)(d + o_rm); cd.vals_m = (const u32*)(d + o_vm);
    cd.colmap = (const int*)(d + o_cmp); cd.a_l = (const u32*)(d + o_al); cd.a_m = (const u32*)(d + o_am);
    cd.inst_l = (const int*)(d + o_il); cd.inst_m = (const int*)(d + o_im); cd.inst_vals = r.inst_vals;
    p.N = n; p.NG = c->ng; p.NH = c->nh; p.n_rnd = (int)(18 + nv + nm); p.rnd_stride = n_rnd * 32; p.part = (const int*)(d + o_part);
    p.transcript_preloaded = 1;
    p.v_pts = r.cp_vpts; p.v = r.cp_v; p.s_v = r.cp_sv; p.w_l = r.digits; p.w_r = r.cp_wr; p.w_o = r.m; p.rnd = r.rnd + 32;
    p.proof_head = d + o_head; p.status = r.status; p.tstate = r.tstate;
    u32* r9 = (u32*)(d + o_r9);
    p.ro = r9; p.rl = r9 + 72 * n; p.rr = r9 + 144 * n; p.rs = r9 + 216 * n;
    u32* lv = (u32*)(d + o_lv);
    p.lo = lv; p.ll = lv + nv * 8 * n; p.lr = lv + 2 * nv * 8 * n; p.ls = lv + 3 * nv * 8 * n; p.v1 = lv + 4 * nv * 8 * n; p.cl0 = lv + 5 * nv * 8 * n;
    u32* nvv = (u32*)(d + o_nv);
    p.no = nvv; p.nl = nvv + nm * 8 * n; p.nr = nvv + 2 * nm * 8 * n; p.ns = nvv + 3 * nm * 8 * n;
    p.lamv = (u32*)(d + o_lam); p.muv = (u32*)(d + o_muv); p.coef = (u32*)(d + o_coef); p.misc = (u32*)(d + o_misc);
    p.msc = r.msc; p.pbuf = r.pbuf;
    p.wn_commit = d + o_wc; p.wn_c = d + o_wcv; p.wn_rho = d + o_rho; p.wn_mu = d + o_mu; p.wn_l = d + o_wlv; p.wn_n = d + o_wnv;
    p.fb = r.fb;
    WnlaProveWs w;
    std::memset(&w, 0, sizeof w);
    w.N = n; w.ng = c->ng; w.nh = c->nh; w.nl = (int)NH; w.nn = (int)NG; w.rounds = (int)rounds; w.nl_f = (int)nl_f; w.nn_f = (int)nn_f;
    w.transcript_preloaded = 1;
    w.base = r.base; w.tio = r.tio; w.divergent_positions = r.divergent_positions;
    w.commitments = p.wn_commit; w.c = p.wn_c; w.rho = p.wn_rho; w.mu = p.wn_mu; w.l_in = p.wn_l; w.n_in = p.wn_n;
    w.proof_r = d + o_pr; w.proof_x = d + o_px; w.proof_l = d + o_pl; w.proof_n = d + o_pn;
    w.status = r.status; w.tstate = r.tstate; w.vl = (u32*)(d + o_vl2); w.vn = (u32*)(d + o_vn2); w.vc = (u32*)(d + o_vc);
    w.ch = (u32*)(d + o_ch); w.cg = (u32*)(d + o_cg); w.prm = (u32*)(d + o_prm); w.com = (u32*)(d + o_cm); w.msc = p.msc; w.pbuf = p.pbuf;
    w.fb = p.fb;
    const unsigned blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    const unsigned fb_blocks = (unsigned)((n * BPPP_FB_LANES + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    HIP_TRY(hipMemsetAsync(p.msc, 0, 3 * NB * 8 * n * 4, s));
    k_rprove_stage_r1<<<blocks, BPPP_BLOCK, 0, s>>>(r);
    k_rprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(r);
    k_rprove_stage_r2<<<blocks, BPPP_BLOCK, 0, s>>>(r);
    HIP_TRY(hipMemsetAsync(p.msc, 0, 3 * NB * 8 * n * 4, s));
    k_cprove_stage_a<<<blocks, BPPP_BLOCK, 0, s>>>(p);
    for (int set = 0; set < 3; set++) k_cprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(p, set, 0);
    k_cprove_stage_b<<<blocks, BPPP_BLOCK, 0, s>>>(p);
    k_cprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(p, 0, 0);
    k_cprove_stage_c<<<blocks, BPPP_BLOCK, 0, s>>>(p);
    k_cprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(p, 0, 1);
    k_cprove_stage_d<<<blocks, BPPP_BLOCK, 0, s>>>(p);
    k_wprove_init<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    for (int kk = 0; kk < (int)rounds; kk++) {
        k_wprove_round_scalars<<<blocks, BPPP_BLOCK, 0, s>>>(w, kk);
        k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 0);
        k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 1);
        k_wprove_round_fold<<<blocks, BPPP_BLOCK, 0, s>>>(w, kk);
        if (kk + 1 < (int)rounds) k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 2);
    }
    k_wprove_finish<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    HIP_TRY(hipGetLastError());
    std::vector<uint8_t> head(n * 256), prr(n * 64), pr(n * rounds * 64 + 1), px(n * rounds * 64 + 1), pl(n * nl_f * 32 + 1), pn(n * nn_f * 32 + 1);
    std::vector<int32_t> st(n);
    HIP_TRY(hipMemcpyAsync(head.data(), d + o_head, n * 256, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(prr.data(), d + o_prr, n * 64, hipMemcpyDeviceToHost, s));
    if (rounds) {
        HIP_TRY(hipMemcpyAsync(pr.data(), d + o_pr, n * rounds * 64, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(px.data(), d + o_px, n * rounds * 64, hipMemcpyDeviceToHost, s));
    }
    if (nl_f) HIP_TRY(hipMemcpyAsync(pl.data(), d + o_pl, n * nl_f * 32, hipMemcpyDeviceToHost, s));
    if (nn_f) HIP_TRY(hipMemcpyAsync(pn.data(), d + o_pn, n * nn_f * 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(st.data(), d + o_st, n * 4, hipMemcpyDeviceToHost, s));
    rc = txd.finish(tx, r.tio, r.base, r.tstate, n, r.status, s);
    if (rc != BPPP_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    for (size_t i = 0; i < n; i++) {
        uint8_t* o = proofs + i * proof_bytes;
        if (st[i] != 0) { std::memset(o, 0, proof_bytes); continue; }
        std::memcpy(o, &head[i * 256], 256); o += 256;
        std::memcpy(o, &pr[i * rounds * 64], rounds * 64); o += rounds * 64;
        std::memcpy(o, &px[i * rounds * 64], rounds * 64); o += rounds * 64;
        std::memcpy(o, &prr[i * 64], 64); o += 64;
        std::memcpy(o, &pl[i * nl_f * 32], nl_f * 32); o += nl_f * 32;
        std::memcpy(o, &pn[i * nn_f * 32], nn_f * 32);
    }
    if (status) std::memcpy(status, st.data(), n * 4);
    return BPPP_OK;
}
int bppp_reciprocal_prove_batch(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                const uint8_t* commitments, const uint8_t* x, const uint8_t* sblind, const uint8_t* digits, const uint8_t* m,
                                const uint8_t* rnd, uint8_t* proofs, int32_t* status) {
    CtxLock lock_(c);
    return recip_prove_impl(c, label, label_len, nullptr, n, dim_nd, dim_np, commitments, x, sblind, digits, m, rnd, proofs, status);
}
int bppp_reciprocal_prove_batch_transcript(bppp_ctx* c, size_t n, const uint8_t* states, size_t n_states, size_t dim_nd, size_t dim_np,
                                           const uint8_t* commitments, const uint8_t* x, const uint8_t* sblind, const uint8_t* digits,
                                           const uint8_t* m, const uint8_t* rnd, uint8_t* proofs, int32_t* status, uint8_t* states_out) {
    CtxLock lock_(c);
    if (!states) return BPPP_ERR_INVALID_ARG;
    HostTranscripts tx = {states, n_states, states_out};
    return recip_prove_impl(c, nullptr, 0, &tx, n, dim_nd, dim_np, commitments, x, sblind, digits, m, rnd, proofs, status);
}

#if defined(BPPP_PHASE_TIMING)
// diagnostic builds only (not declared in include/bppp.h): copy the phase stamps out
BPPP_API int bppp_debug_read_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(bppp::g_bppp_stamps), sizeof(unsigned long long) * 1024 * 32) == hipSuccess ? 0 : -1;
}
#endif


// ---------------------------------------------------------------- setup: generator derivation, table artefact, shared tables
// SHAKE256 (FIPS 202) on the host, over the same Keccak-f[1600] the device transcripts use
static void shake256(const uint8_t* msg, size_t len, uint8_t* out, size_t outlen) {
    u64 st[25];
    for (int i = 0; i < 25; i++) st[i] = 0;
    const size_t R = 136;
    auto xor_byte = [&](size_t pos, uint8_t b) { st[pos >> 3] ^= (u64)b << (8 * (pos & 7)); };
    size_t pos = 0;
    for (size_t i = 0; i < len; i++) {
        xor_byte(pos++, msg[i]);
        if (pos == R) { keccak_f1600(st); pos = 0; }
    }
    xor_byte(pos, 0x1F);
    xor_byte(R - 1, 0x80);
    keccak_f1600(st);
    pos = 0;
    for (size_t i = 0; i < outlen; i++) {
        if (pos == R) { keccak_f1600(st); pos = 0; }
        out[i] = (uint8_t)(st[pos >> 3] >> (8 * (pos & 7)));
        pos++;
    }
}

// Nothing-up-my-sleeve generators (the step before the path: benches/range_proof.rs:18-20 draws random points; a deployment
// needs reproducible ones whose discrete logarithms nobody knows).  Try-and-increment: candidate x = SHAKE256(seed || "bppp-gen"
// || u32le(index) || u32le(counter)) read big-endian; accepted when x < p and x^3 + 7 is a square; y = the EVEN root.  Host only.
int bppp_derive_generators(const uint8_t* seed, size_t seed_len, size_t first_index, size_t n, uint8_t* out /* n x 64 */) {
    if ((!seed && seed_len) || !out || seed_len > 4096) return BPPP_ERR_INVALID_ARG;
    std::vector<uint8_t> msg(seed_len + 8 + 8);
    if (seed_len) std::memcpy(msg.data(), seed, seed_len);
    std::memcpy(msg.data() + seed_len, "bppp-gen", 8);
    for (size_t i = 0; i < n; i++) {
        const uint32_t idx = (uint32_t)(first_index + i);
        for (uint32_t ctr = 0;; ctr++) {
            for (int k = 0; k < 4; k++) { msg[seed_len + 8 + k] = (uint8_t)(idx >> (8 * k)); msg[seed_len + 12 + k] = (uint8_t)(ctr >> (8 * k)); }
            uint8_t xb[32];
            shake256(msg.data(), msg.size(), xb, 32);
            fe x, rhs, y, y2, seven;
            if (!fe_from_be(x, xb)) continue;
            fe_sqr(rhs, x);
            fe_mul(rhs, rhs, x);
            fe_set_u32(seven, 7);
            fe_add(rhs, rhs, seven);
            fe_sqrt_candidate(y, rhs);
            fe_sqr(y2, y);
            if (!fe_eq(y2, rhs)) continue;
            if (fe_is_odd(y)) { fe ny; fe_neg_m<1>(ny, y); y = ny; }
            fe_to_be(out + 64 * i, x);
            fe_to_be(out + 64 * i + 32, y);
            break;
        }
    }
    return BPPP_OK;
}

// ---- fixed-base tables as an artefact.  File = header | generators (nbases x 64 B, the device's decoded form re-encoded) | table.
struct TableFileHeader {
    char magic[8];             // "BPPPTAB1"
    uint32_t nbases, ng, nh, window_bits, nwin, reserved;
    uint64_t per_win, table_bytes;
};
int bppp_ctx_save_tables(bppp_ctx* c, const char* path) {
    CtxLock lock_(c);
    if (!c || !path) return BPPP_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    FILE* f = std::fopen(path, "wb");
    if (!f) { g_last_error = std::string("cannot open ") + path; return BPPP_ERR_INVALID_ARG; }
    TableFileHeader h;
    std::memset(&h, 0, sizeof h);
    std::memcpy(h.magic, "BPPPTAB1", 8);
    h.nbases = (uint32_t)c->nbases; h.ng = (uint32_t)c->ng; h.nh = (uint32_t)c->nh; h.window_bits = (uint32_t)c->fb_w;
    h.nwin = (uint32_t)fb_nwin(c->fb_w); h.per_win = fb_per_win(c->fb_w); h.table_bytes = c->table_bytes;
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
    std::vector<apt> gens(c->nbases);
    if (hipMemcpy(gens.data(), c->d_gens, gens.size() * sizeof(apt), hipMemcpyDeviceToHost) != hipSuccess) ok = false;
    ok = ok && std::fwrite(gens.data(), sizeof(apt), gens.size(), f) == gens.size();
    const size_t CH = (size_t)256 << 20;
    std::vector<uint8_t> buf(c->table_bytes < CH ? c->table_bytes : CH);
    for (size_t off = 0; ok && off < c->table_bytes; off += CH) {
        const size_t m = c->table_bytes - off < CH ? c->table_bytes - off : CH;
        if (hipMemcpy(buf.data(), (const uint8_t*)c->d_table + off, m, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
        ok = ok && std::fwrite(buf.data(), 1, m, f) == m;
    }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) { g_last_error = std::string("writing ") + path + " failed"; return BPPP_ERR_HIP; }
    return BPPP_OK;
}
static int ctx_alloc_common(bppp_ctx* c) {
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIP_TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    HIP_TRY(hipMalloc(&c->d_flags, sizeof(int)));
    c->serial_c0 = std::getenv("BPPP_SERIAL_C0") != nullptr;
    c->force_pairs = std::getenv("BPPP_FORCE_LANE_PAIRS") != nullptr;   // diagnostic: rounds on two lanes per proof at every batch size
    c->no_lane_groups = std::getenv("BPPP_NO_LANE_GROUPS") != nullptr;   // diagnostic: one lane per proof at every batch size
    c->no_small = std::getenv("BPPP_NO_SMALL_KERNELS") != nullptr;       // diagnostic: the 256-VGPR builds (two wavefronts per SIMD) at every batch size
    if (const char* e = std::getenv("BPPP_FB_ONE_LANE")) c->fb_one_lane_mode = e[0] == '0' ? 0 : 1;
    c->generic_slow_rounds = std::getenv("BPPP_GENERIC_SLOW_ROUNDS") != nullptr;   // diagnostic: projective tables + complete additions
    c->rlc_debug = std::getenv("BPPP_RLC_DEBUG") != nullptr;
    c->n_simds = device_simds(c->device);
    return BPPP_OK;
}
int bppp_ctx_create_from_tables(bppp_ctx** out, const char* path, int device) {
    if (!out || !path) return BPPP_ERR_INVALID_ARG;
    *out = nullptr;
    int rc = check_device(device);
    if (rc != BPPP_OK) return rc;
    FILE* f = std::fopen(path, "rb");
    if (!f) { g_last_error = std::string("cannot open ") + path; return BPPP_ERR_INVALID_ARG; }
    TableFileHeader h;
    bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, "BPPPTAB1", 8) == 0;
    const int W = (int)h.window_bits;
    ok = ok && (W == 4 || W == 8 || W == 10 || W == 16 || W == 20 || W == 22) && h.nbases == 1 + h.ng + h.nh && h.nbases <= 8193 &&
         h.nwin == (uint32_t)fb_nwin(W) && h.per_win == fb_per_win(W) && h.table_bytes == (uint64_t)h.nbases * h.nwin * h.per_win * sizeof(apt_packed);
    if (!ok) { std::fclose(f); g_last_error = std::string(path) + " is not a table file of this library"; return BPPP_ERR_INVALID_ARG; }
    HIP_TRY(hipSetDevice(device));
    bppp_ctx* c = new (std::nothrow) bppp_ctx();
    if (!c) { std::fclose(f); return BPPP_ERR_NOMEM; }
    c->device = device; c->fb_w = W; c->ng = (int)h.ng; c->nh = (int)h.nh; c->nbases = (int)h.nbases; c->table_bytes = h.table_bytes;
    auto fail = [&](int code) { std::fclose(f); bppp_ctx_destroy(c); return code; };
    rc = ctx_alloc_common(c);
    if (rc != BPPP_OK) return fail(rc);
    std::vector<apt> gens(h.nbases);
    if (std::fread(gens.data(), sizeof(apt), gens.size(), f) != gens.size()) return fail(BPPP_ERR_INVALID_ARG);
    if (hipMalloc(&c->d_gens, gens.size() * sizeof(apt)) != hipSuccess || hipMalloc(&c->d_table, c->table_bytes) != hipSuccess) return fail(BPPP_ERR_NOMEM);
    if (hipMemcpy(c->d_gens, gens.data(), gens.size() * sizeof(apt), hipMemcpyHostToDevice) != hipSuccess) return fail(BPPP_ERR_HIP);
    const size_t CH = (size_t)256 << 20;
    std::vector<uint8_t> buf(c->table_bytes < CH ? c->table_bytes : CH);
    for (size_t off = 0; off < c->table_bytes; off += CH) {
        const size_t m = c->table_bytes - off < CH ? c->table_bytes - off : CH;
        if (std::fread(buf.data(), 1, m, f) != m) return fail(BPPP_ERR_INVALID_ARG);
        if (hipMemcpy((uint8_t*)c->d_table + off, buf.data(), m, hipMemcpyHostToDevice) != hipSuccess) return fail(BPPP_ERR_HIP);
    }
    std::fclose(f);
    *out = c;
    return BPPP_OK;
}
// A second context on the SAME device that shares `parent`'s generators and fixed-base tables (read-only data) and owns its
// streams and workspaces: several host threads can then verify concurrently on one GPU without a second 21 GB table.  The
// parent must outlive its children.
int bppp_ctx_create_shared(bppp_ctx** out, bppp_ctx* parent) {
    if (!out || !parent) return BPPP_ERR_INVALID_ARG;
    *out = nullptr;
    HIP_TRY(hipSetDevice(parent->device));
    bppp_ctx* c = new (std::nothrow) bppp_ctx();
    if (!c) return BPPP_ERR_NOMEM;
    c->device = parent->device; c->fb_w = parent->fb_w; c->ng = parent->ng; c->nh = parent->nh; c->nbases = parent->nbases;
    c->d_gens = parent->d_gens; c->d_table = parent->d_table; c->table_bytes = 0; c->borrows_tables = true;
    int rc = ctx_alloc_common(c);
    if (rc != BPPP_OK) { bppp_ctx_destroy(c); return rc; }
    *out = c;
    return BPPP_OK;
}

// ---------------------------------------------------------------- one batch over the GPUs of a node
// RCCL through dlopen: no link-time dependency, and whichever librccl the process already holds (e.g. torch's) serves.
namespace {
typedef void* rcclComm;
struct RcclApi {
    void* handle = nullptr;
    int (*CommInitAll)(rcclComm*, int, const int*) = nullptr;
    int (*CommDestroy)(rcclComm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
const int kRcclInt32 = 2, kRcclSum = 0;   // ncclInt32, ncclSum (rccl.h)
bool rccl_load(RcclApi& a) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (a.handle) break;
    }
    if (!a.handle) { g_last_error = std::string("dlopen(librccl): ") + (dlerror() ? dlerror() : "not found"); return false; }
    a.CommInitAll = (int (*)(rcclComm*, int, const int*))dlsym(a.handle, "ncclCommInitAll");
    a.CommDestroy = (int (*)(rcclComm))dlsym(a.handle, "ncclCommDestroy");
    a.AllReduce = (int (*)(const void*, void*, size_t, int, int, rcclComm, hipStream_t))dlsym(a.handle, "ncclAllReduce");
    a.GroupStart = (int (*)())dlsym(a.handle, "ncclGroupStart");
    a.GroupEnd = (int (*)())dlsym(a.handle, "ncclGroupEnd");
    a.GetErrorString = (const char* (*)(int))dlsym(a.handle, "ncclGetErrorString");
    if (!a.CommInitAll || !a.CommDestroy || !a.AllReduce || !a.GroupStart || !a.GroupEnd) {
        g_last_error = "librccl lacks a required symbol";
        return false;
    }
    return true;
}
}  // namespace

struct bppp_group {
    std::vector<int> devices;
    std::vector<bppp_ctx*> ctx;
    std::vector<rcclComm> comm;          // empty when the accept-reduce needs no collective (one device)
    std::vector<int*> d_rej;             // per device: int32 reject counter of the host-buffer entry point
    RcclApi rccl;
};

void bppp_shard_range(size_t n_total, int rank, int world, size_t* lo, size_t* hi) {
    if (world <= 0 || rank < 0 || rank >= world) { if (lo) *lo = 0; if (hi) *hi = 0; return; }
    // n_total * rank may exceed 64 bits only for absurd sizes; use the quotient / remainder form
    const size_t q = n_total / (size_t)world, r = n_total % (size_t)world;
    auto at = [&](size_t k) { return q * k + (r * k) / (size_t)world; };   // floor(n_total * k / world)
    if (lo) *lo = at((size_t)rank);
    if (hi) *hi = at((size_t)rank + 1);
}

int bppp_group_size(const bppp_group* grp) { return grp ? (int)grp->devices.size() : 0; }
bppp_ctx* bppp_group_ctx(bppp_group* grp, int rank) { return (grp && rank >= 0 && rank < (int)grp->ctx.size()) ? grp->ctx[rank] : nullptr; }

void bppp_group_destroy(bppp_group* grp) {
    if (!grp) return;
    for (size_t r = 0; r < grp->comm.size(); r++)
        if (grp->comm[r]) { (void)hipSetDevice(grp->devices[r]); (void)grp->rccl.CommDestroy(grp->comm[r]); }
    for (size_t r = 0; r < grp->ctx.size(); r++) {
        if (r < grp->d_rej.size() && grp->d_rej[r]) { (void)hipSetDevice(grp->devices[r]); (void)hipFree(grp->d_rej[r]); }
        bppp_ctx_destroy(grp->ctx[r]);
    }
    delete grp;
}

int bppp_group_create(bppp_group** out, const uint8_t g[64], const uint8_t* g_vec, const uint8_t* h_vec, const int* devices, int n_devices,
                      int fb_window_bits) {
    if (!out || !g || !g_vec || !h_vec || !devices || n_devices <= 0 || n_devices > 64) return BPPP_ERR_INVALID_ARG;
    *out = nullptr;
    for (int i = 0; i < n_devices; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) return BPPP_ERR_INVALID_ARG;
    bppp_group* grp = new (std::nothrow) bppp_group();
    if (!grp) return BPPP_ERR_NOMEM;
    grp->devices.assign(devices, devices + n_devices);
    grp->ctx.assign(n_devices, nullptr);
    grp->d_rej.assign(n_devices, nullptr);
    // contexts (fixed-base tables) are built concurrently, one host thread per device
    std::vector<int> rcs(n_devices, BPPP_OK);
    std::vector<std::string> errs(n_devices);
    {
        std::vector<std::thread> th;
        for (int r = 0; r < n_devices; r++)
            th.emplace_back([&, r]() {
                rcs[r] = bppp_ctx_create(&grp->ctx[r], g, g_vec, h_vec, devices[r], fb_window_bits);
                if (rcs[r] == BPPP_OK && hipMalloc(&grp->d_rej[r], sizeof(int)) != hipSuccess) rcs[r] = BPPP_ERR_NOMEM;
                if (rcs[r] != BPPP_OK) errs[r] = g_last_error;
            });
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < n_devices; r++)
        if (rcs[r] != BPPP_OK) { g_last_error = errs[r]; int rc = rcs[r]; bppp_group_destroy(grp); return rc; }
    if (n_devices > 1 || std::getenv("BPPP_FORCE_RCCL")) {
        if (!rccl_load(grp->rccl)) { bppp_group_destroy(grp); return BPPP_ERR_RCCL; }
        grp->comm.assign(n_devices, nullptr);
        const int e = grp->rccl.CommInitAll(grp->comm.data(), n_devices, grp->devices.data());
        if (e != 0) {
            g_last_error = std::string("ncclCommInitAll: ") + (grp->rccl.GetErrorString ? grp->rccl.GetErrorString(e) : "failed");
            grp->comm.clear();
            bppp_group_destroy(grp);
            return BPPP_ERR_RCCL;
        }
    }
    *out = grp;
    return BPPP_OK;
}

// one device's part of a sharded call: verify the shard, then take part in the accept-reduce; everything on the context's stream
static int group_rank_verify(bppp_group* grp, int r, const uint8_t* label, size_t label_len, size_t n_r, const void* d_c, const void* d_p,
                             void* d_a, void* d_s, void* d_rej) {
    bppp_ctx* c = grp->ctx[r];
    HIP_TRY(hipSetDevice(grp->devices[r]));
    if (n_r) {
        int rc = verify_device_impl(c, label, label_len, n_r, d_c, d_p, d_a, d_s, nullptr, d_rej, nullptr);
        if (rc != BPPP_OK) return rc;
    } else {
        HIP_TRY(hipMemsetAsync(d_rej, 0, sizeof(int), c->stream));
    }
    if (!grp->comm.empty()) {
        const int e = grp->rccl.AllReduce(d_rej, d_rej, 1, kRcclInt32, kRcclSum, grp->comm[r], c->stream);
        if (e != 0) {
            g_last_error = std::string("ncclAllReduce: ") + (grp->rccl.GetErrorString ? grp->rccl.GetErrorString(e) : "failed");
            return BPPP_ERR_RCCL;
        }
    }
    return BPPP_OK;
}

int bppp_u64_verify_batch_sharded_device(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, const void* const* d_commitments,
                                         const void* const* d_proofs, void* const* d_accept, void* const* d_status,
                                         void* const* d_reject_count) {
    if (!grp || (!label && label_len) || !d_commitments || !d_proofs || !d_accept || !d_reject_count) return BPPP_ERR_INVALID_ARG;
    const int G = (int)grp->devices.size();
    for (int r = 0; r < G; r++) {
        size_t lo, hi;
        bppp_shard_range(n, r, G, &lo, &hi);
        if (!d_reject_count[r] || (hi > lo && (!d_commitments[r] || !d_proofs[r] || !d_accept[r]))) return BPPP_ERR_INVALID_ARG;
    }
    std::vector<int> rcs(G, BPPP_OK);
    std::vector<std::string> errs(G);
    std::vector<std::thread> th;
    for (int r = 0; r < G; r++)
        th.emplace_back([&, r]() {
            size_t lo, hi;
            bppp_shard_range(n, r, G, &lo, &hi);
            CtxLock lock_(grp->ctx[r]);
            rcs[r] = group_rank_verify(grp, r, label, label_len, hi - lo, d_commitments[r], d_proofs[r], d_accept[r],
                                       d_status ? d_status[r] : nullptr, d_reject_count[r]);
            if (rcs[r] == BPPP_OK) rcs[r] = bppp_ctx_synchronize(grp->ctx[r]);
            if (rcs[r] != BPPP_OK) errs[r] = g_last_error;
        });
    for (auto& t : th) t.join();
    for (int r = 0; r < G; r++)
        if (rcs[r] != BPPP_OK) { g_last_error = errs[r]; return rcs[r]; }
    return BPPP_OK;
}

int bppp_u64_verify_batch_sharded(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                                  const uint8_t* proofs, uint8_t* accept, int32_t* status, int32_t* reject_count) {
    if (!grp || (!label && label_len) || !commitments || !proofs || !accept) return BPPP_ERR_INVALID_ARG;
    if (reject_count) *reject_count = 0;
    if (n == 0) return BPPP_OK;
    const int G = (int)grp->devices.size();
    std::vector<int> rcs(G, BPPP_OK), counts(G, 0);
    std::vector<std::string> errs(G);
    std::vector<std::thread> th;
    for (int r = 0; r < G; r++)
        th.emplace_back([&, r]() {
            auto run = [&]() -> int {
                size_t lo, hi;
                bppp_shard_range(n, r, G, &lo, &hi);
                const size_t m = hi - lo;
                bppp_ctx* c = grp->ctx[r];
                CtxLock lock_(c);
                HIP_TRY(hipSetDevice(grp->devices[r]));
                // the shard goes through the context's persistent I/O staging, exactly as bppp_u64_verify_batch does
                const size_t o_c = 0, o_p = align16(o_c + m * 64), o_a = align16(o_p + m * (size_t)BPPP_U64_PROOF_BYTES), o_s = align16(o_a + m),
                             need = align16(o_s + m * sizeof(int32_t)) + 16;
                if (need > c->io_bytes) {
                    if (c->d_io) { (void)hipFree(c->d_io); c->d_io = nullptr; c->io_bytes = 0; }
                    HIP_TRY(hipMalloc(&c->d_io, need));
                    c->io_bytes = need;
                }
                uint8_t *d_c = c->d_io + o_c, *d_p = c->d_io + o_p, *d_a = c->d_io + o_a;
                int32_t* d_s = (int32_t*)(c->d_io + o_s);
                if (m) {
                    HIP_TRY(hipMemcpyAsync(d_c, commitments + lo * 64, m * 64, hipMemcpyHostToDevice, c->stream));
                    HIP_TRY(hipMemcpyAsync(d_p, proofs + lo * (size_t)BPPP_U64_PROOF_BYTES, m * (size_t)BPPP_U64_PROOF_BYTES, hipMemcpyHostToDevice,
                                           c->stream));
                }
                int rc = group_rank_verify(grp, r, label, label_len, m, d_c, d_p, d_a, d_s, grp->d_rej[r]);
                if (rc != BPPP_OK) return rc;
                if (m) {
                    HIP_TRY(hipMemcpyAsync(accept + lo, d_a, m, hipMemcpyDeviceToHost, c->stream));
                    if (status) HIP_TRY(hipMemcpyAsync(status + lo, d_s, m * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
                }
                HIP_TRY(hipMemcpyAsync(&counts[r], grp->d_rej[r], sizeof(int), hipMemcpyDeviceToHost, c->stream));
                return bppp_ctx_synchronize(c);
            };
            rcs[r] = run();
            if (rcs[r] != BPPP_OK) errs[r] = g_last_error;
        });
    for (auto& t : th) t.join();
    for (int r = 0; r < G; r++)
        if (rcs[r] != BPPP_OK) { g_last_error = errs[r]; return rcs[r]; }
    if (reject_count) {
        // with a communicator every device already holds the global count; without one (a single device) it is the local one
        int total = 0;
        if (!grp->comm.empty()) total = counts[0];
        else for (int r = 0; r < G; r++) total += counts[r];
        *reject_count = total;
    }
    return BPPP_OK;
}

}  // extern "C"
