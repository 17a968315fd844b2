// libbppp_hip.so, host side: contexts (generators -> fixed-base tables in HBM, streams, workspace bookkeeping), options, kernel timing, merlin on
// serialized states (host only), generator derivation and the table artefact.
#include "host.h"
#include "plan_core.h"
#if defined(BPPP_PHASE_TIMING)
unsigned long long* g_bppp_stamps_dev = nullptr;      // diagnostic builds: the phase-stamp rows (verify_ws.h: BPPP_STAMP), allocated at the first verify call
#endif

thread_local std::string g_last_error;

// Diagnostic switches (A/B measurements), read from the environment once per context, here and nowhere else.
static void read_diagnostics(bppp_ctx* c) {
    c->no_lane_groups = std::getenv("BPPP_NO_LANE_GROUPS") != nullptr;    // one lane per proof at every batch size
    if (const char* e = std::getenv("BPPP_GENERIC_LANE_GROUP")) {
        const int g = std::atoi(e);
        c->generic_lane_group = (g == 2 || g == 4) ? g : 0;
    }
    c->no_small = std::getenv("BPPP_NO_SMALL_KERNELS") != nullptr;        // the 256-VGPR builds (two wavefronts per SIMD) at every batch size
    c->no_split = std::getenv("BPPP_NO_SPLIT") != nullptr;                // no half-stream lanes / per-table lanes for calls of <= one proof per SIMD
    if (const char* e = std::getenv("BPPP_FB_ONE_LANE")) c->fb_one_lane_mode = e[0] == '0' ? 0 : 1;
    if (const char* e = std::getenv("BPPP_LANE_FORMS_MAX")) c->lane_forms_max = std::atol(e);
    if (const char* e = std::getenv("BPPP_NEXT_OVERLAP")) c->next_overlap = std::atoi(e);
    if (const char* e = std::getenv("BPPP_SCAL_PARTS_MAX")) c->scal_parts_max = std::atol(e);
    if (const char* e = std::getenv("BPPP_LANE4_MAX")) c->lane4_max = std::atol(e);
    if (const char* e = std::getenv("BPPP_TAIL_BESIDE")) c->tail_beside = std::atoi(e);
    if (const char* e = std::getenv("BPPP_TABLES_BESIDE")) c->tables_beside = std::atoi(e);
    if (const char* e = std::getenv("BPPP_SHARED_INV")) c->shared_inv = std::atoi(e);
    if (const char* e = std::getenv("BPPP_GENERIC_PARTS")) c->generic_parts = std::atoi(e);
    if (const char* e = std::getenv("BPPP_RECIP_BESIDE")) c->recip_beside = std::atoi(e) != 0;
    if (const char* e = std::getenv("BPPP_GENERIC_FB_WIDE_MAX")) c->generic_fb_wide_max = std::atol(e);
    if (const char* e = std::getenv("BPPP_RECIP_P1_GROUP")) { const int g = std::atoi(e); c->recip_p1_group = (g == 1 || g == 2 || g == 4 || g == 8) ? g : 0; }
    if (const char* e = std::getenv("BPPP_TWIN")) c->twin = std::atoi(e);
    if (const char* e = std::getenv("BPPP_TWIN_STREAMS")) c->twin_stream_kind = std::atoi(e);
    if (const char* e = std::getenv("BPPP_PACE")) c->pace = std::atoi(e);
    if (const char* e = std::getenv("BPPP_NEXT_MSM_MAX")) c->next_msm_max = std::atol(e);
    c->generic_u64_shape = std::getenv("BPPP_GENERIC_U64_SHAPE") != nullptr;     // reciprocal (16, 16) calls stay on the generic kernels
    c->generic_slow_rounds = std::getenv("BPPP_GENERIC_SLOW_ROUNDS") != nullptr;   // projective tables + complete additions
}

// fb_window_bits = 0: the FEWEST windows per scalar whose tables fit the HBM that is FREE when the context is created.  Windows come in
// two widths sized to the bit (fb_core.h: fb_wb): n windows are floor(258 / n)-bit windows with 258 mod n of them one bit wider, e.g.
//   11 windows: code  523 (5 x 24 + 6 x 23 bits), 4.29 GB per generator      14 windows: code 618, 168 MB per generator
//   12 windows: code  621 (6 x 22 + 6 x 21),      1.21 GB                    15 windows: code 317,  75 MB
//   13 windows: code 1119 (11 x 20 + 2 x 19),     403 MB                     16 windows: code 216,  38 MB   ...  32 windows: code 208, 278 KB
// (a uniform table of the same count always takes more: 22-bit windows 1.6 GB for 12 additions, 18-bit windows 126 MB for 15).  A table
// fits when it takes at most 76 % of the free memory, leaves 50 GB of it (or half, if less is free) for workspaces, and its build scratch
// fits beside it.  On an otherwise empty MI355X: 11 windows for the u64 protocol's 49 generators (210 GB; 726 table additions per proof),
// 14 windows for the 769 generators of BASELINE configs[4]'s shape (129 GB).
static int two_width_code_for(int nwin) { const int W = 258 / nwin; return W + 100 * (258 - W * nwin); }
static double table_bytes_for(int nbases, int W) { return (double)nbases * (double)fb_per_base(W) * sizeof(apt_packed); }
// bases per build pass: scratch (x, y, z, prefix: 160 B per entry) within `budget` bytes (at most ~32 GB), one base at least
static size_t build_group_for(int nbases, int W, size_t budget = (size_t)32 << 30) {
    const size_t per_base = fb_per_base(W);
    if (budget > ((size_t)32 << 30)) budget = (size_t)32 << 30;
    size_t group = budget / (per_base * 4 * sizeof(fe));
    if (group < 1) group = 1;
    if (group > (size_t)nbases) group = (size_t)nbases;
    return group;
}
static bool tables_fit(double tb, double scratch, size_t free_bytes) {
    const double fr = (double)free_bytes, room = fr * 0.5 < 50e9 ? fr * 0.5 : 50e9;
    return tb <= 0.76 * fr && tb + scratch <= 0.92 * fr && fr - tb >= room;
}
static bool window_fits(int nbases, int W, size_t free_bytes) {
    const double tb = table_bytes_for(nbases, W);
    const double scratch = (double)fb_per_base(W) * 4 * sizeof(fe);          // of a pass over one generator (the build takes more when there is room)
    return tables_fit(tb, scratch, free_bytes);
}
// Build the fixed-base tables of the generators first .. first + nb - 1 of c->d_gens at window width W into a fresh allocation.
// On failure nothing stays allocated.
static int build_table_range(bppp_ctx* c, int W, int first, int nb_total, apt_packed** out, size_t* out_bytes) {
    const int nwin = fb_nwin(W);
    const size_t per_base = fb_per_base(W);
    const size_t bytes = (size_t)nb_total * per_base * sizeof(apt_packed);
    apt_packed* d_table = nullptr;
    fe* d_tmp = nullptr;
    auto release = [&]() { if (d_tmp) (void)hipFree(d_tmp); if (d_table) (void)hipFree(d_table); };
#define HIP_TRY_T(expr)                                                             \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_);       \
            (void)hipGetLastError();                                                \
            release();                                                              \
            return e_ == hipErrorOutOfMemory ? BPPP_ERR_NOMEM : BPPP_ERR_HIP;       \
        }                                                                           \
    } while (0)
    if (c->inject_alloc_fault > 0 && --c->inject_alloc_fault == 0) { g_last_error = "injected allocation failure (tables)"; return BPPP_ERR_NOMEM; }
    HIP_TRY_T(hipMalloc(&d_table, bytes));
    size_t free_now = 0, total_now = 0;           // the build's scratch: up to 32 GB of what is free beside the table
    if (hipMemGetInfo(&free_now, &total_now) != hipSuccess) { (void)hipGetLastError(); free_now = (size_t)40 << 30; }
    const size_t group = build_group_for(nb_total, W, (size_t)(0.8 * (double)free_now));
    const size_t gentries = group * per_base;
    HIP_TRY_T(hipMalloc(&d_tmp, gentries * 4 * sizeof(fe)));
    for (size_t b0 = 0; b0 < (size_t)nb_total; b0 += group) {
        const size_t nb = (size_t)nb_total - b0 < group ? (size_t)nb_total - b0 : group;
        FbBuild fb{c->d_gens, c->nbases, W, d_table, d_tmp, d_tmp + gentries, d_tmp + 2 * gentries, d_tmp + 3 * gentries, first + (int)b0, (int)nb, first};
        size_t nthreads = nb * nwin * fb_chunks_per_window(W);
        unsigned blocks = (unsigned)((nthreads + BPPP_BLOCK - 1) / BPPP_BLOCK);
        k_fb_build_pass1<<<blocks, BPPP_BLOCK, 0, c->stream>>>(fb, nthreads);
        k_fb_build_pass2<<<blocks, BPPP_BLOCK, 0, c->stream>>>(fb, nthreads);
    }
    HIP_TRY_T(hipGetLastError());
    HIP_TRY_T(hipStreamSynchronize(c->stream));
#undef HIP_TRY_T
    (void)hipFree(d_tmp);
    *out = d_table;
    *out_bytes = bytes;
    return BPPP_OK;
}
// ... of every generator at width W into c->d_table (ct = false) or into c->d_table_ct (the 4-bit table of the "ct_prover" mode)
static int build_tables(bppp_ctx* c, int W, bool ct = false) {
    apt_packed* d_table = nullptr;
    size_t bytes = 0;
    const int rc = build_table_range(c, W, 0, c->nbases, &d_table, &bytes);
    if (rc != BPPP_OK) return rc;
    if (ct) { c->d_table_ct = d_table; c->table_ct_bytes = bytes; return BPPP_OK; }
    c->d_table = d_table;
    c->table_bytes = bytes;
    c->fb_w = W;
    return BPPP_OK;
}
// Two regions (fb_core.h: FbTable; round 5), the u64 shape's second choice when 11 windows for all 49 generators do not fit: the
// u64 protocol's g and g_vec -- the 17 generators that BOTH fixed-base sums of a verify run over (C0's fixed half and the final check) --
// at 11 windows (code 523, 73 GB), h_vec at 12 (code 621, 39 GB): 758 table additions per proof against 726 / 792.
// BPPP_NO_WIDE_TABLES=1 skips the first choice, BPPP_NO_MIXED_WINDOWS=1 both (one table, the general rule).
static const int kWideCode = 523, kMixedHiW = 523, kMixedLoW = 621;
static bool mixed_fits(int ng, int nh, size_t free_bytes) {
    const double tb = table_bytes_for(1 + ng, kMixedHiW) + table_bytes_for(nh, kMixedLoW);
    const double scratch = (double)fb_per_base(kMixedHiW) * 4 * sizeof(fe);
    return tables_fit(tb, scratch, free_bytes);
}
// a window code the library can build: one of the uniform widths, or two widths that tile 258 bits with windows of at most 24 bits
static bool window_code_valid(int code) {
    const int W = fb_wb(code), ka = fb_ka(code);
    if (ka == 0) return W == 4 || W == 8 || W == 16 || W == 10 || W == 18 || W == 19 || W == 20 || W == 22;
    return code > 0 && code < 3200 && W >= 8 && W <= 23 && ka <= fb_nwin(code) && W * fb_nwin(code) + ka >= 258;
}
static int build_tables_mixed(bppp_ctx* c) {
    const int hi = 1 + c->ng;
    apt_packed *t_hi = nullptr, *t_lo = nullptr;
    size_t b_hi = 0, b_lo = 0;
    int rc = build_table_range(c, kMixedHiW, 0, hi, &t_hi, &b_hi);
    if (rc != BPPP_OK) return rc;
    rc = build_table_range(c, kMixedLoW, hi, c->nbases - hi, &t_lo, &b_lo);
    if (rc != BPPP_OK) { (void)hipFree(t_hi); return rc; }
    c->d_table_hi = t_hi; c->table_hi_bytes = b_hi; c->fb_w_hi = kMixedHiW; c->fb_hi_bases = hi;
    c->d_table = t_lo; c->table_bytes = b_lo; c->fb_w = kMixedLoW;
    return BPPP_OK;
}
int ensure_ct_table(bppp_ctx* c) {
    if (c->d_table_ct) return BPPP_OK;
    if (c->fb_w == 4 && c->d_table) {          // the context's own tables already have that shape
        c->d_table_ct = c->d_table;
        c->borrows_table_ct = true;
        return BPPP_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    return build_tables(c, 4, true);
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
        case BPPP_ERR_CLOSED: return "the context is being destroyed";
        default: return "unknown error";
    }
}
const char* bppp_last_error(void) { return g_last_error.c_str(); }

int bppp_ctx_create(bppp_ctx** out, const uint8_t g[64], const uint8_t* g_vec, const uint8_t* h_vec, int device, int fb_window_bits) {
    return bppp_wnla_ctx_create(out, g, g_vec, 16, h_vec, 32, device, fb_window_bits);
}

int bppp_wnla_ctx_create(bppp_ctx** out, const uint8_t g[64], const uint8_t* g_vec, size_t ng, const uint8_t* h_vec, size_t nh, int device,
                         int fb_window_bits) {
    return bppp_wnla_ctx_create_budget(out, g, g_vec, ng, h_vec, nh, device, fb_window_bits, 0);
}
// the largest part of a verify call whose workspace (about 30 KB per proof) takes at most 70 % of the device memory free NOW
static void shrink_max_batch_to_free(bppp_ctx* c) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
    while (c->max_batch > ((size_t)1 << 16) && (double)c->max_batch * 30e3 > 0.70 * (double)free_b) c->max_batch >>= 1;
}

int bppp_wnla_ctx_create_budget(bppp_ctx** out, const uint8_t g[64], const uint8_t* g_vec, size_t ng, const uint8_t* h_vec, size_t nh, int device,
                                int fb_window_bits, uint64_t fb_table_budget_bytes) {
    if (!out || !g || (!g_vec && ng) || (!h_vec && nh) || ng > 4096 || nh > 4096) return BPPP_ERR_INVALID_ARG;
    *out = nullptr;
    const int NB = 1 + (int)ng + (int)nh;
    const int W0 = fb_window_bits;
    if (W0 != 0 && !window_code_valid(W0)) return BPPP_ERR_INVALID_ARG;
    int rc = check_device(device);
    if (rc != BPPP_OK) return rc;
    HIP_TRY(hipSetDevice(device));
    bppp_ctx* c = new (std::nothrow) bppp_ctx();
    if (!c) return BPPP_ERR_NOMEM;
    c->device = device;
    c->n_simds = device_simds(device);
    read_diagnostics(c);
    c->ng = (int)ng; c->nh = (int)nh; c->nbases = NB;
    c->fb_table_budget = fb_table_budget_bytes;
    uint8_t* d_raw = nullptr;
    auto fail = [&](int code) { if (d_raw) (void)hipFree(d_raw); bppp_ctx_destroy(c); return code; };
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
    HIP_TRY_C(hipEventCreateWithFlags(&c->ev_tab, hipEventDisableTiming));
    HIP_TRY_C(hipMalloc(&c->d_gens, NB * sizeof(apt)));
    HIP_TRY_C(hipMalloc(&c->d_flags, sizeof(int)));
    HIP_TRY_C(hipMemsetAsync(c->d_flags, 0, sizeof(int), c->stream));
    // upload + decode generators
    std::vector<uint8_t> hg;
    try { hg.resize((size_t)NB * 64); } catch (...) { return fail(BPPP_ERR_NOMEM); }
    std::memcpy(hg.data(), g, 64);
    if (ng) std::memcpy(hg.data() + 64, g_vec, ng * 64);
    if (nh) std::memcpy(hg.data() + (1 + ng) * 64, h_vec, nh * 64);
    HIP_TRY_C(hipMalloc(&d_raw, hg.size()));
    HIP_TRY_C(hipMemcpyAsync(d_raw, hg.data(), hg.size(), hipMemcpyHostToDevice, c->stream));
    k_decode_generators<<<(NB + 63) / 64, 64, 0, c->stream>>>(d_raw, c->d_gens, NB, c->d_flags);
    int flags = 0;
    HIP_TRY_C(hipMemcpyAsync(&flags, c->d_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY_C(hipStreamSynchronize(c->stream));
    (void)hipFree(d_raw);
    d_raw = nullptr;
    if (flags) return fail(BPPP_ERR_ENCODING);
#undef HIP_TRY_C
    // fixed-base tables: the requested width, or (0) the widest that fits the HBM free right now -- and if even that allocation fails
    // (another process took the memory meanwhile), the next narrower one
    // a table budget (0 = none) bounds what the tables may take, whichever way their layout is chosen
    const double budget = fb_table_budget_bytes ? (double)fb_table_budget_bytes : 1e30;
    if (W0 != 0) {
        if (table_bytes_for(NB, W0) > budget) { g_last_error = "fb_window_bits asks for tables beyond fb_table_budget_bytes"; return fail(BPPP_ERR_INVALID_ARG); }
        rc = build_tables(c, W0);
        if (rc != BPPP_OK) return fail(rc);
    } else {
        size_t free_b = 0, total_b = 0;
        // (re-read after every failed attempt: whoever took the memory meanwhile still has it)
        auto read_free = [&]() {
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
            if (const char* e = std::getenv("BPPP_ASSUME_FREE_GB")) free_b = (size_t)(std::atof(e) * 1e9);      // diagnostic: exercise the choice
        };
        read_free();
        rc = BPPP_ERR_NOMEM;
        const bool u64_shape = ng == 16 && nh == 32 && free_b && !std::getenv("BPPP_NO_MIXED_WINDOWS");
        if (u64_shape && !std::getenv("BPPP_NO_WIDE_TABLES") && table_bytes_for(NB, kWideCode) <= budget && window_fits(NB, kWideCode, free_b)) {
            rc = build_tables(c, kWideCode);
            if (rc != BPPP_OK && rc != BPPP_ERR_NOMEM) return fail(rc);
            if (rc != BPPP_OK) read_free();
        }
        if (rc != BPPP_OK && u64_shape && table_bytes_for(1 + (int)ng, kMixedHiW) + table_bytes_for((int)nh, kMixedLoW) <= budget && mixed_fits((int)ng, (int)nh, free_b)) {
            rc = build_tables_mixed(c);
            if (rc != BPPP_OK && rc != BPPP_ERR_NOMEM) return fail(rc);
            if (rc != BPPP_OK) read_free();
        }
        // the general rule: the fewest windows that fit (and if that allocation fails anyway -- another process took the memory meanwhile
        // -- the next count); without a reading of the free memory, or below every two-width table, the small uniform ones
        if (rc != BPPP_OK && free_b)
            for (int nwin = 11; nwin <= 32; nwin++) {
                const int code = two_width_code_for(nwin);
                if (!window_code_valid(code) || table_bytes_for(NB, code) > budget || !window_fits(NB, code, free_b)) continue;
                rc = build_tables(c, code);
                if (rc == BPPP_OK) break;
                if (rc != BPPP_ERR_NOMEM) return fail(rc);          // a HIP error is not a reason to try a smaller table: report it
                read_free();
            }
        if (rc != BPPP_OK)
            for (int W : {8, 4}) {
                if (table_bytes_for(NB, W) > budget) continue;
                rc = build_tables(c, W);
                if (rc != BPPP_ERR_NOMEM) break;
            }
        if (rc != BPPP_OK) {
            if (rc == BPPP_ERR_NOMEM && fb_table_budget_bytes && table_bytes_for(NB, 4) > budget) g_last_error = "fb_table_budget_bytes is below the smallest table (4-bit windows)";
            return fail(rc);
        }
    }
    // max_batch: the largest part of a verify call whose workspace (about 30 KB per proof) takes at most 70 % of what the tables left
    // free -- still 2^21 proofs beside the 210 GB an empty MI355X gets (288 GiB = 309 GB), fewer on a device that is shared.  Only a
    // first guess: memory that disappears later makes the call shrink its parts (bppp_u64.hip: verify_device_impl), not fail.
    shrink_max_batch_to_free(c);
    *out = c;
    return BPPP_OK;
}

void bppp_ctx_destroy(bppp_ctx* c) {
    if (!c) return;
    // single-proof callers still inside complete first (the front ends' contexts borrow this one's tables); callers on their way in or
    // out of a *_one entry point get BPPP_ERR_CLOSED, and nothing is freed before the last of them has left
    bppp_fronts_teardown(c, true);
    bppp_fronts_delete(c);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& tl : c->pending) { (void)hipEventDestroy(tl.a); (void)hipEventDestroy(tl.b); }
    for (auto& ev : c->event_pool) (void)hipEventDestroy(ev);
    if (c->d_gens && !c->borrows_tables) (void)hipFree(c->d_gens);
    if (c->d_table && !c->borrows_tables) (void)hipFree(c->d_table);
    if (c->d_table_hi && !c->borrows_tables) (void)hipFree(c->d_table_hi);
    if (c->d_table_ct && !c->borrows_table_ct) (void)hipFree(c->d_table_ct);
    if (c->d_ws) (void)hipFree(c->d_ws);
    if (c->d_straus) (void)hipFree(c->d_straus);
    if (c->d_rlc) (void)hipFree(c->d_rlc);
    if (c->d_bkt) (void)hipFree(c->d_bkt);
    if (c->d_atab) (void)hipFree(c->d_atab);
    if (c->d_tscr) (void)hipFree(c->d_tscr);
    if (c->d_pws) (void)hipFree(c->d_pws);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->d_io) (void)hipFree(c->d_io);
    if (c->d_blob) (void)hipFree(c->d_blob);
    if (c->d_txio) (void)hipFree(c->d_txio);
    if (c->d_gws) (void)hipFree(c->d_gws);
    if (c->d_gtab) (void)hipFree(c->d_gtab);
    if (c->d_expand) (void)hipFree(c->d_expand);
    if (c->d_flags) (void)hipFree(c->d_flags);
    if (c->d_rlc_hist) (void)hipFree(c->d_rlc_hist);
    if (c->h_rlc_hist) (void)hipHostFree(c->h_rlc_hist);
    if (c->ev_rlc_hist) (void)hipEventDestroy(c->ev_rlc_hist);
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->twin_stream) { (void)hipStreamSynchronize(c->twin_stream); (void)hipStreamDestroy(c->twin_stream); }
    if (c->twin_aux) { (void)hipStreamSynchronize(c->twin_aux); (void)hipStreamDestroy(c->twin_aux); }
    for (hipEvent_t e : {c->ev_twin_fork, c->ev_twin_join, c->ev2_fork, c->ev2_join, c->ev2_tab}) if (e) (void)hipEventDestroy(e);
    if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_tab) (void)hipEventDestroy(c->ev_tab);
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
        c->rlc_super_auto = false;           // an explicit size is taken as it is (0 switches the stage off)
        return BPPP_OK;
    }
    // RLC mode: proofs per chunk after the bucket stage (8, 32; 0 = chosen per call from the previous call's reject rate), and
    // "rlc_history" = 0: forget that rate (the next call plans as a first call does)
    if (std::strcmp(name, "rlc_chunk") == 0) {
        if (value != 0 && value != 8 && value != 32) return BPPP_ERR_INVALID_ARG;
        c->rlc_chunk_opt = (int)value;
        return BPPP_OK;
    }
    if (std::strcmp(name, "rlc_history") == 0) {
        if (value != 0) return BPPP_ERR_INVALID_ARG;
        c->rlc_rate = -1.0;
        c->rlc_hist_n = 0;
        return BPPP_OK;
    }
    if (std::strcmp(name, "max_batch") == 0) {
        if (value < 1024 || (value & 63)) return BPPP_ERR_INVALID_ARG;
        c->max_batch = (size_t)value;
        return BPPP_OK;
    }
    // the single-proof front end (bppp_u64_verify_one / bppp_u64_prove_one); a change drains the running front ends, the next call
    // starts new ones
    if (std::strcmp(name, "coalesce_max") == 0 || std::strcmp(name, "coalesce_us") == 0 || std::strcmp(name, "coalesce_lanes") == 0) {
        const char k = name[9];      // 'm' / 'u' / 'l'
        if ((k == 'm' && (value < 1 || value > 65536)) || (k == 'u' && (value < 0 || value > 1000000)) || (k == 'l' && (value < 1 || value > 8)))
            return BPPP_ERR_INVALID_ARG;
        if (k == 'm') c->coalesce_max = value;           // first the value, then the drain: a front end started meanwhile has it
        else if (k == 'u') c->coalesce_us = value;
        else c->coalesce_lanes = (int)value;
        bppp_fronts_teardown(c, false);
        return BPPP_OK;
    }
    // the provers' secret-scalar sums in the constant-address form (include/bppp.h); the running single-proof front ends are drained
    // so that their lane contexts pick the setting up
    if (std::strcmp(name, "ct_prover") == 0) {
        if (value != 0 && value != 1) return BPPP_ERR_INVALID_ARG;
        if (value) {
            int rc = ensure_ct_table(c);
            if (rc != BPPP_OK) return rc;
        }
        c->ct_prover = value != 0;
        bppp_fronts_teardown(c, false);
        return BPPP_OK;
    }
    // testing aid: the value-th device allocation from now on (workspaces, staging, tables) fails with BPPP_ERR_NOMEM; 0 clears it
    if (std::strcmp(name, "inject_alloc_fault") == 0) {
        if (value < 0) return BPPP_ERR_INVALID_ARG;
        c->inject_alloc_fault = (int)value;
        return BPPP_OK;
    }
    // parts of a generic reciprocal verify call on device buffers (bppp_generic.hip: generic_parts_for): 0 = by size, 1 .. 4 = that many
    if (std::strcmp(name, "generic_parts") == 0) {
        if (value < 0 || value > 4) return BPPP_ERR_INVALID_ARG;
        c->generic_parts = (int)value;
        return BPPP_OK;
    }
    // how the parts' chains start: 0 together, 1 .. 3 each behind the one before's phase 1 / C0 stage / rounds (host.h: generic_stagger)
    if (std::strcmp(name, "generic_stagger") == 0) {
        if (value < 0 || value > 3) return BPPP_ERR_INVALID_ARG;
        c->generic_stagger = (int)value;
        return BPPP_OK;
    }
    if (std::strcmp(name, "host_chunk") == 0) {
        if (value != 0 && (value < 1024 || (value & 63))) return BPPP_ERR_INVALID_ARG;
        c->host_chunk = (size_t)value;
        return BPPP_OK;
    }
    return BPPP_ERR_INVALID_ARG;
}
// read back a tunable, or one of the read-only facts "fb_window_bits" (the window width in use -- the library's choice when the
// context was created with 0), "device", "n_generators", "last_verify_plan", "last_prove_plan"
long bppp_ctx_get_option(bppp_ctx* c, const char* name) {
    CtxLock lock_(c);
    if (!c || !name) return BPPP_ERR_INVALID_ARG;
    if (std::strcmp(name, "fb_window_bits") == 0) return c->fb_w;
    // a table in two regions (the u64 generator shape with fb_window_bits = 0 on an empty MI355X): the first "fb_hi_bases" generators at
    // "fb_window_bits_hi" bits, the rest at "fb_window_bits"; 0 / 0 = one table
    if (std::strcmp(name, "fb_window_bits_hi") == 0) return c->fb_w_hi;
    if (std::strcmp(name, "fb_hi_bases") == 0) return c->fb_hi_bases;
    // the same layout in plain numbers: table additions per scalar, and the widest window in bits
    if (std::strcmp(name, "fb_windows") == 0) return fb_nwin(c->fb_w);
    if (std::strcmp(name, "fb_window_bits_widest") == 0) return fb_wb(c->fb_w) + (fb_ka(c->fb_w) ? 1 : 0);
    if (std::strcmp(name, "fb_table_bytes") == 0) return (long)(c->table_bytes + c->table_hi_bytes);      // 0 on a context that borrows its parent's tables
    if (std::strcmp(name, "fb_table_budget_bytes") == 0) return (long)c->fb_table_budget;
    if (std::strcmp(name, "device") == 0) return c->device;
    if (std::strcmp(name, "n_generators") == 0) return c->nbases;
    if (std::strcmp(name, "rlc_superchunk") == 0) return (long)c->rlc_super_m;
    if (std::strcmp(name, "rlc_chunk") == 0) return (long)c->rlc_chunk_opt;
    // what the last RLC call on this context used, and whether / which reject rate (parts per million) the next one will plan with
    if (std::strcmp(name, "last_rlc_superchunk") == 0) return (long)c->last_rlc_super_m;
    if (std::strcmp(name, "last_rlc_chunk") == 0) return (long)c->last_rlc_chunk;
    if (std::strcmp(name, "rlc_has_history") == 0 || std::strcmp(name, "rlc_reject_ppm") == 0) {
        if (c->rlc_hist_n && c->ev_rlc_hist && hipEventQuery(c->ev_rlc_hist) == hipSuccess) {
            c->rlc_rate = (double)*c->h_rlc_hist / (double)c->rlc_hist_n;
            c->rlc_hist_n = 0;
        }
        (void)hipGetLastError();
        if (name[4] == 'h') return c->rlc_rate < 0 ? 0 : 1;
        return c->rlc_rate < 0 ? 0 : (long)(c->rlc_rate * 1e6 + 0.5);
    }
    if (std::strcmp(name, "max_batch") == 0) return (long)c->max_batch;
    if (std::strcmp(name, "host_chunk") == 0) return (long)c->host_chunk;
    if (std::strcmp(name, "generic_parts") == 0) return (long)c->generic_parts;
    if (std::strcmp(name, "generic_stagger") == 0) return (long)c->generic_stagger;
    if (std::strcmp(name, "coalesce_max") == 0) return c->coalesce_max;
    if (std::strcmp(name, "coalesce_us") == 0) return c->coalesce_us;
    if (std::strcmp(name, "coalesce_lanes") == 0) return c->coalesce_lanes;
    if (std::strcmp(name, "ct_prover") == 0) return c->ct_prover ? 1 : 0;
    // the kernels the last u64 verify / prove call on this context ran (plan_core.h; text form: bppp_plan_describe)
    if (std::strcmp(name, "last_verify_plan") == 0) return (long)c->last_verify_plan;
    if (std::strcmp(name, "last_prove_plan") == 0) return (long)c->last_prove_plan;
    return BPPP_ERR_INVALID_ARG;
}
long bppp_u64_plan(int prove, size_t n, int n_simds, int flags) {
    if (n_simds < 1 || (prove != 0 && prove != 1) || flags < 0 || flags > 3) return BPPP_ERR_INVALID_ARG;
    bppp_host::PlanKnobs k;
    k.n_simds = n_simds;
    k.timing = (flags & 2) != 0;
    if (prove) return (long)bppp_host::plan_prove(n, k, (flags & 1) != 0).code();
    bppp_host::VerifyPlan p = bppp_host::plan_verify(n, k, (flags & 1) != 0);
    if (p.twin == 2) p = bppp_host::plan_verify_half(bppp_host::twin_first_half(n), k, p.pace);      // what each of the two chains runs
    return (long)p.code();
}
int bppp_plan_describe(long code, int prove, char* buf, size_t cap) {
    if (code < 0 || code > 0xFFFFFFFFl || (!buf && cap)) return BPPP_ERR_INVALID_ARG;
    return bppp_host::plan_describe((uint32_t)code, prove != 0, buf, cap);
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
    return c->table_bytes + c->table_hi_bytes + (c->borrows_table_ct ? 0 : c->table_ct_bytes) + c->ws_bytes + c->straus_bytes + c->vtab_bytes + c->rlc_bytes + c->bkt_bytes + c->pws_bytes + c->stage_bytes + c->io_bytes + c->blob_bytes + c->txio_bytes + c->gws_bytes + c->gtab_bytes + (size_t)c->nbases * sizeof(apt);
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
// merlin::Transcript as 203 serialized bytes, on the host: lets a C caller build the pre-loaded states without merlin and lets
// the tests follow the reference's `t: &mut Transcript` contract end to end (no GPU involved)
int bppp_transcript_new(const uint8_t* label, size_t label_len, uint8_t state_out[203]) {
    if (!label_ok(label, label_len) || !state_out) return BPPP_ERR_INVALID_ARG;
    strobe t;
    t_new(t, label, (u32)label_len);
    strobe_to_bytes(state_out, t, 2);       // the last operation of Transcript::new is the AD of the dom-sep message
    return BPPP_OK;
}
int bppp_transcript_append_message(uint8_t state[203], const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len) {
    if (!state || !label_ok(label, label_len) || (!msg && msg_len) || msg_len > 0xFFFFFFFFu) return BPPP_ERR_INVALID_ARG;
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
    if (!state || !label_ok(label, label_len) || (!out && n) || n > 0xFFFFFFFFu) return BPPP_ERR_INVALID_ARG;
    strobe t;
    if (!strobe_from_bytes(t, state)) return BPPP_ERR_INVALID_ARG;
    uint8_t len4[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    strobe_meta_ad(t, label, (u32)label_len, false);
    strobe_meta_ad(t, len4, 4, true);
    strobe_prf(t, out, (u32)n);
    strobe_to_bytes(state, t, 7);
    return BPPP_OK;
}
#if defined(BPPP_PHASE_TIMING)
// diagnostic builds only (not declared in include/bppp.h): copy the phase stamps out
BPPP_API int bppp_debug_read_stamps(unsigned long long* out) {
    if (!g_bppp_stamps_dev) return -1;
    return hipMemcpy(out, g_bppp_stamps_dev, sizeof(unsigned long long) * BPPP_STAMP_WAVES * 32, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
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
    char magic[8];             // "BPPPTAB3"
    uint32_t nbases, ng, nh, window_bits, nwin, reserved;
    uint64_t per_win, table_bytes;
    uint64_t checksum;         // TableChecksum over the header (this field zero), the generator block and the table body, in file order
};
static_assert(sizeof(TableFileHeader) % 8 == 0, "the checksum runs over 8-byte words");
// A verifier running on a truncated, stale or tampered table would accept bad proofs and nothing would report it, so the file
// carries a checksum of everything after the header: four interleaved 64-bit FNV-1a lanes over little-endian words (host speed
// of a few GB/s -- the disk is slower), folded at the end.  Not a MAC: it catches damage and mix-ups, not an adversary who can
// also rewrite the header; a deployment that distrusts its storage rebuilds the tables (0.2-0.5 s on the GPU) instead.
struct TableChecksum {
    uint64_t h[4] = {0xcbf29ce484222325ull, 0x84222325cbf29ce4ull, 0x9ce484222325cbf2ull, 0x2325cbf29ce48422ull};
    size_t words = 0;
    void update(const void* data, size_t bytes) {                 // bytes is a multiple of 8 (apt = 80 B, apt_packed = 64 B)
        const uint8_t* p = (const uint8_t*)data;
        for (size_t i = 0; i + 8 <= bytes; i += 8, words++) {
            uint64_t w;
            std::memcpy(&w, p + i, 8);
            uint64_t& x = h[words & 3];
            x = (x ^ w) * 0x100000001b3ull;
        }
    }
    uint64_t digest() const {
        uint64_t d = (uint64_t)words;
        for (int i = 0; i < 4; i++) d = (d ^ h[i]) * 0x100000001b3ull;
        return d;
    }
};
int bppp_ctx_save_tables(bppp_ctx* c, const char* path) {
    CtxLock lock_(c);
    if (!c || !path) return BPPP_ERR_INVALID_ARG;
    if (c->fb_hi_bases || !c->d_table || c->borrows_tables) {
        g_last_error = c->fb_hi_bases ? "this context's tables are laid out in two regions (the automatic choice when 11 windows for every generator do not fit); "
                                        "create it with an explicit fb_window_bits (or a table budget that selects one region) to save them"
                                      : "this context has no tables of its own to save";
        return BPPP_ERR_INVALID_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    FILE* f = std::fopen(path, "wb");
    if (!f) { g_last_error = std::string("cannot open ") + path; return BPPP_ERR_INVALID_ARG; }
    TableFileHeader h;
    std::memset(&h, 0, sizeof h);
    std::memcpy(h.magic, fb_ka(c->fb_w) ? "BPPPTAB4" : "BPPPTAB3", 8);       // TAB4: window_bits holds a two-width window code (Wb + 100 ka)
    h.nbases = (uint32_t)c->nbases; h.ng = (uint32_t)c->ng; h.nh = (uint32_t)c->nh; h.window_bits = (uint32_t)c->fb_w;
    h.nwin = (uint32_t)fb_nwin(c->fb_w); h.per_win = fb_per_narrow(c->fb_w); h.table_bytes = c->table_bytes;
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
    std::vector<apt> gens;
    std::vector<uint8_t> buf;
    const size_t CH = (size_t)256 << 20;
    try {
        gens.resize(c->nbases);
        buf.resize(c->table_bytes < CH ? c->table_bytes : CH);
    } catch (...) { std::fclose(f); g_last_error = "out of host memory"; return BPPP_ERR_NOMEM; }
    if (hipMemcpy(gens.data(), c->d_gens, gens.size() * sizeof(apt), hipMemcpyDeviceToHost) != hipSuccess) ok = false;
    ok = ok && std::fwrite(gens.data(), sizeof(apt), gens.size(), f) == gens.size();
    TableChecksum sum;
    sum.update(&h, sizeof h);                                  // the header too (checksum field still zero): ng / nh / window width are covered
    sum.update(gens.data(), gens.size() * sizeof(apt));
    for (size_t off = 0; ok && off < c->table_bytes; off += CH) {
        const size_t m = c->table_bytes - off < CH ? c->table_bytes - off : CH;
        if (hipMemcpy(buf.data(), (const uint8_t*)c->d_table + off, m, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
        sum.update(buf.data(), m);
        ok = ok && std::fwrite(buf.data(), 1, m, f) == m;
    }
    h.checksum = sum.digest();                               // known only now: the header is written a second time, complete
    ok = ok && std::fseek(f, 0, SEEK_SET) == 0 && std::fwrite(&h, sizeof h, 1, f) == 1;
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
    HIP_TRY(hipEventCreateWithFlags(&c->ev_tab, hipEventDisableTiming));
    HIP_TRY(hipMalloc(&c->d_flags, sizeof(int)));
    read_diagnostics(c);
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
    bool ok = std::fread(&h, sizeof h, 1, f) == 1;
    const bool tab3 = ok && std::memcmp(h.magic, "BPPPTAB3", 8) == 0, tab4 = ok && std::memcmp(h.magic, "BPPPTAB4", 8) == 0;
    const int W = ok ? (int)h.window_bits : 0;
    // TAB3: one of the uniform widths; TAB4: a two-width window code (include/bppp.h) -- in both the geometry fields must be the ones the
    // code implies, so that a header edited to another layout of the same size is refused before the checksum is even computed
    ok = ok && (tab3 || tab4) && h.window_bits < 3200 && window_code_valid(W) && (fb_ka(W) > 0) == tab4 && h.ng <= 4096 && h.nh <= 4096 && h.nbases == 1 + h.ng + h.nh &&
         h.nwin == (uint32_t)fb_nwin(W) && h.per_win == fb_per_narrow(W) && h.table_bytes == (uint64_t)h.nbases * fb_per_base(W) * sizeof(apt_packed);
    if (!ok) { std::fclose(f); g_last_error = std::string(path) + " is not a table file of this library"; return BPPP_ERR_INVALID_ARG; }
    if (hipSetDevice(device) != hipSuccess) { std::fclose(f); g_last_error = "hipSetDevice failed"; (void)hipGetLastError(); return BPPP_ERR_HIP; }
    bppp_ctx* c = new (std::nothrow) bppp_ctx();
    if (!c) { std::fclose(f); return BPPP_ERR_NOMEM; }
    c->device = device; c->fb_w = W; c->ng = (int)h.ng; c->nh = (int)h.nh; c->nbases = (int)h.nbases; c->table_bytes = h.table_bytes;
    auto fail = [&](int code) { std::fclose(f); bppp_ctx_destroy(c); return code; };
    rc = ctx_alloc_common(c);
    if (rc != BPPP_OK) return fail(rc);
    std::vector<apt> gens;
    std::vector<uint8_t> buf;
    const size_t CH = (size_t)256 << 20;
    try {
        gens.resize(h.nbases);
        buf.resize(c->table_bytes < CH ? c->table_bytes : CH);
    } catch (...) { g_last_error = "out of host memory"; return fail(BPPP_ERR_NOMEM); }
    if (std::fread(gens.data(), sizeof(apt), gens.size(), f) != gens.size()) return fail(BPPP_ERR_INVALID_ARG);
    // the generators get the validation bppp_ctx_create gives them (k_decode_generators): canonical limbs, on the curve
    for (const apt& a : gens) {
        uint8_t xy[64];
        apt back;
        apt_to_xy64(xy, a);                                   // through the wire form and back: the stored limbs must be exactly the
        if (!apt_from_xy64(back, xy) || std::memcmp(&back, &a, sizeof(apt)) != 0) {                        // decoder's, on the curve
            g_last_error = std::string(path) + ": a stored generator is not a valid secp256k1 point";
            return fail(BPPP_ERR_ENCODING);
        }
    }
    TableChecksum sum;
    {
        TableFileHeader hz = h;
        hz.checksum = 0;
        sum.update(&hz, sizeof hz);
    }
    sum.update(gens.data(), gens.size() * sizeof(apt));
    if (hipMalloc(&c->d_gens, gens.size() * sizeof(apt)) != hipSuccess || hipMalloc(&c->d_table, c->table_bytes) != hipSuccess) { (void)hipGetLastError(); return fail(BPPP_ERR_NOMEM); }
    if (hipMemcpy(c->d_gens, gens.data(), gens.size() * sizeof(apt), hipMemcpyHostToDevice) != hipSuccess) return fail(BPPP_ERR_HIP);
    for (size_t off = 0; off < c->table_bytes; off += CH) {
        const size_t m = c->table_bytes - off < CH ? c->table_bytes - off : CH;
        if (std::fread(buf.data(), 1, m, f) != m) { g_last_error = std::string(path) + " is truncated"; return fail(BPPP_ERR_INVALID_ARG); }
        sum.update(buf.data(), m);
        if (hipMemcpy((uint8_t*)c->d_table + off, buf.data(), m, hipMemcpyHostToDevice) != hipSuccess) return fail(BPPP_ERR_HIP);
    }
    if (sum.digest() != h.checksum) {
        g_last_error = std::string(path) + ": checksum mismatch (the header, the generators or the table body differ from what was saved)";
        return fail(BPPP_ERR_INVALID_ARG);
    }
    std::fclose(f);
    shrink_max_batch_to_free(c);         // like a context built from generators: parts sized to what the tables left free
    *out = c;
    return BPPP_OK;
}
// A second context on the SAME device that shares `parent`'s generators and fixed-base tables (read-only data) and owns its
// streams and workspaces: several host threads can then verify concurrently on one GPU without a second 21 GB table.  The
// parent must outlive its children.
int bppp_ctx_create_shared(bppp_ctx** out, bppp_ctx* parent) {
    if (!out || !parent) return BPPP_ERR_INVALID_ARG;
    CtShare ct;
    {       // the "ct_prover" option may be changing on another thread: its two fields are read under the parent's lock
        std::lock_guard<std::recursive_mutex> lk(parent->mu);
        ct.d_table_ct = parent->d_table_ct;
        ct.ct_prover = parent->ct_prover;
    }
    return ctx_create_shared_with(out, parent, ct);
}

}  // extern "C"

// (everything else a child reads from its parent -- device, generator counts, window width, table pointers -- is fixed at creation)
int ctx_create_shared_with(bppp_ctx** out, bppp_ctx* parent, const CtShare& ct) {
    if (!out || !parent) return BPPP_ERR_INVALID_ARG;
    *out = nullptr;
    HIP_TRY(hipSetDevice(parent->device));
    bppp_ctx* c = new (std::nothrow) bppp_ctx();
    if (!c) return BPPP_ERR_NOMEM;
    c->device = parent->device; c->fb_w = parent->fb_w; c->ng = parent->ng; c->nh = parent->nh; c->nbases = parent->nbases;
    c->d_gens = parent->d_gens; c->d_table = parent->d_table; c->table_bytes = 0; c->borrows_tables = true;
    c->d_table_hi = parent->d_table_hi; c->table_hi_bytes = 0; c->fb_w_hi = parent->fb_w_hi; c->fb_hi_bases = parent->fb_hi_bases;
    if (ct.d_table_ct) { c->d_table_ct = ct.d_table_ct; c->borrows_table_ct = true; c->ct_prover = ct.ct_prover; }
    // the part size the parent settled on (from the memory its tables left, or from a call that ran out of it), shrunk again to what
    // is free now: the child's workspaces come out of the same device memory
    c->max_batch = parent->max_batch; c->host_chunk = parent->host_chunk; c->fb_table_budget = parent->fb_table_budget;
    shrink_max_batch_to_free(c);
    int rc = ctx_alloc_common(c);
    if (rc != BPPP_OK) { bppp_ctx_destroy(c); return rc; }
    *out = c;
    return BPPP_OK;
}
