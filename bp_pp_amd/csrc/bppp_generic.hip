// libbppp_hip.so, host side: generic WeightNormLinearArgument / ReciprocalRangeProofProtocol / ArithmeticCircuit verifiers and provers, and the
// generic fixed-base linear combination (the crate's commit functions).
#include "host.h"

extern "C" {

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
    uint8_t *d_in = nullptr, *d_out = nullptr;      // carved out of the context's grow-only transcript staging
    int begin(bppp_ctx* c, const HostTranscripts* tx, size_t n, hipStream_t s, TranscriptIo& io, int& divergent) {
        io.states = nullptr; io.n_states = 0; io.states_out = nullptr; io.no_ops = 0;
        divergent = 0;
        if (!tx) return BPPP_OK;
        int rc = check_host_transcripts(tx, n);
        if (rc != BPPP_OK) return rc;
        const size_t b_in = align16(tx->n_states * 203);
        rc = ensure_buffer(c, c->d_txio, c->txio_bytes, b_in + (tx->states_out ? n * 203 : 0));
        if (rc != BPPP_OK) return rc;
        d_in = c->d_txio;
        if (tx->states_out) d_out = c->d_txio + b_in;
        HIP_TRY(hipMemcpyAsync(d_in, tx->states, tx->n_states * 203, hipMemcpyHostToDevice, s));
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
// running products of their build (BPPP_TSCR_PER_POINT per point) and the decoded round points -- 1.2 KB + 0.55 KB + 64 B per point, grow-only
// extra_points: tables of that many more points per instance behind the round points' (the reciprocal verifier's five C0 points)
// Sets of round-point tables a call of n instances builds (WnlaWs::tab_parts): calls that leave the chip EMPTY -- at most four instances
// per SIMD -- cut every 26-window stream in 2 (4: at most one instance per SIMD) parts over tables of P and 2^65 P (P, 2^35 P, 2^70 P,
// 2^100 P), a lane per table, and walk a round's sum on 8 (16) lanes: what such a call takes is the length of ONE instance's chain
// (round 6: one WNLA verify of the u64 size 4.1 -> ms).  The u64 verifier has done the same since round 3 (plan_core.h: split).
static int wnla_table_parts(const bppp_ctx* c, size_t n, size_t rounds) {
    if (rounds == 0 || c->generic_slow_rounds || c->no_lane_groups || c->no_split || c->generic_lane_group) return 1;
    const size_t S = (size_t)(c->n_simds > 0 ? c->n_simds : 1);
    return n <= S ? 4 : n <= 4 * S ? 2 : 1;
}
static size_t wnla_fast_bytes(size_t n, size_t rounds, size_t extra_points, size_t parts = 1) {
    const size_t np = 2 * rounds * parts + extra_points;
    return align16(np * 16 * sizeof(apt_packed) * n) + align16((size_t)BPPP_TSCR_PER_POINT * np * 10 * sizeof(u32) * n) + align16(np * 16 * sizeof(u32) * n);
}
// base: where this call's (or this part's) share of the table buffer starts; null = the context's buffer, grown to what the call needs
// parts: sets of round-point tables (wnla_table_parts); the extra points' tables follow them, at entry 2 x rounds x 16 x parts
static int wnla_fast_setup(bppp_ctx* c, WnlaWs& w, size_t n, size_t rounds, size_t extra_points = 0, uint8_t* base = nullptr, size_t parts = 1) {
    w.atab = nullptr; w.tscr = nullptr; w.rpts = nullptr; w.tab_parts = 1;
    if (rounds == 0 || c->generic_slow_rounds) return BPPP_OK;
    const size_t np = 2 * rounds * parts + extra_points;
    const size_t b_tab = align16(np * 16 * sizeof(apt_packed) * n), b_scr = align16((size_t)BPPP_TSCR_PER_POINT * np * 10 * sizeof(u32) * n);
    if (!base) {
        const int rc_t = ensure_buffer(c, c->d_gtab, c->gtab_bytes, wnla_fast_bytes(n, rounds, extra_points, parts));
        if (rc_t != BPPP_OK) return rc_t;
        base = c->d_gtab;
    }
    w.atab = (apt_packed*)base;
    w.tscr = (u32*)(base + b_tab);
    w.rpts = (u32*)(base + b_tab + b_scr);
    w.tab_parts = (int)parts;
    return BPPP_OK;
}
// lanes per instance for the generic rounds: 16 / 8 over tables in 4 / 2 parts (calls that leave the chip empty), else 4 or 2 while that
// still leaves wavefront slots free (and the fast path's tables exist)
static int wnla_round_group(const bppp_ctx* c, const WnlaWs& w, unsigned blocks) {
    if (!w.atab || c->no_lane_groups) return 1;
    if (w.tab_parts == 4) return 16;
    if (w.tab_parts == 2) return 8;
    if (c->generic_lane_group) return c->generic_lane_group;
    if (4 * (size_t)blocks <= (size_t)c->n_simds) return 4;
    if (2 * (size_t)blocks <= (size_t)c->n_simds) return 2;
    return 1;
}
// the fixed-base sums of a generic verify call on a WAVEFRONT per instance instead of 8 lanes: calls of up to eight instances per SIMD, where
// 8 lanes per instance are at most one wavefront per SIMD and the call waits for one lane's chain of (bases x windows) / 8 dependent
// table additions -- one instance of configs[4]'s shape: k_wnla_msm 7.6 -> 0.86 ms, k_recip_c0_fixed 2.5 -> 0.32 ms; with phase 1 on lane groups the call
// 16.7 -> 6.1 ms (round 6, profiles/r06/r06_p4_latency_recip256.txt).  By size, same shape on 16-bit tables: 2,048 instances 14.1 -> 8.4 ms,
// 4,096 14.9 -> 11.2, 8,192 19.3 -> 18.6, 16,384 28.2 -> 28.7 (profiles/r06/r06_p5_fb_wide_sizes.txt): up to 8 S
static bool generic_fb_wide(const bppp_ctx* c, size_t n) {
    if (c->generic_fb_wide_max >= 0) return n <= (size_t)c->generic_fb_wide_max;
    return !c->no_lane_groups && !c->no_split && n <= 8 * (size_t)c->n_simds;
}
static unsigned fb64_blocks_of(size_t n) { return (unsigned)((n * 64 + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK); }
// the round points' window tables: a lane per instance, or a lane per (point, part) table in a call that leaves the chip empty
static void launch_wnla_tables(const WnlaWs& w, size_t n, unsigned blocks, hipStream_t s) {
    if (w.tab_parts > 1) {
        int lp = 2;
        while (lp < 2 * w.rounds) lp *= 2;
        k_wnla_tables_split<<<(unsigned)(((size_t)lp * w.tab_parts * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(w, w.tab_parts, lp);
    }
    else k_wnla_tables<<<blocks, BPPP_BLOCK, 0, s>>>(w);
}
// lanes per instance for the final scalars (k_wnla_final_scalars_grp), as log2: up to 8 while the launch stays within four wavefronts
// per SIMD.  The work is independent per generator, so unlike the rounds it divides by the full group size.
static int wnla_final_scalars_group_lg(const bppp_ctx* c, unsigned rounds, unsigned blocks) {
    if (c->no_lane_groups) return 0;
    if (c->generic_lane_group) return wnla_final_scalars_lg((int)rounds, c->generic_lane_group == 2 ? 1 : 3);   // (tests: 2 or 8 parts)
    int lg = 0;
    while (lg < 3 && ((size_t)blocks << (lg + 1)) <= 4 * (size_t)c->n_simds) lg++;
    return wnla_final_scalars_lg((int)rounds, lg);
}
static void launch_wnla_final_scalars(const bppp_ctx* c, const WnlaWs& w, unsigned rounds, size_t n, unsigned blocks, hipStream_t s, unsigned call_blocks = 0) {
    const int lg = wnla_final_scalars_group_lg(c, rounds, call_blocks ? call_blocks : blocks);
    if (lg > 0) {
        k_wnla_final_scalars_grp<<<blocks << lg, BPPP_BLOCK, 0, s>>>(w, lg);
        k_wnla_final_scalars_join<<<blocks, BPPP_BLOCK, 0, s>>>(w, lg);
    } else k_wnla_final_scalars<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    (void)n;
}
// ---- generic WeightNormLinearArgument entry points (host pointers; one device blob per call)
// (the context's grow-only buffer: no allocator round trip per call.  Whatever way the call ends, nothing of it is still running
// when the blob goes out of scope -- on the success path the stream has just been waited for and this costs nothing)
struct WnlaBlob {
    bppp_ctx* c = nullptr;
    uint8_t* d = nullptr;
    int take(bppp_ctx* ctx, size_t bytes) {
        c = ctx;
        int rc = ensure_blob(ctx, bytes);
        d = ctx->d_blob;
        return rc;
    }
    ~WnlaBlob() { if (c) quiesce(c); }
};

static int wnla_run(bppp_ctx* c, bool commit, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                    const uint8_t* cvec, const uint8_t* rho, const uint8_t* mu, size_t rounds, const uint8_t* proof_r,
                    const uint8_t* proof_x, const uint8_t* proof_l, size_t nl, const uint8_t* proof_n, size_t nn, uint8_t* out_points,
                    uint8_t* accept, int32_t* status, const HostTranscripts* tx = nullptr, bool device_io = false) {
    // device_io (bppp_wnla_verify_batch_device): every pointer above is DEVICE memory, read and written in place; the call is asynchronous
    // on the context's stream and only the workspace comes out of the context's buffer
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
    if (device_io) { const int rc_b = ensure_blob(c, off + 16); if (rc_b != BPPP_OK) return rc_b; }       // (no sync when the call returns)
    else { const int rc_b = blob.take(c, off + 16); if (rc_b != BPPP_OK) return rc_b; }
    uint8_t* d = c->d_blob;
    hipStream_t s = c->stream;
    auto up = [&](size_t o, const uint8_t* src, size_t bytes) -> hipError_t {
        return (src && bytes && !device_io) ? hipMemcpyAsync(d + o, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
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
    auto in = [&](size_t o, const uint8_t* p) -> const uint8_t* { return device_io ? p : d + o; };
    WnlaWs w;
    std::memset(&w, 0, sizeof w);
    w.N = n; w.ng = c->ng; w.nh = c->nh; w.rounds = (int)rounds; w.nl = (int)nl; w.nn = (int)nn;
    w.commitments = in(o_com, commitments); w.c = in(o_c, cvec); w.rho = in(o_rho, rho); w.mu = in(o_mu, mu); w.proof_r = in(o_r, proof_r); w.proof_x = in(o_x, proof_x);
    w.proof_l = in(o_l, proof_l); w.proof_n = in(o_n, proof_n); w.out_points = d + o_out;
    w.accept = device_io ? accept : d + o_acc; w.status = (device_io && status) ? status : (int32_t*)(d + o_st);
    w.tstate = (u32*)(d + o_ts); w.acc = (u32*)(d + o_a); w.pfix = (u32*)(d + o_pf); w.ys = (u32*)(d + o_ys);
    w.tab = (u32*)(d + o_tab); w.msc = (u32*)(d + o_msc);
    w.stride_r = rounds * 64; w.stride_x = rounds * 64; w.stride_l = nl * 32; w.stride_n = nn * 32;
    w.straus = c->d_straus;
    w.fb = fb_table_of(c, n);
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
        rc = wnla_fast_setup(c, w, n, rounds, 0, nullptr, (size_t)wnla_table_parts(c, n, rounds));
        if (rc != BPPP_OK) return rc;
#define WLAUNCH(id, ...)                                       \
    do {                                                       \
        rc = timed(c, id, s, [&]() { __VA_ARGS__; });          \
        if (rc != BPPP_OK) return rc;                          \
    } while (0)
        WLAUNCH(K_WNLA_BEGIN, k_wnla_begin<<<blocks, BPPP_BLOCK, 0, s>>>(w));
        if (w.atab) WLAUNCH(K_WNLA_TABLES, launch_wnla_tables(w, n, blocks, s));
        {
            const int grp = wnla_round_group(c, w, blocks);
            for (int k = 1; k <= (int)rounds; k++) {
                if (grp > 1) WLAUNCH(K_WNLA_ROUND, k_wnla_round_grp<<<(unsigned)(((size_t)grp * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(w, k, grp));
                else WLAUNCH(K_WNLA_ROUND, k_wnla_round<<<blocks, BPPP_BLOCK, 0, s>>>(w, k));
            }
        }
        WLAUNCH(K_WNLA_FINAL_SCALARS, launch_wnla_final_scalars(c, w, (unsigned)rounds, n, blocks, s));
        if (generic_fb_wide(c, n)) WLAUNCH(K_WNLA_MSM, k_wnla_msm_l64<<<fb64_blocks_of(n), BPPP_FB_BLOCK, 0, s>>>(w));
        else WLAUNCH(K_WNLA_MSM, k_wnla_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 0));
        WLAUNCH(K_WNLA_ACCEPT, k_wnla_accept<<<blocks, BPPP_BLOCK, 0, s>>>(w));
#undef WLAUNCH
        if (w.tio.states_out) k_generic_export_states<<<blocks, BPPP_BLOCK, 0, s>>>(w);
        HIP_TRY(hipGetLastError());
        if (device_io) return BPPP_OK;
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
    if (!c || !label_ok(label, label_len) || !commitments || !cvec || !rho || !mu || (rounds && (!proof_r || !proof_x)) || (!proof_l && nl) ||
        (!proof_n && nn) || !accept)
        return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    return wnla_run(c, false, label, label_len, n, commitments, cvec, rho, mu, rounds, proof_r, proof_x, proof_l, nl, proof_n, nn, nullptr,
                    accept, status);
}

int bppp_wnla_verify_batch_device(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments, const void* d_c,
                                  const void* d_rho, const void* d_mu, size_t rounds, const void* d_proof_r, const void* d_proof_x,
                                  const void* d_proof_l, size_t nl, const void* d_proof_n, size_t nn, void* d_accept, void* d_status) {
    CtxLock lock_(c);
    if (!c || !label_ok(label, label_len) || !d_commitments || !d_c || !d_rho || !d_mu || (rounds && (!d_proof_r || !d_proof_x)) || (!d_proof_l && nl) ||
        (!d_proof_n && nn) || !d_accept)
        return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    return wnla_run(c, false, label, label_len, n, (const uint8_t*)d_commitments, (const uint8_t*)d_c, (const uint8_t*)d_rho, (const uint8_t*)d_mu, rounds,
                    (const uint8_t*)d_proof_r, (const uint8_t*)d_proof_x, (const uint8_t*)d_proof_l, nl, (const uint8_t*)d_proof_n, nn, nullptr,
                    (uint8_t*)d_accept, (int32_t*)d_status, nullptr, true);
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
// one part of a multi-part call (recip_verify_device_entry): its stream and its shares of the context's buffers
// (started / stage: an event recorded behind the part's stage-th milestone -- 1 phase 1, 2 the C0 stage, 3 the rounds -- that the NEXT
// part's chain waits for, so that the chains run out of step: one part's fixed-base sums under another's one-lane kernels)
struct GenericPart { hipStream_t s; uint8_t* gtab; pt_slot* straus; unsigned call_blocks; hipEvent_t started; int stage; };
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
                                    int32_t* d_st, uint8_t* d_ws, const TranscriptIo* dtio = nullptr, const uint8_t* rlc_seed = nullptr,
                                    const GenericPart* part = nullptr) {
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
    hipStream_t s = part ? part->s : c->stream;
    pt_slot* const straus = part ? part->straus : c->d_straus;
    RecipWs r;
    std::memset(&r, 0, sizeof r);
    r.N = n; r.nd = (int)dim_nd; r.np = (int)dim_np; r.rounds = (int)rounds; r.nl = (int)nl; r.nn = (int)nn;
    r.NG = c->ng; r.NH = c->nh; r.proof_bytes = proof_bytes;
    r.commitments = d_com; r.proofs = d_proofs; r.status = d_st; r.tstate = (u32*)(d + o_ts);
    r.sc0 = (u32*)(d + o_sc0); r.pts = (u32*)(d + o_pts); r.acc = (u32*)(d + o_a); r.pfix = (u32*)(d + o_pf); r.inv = (u32*)(d + o_inv);
    r.straus = straus;
    r.wn_commit = d + o_wc; r.wn_c = d + o_wcv; r.wn_rho = d + o_rho; r.wn_mu = d + o_mu;
    r.fb = fb_table_of(c, n);
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
    w.straus = straus;
    w.fb = r.fb;
    const unsigned blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    const unsigned fb_blocks = (unsigned)((n * BPPP_FB_LANES + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    // (the lane-group choices below go by the wavefronts of the WHOLE call: the parts of a multi-part call share the chip)
    const unsigned call_blocks = part ? part->call_blocks : blocks;
    int rc;
#define GLAUNCH(id, ...)                                       \
    do {                                                       \
        rc = timed(c, id, s, [&]() { __VA_ARGS__; });          \
        if (rc != BPPP_OK) return rc;                          \
    } while (0)
    // the WNLA stage's table buffer with room for the five C0 points' tables behind the round points': the variable-base part of C0 on
    // affine window tables too (and on lane groups while one lane per instance leaves wavefront slots free)
    rc = wnla_fast_setup(c, w, n, rounds, 5, part ? part->gtab : nullptr, part ? 1 : (size_t)wnla_table_parts(c, n, rounds));
    if (rc != BPPP_OK) return rc;
    r.atab = w.atab; r.tscr = w.tscr; r.atab_first = (int)(2 * rounds * 16 * (size_t)w.tab_parts);
    // What needs nothing but the proof bytes -- the round points' window tables -- and what needs only phase 1 -- the C0 points' tables
    // and C0's variable-base sum, one lane (or a lane group) per instance -- runs on the HELPER stream beside phase 1 and the fixed-base
    // half of C0 (8 lanes per instance: the kernel that fills the chip); round 6: 2^15 instances of configs[4]'s shape, where the
    // one-lane kernels are half a wavefront per SIMD, 46.6 -> 45.0 ms per batch.  With kernel timing on everything stays on one stream so
    // that the per-kernel times add up; the parts of a multi-part call are chains of their own.
    // the two fixed-base sums: 8 lanes per instance, or one from the size at which one lane per instance fills the SIMDs twice over
    const bool fb_one_lane = c->fb_one_lane_mode >= 0 ? c->fb_one_lane_mode == 1 : n >= (size_t)128 * (size_t)c->n_simds;
    // (only while the one-lane kernels are at most half a wavefront per SIMD: beyond that the kernels fill the chip by themselves and side
    // by side they take LONGER than one after the other, as in the u64 verifier -- 2^16 instances 79.6 ms on two streams against 79.2 on
    // one, 2^17 154.2 / 153.0, 2^18 314.3 / 301.4: profiles/r06/r06_b1_recip_beside_sizes.txt)
    const bool beside = w.atab && !c->timing && !part && (c->recip_beside >= 0 ? c->recip_beside == 1 : 2 * (size_t)call_blocks <= (size_t)c->n_simds);
    hipStream_t a = beside ? c->aux_stream : s;
#define GLAUNCH_ON(st, id, ...)                                 \
    do {                                                        \
        rc = timed(c, id, st, [&]() { __VA_ARGS__; });          \
        if (rc != BPPP_OK) return rc;                           \
    } while (0)
    if (beside) {
        HIP_TRY(hipEventRecord(c->ev_tab, s));               // (the call's inputs are ready on s)
        HIP_TRY(hipStreamWaitEvent(a, c->ev_tab, 0));
        GLAUNCH_ON(a, K_WNLA_TABLES, launch_wnla_tables(w, n, blocks, a));
    }
    {
        // lanes per instance for phase 1's two loops over the digits: as many (up to 8) as keep the launch within ONE wavefront per SIMD --
        // the kernel is an uncapped build (one wavefront per SIMD fits), and a group's lanes each repeat the head (transcript, inversions:
        // a seventh of the one-lane kernel).  Round 6, configs[4]'s shape, the kernel alone: 2^15 instances 3.48 -> 2.10 ms on 2 lanes
        // (2.65 on 4, 3.8 on 8: two and four generations of wavefronts); the call 45.71 -> 45.30 ms, because the round-point tables
        // that ran beside the half-empty one-lane kernel now share its SIMDs (profiles/r06/r06_p2_recip_phase1_groups.txt)
        int G = 1;
        if (c->recip_p1_group) G = c->recip_p1_group;
        else if (!c->no_lane_groups) {
            if (c->generic_lane_group) G = c->generic_lane_group == 2 ? 2 : 8;      // (tests: the smallest and the largest split at any size)
            else while (G < 8 && 2 * (size_t)G * call_blocks <= (size_t)c->n_simds) G *= 2;
        }
        if (G > 1) GLAUNCH(K_RECIP_PHASE1, k_recip_phase1_grp<<<(unsigned)(((size_t)G * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(r, G));
        else GLAUNCH(K_RECIP_PHASE1, k_recip_phase1<<<blocks, BPPP_BLOCK, 0, s>>>(r));
    }
    if (part && part->started && part->stage == 1) HIP_TRY(hipEventRecord(part->started, s));
    if (beside) {
        HIP_TRY(hipEventRecord(c->ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(a, c->ev_fork, 0));
    }
    const unsigned fb1_blocks = (unsigned)((n + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    const bool fb_wide = !fb_one_lane && !part && generic_fb_wide(c, n);
    if (fb_one_lane) GLAUNCH(K_RECIP_C0_FIXED, k_recip_c0_fixed_l1<<<fb1_blocks, BPPP_FB_BLOCK, 0, s>>>(r));
    else if (fb_wide) GLAUNCH(K_RECIP_C0_FIXED, k_recip_c0_fixed_l64<<<fb64_blocks_of(n), BPPP_FB_BLOCK, 0, s>>>(r));
    else GLAUNCH(K_RECIP_C0_FIXED, k_recip_c0_fixed<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(r));
    if (r.atab) {
        const int grp_r = wnla_round_group(c, w, call_blocks), grp = grp_r > 4 ? 4 : grp_r;      // (C0's sum: lane groups of 2 or 4)
        GLAUNCH_ON(a, K_RECIP_C0_VAR, {
            k_recip_c0_tables<<<blocks, BPPP_BLOCK, 0, a>>>(r);
            if (grp > 1) k_recip_c0_var_grp<<<(unsigned)(((size_t)grp * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, a>>>(r, grp);
            else k_recip_c0_var<<<blocks, BPPP_BLOCK, 0, a>>>(r);
        });
    } else GLAUNCH(K_RECIP_C0_VAR, k_recip_c0_var<<<blocks, BPPP_BLOCK, 0, s>>>(r));
    if (beside) {
        HIP_TRY(hipEventRecord(c->ev_join, a));
        HIP_TRY(hipStreamWaitEvent(s, c->ev_join, 0));
    }
#undef GLAUNCH_ON
    GLAUNCH(K_RECIP_C0_FINISH, k_recip_c0_finish<<<blocks, BPPP_BLOCK, 0, s>>>(r));
    if (part && part->started && part->stage == 2) HIP_TRY(hipEventRecord(part->started, s));
    GLAUNCH(K_WNLA_BEGIN, k_wnla_begin<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    if (w.atab && !beside) GLAUNCH(K_WNLA_TABLES, launch_wnla_tables(w, n, blocks, s));
    {
        const int grp = wnla_round_group(c, w, call_blocks);
        for (int k = 1; k <= (int)rounds; k++) {
            if (grp > 1) GLAUNCH(K_WNLA_ROUND, k_wnla_round_grp<<<(unsigned)(((size_t)grp * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(w, k, grp));
            else GLAUNCH(K_WNLA_ROUND, k_wnla_round<<<blocks, BPPP_BLOCK, 0, s>>>(w, k));
        }
    }
    if (part && part->started && part->stage == 3) HIP_TRY(hipEventRecord(part->started, s));
    GLAUNCH(K_WNLA_FINAL_SCALARS, launch_wnla_final_scalars(c, w, (unsigned)rounds, n, blocks, s, call_blocks));
    if (!rlc_seed) {
        if (fb_one_lane) GLAUNCH(K_WNLA_MSM, k_wnla_msm_l1<<<fb1_blocks, BPPP_FB_BLOCK, 0, s>>>(w));
        else if (fb_wide) GLAUNCH(K_WNLA_MSM, k_wnla_msm_l64<<<fb64_blocks_of(n), BPPP_FB_BLOCK, 0, s>>>(w));
        else GLAUNCH(K_WNLA_MSM, k_wnla_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 0));
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
        if (const unsigned SM = bucket_superchunk_for(c, n)) {
            // bucket (Pippenger) stage first, as in the u64 verifier: superchunks of SM instances, the weighted commitments summed by
            // bucket accumulation and ONE 1 + ng + nh-base MSM per superchunk; the chunk-of-8 kernels only see what failed it
            BucketWs bw;
            rc = launch_bucket_stage(c, bw, n, SM, rl.seed, w.status, w.acc, w.msc, (int)NB, d_acc, s,
                                     [&](int id, auto&& f) { return timed(c, id, s, f); });
            if (rc != BPPP_OK) return rc;
            rl.sflag = bw.sflag;
            rl.super_m = SM;
        }
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
// parts of a reciprocal verify call (see recip_verify_device_entry): by how far one lane per instance under-fills the chip; one part
// with kernel timing on (the per-kernel times must add up), in RLC mode (its stages work on the whole batch) and for small calls
static int generic_parts_for(const bppp_ctx* c, size_t n, bool rlc) {
    if (rlc || c->timing || n < 2 * BPPP_BLOCK) return 1;
    if (c->generic_parts > 0) return c->generic_parts > 4 ? 4 : c->generic_parts;      // forced (A/B runs, tests at small sizes)
    return 1;
}
static int recip_verify_check_args(const bppp_ctx* c, size_t dim_nd, size_t dim_np, size_t rounds, size_t nl, size_t nn) {
    if (dim_nd == 0 || dim_np == 0 || dim_nd > (size_t)c->ng || dim_nd + 10 > (size_t)c->nh || dim_np > dim_nd + 1 || rounds > 12 ||
        nl > 4096 || nn > 4096)
        return BPPP_ERR_INVALID_ARG;
    return BPPP_OK;
}
// ReciprocalRangeProofProtocol { dim_nd: 16, dim_np: 16 } over 16 + 32 generators with a standard-shape proof IS the u64 protocol
// (u64_proof.rs:42-54 builds exactly this and calls reciprocal verify), and its proof layout is the 928-byte u64 form: the specialised
// kernels (closed-form scalars, affine window tables, fixed-base tables shared with the generic path) take such calls -- an order of
// magnitude faster than the generic kernels, same verdicts and statuses.  BPPP_GENERIC_U64_SHAPE=1 keeps the generic kernels (A/B, tests).
static bool recip_is_u64_shape(const bppp_ctx* c, size_t dim_nd, size_t dim_np, size_t rounds, size_t nl, size_t nn) {
    return !c->generic_u64_shape && c->ng == 16 && c->nh == 32 && dim_nd == 16 && dim_np == 16 && rounds == 4 && nl == 2 && nn == 1;
}
int recip_verify_device_entry(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                              const void* d_commitments, const void* d_proofs, size_t rounds, size_t nl, size_t nn, void* d_accept,
                              void* d_status, const uint8_t* rlc_seed, void* d_reject_count) {
    if (!c || !label_ok(label, label_len) || !d_commitments || !d_proofs || !d_accept || !d_status) return BPPP_ERR_INVALID_ARG;
    int rc = recip_verify_check_args(c, dim_nd, dim_np, rounds, nl, nn);
    if (rc != BPPP_OK) return rc;
    if (recip_is_u64_shape(c, dim_nd, dim_np, rounds, nl, nn))
        return verify_device_impl(c, label, label_len, n, d_commitments, d_proofs, d_accept, d_status, nullptr, d_reject_count, rlc_seed, nullptr);
    HIP_TRY(hipSetDevice(c->device));
    if (d_reject_count) HIP_TRY(hipMemsetAsync(d_reject_count, 0, sizeof(int), c->stream));
    if (n == 0) return BPPP_OK;
    rc = ensure_straus_capacity(c, n);
    if (rc != BPPP_OK) return rc;
    // A call that leaves the chip under-filled runs as K PARTS on K streams.  One lane per instance gives 2^15 instances of BASELINE
    // configs[4]'s shape 512 wavefronts on 1,024 SIMDs for phase 1, the C0 tables and sum, the round tables and the eight rounds -- a
    // third of the step waiting on lone wavefronts' dependent chains -- while the two fixed-base sums (8 lanes per instance: 769 and 263
    // bases) fill it.  Instances are independent, so the parts need no ordering among themselves: one part's fixed-base sums run under
    // another part's one-lane kernels, and the wavefront slots the chains leave idle do the sums' work.
    const int K = generic_parts_for(c, n, rlc_seed != nullptr);
    if (K > 1) {
        rc = bppp_ensure_twin_lanes(c);
        if (rc != BPPP_OK) return rc;
        size_t m[4], lo[4], ws_off[4], gt_off[4], ws_total = 0, gt_total = 0;
        const size_t per = ((n + (size_t)K - 1) / (size_t)K + BPPP_BLOCK - 1) / BPPP_BLOCK * BPPP_BLOCK;
        int parts = 0;
        for (size_t a = 0; a < n; a += per, parts++) {
            lo[parts] = a; m[parts] = n - a < per ? n - a : per;
            ws_off[parts] = ws_total; ws_total += align16(recip_verify_ws_bytes(c, m[parts], dim_nd, dim_np, rounds, false)) + 256;
            gt_off[parts] = gt_total; gt_total += wnla_fast_bytes(m[parts], rounds, 5) + 256;
        }
        rc = ensure_buffer(c, c->d_gws, c->gws_bytes, ws_total);
        if (rc != BPPP_OK) return rc;
        const bool fast = rounds != 0 && !c->generic_slow_rounds;
        if (fast) {
            rc = ensure_buffer(c, c->d_gtab, c->gtab_bytes, gt_total);
            if (rc != BPPP_OK) return rc;
        }
        hipStream_t streams[4] = {c->stream, c->twin_stream, c->aux_stream, c->twin_aux};
        hipEvent_t joins[4] = {nullptr, c->ev_twin_join, c->ev2_fork, c->ev2_join};
        hipEvent_t started[4] = {c->ev_tab, c->ev_fork, c->ev_join, nullptr};      // (free here: a part's kernels are one chain on one stream)
        const int stage = c->generic_stagger;
        const size_t proof_bytes = 64 * (5 + 2 * rounds) + 32 * (nl + nn);
        const unsigned call_blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
        HIP_TRY(hipEventRecord(c->ev_twin_fork, c->stream));
        for (int i = 1; i < parts; i++) HIP_TRY(hipStreamWaitEvent(streams[i], c->ev_twin_fork, 0));
        int rc_parts = BPPP_OK;
        for (int i = 0; i < parts && rc_parts == BPPP_OK; i++) {
            const GenericPart gp = {streams[i], fast ? c->d_gtab + gt_off[i] : nullptr, c->d_straus + lo[i] * 5 * BPPP_STRAUS_ENTRIES, call_blocks,
                                    stage && i + 1 < parts ? started[i] : nullptr, stage};
            if (stage && i > 0) HIP_TRY(hipStreamWaitEvent(streams[i], started[i - 1], 0));
            rc_parts = recip_verify_device_impl(c, label, label_len, m[i], dim_nd, dim_np, (const uint8_t*)d_commitments + 64 * lo[i],
                                                (const uint8_t*)d_proofs + proof_bytes * lo[i], rounds, nl, nn, (uint8_t*)d_accept + lo[i],
                                                (int32_t*)d_status + lo[i], c->d_gws + ws_off[i], nullptr, nullptr, &gp);
        }
        // (joined even after a failed launch: nothing of this call may outlive it on a stream the caller does not know)
        for (int i = 1; i < parts; i++) {
            HIP_TRY(hipEventRecord(joins[i], streams[i]));
            HIP_TRY(hipStreamWaitEvent(c->stream, joins[i], 0));
        }
        rc = rc_parts;
    } else {
        // persistent, grow-only workspace (the host-pointer entry point allocates per call instead)
        const size_t need = recip_verify_ws_bytes(c, n, dim_nd, dim_np, rounds, rlc_seed != nullptr);
        rc = ensure_buffer(c, c->d_gws, c->gws_bytes, need);
        if (rc != BPPP_OK) return rc;
        rc = recip_verify_device_impl(c, label, label_len, n, dim_nd, dim_np, (const uint8_t*)d_commitments, (const uint8_t*)d_proofs, rounds, nl,
                                      nn, (uint8_t*)d_accept, (int32_t*)d_status, c->d_gws, nullptr, rlc_seed);
    }
    if (rc != BPPP_OK || !d_reject_count) return rc;
    k_count_rejects<<<(unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), 256, 0, c->stream>>>((const uint8_t*)d_accept, n, (int*)d_reject_count);
    HIP_TRY(hipGetLastError());
    return BPPP_OK;
}
int bppp_reciprocal_verify_batch_device(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                        const void* d_commitments, const void* d_proofs, size_t rounds, size_t nl, size_t nn,
                                        void* d_accept, void* d_status) {
    CtxLock lock_(c);
    return recip_verify_device_entry(c, label, label_len, n, dim_nd, dim_np, d_commitments, d_proofs, rounds, nl, nn, d_accept, d_status, nullptr, nullptr);
}
int bppp_reciprocal_verify_batch_rlc_device(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                            const void* d_commitments, const void* d_proofs, size_t rounds, size_t nl, size_t nn,
                                            void* d_accept, void* d_status, const uint8_t seed[32]) {
    CtxLock lock_(c);
    if (!seed) return BPPP_ERR_INVALID_ARG;
    return recip_verify_device_entry(c, label, label_len, n, dim_nd, dim_np, d_commitments, d_proofs, rounds, nl, nn, d_accept, d_status, seed, nullptr);
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
    if (!c || !label_ok(label, label_len) || !commitments || !proofs || !accept) return BPPP_ERR_INVALID_ARG;
    int rc = recip_verify_check_args(c, dim_nd, dim_np, rounds, nl, nn);
    if (rc != BPPP_OK) return rc;
    if (n == 0) return BPPP_OK;
    rc = check_host_transcripts(tx, n);
    if (rc != BPPP_OK) return rc;
    if (recip_is_u64_shape(c, dim_nd, dim_np, rounds, nl, nn)) {      // the u64 protocol under its generic name: the specialised path
        if (tx) return bppp_u64_verify_batch_transcript(c, n, tx->states, tx->n_states, commitments, proofs, accept, status, tx->states_out);
        if (rlc_seed) return bppp_u64_verify_batch_rlc(c, label, label_len, n, commitments, proofs, accept, status, rlc_seed);
        return bppp_u64_verify_batch(c, label, label_len, n, commitments, proofs, accept, status);
    }
    HIP_TRY(hipSetDevice(c->device));
    rc = ensure_straus_capacity(c, n);
    if (rc != BPPP_OK) return rc;
    const size_t proof_bytes = 64 * (5 + 2 * rounds) + 32 * (nl + nn);
    const size_t o_com = 0, o_pr = align16(n * 64), o_acc = align16(o_pr + n * proof_bytes), o_st = align16(o_acc + n),
                 o_ti = align16(o_st + n * 4), o_to = align16(o_ti + (tx ? tx->n_states * 203 : 0)),
                 o_ws = align16(o_to + (tx && tx->states_out ? n * 203 : 0)),
                 total = o_ws + recip_verify_ws_bytes(c, n, dim_nd, dim_np, rounds, rlc_seed != nullptr);
    WnlaBlob blob;
    { const int rc_b = blob.take(c, total); if (rc_b != BPPP_OK) return rc_b; }
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
    hipError_t e = ctx_malloc(c, (void**)&q->d_blob, off);
    if (e != hipSuccess) {
        delete q;
        g_last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
        (void)hipGetLastError();
        return e == hipErrorOutOfMemory ? BPPP_ERR_NOMEM : BPPP_ERR_HIP;
    }
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
                                    int32_t* status, const HostTranscripts* tx, bool device_io = false);
// ArithmeticCircuit::verify over DEVICE buffers (commitments n x k x 64, proofs, accept n, status n or null), asynchronous on the
// context's stream: the resident form of bppp_circuit_verify_batch
int bppp_circuit_verify_batch_device(bppp_ctx* c, const bppp_circuit* q, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments,
                                     const void* d_proofs, size_t rounds, size_t nl, size_t nn, void* d_accept, void* d_status) {
    CtxLock lock_(c);
    return circuit_verify_host_impl(c, q, label, label_len, n, (const uint8_t*)d_commitments, (const uint8_t*)d_proofs, rounds, nl, nn,
                                    (uint8_t*)d_accept, (int32_t*)d_status, nullptr, true);
}
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
                                    int32_t* status, const HostTranscripts* tx, bool device_io) {
    if (!c || !q || !label_ok(label, label_len) || !commitments || !proofs || !accept) return BPPP_ERR_INVALID_ARG;
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
    if (device_io) { const int rc_b = ensure_blob(c, off + 16); if (rc_b != BPPP_OK) return rc_b; }       // (no sync when the call returns)
    else { const int rc_b = blob.take(c, off + 16); if (rc_b != BPPP_OK) return rc_b; }
    uint8_t* d = c->d_blob;
    hipStream_t s = c->stream;
    if (tx) HIP_TRY(hipMemcpyAsync(d + o_ti, tx->states, tx->n_states * 203, hipMemcpyHostToDevice, s));
    if (!device_io) {
        HIP_TRY(hipMemcpyAsync(d + o_com, commitments, n * k * 64, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d + o_pr, proofs, n * proof_bytes, hipMemcpyHostToDevice, s));
    }
    CircuitWs r;
    std::memset(&r, 0, sizeof r);
    r.N = n; r.cd = cd; r.rounds = (int)rounds; r.NG = c->ng; r.NH = c->nh; r.proof_bytes = proof_bytes;
    r.commitments = device_io ? commitments : d + o_com; r.proofs = device_io ? proofs : d + o_pr;
    r.status = (device_io && status) ? status : (int32_t*)(d + o_st); r.tstate = (u32*)(d + o_ts);
    r.lamv = (u32*)(d + o_lam); r.muv = (u32*)(d + o_muv); r.coef = (u32*)(d + o_coef); r.sc0 = (u32*)(d + o_sc0); r.pts = (u32*)(d + o_pts);
    r.acc = (u32*)(d + o_a); r.pfix = (u32*)(d + o_pf);
    r.straus = c->d_straus;
    r.wn_commit = d + o_wc; r.wn_c = d + o_wcv; r.wn_rho = d + o_rho; r.wn_mu = d + o_mu;
    r.fb = fb_table_of(c, n);
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
    w.accept = device_io ? accept : d + o_acc; w.status = r.status; w.tstate = r.tstate; w.acc = r.acc; w.pfix = r.pfix;
    w.ys = (u32*)(d + o_ys); w.tab = (u32*)(d + o_tab); w.msc = (u32*)(d + o_msc);
    w.straus = c->d_straus;
    w.fb = r.fb;
    const unsigned blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    const unsigned fb_blocks = (unsigned)((n * BPPP_FB_LANES + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
#define CLAUNCH(id, ...)                                       \
    do {                                                       \
        rc = timed(c, id, s, [&]() { __VA_ARGS__; });          \
        if (rc != BPPP_OK) return rc;                          \
    } while (0)
    {   // the WNLA stage's table buffer with room for the 4 + k points of C0's variable-base part behind the round points' tables
        int rcf = wnla_fast_setup(c, w, n, rounds, 4 + k, nullptr, (size_t)wnla_table_parts(c, n, rounds));
        if (rcf != BPPP_OK) return rcf;
        r.atab = w.atab; r.tscr = w.tscr; r.atab_first = (int)(2 * rounds * 16 * (size_t)w.tab_parts);
    }
    CLAUNCH(K_CIRCUIT_PHASE1, k_circuit_phase1<<<blocks, BPPP_BLOCK, 0, s>>>(r));
    if (generic_fb_wide(c, n)) CLAUNCH(K_CIRCUIT_C0_FIXED, k_circuit_c0_fixed_l64<<<fb64_blocks_of(n), BPPP_FB_BLOCK, 0, s>>>(r));
    else CLAUNCH(K_CIRCUIT_C0_FIXED, k_circuit_c0_fixed<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(r));
    {
        // C0's variable-base sum: a lane per point (L lanes per instance, tables and sum in one launch) while that stays within two
        // wavefronts per SIMD, else the one-lane kernels (five points per shared-doubling pass).  Round 6, `mixed_k2` (6 points): one
        // verify 4.74 -> 2.76 ms (this stage 3.0 -> 0.98), 8,192 instances 1.37 ms where 16,384 on the one-lane kernels take 2.64
        int L = 8;
        while (L < 4 + (int)k) L *= 2;
        const bool per_point = r.atab && !c->no_lane_groups && !c->no_split && L <= 64 && (size_t)L * blocks <= 2 * (size_t)c->n_simds;
        CLAUNCH(K_CIRCUIT_C0_VAR, {
            if (per_point) k_circuit_c0_var_pts<<<(unsigned)(((size_t)L * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(r, L);
            else {
                if (r.atab) k_circuit_c0_tables<<<blocks, BPPP_BLOCK, 0, s>>>(r);
                k_circuit_c0_var<<<blocks, BPPP_BLOCK, 0, s>>>(r);
            }
        });
    }
    CLAUNCH(K_CIRCUIT_C0_FINISH, k_circuit_c0_finish<<<blocks, BPPP_BLOCK, 0, s>>>(r));
    CLAUNCH(K_WNLA_BEGIN, k_wnla_begin<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    if (w.atab) CLAUNCH(K_WNLA_TABLES, launch_wnla_tables(w, n, blocks, s));
    {
        const int grp = wnla_round_group(c, w, blocks);
        for (int kk = 1; kk <= (int)rounds; kk++) {
            if (grp > 1) CLAUNCH(K_WNLA_ROUND, k_wnla_round_grp<<<(unsigned)(((size_t)grp * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(w, kk, grp));
            else CLAUNCH(K_WNLA_ROUND, k_wnla_round<<<blocks, BPPP_BLOCK, 0, s>>>(w, kk));
        }
    }
    CLAUNCH(K_WNLA_FINAL_SCALARS, launch_wnla_final_scalars(c, w, (unsigned)rounds, n, blocks, s));
    if (generic_fb_wide(c, n)) CLAUNCH(K_WNLA_MSM, k_wnla_msm_l64<<<fb64_blocks_of(n), BPPP_FB_BLOCK, 0, s>>>(w));
    else CLAUNCH(K_WNLA_MSM, k_wnla_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 0));
    CLAUNCH(K_WNLA_ACCEPT, k_wnla_accept<<<blocks, BPPP_BLOCK, 0, s>>>(w));
#undef CLAUNCH
    if (w.tio.states_out) k_generic_export_states<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    HIP_TRY(hipGetLastError());
    if (device_io) return BPPP_OK;
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
    { const int rc_b = blob.take(c, off); if (rc_b != BPPP_OK) return rc_b; }
    uint8_t* d = blob.d;
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(d + o_sc, scalars, n * nterms * 32, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + o_runs, runs.data(), runs.size() * 4, hipMemcpyHostToDevice, s));
    MsmWs w;
    std::memset(&w, 0, sizeof w);
    w.N = n; w.nterms = (int)nterms; w.nruns = (int)(runs.size() / 3);
    w.scalars = d + o_sc; w.runs = (const int*)(d + o_runs); w.msc = (u32*)(d + o_msc); w.pfix = (u32*)(d + o_pf);
    w.status = (int32_t*)(d + o_st); w.out = d + o_out;
    w.fb = fb_table_of(c, n);
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
// "ct_prover" for the generic provers: the 4-bit table over THIS context's generators (built at the first use) for the sums over secret
// scalars -- every entry of every window read and selected by mask, complete additions (fb_core.h: fb_lookup_add_ct)
static int ct_setup(bppp_ctx* c, FbTable& fb_ct, int& ct, size_t n) {
    ct = 0;
    if (!c->ct_prover) return BPPP_OK;
    const int rc = ensure_ct_table(c);
    if (rc != BPPP_OK) return rc;
    fb_ct.table = c->d_table_ct; fb_ct.W = 4; fb_ct.N = n;
    ct = 1;
    return BPPP_OK;
}
static int wnla_prove_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, const HostTranscripts* tx, size_t n, const uint8_t* commitments,
                           const uint8_t* cvec, const uint8_t* rho, const uint8_t* mu, const uint8_t* l, size_t nl, const uint8_t* nvec, size_t nn,
                           uint8_t* proof_r, uint8_t* proof_x, uint8_t* proof_l, uint8_t* proof_n, int32_t* status) {
    if (!c || !label_ok(label, label_len) || !commitments || !cvec || !rho || !mu || (!l && nl) || (!nvec && nn) || nl > 65536 || nn > 65536)
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
    { const int rc_b = blob.take(c, off); if (rc_b != BPPP_OK) return rc_b; }
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
    { const int rc_s = ensure_straus_capacity(c, n); if (rc_s != BPPP_OK) return rc_s; }   // window tables of X, R (next commitment by the relation)
    w.straus = c->d_straus;
    w.N = n; w.ng = c->ng; w.nh = c->nh; w.nl = (int)nl; w.nn = (int)nn; w.rounds = (int)rounds; w.nl_f = (int)nl_f; w.nn_f = (int)nn_f;
    w.commitments = d + o_com; w.c = d + o_c; w.rho = d + o_rho; w.mu = d + o_mu; w.l_in = d + o_l; w.n_in = d + o_n;
    w.proof_r = d + o_pr; w.proof_x = d + o_px; w.proof_l = d + o_pl; w.proof_n = d + o_pn;
    w.status = (int32_t*)(d + o_st); w.tstate = (u32*)(d + o_ts); w.vl = (u32*)(d + o_vl); w.vn = (u32*)(d + o_vn); w.vc = (u32*)(d + o_vc);
    w.ch = (u32*)(d + o_ch); w.cg = (u32*)(d + o_cg); w.prm = (u32*)(d + o_prm); w.com = (u32*)(d + o_cm); w.msc = (u32*)(d + o_msc);
    w.pbuf = (u32*)(d + o_pb);
    w.fb = fb_table_of(c, n);
    { const int rc_ct = ct_setup(c, w.fb_ct, w.ct, n); if (rc_ct != BPPP_OK) return rc_ct; }      // l, n are the caller's secrets here (wnla.rs:152-160)
    t_new(w.base, label, (u32)label_len);
    TxDev txd;
    int rc = txd.begin(c, tx, n, s, w.tio, w.divergent_positions);
    if (rc != BPPP_OK) return rc;
    w.tio.no_ops = rounds == 0;
    const unsigned blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    const unsigned fb_blocks = (unsigned)((n * BPPP_FB_LANES + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    k_wprove_init<<<blocks, BPPP_BLOCK, 0, s>>>(w);
    for (int k = 0; k < (int)rounds; k++) {
        k_wprove_round_scalars<<<blocks, BPPP_BLOCK, 0, s>>>(w, k);
        k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 0, -1);
        k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 1, k);
        k_wprove_round_fold<<<blocks, BPPP_BLOCK, 0, s>>>(w, k);
        if (k == 0 && rounds > 1) k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 2, -1);   // level 1's commitment; later levels by the relation
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
    if (!c || !q || !label_ok(label, label_len) || !v_commitments || !v || !s_v || !w_l || !w_r || !rnd || !proofs) return BPPP_ERR_INVALID_ARG;
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
    { const int rc_b = blob.take(c, off); if (rc_b != BPPP_OK) return rc_b; }
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
    p.fb = fb_table_of(c, n);
    { const int rc_ct = ct_setup(c, p.fb_ct, p.ct, n); if (rc_ct != BPPP_OK) return rc_ct; }      // c_l, c_r, c_o, c_s: witness and blindings (circuit.rs:336-345, 469-470)
    t_new(p.base, label, (u32)label_len);
    TxDev txd;
    int rc = txd.begin(c, tx, n, s, p.tio, p.divergent_positions);
    if (rc != BPPP_OK) return rc;
    WnlaProveWs w;
    std::memset(&w, 0, sizeof w);
    { const int rc_s = ensure_straus_capacity(c, n); if (rc_s != BPPP_OK) return rc_s; }   // window tables of X, R (next commitment by the relation)
    w.straus = c->d_straus;
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
        k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 0, -1);
        k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 1, kk);
        k_wprove_round_fold<<<blocks, BPPP_BLOCK, 0, s>>>(w, kk);
        if (kk == 0 && rounds > 1) k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 2, -1);   // level 1's commitment; later levels by the relation
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
    if (!c || !label_ok(label, label_len) || !commitments || !x || !sblind || !digits || !m || !rnd || !proofs) return BPPP_ERR_INVALID_ARG;
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
    { const int rc_b = blob.take(c, off); if (rc_b != BPPP_OK) return rc_b; }
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
    r.fb = fb_table_of(c, n);
    { const int rc_ct = ct_setup(c, r.fb_ct, r.ct, n); if (rc_ct != BPPP_OK) return rc_ct; }      // the reciprocals 1 / (e + d_i) (reciprocal.rs:118)
    t_new(r.base, label, (u32)label_len);
    TxDev txd;
    int rc = txd.begin(c, tx, n, s, r.tio, r.divergent_positions);
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
    p.fb = r.fb; p.fb_ct = r.fb_ct; p.ct = r.ct;
    WnlaProveWs w;
    std::memset(&w, 0, sizeof w);
    { const int rc_s = ensure_straus_capacity(c, n); if (rc_s != BPPP_OK) return rc_s; }   // window tables of X, R (next commitment by the relation)
    w.straus = c->d_straus;
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
        k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 0, -1);
        k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 1, kk);
        k_wprove_round_fold<<<blocks, BPPP_BLOCK, 0, s>>>(w, kk);
        if (kk == 0 && rounds > 1) k_wprove_msm<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(w, 2, -1);   // level 1's commitment; later levels by the relation
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


}  // extern "C"
