// libbppp_hip.so, host side: launch sequences of the u64 range-proof verifier (exact and RLC modes), commit_value and the batch prover.
// One lane per proof; 64-thread workgroups (one wavefront) so that a 2^16-proof batch yields 1024 workgroups (4 per CU) and no
// lane ever waits on a workgroup barrier.  The per-lane work is in verify_core.h / prove_core.h.
#include "host.h"
#include "plan_core.h"

// host-buffer verify calls: each pipelined part is this many times the previous one (what crosses PCIe while a part is verified)
#define BPPP_HOST_PART_GROWTH 7
#if defined(BPPP_PHASE_TIMING)
extern unsigned long long* g_bppp_stamps_dev;      // bppp_ctx.hip
#endif
static_assert(bppp_host::PLAN_BLOCK == BPPP_BLOCK, "plan_core.h counts workgroups of BPPP_BLOCK lanes");
// the switches of a context that the plans depend on
struct VerifyLanes { hipStream_t s, a; hipEvent_t ev_fork, ev_join, ev_tab; hipEvent_t ev_started; };      // a: null = no helper stream; ev_started: recorded after the first kernel (or null)
// the second stream pair of a twin call (plan_core.h: twin), created at the first such call
int bppp_ensure_twin_lanes(bppp_ctx* c) {
    if (c->twin_stream) return BPPP_OK;
    // The second chain's streams.  What kind they are made no measurable difference (normal priority, high priority, a CU mask of all
    // CUs -- i.e. a hardware queue of their own: profiles/r06/r06_e_*, r06_f): the twin form's failures were never queue aliasing but
    // the two chains falling into step (bppp_u64.hip: verify_device_part).  High priority is kept: such streams come from another set of
    // hardware queues than the context's own whatever else lives in the process, and the second chain -- started later -- then gets the
    // wavefront slots first when both have a kernel waiting.
    int prio_least = 0, prio_greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    const int kind = c->twin_stream_kind;                                            // 0 normal priority | 1 high priority | 2 a CU mask of all CUs (a queue of its own)
    if (kind == 2) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, c->device));
        std::vector<uint32_t> mask((size_t)(prop.multiProcessorCount + 31) / 32, 0xFFFFFFFFu);
        if (prop.multiProcessorCount % 32) mask.back() = (1u << (prop.multiProcessorCount % 32)) - 1u;
        HIP_TRY(hipExtStreamCreateWithCUMask(&c->twin_stream, (uint32_t)mask.size(), mask.data()));
        HIP_TRY(hipExtStreamCreateWithCUMask(&c->twin_aux, (uint32_t)mask.size(), mask.data()));
    } else {
        const int prio = kind == 1 ? prio_greatest : 0;
        HIP_TRY(hipStreamCreateWithPriority(&c->twin_stream, hipStreamNonBlocking, prio));
        HIP_TRY(hipStreamCreateWithPriority(&c->twin_aux, hipStreamNonBlocking, prio));
    }
    for (hipEvent_t* e : {&c->ev_twin_fork, &c->ev_twin_join, &c->ev2_fork, &c->ev2_join, &c->ev2_tab}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return BPPP_OK;
}
static bppp_host::PlanKnobs knobs_of(const bppp_ctx* c) {
    bppp_host::PlanKnobs k;
    k.n_simds = c->n_simds;
    k.no_small = c->no_small; k.no_lane_groups = c->no_lane_groups; k.no_split = c->no_split; k.timing = c->timing;
    k.tables_beside = c->tables_beside; k.tail_beside = c->tail_beside; k.fb_one_lane_mode = c->fb_one_lane_mode; k.next_overlap = c->next_overlap;
    k.shared_inv = c->shared_inv; k.twin = c->twin; k.pace = c->pace;
    k.next_msm_max = c->next_msm_max; k.lane_forms_max = c->lane_forms_max; k.lane4_max = c->lane4_max; k.scal_parts_max = c->scal_parts_max;
    return k;
}

// 1 / in[t] for all t < n, one inversion per G elements (k_verify_misc.hip)
static void launch_fe_batch_inv(int G, const u32* in, u32* out, size_t n, hipStream_t s) {
    const size_t lanes = (n + (size_t)G - 1) / (size_t)G;
    const unsigned b = (unsigned)((lanes + BPPP_BLOCK - 1) / BPPP_BLOCK);
    switch (G) {
    case 16: k_verify_shared_inv16<<<b, BPPP_BLOCK, 0, s>>>(in, out, n); break;
    case 8: k_verify_shared_inv8<<<b, BPPP_BLOCK, 0, s>>>(in, out, n); break;
    case 4: k_verify_shared_inv4<<<b, BPPP_BLOCK, 0, s>>>(in, out, n); break;
    default: k_verify_shared_inv2<<<b, BPPP_BLOCK, 0, s>>>(in, out, n); break;
    }
}

extern "C" {

static int verify_device_part(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments,
                              const void* d_proofs, void* d_accept, void* d_status, void* d_trace, void* d_reject_count,
                              const uint8_t* rlc_seed, const VerifyTranscripts* tx, bool reset_reject_count);
// One call = one batch for the caller; internally a batch larger than max_batch proofs runs as consecutive parts on the same
// stream, so the per-proof workspace (~30 KB per proof) is bounded by max_batch whatever n is.  Proofs are independent, the reject
// counter accumulates across parts, and every per-proof array is simply offset.
int verify_device_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments, const void* d_proofs,
                       void* d_accept, void* d_status, void* d_trace, void* d_reject_count, const uint8_t* rlc_seed,
                       const VerifyTranscripts* tx) {
    if (!c) return BPPP_ERR_INVALID_ARG;
    if (!d_commitments || !d_proofs || !d_accept)
        return verify_device_part(c, label, label_len, n, d_commitments, d_proofs, d_accept, d_status, d_trace, d_reject_count, rlc_seed, tx, true);
    if (tx && tx->d_states && tx->n_states != 1 && tx->n_states != n) return BPPP_ERR_INVALID_ARG;
    const size_t SB = BPPP_TRANSCRIPT_STATE_BYTES;
    // max_batch is a guess made from the memory that was free when the context was created; another context, another process or the
    // caller's own allocations may have taken it since.  A part whose workspace cannot be allocated is therefore not the end of the
    // call: what the failed attempt holds is released, the part size is halved (and stays halved for the calls that follow) and the
    // part is run again.  Nothing of a part is visible before its last kernel (k_verify_accept*: accept bytes, reject count), and
    // every allocation of a part precedes that kernel, so a retried part leaves no trace of its first attempt.
    const size_t min_part = (size_t)1 << 12;
    for (size_t lo = 0; lo < n || (n == 0 && lo == 0);) {
        const size_t cap = c->max_batch;
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
        if (rc == BPPP_ERR_NOMEM && m > min_part) {
            quiesce(c);
            release_workspaces(c);
            size_t half = (m / 2 + BPPP_BLOCK - 1) / BPPP_BLOCK * BPPP_BLOCK;
            if (half < min_part) half = min_part;
            c->max_batch = half;
            continue;
        }
        if (rc != BPPP_OK) return rc;
        if (n == 0) break;
        lo += m;
    }
    return BPPP_OK;
}
static int verify_device_part(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments,
                              const void* d_proofs, void* d_accept, void* d_status, void* d_trace, void* d_reject_count,
                              const uint8_t* rlc_seed, const VerifyTranscripts* tx, bool reset_reject_count) {
    if (!c || !label_ok(label, label_len) || !d_commitments || !d_proofs || !d_accept) return BPPP_ERR_INVALID_ARG;
    if (c->ng != 16 || c->nh != 32) return BPPP_ERR_INVALID_ARG;   // u64 entry points need the u64 generator shape
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_capacity(c, n);
    if (rc != BPPP_OK) return rc;
    // which kernels this size runs: ONE pure function of (n, SIMDs, switches) -- plan_core.h, where the regimes are described
    const bppp_host::VerifyPlan plan = bppp_host::plan_verify(n, knobs_of(c), rlc_seed != nullptr);
    c->last_verify_plan = plan.code();
    rc = ensure_vtab_capacity(c, plan.vtab_sets * n);
    if (rc != BPPP_OK) return rc;
    VerifyWs ws;
    std::memset(&ws, 0, sizeof ws);
    carve(c, ws, n);
    ws.atab = c->d_atab;
    ws.tscr = c->d_tscr;
    if (plan.shared_inv) ws.zinv = c->d_zinv;
#if defined(BPPP_PHASE_TIMING)
    if (!g_bppp_stamps_dev) {
        HIP_TRY(hipMalloc(&g_bppp_stamps_dev, sizeof(unsigned long long) * BPPP_STAMP_WAVES * 32));
        HIP_TRY(hipMemset(g_bppp_stamps_dev, 0, sizeof(unsigned long long) * BPPP_STAMP_WAVES * 32));
    }
    ws.stamps = g_bppp_stamps_dev;
    ws.stamp_stride = (unsigned)(((n + BPPP_BLOCK - 1) / BPPP_BLOCK + BPPP_STAMP_WAVES - 1) / BPPP_STAMP_WAVES);
    if (ws.stamp_stride == 0) ws.stamp_stride = 1;
#endif
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
    // the launch sequence of one batch -- or of one half of a twin call -- on one stream pair
    auto sequence = [&](const VerifyWs& ws, size_t n, const bppp_host::VerifyPlan& plan, const VerifyLanes& L) -> int {
    int rc = BPPP_OK;
    const unsigned blocks = (unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    hipStream_t s = L.s;
#define LAUNCH_ON(st, id, ...)                                  \
    do {                                                        \
        rc = timed(c, id, st, [&]() { __VA_ARGS__; });          \
        if (rc != BPPP_OK) return rc;                           \
    } while (0)
#define LAUNCH(id, ...) LAUNCH_ON(s, id, __VA_ARGS__)
    // (the helper stream: not with kernel timing on, not for a chain of a twin call, and not where the fixed-base sums run one lane per
    // proof (fb = l1, from 128 S proofs): there both halves of C0 fill the chip by themselves, and side by side they take LONGER than one
    // after the other -- round 6: 2^20 proofs 143.3 ms with the halves on two streams against 141.65 back to back, events included)
    hipStream_t a = (c->timing || !L.a || plan.fb == bppp_host::FB_L1) ? s : L.a;
    // (tables: a lane per point on the helper stream while phase 1 runs, or the one-lane kernel beside phase 1 -- both decode their points
    // themselves -- or after phase 1 on the main stream)
    const bool tables_aside = plan.tables == bppp_host::TABLES_ASIDE, tables_beside = plan.tables == bppp_host::TABLES_BESIDE;
    const int tparts = plan.tparts;
    const unsigned wg4_blocks = (unsigned)((n + BPPP_C0VAR_SMALL_BLOCK - 1) / BPPP_C0VAR_SMALL_BLOCK);
    if (tables_beside) {
        HIP_TRY(hipEventRecord(L.ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(a, L.ev_fork, 0));
        LAUNCH_ON(a, K_TABLES, k_verify_tables_own<<<wg4_blocks, BPPP_C0VAR_SMALL_BLOCK, 0, a>>>(ws));
        HIP_TRY(hipEventRecord(L.ev_tab, a));
    }
    if (tables_aside) {
        // the table kernel decodes its points itself, so it runs on the helper stream beside phase 1
        HIP_TRY(hipEventRecord(L.ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(a, L.ev_fork, 0));
        const unsigned tb = (unsigned)((16 * (size_t)tparts * n + BPPP_BLOCK - 1) / BPPP_BLOCK);
        if (tparts == 4) LAUNCH_ON(a, K_TABLES, k_verify_tables_split4<<<tb, BPPP_BLOCK, 0, a>>>(ws));
        else if (tparts == 2) LAUNCH_ON(a, K_TABLES, k_verify_tables_split2<<<tb, BPPP_BLOCK, 0, a>>>(ws));
        else LAUNCH_ON(a, K_TABLES, k_verify_tables_split1<<<tb, BPPP_BLOCK, 0, a>>>(ws));
        HIP_TRY(hipEventRecord(L.ev_tab, a));
    }
    switch (plan.phase1) {
    case bppp_host::P1_G16: LAUNCH(K_PHASE1, k_verify_phase1_g16<<<(unsigned)((16 * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(ws)); break;
    case bppp_host::P1_WG4: LAUNCH(K_PHASE1, k_verify_phase1_wg4<<<wg4_blocks, BPPP_C0VAR_SMALL_BLOCK, 0, s>>>(ws)); break;
    case bppp_host::P1_SMALL: LAUNCH(K_PHASE1, k_verify_phase1_small<<<blocks, BPPP_BLOCK, 0, s>>>(ws)); break;
    default: LAUNCH(K_PHASE1, k_verify_phase1<<<blocks, BPPP_BLOCK, 0, s>>>(ws)); break;
    }
    if (L.ev_started) HIP_TRY(hipEventRecord(L.ev_started, s));
    const unsigned fb_blocks = (unsigned)((n * BPPP_FB_LANES + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    // C0 = variable-base half (window tables of the proof points, then the shared-doubling sum: one lane per proof, 1 wave
    // per SIMD) + fixed-base half (8 lanes per proof): independent, so they run concurrently on two streams and share the
    // SIMDs; round 1 adds the halves.
    // with per-kernel timing on, the two halves run back to back so that the kernel times add up to the step (overlapped, each
    // half's events also cover the other's share of the SIMDs; the overlap itself buys nothing at 2 waves/SIMD: DESIGN.md 4)
    if (tables_aside || tables_beside) HIP_TRY(hipStreamWaitEvent(s, L.ev_tab, 0));
    else if (plan.shared_inv) {
        // five passes, the running product of each inverted once per G proofs in between (in place in ws.zinv)
        LAUNCH(K_TABLES, k_verify_tables_pass0<<<blocks, BPPP_BLOCK, 0, s>>>(ws));
        LAUNCH(K_SHARED_INV, launch_fe_batch_inv(plan.shared_inv, ws.zinv, ws.zinv, n, s));
        LAUNCH(K_TABLES, k_verify_tables_pass1<<<blocks, BPPP_BLOCK, 0, s>>>(ws));
        LAUNCH(K_SHARED_INV, launch_fe_batch_inv(plan.shared_inv, ws.zinv, ws.zinv, n, s));
        LAUNCH(K_TABLES, k_verify_tables_pass2<<<blocks, BPPP_BLOCK, 0, s>>>(ws));
        LAUNCH(K_SHARED_INV, launch_fe_batch_inv(plan.shared_inv, ws.zinv, ws.zinv, n, s));
        LAUNCH(K_TABLES, k_verify_tables_pass3<<<blocks, BPPP_BLOCK, 0, s>>>(ws));
        LAUNCH(K_SHARED_INV, launch_fe_batch_inv(plan.shared_inv, ws.zinv, ws.zinv, n, s));
        LAUNCH(K_TABLES, k_verify_tables_pass4<<<blocks, BPPP_BLOCK, 0, s>>>(ws));
    } else LAUNCH(K_TABLES, k_verify_tables<<<blocks, BPPP_BLOCK, 0, s>>>(ws));
    HIP_TRY(hipEventRecord(L.ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(a, L.ev_fork, 0));
    // the two fixed-base sums: a wavefront per sum in a small call, 8 lanes per proof, or one from the size at which one lane per proof
    // fills the SIMDs twice over
    const unsigned fb1_blocks = (unsigned)((n + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    const unsigned fb64_blocks = (unsigned)((n * 64 + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    if (plan.fb == bppp_host::FB_L64) LAUNCH_ON(a, K_C0_FIXED, k_verify_c0_fixed_l64<<<fb64_blocks, BPPP_FB_BLOCK, 0, a>>>(ws));
    else if (plan.fb == bppp_host::FB_L1) LAUNCH_ON(a, K_C0_FIXED, k_verify_c0_fixed_l1<<<fb1_blocks, BPPP_FB_BLOCK, 0, a>>>(ws));
    else LAUNCH_ON(a, K_C0_FIXED, k_verify_c0_fixed<<<fb_blocks, BPPP_FB_BLOCK, 0, a>>>(ws));
    HIP_TRY(hipEventRecord(L.ev_join, a));
    // the variable-base half: 64 / 32 lanes per proof in a small call, lane groups of 4 while the grid leaves the SIMDs under-filled,
    // else one lane per proof (uncapped build: 256-thread workgroups, their four wavefronts land one per SIMD -- see k_verify_var.hip)
    const unsigned g4_blocks = (unsigned)((4 * n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    switch (plan.c0var) {
    case bppp_host::C0V_G64: LAUNCH(K_C0_VAR, k_verify_c0_var_g64<<<(unsigned)((64 * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(ws)); break;
    case bppp_host::C0V_G32: LAUNCH(K_C0_VAR, k_verify_c0_var_g32<<<(unsigned)((32 * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(ws)); break;
    case bppp_host::C0V_G4: LAUNCH(K_C0_VAR, k_verify_c0_var_g4<<<g4_blocks, BPPP_BLOCK, 0, s>>>(ws)); break;
    case bppp_host::C0V_SMALL: LAUNCH(K_C0_VAR, k_verify_c0_var_small<<<wg4_blocks, BPPP_C0VAR_SMALL_BLOCK, 0, s>>>(ws)); break;
    default: LAUNCH(K_C0_VAR, k_verify_c0_var<<<blocks, BPPP_BLOCK, 0, s>>>(ws)); break;
    }
    HIP_TRY(hipStreamWaitEvent(s, L.ev_join, 0));
    // The last round's two-point sum beside the final fixed-base sum (which needs the challenges, not C_4): for batches whose one-lane
    // kernels are a lone wavefront per SIMD and not on lane groups (2^15 < n <= 2^16), exact mode.  The round then goes out as head and
    // tail (k_verify_var.hip); the final scalars and the final sum follow the head on the helper stream; k_verify_accept waits for both.
    const bool tail_beside = plan.tail_beside;
    const int parts = plan.parts;
    (void)parts;
    for (int k = 1; k <= 4; k++) {
        if (k == 4 && tail_beside) {
            LAUNCH(K_ROUND, k_verify_round_head_small<<<blocks, BPPP_BLOCK, 0, s>>>(ws, k));
            HIP_TRY(hipEventRecord(L.ev_fork, s));
            HIP_TRY(hipStreamWaitEvent(a, L.ev_fork, 0));
            LAUNCH(K_ROUND, k_verify_round_tail<<<wg4_blocks, BPPP_C0VAR_SMALL_BLOCK, 0, s>>>(ws, k));
            continue;
        }
        switch (plan.round) {
        case bppp_host::R_G16: LAUNCH(K_ROUND, k_verify_round_g16<<<(unsigned)((16 * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(ws, k)); break;
        case bppp_host::R_G8: LAUNCH(K_ROUND, k_verify_round_g8<<<(unsigned)((8 * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(ws, k)); break;
        case bppp_host::R_G4: LAUNCH(K_ROUND, k_verify_round_g4<<<g4_blocks, BPPP_BLOCK, 0, s>>>(ws, k)); break;
        case bppp_host::R_G2: LAUNCH(K_ROUND, k_verify_round_g2<<<(unsigned)((2 * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(ws, k)); break;
        case bppp_host::R_SMALL: LAUNCH(K_ROUND, k_verify_round_small<<<blocks, BPPP_BLOCK, 0, s>>>(ws, k)); break;
        default:
            if (plan.shared_inv) {
                // 1 / Z of C_{k-1} for every proof from n / G inversions (round 1: after the halves of C0 are added)
                LAUNCH(K_SHARED_INV, {
                    if (k == 1) k_verify_c0_join<<<blocks, BPPP_BLOCK, 0, s>>>(ws);
                    launch_fe_batch_inv(plan.shared_inv, ws.acc + 20 * n, ws.zinv, n, s);
                });
                LAUNCH(K_ROUND, k_verify_round<<<blocks, BPPP_BLOCK, 0, s>>>(ws, k));
            } else LAUNCH(K_ROUND, k_verify_round<<<blocks, BPPP_BLOCK, 0, s>>>(ws, k));
            break;
        }
    }
    // the 49 unrolled generator coefficients: sixteen lanes per proof while that still leaves the chip under-filled
    hipStream_t fs = tail_beside ? a : s;      // where the final scalars and the final sum go
    if (plan.final_scalars_g16) LAUNCH(K_FINAL_SCALARS, k_verify_final_scalars_g16<<<(unsigned)((16 * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(ws));
    else LAUNCH_ON(fs, K_FINAL_SCALARS, k_verify_final_scalars<<<blocks, BPPP_BLOCK, 0, fs>>>(ws));
    if (!rlc_seed) {
        if (plan.fb == bppp_host::FB_L64) LAUNCH(K_FINAL_CHECK, k_verify_final_check_l64<<<fb64_blocks, BPPP_FB_BLOCK, 0, s>>>(ws));
        else if (plan.fb == bppp_host::FB_L1) LAUNCH_ON(fs, K_FINAL_CHECK, k_verify_final_check_l1<<<fb1_blocks, BPPP_FB_BLOCK, 0, fs>>>(ws));
        else LAUNCH_ON(fs, K_FINAL_CHECK, k_verify_final_check<<<fb_blocks, BPPP_FB_BLOCK, 0, fs>>>(ws));
        if (tail_beside) { HIP_TRY(hipEventRecord(L.ev_join, a)); HIP_TRY(hipStreamWaitEvent(s, L.ev_join, 0)); }
        LAUNCH(K_ACCEPT, k_verify_accept<<<blocks, BPPP_BLOCK, 0, s>>>(ws, (int*)d_reject_count));
    } else {
        // How the final checks are grouped -- superchunks of the bucket stage, then chunks of 8 or 32 -- follows what the previous RLC
        // call on this context rejected (plan_core.h: plan_rlc).  That call's counter was copied to pinned host memory when it ended;
        // if the copy has landed, it is the rate this call plans with.
        if (c->rlc_hist_n && c->ev_rlc_hist && hipEventQuery(c->ev_rlc_hist) == hipSuccess) {
            c->rlc_rate = (double)*c->h_rlc_hist / (double)c->rlc_hist_n;
            c->rlc_hist_n = 0;
        }
        (void)hipGetLastError();      // (hipErrorNotReady from the query is not an error of this call)
        const bppp_host::RlcPlan rp = bppp_host::plan_rlc(bucket_superchunk_for(c, n), c->rlc_super_auto, c->rlc_chunk_opt, c->rlc_rate);
        c->last_rlc_super_m = rp.super_m; c->last_rlc_chunk = rp.chunk;
        rl.chunk = rp.chunk;
        // combined check per chunk; chunks that fail it (or hold a flagged proof) fall through to the exact kernels
        const size_t nchunks = (n + rp.chunk - 1) / rp.chunk;
        const unsigned chunk_blocks = (unsigned)((nchunks * rp.chunk + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
        HIP_TRY(hipMemsetAsync(d_accept, 0, n, s));
        HIP_TRY(hipMemsetAsync(rl.count, 0, sizeof(int), s));
        rc = ensure_rlc_history(c);
        if (rc != BPPP_OK) return rc;
        HIP_TRY(hipMemsetAsync(c->d_rlc_hist, 0, sizeof(int), s));
        if (const unsigned SM = rp.super_m) {
            // bucket stage first: superchunks of SM proofs, one combined check each; the chunk-of-8 kernels below only see the proofs
            // of superchunks that failed it
            BucketWs bw;
            rc = launch_bucket_stage(c, bw, n, SM, rl.seed, ws.status, ws.acc, ws.fsc, BPPP_NG, ws.accept, s,
                                     [&](int id, auto&& f) { return timed(c, id, s, f); });
            if (rc != BPPP_OK) return rc;
            rl.sflag = bw.sflag;
            rl.super_m = SM;
        }
        LAUNCH(K_RLC_LHS, k_rlc_lhs<<<blocks, BPPP_BLOCK, 0, s>>>(ws, rl));
        if (rp.chunk == 32) LAUNCH(K_RLC_CHUNK, k_rlc_chunk_c32<<<chunk_blocks, BPPP_FB_BLOCK, 0, s>>>(ws, rl));
        else LAUNCH(K_RLC_CHUNK, k_rlc_chunk<<<chunk_blocks, BPPP_FB_BLOCK, 0, s>>>(ws, rl));
        LAUNCH(K_FINAL_CHECK, k_verify_final_check_flagged<<<1024, 64, 0, s>>>(ws, rl));
        LAUNCH(K_FINAL_CHECK, k_verify_final_check_flagged_l8<<<2 * (unsigned)c->n_simds / 4, BPPP_FB_BLOCK, 0, s>>>(ws, rl));
        LAUNCH(K_FINAL_CHECK, k_verify_final_check_flagged_dense<<<fb_blocks, BPPP_FB_BLOCK, 0, s>>>(ws, rl));
        LAUNCH(K_ACCEPT, k_verify_accept_flagged<<<blocks, BPPP_BLOCK, 0, s>>>(ws, rl, (int*)d_reject_count, c->d_rlc_hist));
        // this call's reject count, for the next call's plan (a call that runs in parts leaves the last part's)
        HIP_TRY(hipMemcpyAsync(c->h_rlc_hist, c->d_rlc_hist, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(c->ev_rlc_hist, s));
        c->rlc_hist_n = n;
    }
#undef LAUNCH
#undef LAUNCH_ON
    return BPPP_OK;
    };
    if (plan.twin == 2) {
        // two halves, two stream pairs (plan_core.h: twin): the second half starts when the call's inputs are ready on the main stream
        // and is joined back into it, so callers see one asynchronous call on c->stream as before
        rc = bppp_ensure_twin_lanes(c);
        if (rc != BPPP_OK) return rc;
        const size_t n0 = bppp_host::twin_first_half(n), n1 = n - n0;
        const bppp_host::VerifyPlan ph0 = bppp_host::plan_verify_half(n0, knobs_of(c), plan.pace), ph1 = bppp_host::plan_verify_half(n1, knobs_of(c), plan.pace);
        c->last_verify_plan = ph0.code();        // what a half runs (twin=2 says there are two of them)
        VerifyWs w0 = ws, w1 = ws;
        u32* z0 = carve_from(c, w0, n0, c->d_ws);
        u32* z1 = carve_from(c, w1, n1, c->d_ws + WS_WORDS_PER_PROOF * n0);
        w0.zinv = ph0.shared_inv ? z0 : nullptr;
        w1.zinv = ph1.shared_inv ? z1 : nullptr;
        w0.pace = ph0.pace; w1.pace = ph1.pace;
        w1.atab = ws.atab + (size_t)BPPP_ATAB_PER_PROOF * n0;
        w1.tscr = ws.tscr + (size_t)(BPPP_TSCR_FE * 10) * n0;
        w1.commitments = ws.commitments + 64 * n0;
        w1.proofs = ws.proofs + (size_t)BPPP_U64_PROOF_BYTES * n0;
        w1.accept = ws.accept + n0;
        w1.status = ws.status + n0;
        if (ws.trace) w1.trace = ws.trace + (size_t)BPPP_U64_TRACE_BYTES * n0;
        if (ws.states && ws.n_states != 1) { w0.n_states = n0; w1.states = ws.states + (size_t)BPPP_TRANSCRIPT_STATE_BYTES * n0; w1.n_states = n1; }
        if (ws.states_out) w1.states_out = ws.states_out + (size_t)BPPP_TRANSCRIPT_STATE_BYTES * n0;
        // Each half is ONE chain of kernels on ONE stream (the fixed-base half of C0 ahead of the variable-base half: no helper streams),
        // and the second chain starts when the first has finished its first kernel.  Every SIMD then holds an older wavefront of one
        // chain and a younger one of the other; the arbiter serves the older first, it ends early, its chain's next kernel moves in as
        // the younger one: the chains leapfrog.  Started TOGETHER the two would mix old and young wavefronts of both chains on the
        // SIMDs, every kernel would take what the slow role takes, both chains would stay in step for good (measured: the rounds of both
        // halves starting within 30 us of each other, 1.95 ms each instead of 1.75 with 0.9 ms between them) and the split would buy nothing.
        const VerifyLanes L0 = {c->stream, nullptr, c->ev_fork, c->ev_join, c->ev_tab, c->ev_twin_fork};
        const VerifyLanes L1 = {c->twin_stream, nullptr, c->ev2_fork, c->ev2_join, c->ev2_tab, nullptr};
        rc = sequence(w0, n0, ph0, L0);
        if (rc == BPPP_OK) {
            HIP_TRY(hipStreamWaitEvent(c->twin_stream, c->ev_twin_fork, 0));
            rc = sequence(w1, n1, ph1, L1);
        }
        if (rc == BPPP_OK && ws.states_out) {
            k_verify_export_states<<<(unsigned)((n0 + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, c->stream>>>(w0);
            k_verify_export_states<<<(unsigned)((n1 + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, c->twin_stream>>>(w1);
        }
        // (joined even after a failed launch: nothing of this call may outlive it on a stream the caller does not know)
        HIP_TRY(hipEventRecord(c->ev_twin_join, c->twin_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_twin_join, 0));
        if (rc != BPPP_OK) return rc;
        HIP_TRY(hipGetLastError());
        return BPPP_OK;
    }
    {
        const VerifyLanes L0 = {c->stream, c->aux_stream, c->ev_fork, c->ev_join, c->ev_tab, nullptr};
        ws.pace = plan.pace;
        rc = sequence(ws, n, plan, L0);
        if (rc != BPPP_OK) return rc;
    }
    if (ws.states_out) k_verify_export_states<<<(unsigned)((n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, c->stream>>>(ws);
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
    const size_t SB = BPPP_TRANSCRIPT_STATE_BYTES;
    const size_t o_c = 0, o_p = align16(o_c + n * 64), o_a = align16(o_p + n * (size_t)BPPP_U64_PROOF_BYTES), o_s = align16(o_a + n),
                 o_ti = align16(o_s + n * sizeof(int32_t)), o_to = align16(o_ti + n_states * SB), total = align16(o_to + n * SB);
    int rc = ensure_io(c, total);
    if (rc != BPPP_OK) return rc;
    uint8_t* d = c->d_io;
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
    if (e != hipSuccess || rc != BPPP_OK) quiesce(c);
    if (e != hipSuccess) { g_last_error = std::string("verify_batch_transcript: ") + hipGetErrorString(e); return BPPP_ERR_HIP; }
    return rc;
}
int bppp_u64_verify_batch_device(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments,
                                 const void* d_proofs, void* d_accept, void* d_status, void* d_trace, void* d_reject_count) {
    CtxLock lock_(c);
    return verify_device_impl(c, label, label_len, n, d_commitments, d_proofs, d_accept, d_status, d_trace, d_reject_count, nullptr, nullptr);
}
int bppp_u64_verify_batch_rlc_device(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments,
                                     const void* d_proofs, void* d_accept, void* d_status, void* d_reject_count, const uint8_t seed[32]) {
    CtxLock lock_(c);
    if (!seed) return BPPP_ERR_INVALID_ARG;
    return verify_device_impl(c, label, label_len, n, d_commitments, d_proofs, d_accept, d_status, nullptr, d_reject_count, seed, nullptr);
}

static int verify_host_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                            const uint8_t* proofs, uint8_t* accept, int32_t* status, const uint8_t* rlc_seed) {
    if (!c || !commitments || !proofs || !accept) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    // persistent I/O staging of the host-buffer entry points (grow-only; separate from d_stage, which the device call may use)
    const size_t o_c = 0, o_p = align16(o_c + n * 64), o_a = align16(o_p + n * (size_t)BPPP_U64_PROOF_BYTES), o_s = align16(o_a + n),
                 need = align16(o_s + n * sizeof(int32_t));
    {
        int rc = ensure_io(c, need);
        if (rc != BPPP_OK) return rc;
    }
    uint8_t *d_c = c->d_io + o_c, *d_p = c->d_io + o_p, *d_a = c->d_io + o_a;
    int32_t* d_s = (int32_t*)(c->d_io + o_s);
    const size_t CH = c->host_chunk;
    if (CH == 0 || n <= CH + CH / 2) {
        HIP_TRY(hipMemcpyAsync(d_c, commitments, n * 64, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_p, proofs, n * (size_t)BPPP_U64_PROOF_BYTES, hipMemcpyHostToDevice, c->stream));
        int rc = verify_device_impl(c, label, label_len, n, d_c, d_p, d_a, d_s, nullptr, nullptr, rlc_seed, nullptr);
        if (rc != BPPP_OK) return rc;
    } else {
        // Large batch: proofs are independent, so the batch is verified in parts while the next part crosses PCIe on a second stream.
        // Round 4 cut it into equal chunks of host_chunk proofs and lost 9 of the 13 ms it was behind a resident batch to the chunks
        // themselves: eight 2^17-proof batches take 158 ms where one 2^20-proof batch takes 149 (the tails of kernels that fill the chip
        // exactly once).  The link moves a proof (992 B at 56 GB/s: 18 ns) eight times faster than the chip verifies it (142 ns), so the
        // parts GROW: the first is one host_chunk -- the only upload nothing hides -- and each next one is what can be uploaded while the
        // previous one is verified, 7 times its size: 2^20 proofs = 2^17 + 7 * 2^17, the second part a 917,504-proof batch at the
        // resident rate (profiles/r05/r05_d_hostpath_probe.txt).  (From pageable memory the runtime stages the copy and this thread blocks
        // in it, at the same 56 GB/s; the GPU keeps verifying meanwhile.)
        if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        if (!c->ev_copy) HIP_TRY(hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming));
        // the staging buffer may still be read by kernels of an earlier call on c->stream only if that call returned early on an
        // error; order the first upload after whatever is queued there
        HIP_TRY(hipEventRecord(c->ev_copy, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->ev_copy, 0));
        size_t part = CH;
        for (size_t lo = 0; lo < n;) {
            size_t m = n - lo;
            if (m > part + part / 2) m = part;          // a tail of less than half a part joins the last part rather than running as a sliver
            HIP_TRY(hipMemcpyAsync(d_c + lo * 64, commitments + lo * 64, m * 64, hipMemcpyHostToDevice, c->copy_stream));
            HIP_TRY(hipMemcpyAsync(d_p + lo * (size_t)BPPP_U64_PROOF_BYTES, proofs + lo * (size_t)BPPP_U64_PROOF_BYTES,
                                   m * (size_t)BPPP_U64_PROOF_BYTES, hipMemcpyHostToDevice, c->copy_stream));
            HIP_TRY(hipEventRecord(c->ev_copy, c->copy_stream));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_copy, 0));
            int rc = verify_device_impl(c, label, label_len, m, d_c + lo * 64, d_p + lo * (size_t)BPPP_U64_PROOF_BYTES, d_a + lo, d_s + lo,
                                        nullptr, nullptr, rlc_seed, nullptr);
            if (rc != BPPP_OK) return rc;
            lo += m;
            part = m * BPPP_HOST_PART_GROWTH;
            if (part > c->max_batch) part = c->max_batch;
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
    FbTable ct = {nullptr, 4, n};
    if (c->ct_prover) {
        rc = ensure_ct_table(c);
        if (rc != BPPP_OK) return rc;
        ct.table = c->d_table_ct;
    }
    k_commit_value<<<blocks, BPPP_BLOCK, 0, c->stream>>>(ws, d_x, d_s, d_o, c->d_flags, ct);
    HIP_TRY(hipGetLastError());
    int flags = 0;
    HIP_TRY(hipMemcpyAsync(out, d_o, n * 64, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&flags, c->d_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return flags ? BPPP_ERR_INVALID_ARG : BPPP_OK;
}

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
int prove_device_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_x, const void* d_s, const void* d_rnd,
                      void* d_proofs, void* d_commitments, void* d_status, const VerifyTranscripts* tx) {
    if (!c || !label_ok(label, label_len) || !d_x || !d_s || !d_rnd || !d_proofs || !d_commitments) return BPPP_ERR_INVALID_ARG;
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
    w.fb = fb_table_of(c, n);
    if (c->ct_prover) {
        rc = ensure_ct_table(c);
        if (rc != BPPP_OK) return rc;
        w.ct = 1;
        w.fb_ct.table = c->d_table_ct; w.fb_ct.W = 4; w.fb_ct.N = n;
    }
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
    // which kernels this size runs: ONE pure function of (n, SIMDs, switches) -- plan_core.h
    const bppp_host::ProvePlan plan = bppp_host::plan_prove(n, knobs_of(c), c->ct_prover);
    c->last_prove_plan = plan.code();
    const unsigned fb1_blocks = (unsigned)((n + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    // A level's commitment from fixed-base sums too: its even folded slots as one more sum (prove_core.h: job_e, 25 terms) fused with the
    // next round's X | R, its odd ones being that round's R.  That form wins while the variable-base one is a latency chain on an
    // under-filled chip: up to 32 values per SIMD (2^14 values: 300 more table additions per level against 0.8 ms of chain, of which the
    // helper stream hides a quarter: 10.5 -> 9.8 ms; 2^15: 17.65 -> 17.1; 2^16: 31.0 against 32.1, so not there).
    w.next_by_msm = plan.next_by_msm ? 1 : 0;
    const unsigned fb64_blocks = (unsigned)((n * 64 + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK);
    // The prover's fixed-base sums.  Lanes per proof: a wavefront in a small call; otherwise 8, 4 or 1 -- the fewest that still give
    // every SIMD two wavefronts in the launch (fewer lanes = fewer idle lanes in the short runs and a shorter tree of complete additions
    // per sum: 2^14 values 12.38 -> 12.03 ms with 4 lanes in the fused launches).  The independent sums of a stage go out as ONE launch
    // (blockIdx.y = job): no gap between them, and the short ones (c_o: 84 table additions per proof) fill in beside the long ones.
#define PMSMX(NJ, ...)                                                                                                     \
    do {                                                                                                                    \
        MsmJobs js = {{__VA_ARGS__}};                                                                                        \
        if (plan.fb == bppp_host::FB_L64) PLAUNCH(K_PROVE_MSM, k_prove_msm_l64x<<<dim3(fb64_blocks, NJ), BPPP_FB_BLOCK, 0, s>>>(w, js));          \
        else if (plan.fb == bppp_host::FB_L1) PLAUNCH(K_PROVE_MSM, k_prove_msm_l1x<<<dim3(fb1_blocks, NJ), BPPP_FB_BLOCK, 0, s>>>(w, js));   \
        else if (plan.fb4_from_jobs && (NJ) >= plan.fb4_from_jobs)                                                          \
            PLAUNCH(K_PROVE_MSM, k_prove_msm_l4x<<<dim3((unsigned)((n * 4 + BPPP_FB_BLOCK - 1) / BPPP_FB_BLOCK), NJ), BPPP_FB_BLOCK, 0, s>>>(w, js)); \
        else PLAUNCH(K_PROVE_MSM, k_prove_msm_x<<<dim3(fb_blocks, NJ), BPPP_FB_BLOCK, 0, s>>>(w, js));                       \
    } while (0)
#define PMSM(job) PMSMX(1, job, job, job, job)
    // the sums over the witness and its blindings: as above, or -- "ct_prover" -- in the form that reads every table entry of every window
#define PSECRETX(NJ, ...)                                                                                                  \
    do {                                                                                                                    \
        if (w.ct) {                                                                                                         \
            MsmJobs js = {{__VA_ARGS__}};                                                                                    \
            PLAUNCH(K_PROVE_MSM, k_prove_msm_ct<<<dim3(fb_blocks, NJ), BPPP_FB_BLOCK, 0, s>>>(w, js));                        \
        } else PMSMX(NJ, __VA_ARGS__);                                                                                      \
    } while (0)
#define PSECRET(job) PSECRETX(1, job, job, job, job)
    // a grid that gives every SIMD more than one wavefront runs the 256-register builds of the lane kernels (two wavefronts per SIMD)
    const bool w2 = plan.w2;
    PLAUNCH(K_PROVE_STAGES, k_prove_stage_a<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    PSECRET(job_v());
    if (w2) PLAUNCH(K_PROVE_STAGES, k_prove_stage_b_w2<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    else PLAUNCH(K_PROVE_STAGES, k_prove_stage_b<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    PSECRETX(4, job_rcom(), job_co(), job_cl(), job_cr());
    // a small call waits for one lane's chain: the stages' 16-term loops on sixteen lanes per value (prove_core.h: "lane forms"), on four
    // lanes per value while that still leaves SIMDs idle
    using bppp_host::ST_G16; using bppp_host::ST_G16_W2; using bppp_host::ST_G4; using bppp_host::ST_G4_W2;
    const bool stage_lanes = plan.stage == ST_G16 || plan.stage == ST_G16_W2, g16_w2 = plan.stage == ST_G16_W2 || plan.fold == ST_G16_W2;
    const bool stage_lanes4 = plan.stage == ST_G4 || plan.stage == ST_G4_W2, g4_w2 = plan.stage == ST_G4_W2 || plan.fold == ST_G4_W2;
    const bool fold_lanes = plan.fold == ST_G16 || plan.fold == ST_G16_W2, fold_lanes4 = plan.fold == ST_G4 || plan.fold == ST_G4_W2;
    const unsigned g16_blocks = (unsigned)((16 * n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    const unsigned g4_blocks = (unsigned)((4 * n + BPPP_BLOCK - 1) / BPPP_BLOCK);
    if (stage_lanes && g16_w2) PLAUNCH(K_PROVE_STAGES, k_prove_stage_d_g16_w2<<<g16_blocks, BPPP_BLOCK, 0, s>>>(w));
    else if (stage_lanes) PLAUNCH(K_PROVE_STAGES, k_prove_stage_d_g16<<<g16_blocks, BPPP_BLOCK, 0, s>>>(w));
    else if (stage_lanes4 && g4_w2) PLAUNCH(K_PROVE_STAGES, k_prove_stage_d_g4<2><<<g4_blocks, BPPP_BLOCK, 0, s>>>(w));
    else if (stage_lanes4) PLAUNCH(K_PROVE_STAGES, k_prove_stage_d_g4<1><<<g4_blocks, BPPP_BLOCK, 0, s>>>(w));
    else if (w2) PLAUNCH(K_PROVE_STAGES, k_prove_stage_d_w2<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    else PLAUNCH(K_PROVE_STAGES, k_prove_stage_d<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    PSECRET(job_cs());
    if (stage_lanes && g16_w2) PLAUNCH(K_PROVE_STAGES, k_prove_stage_f_g16_w2<<<g16_blocks, BPPP_BLOCK, 0, s>>>(w));
    else if (stage_lanes) PLAUNCH(K_PROVE_STAGES, k_prove_stage_f_g16<<<g16_blocks, BPPP_BLOCK, 0, s>>>(w));
    else if (stage_lanes4 && g4_w2) PLAUNCH(K_PROVE_STAGES, k_prove_stage_f_g4<2><<<g4_blocks, BPPP_BLOCK, 0, s>>>(w));
    else if (stage_lanes4) PLAUNCH(K_PROVE_STAGES, k_prove_stage_f_g4<1><<<g4_blocks, BPPP_BLOCK, 0, s>>>(w));
    else if (w2) PLAUNCH(K_PROVE_STAGES, k_prove_stage_f_w2<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    else PLAUNCH(K_PROVE_STAGES, k_prove_stage_f<<<blocks, BPPP_BLOCK, 0, s>>>(w));
    // (the lane-per-generator form of the scalar kernel pays only while the chip is empty: at 2^13 ... 2^15 values it costs 1.2 / 2.3 /
    // 4.5 ms per batch against 0.8: profiles/r04/r04_r_size_probe_wide_scalars.txt)
    const bool scal_wide = plan.scalars == bppp_host::SC_WIDE, scal_parts = plan.scalars == bppp_host::SC_PARTS;
    bool pending_cnext = false;
    // Beyond the sizes of next_by_msm: round k's next commitment C_k (prove_core.h: prove_round_next -- window tables of X and R and a
    // two-point GLV Straus sum, 125 dependent doublings) is not needed before round k + 1 appends it to the transcript: it runs on the
    // helper stream, under round k + 1's scalar kernel and X | R sums, and the main stream picks it up just before that round's fold.
    // On the helper stream it runs in its 256-register build: the uncapped one leaves no room on its SIMDs for a wavefront of the sums
    // it is meant to run under (profiles/r04/r04_zd_prove_next_overlap_probe.txt).  (With per-kernel timing on it stays on the main stream
    // so that the kernel times add up to the step.)
    hipStream_t a = plan.overlap_next ? c->aux_stream : s;
    bool next_in_flight = false;
    for (int k = 1; k <= 4; k++) {
        if (k > 1 && plan.fold_leaves_scalars) {}       // the previous round's lane-form fold left this round's scalars (prove_core.h: prove_round_fold_lanes*)
        else if (scal_wide) PLAUNCH(K_PROVE_ROUND_SCALARS, k_prove_round_scalars_wide<<<(unsigned)((64 * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, s>>>(w, k));
        else if (scal_parts) PLAUNCH(K_PROVE_ROUND_SCALARS, k_prove_round_scalars_parts<<<dim3(blocks, 4), BPPP_BLOCK, 0, s>>>(w, k));
        else PLAUNCH(K_PROVE_ROUND_SCALARS, k_prove_round_scalars<<<blocks, BPPP_BLOCK, 0, s>>>(w, k));
        if (pending_cnext || k == 1) PMSMX(3, job_x(), job_r(k), job_e(k), job_x());       // C_{k-1} = E + R (prove_core.h: job_e); C_0 always
        else PMSMX(2, job_x(), job_r(k), job_x(), job_x());
        if (next_in_flight) { HIP_TRY(hipStreamWaitEvent(s, c->ev_join, 0)); next_in_flight = false; }      // C_{k-1} is there
        if (fold_lanes && g16_w2) PLAUNCH(K_PROVE_ROUND_FOLD, k_prove_round_fold_g16_w2<<<g16_blocks, BPPP_BLOCK, 0, s>>>(w, k));
        else if (fold_lanes) PLAUNCH(K_PROVE_ROUND_FOLD, k_prove_round_fold_g16<<<g16_blocks, BPPP_BLOCK, 0, s>>>(w, k));
        else if (fold_lanes4 && g4_w2) PLAUNCH(K_PROVE_ROUND_FOLD, k_prove_round_fold_g4<2><<<g4_blocks, BPPP_BLOCK, 0, s>>>(w, k));
        else if (fold_lanes4) PLAUNCH(K_PROVE_ROUND_FOLD, k_prove_round_fold_g4<1><<<g4_blocks, BPPP_BLOCK, 0, s>>>(w, k));
        else if (w2) PLAUNCH(K_PROVE_ROUND_FOLD, k_prove_round_fold_w2<<<blocks, BPPP_BLOCK, 0, s>>>(w, k));
        else PLAUNCH(K_PROVE_ROUND_FOLD, k_prove_round_fold<<<blocks, BPPP_BLOCK, 0, s>>>(w, k));
        if (!w.next_by_msm && k < 4) {
            if (a != s) { HIP_TRY(hipEventRecord(c->ev_fork, s)); HIP_TRY(hipStreamWaitEvent(a, c->ev_fork, 0)); }
            const bool next_g4 = plan.next_g4;      // (only with BPPP_NEXT_MSM_MAX lowered: A/B runs)
            if (next_g4 && a != s)
                rc = timed(c, K_PROVE_ROUND_NEXT, a, [&]() { k_prove_round_next_g4_w2<<<(unsigned)((4 * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, a>>>(w, k); });
            else if (next_g4)
                rc = timed(c, K_PROVE_ROUND_NEXT, a, [&]() { k_prove_round_next_g4<<<(unsigned)((4 * n + BPPP_BLOCK - 1) / BPPP_BLOCK), BPPP_BLOCK, 0, a>>>(w, k); });
            else if (w2 || a != s)
                rc = timed(c, K_PROVE_ROUND_NEXT, a, [&]() { k_prove_round_next_w2<<<blocks, BPPP_BLOCK, 0, a>>>(w, k); });
            else
                rc = timed(c, K_PROVE_ROUND_NEXT, a, [&]() { k_prove_round_next<<<blocks, BPPP_BLOCK, 0, a>>>(w, k); });
            if (rc != BPPP_OK) return rc;
            if (a != s) { HIP_TRY(hipEventRecord(c->ev_join, a)); next_in_flight = true; }
        }
        pending_cnext = w.next_by_msm && k < 4;     // its scalars are in set 0 now; the sum rides with the next round's X | R
    }
    if (w.states_out) k_prove_export_states<<<blocks, BPPP_BLOCK, 0, s>>>(w);
#undef PSECRETX
#undef PSECRET
#undef PMSMX
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
    const size_t o_x = 0, o_s = align16(o_x + n * 8), o_r = align16(o_s + n * 32), o_p = align16(o_r + n * 52 * 32),
                 o_c = align16(o_p + n * (size_t)BPPP_U64_PROOF_BYTES), o_st = align16(o_c + n * 64), o_ti = align16(o_st + n * sizeof(int32_t)),
                 o_to = align16(o_ti + n_states * SB), total = align16(o_to + n * SB);
    int rc = ensure_io(c, total);
    if (rc != BPPP_OK) return rc;
    uint8_t* d = c->d_io;
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
    if (e != hipSuccess || rc != BPPP_OK) quiesce(c);
    if (e != hipSuccess) { g_last_error = std::string("prove_batch_transcript: ") + hipGetErrorString(e); return BPPP_ERR_HIP; }
    return rc;
}

int bppp_u64_prove_batch(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const uint64_t* x, const uint8_t* s,
                         const uint8_t* rnd, uint8_t* proofs, uint8_t* commitments, int32_t* status) {
    CtxLock lock_(c);
    if (!c || !x || !s || !rnd || !proofs || !commitments) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t o_x = 0, o_s = o_x + n * 8, o_r = o_s + n * 32, o_p = o_r + n * 52 * 32, o_c = o_p + n * (size_t)BPPP_U64_PROOF_BYTES,
                 o_st = o_c + n * 64, total = o_st + n * sizeof(int32_t);
    int rc = ensure_io(c, total);
    if (rc != BPPP_OK) return rc;
    uint8_t* d = c->d_io;
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
    if (e != hipSuccess || rc != BPPP_OK) quiesce(c);
    if (e != hipSuccess) { g_last_error = std::string("prove_batch: ") + hipGetErrorString(e); return BPPP_ERR_HIP; }
    return rc;
}

int verify_sec1_device_impl(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments33, const void* d_proofs525,
                            void* d_accept, void* d_status, void* d_trace, void* d_reject_count) {
    if (!c || !d_commitments33 || !d_proofs525 || !d_accept) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t need = n * (64 + (size_t)BPPP_U64_PROOF_BYTES);
    {
        int rc_e = ensure_buffer(c, c->d_expand, c->expand_bytes, need);
        if (rc_e != BPPP_OK) return rc_e;
    }
    uint8_t* d_c64 = c->d_expand;
    uint8_t* d_p928 = c->d_expand + n * 64;
    const unsigned blocks = (unsigned)((n * 16 + 255) / 256);
    k_sec1_expand<<<blocks, 256, 0, c->stream>>>(d_c64, d_p928, (const uint8_t*)d_commitments33, (const uint8_t*)d_proofs525, n);
    HIP_TRY(hipGetLastError());
    return verify_device_impl(c, label, label_len, n, d_c64, d_p928, d_accept, d_status, d_trace, d_reject_count, nullptr, nullptr);
}
int bppp_u64_verify_batch_sec1_device(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_commitments33,
                                      const void* d_proofs525, void* d_accept, void* d_status, void* d_trace, void* d_reject_count) {
    CtxLock lock_(c);
    return verify_sec1_device_impl(c, label, label_len, n, d_commitments33, d_proofs525, d_accept, d_status, d_trace, d_reject_count);
}

int bppp_u64_verify_batch_sec1(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments33,
                               const uint8_t* proofs525, uint8_t* accept, int32_t* status) {
    CtxLock lock_(c);
    if (!c || !commitments33 || !proofs525 || !accept) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t o_c = 0, o_p = o_c + n * 33, o_a = o_p + n * (size_t)BPPP_U64_PROOF_SEC1_BYTES, o_s = (o_a + n + 3) / 4 * 4,
                 total = o_s + n * sizeof(int32_t);
    int rc = ensure_io(c, total);
    if (rc != BPPP_OK) return rc;
    uint8_t* d = c->d_io;
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
    if (e != hipSuccess || rc != BPPP_OK) quiesce(c);
    if (e != hipSuccess) { g_last_error = std::string("verify_batch_sec1: ") + hipGetErrorString(e); return BPPP_ERR_HIP; }
    return rc;
}

// ---- the prover's output in the wire format (33-byte commitments, 525-byte proofs): proved into the context's 64-byte staging
//      (the buffer the SEC1 verify entry points expand into), compressed on the device
int bppp_u64_prove_batch_sec1_device(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const void* d_x, const void* d_s,
                                     const void* d_rnd, void* d_proofs525, void* d_commitments33, void* d_status) {
    CtxLock lock_(c);
    if (!c || !d_proofs525 || !d_commitments33) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t need = n * (64 + (size_t)BPPP_U64_PROOF_BYTES);
    {
        int rc_e = ensure_buffer(c, c->d_expand, c->expand_bytes, need);
        if (rc_e != BPPP_OK) return rc_e;
    }
    uint8_t* d_c64 = c->d_expand;
    uint8_t* d_p928 = c->d_expand + n * 64;
    int rc = prove_device_impl(c, label, label_len, n, d_x, d_s, d_rnd, d_p928, d_c64, d_status, nullptr);
    if (rc != BPPP_OK) return rc;
    k_sec1_compress<<<(unsigned)((n * 16 + 255) / 256), 256, 0, c->stream>>>((uint8_t*)d_commitments33, (uint8_t*)d_proofs525, d_c64, d_p928, n);
    HIP_TRY(hipGetLastError());
    return BPPP_OK;
}
int bppp_u64_prove_batch_sec1(bppp_ctx* c, const uint8_t* label, size_t label_len, size_t n, const uint64_t* x, const uint8_t* s,
                              const uint8_t* rnd, uint8_t* proofs525, uint8_t* commitments33, int32_t* status) {
    CtxLock lock_(c);
    if (!c || !x || !s || !rnd || !proofs525 || !commitments33) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t o_x = 0, o_s = o_x + n * 8, o_r = o_s + n * 32, o_p = o_r + n * 52 * 32, o_c = o_p + n * (size_t)BPPP_U64_PROOF_SEC1_BYTES,
                 o_st = (o_c + n * 33 + 3) / 4 * 4, total = o_st + n * sizeof(int32_t);
    int rc = ensure_io(c, total);
    if (rc != BPPP_OK) return rc;
    uint8_t* d = c->d_io;
    hipError_t e = hipMemcpyAsync(d + o_x, x, n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_s, s, n * 32, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_r, rnd, n * 52 * 32, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        rc = bppp_u64_prove_batch_sec1_device(c, label, label_len, n, d + o_x, d + o_s, d + o_r, d + o_p, d + o_c, d + o_st);
        if (rc == BPPP_OK) {
            e = hipMemcpyAsync(proofs525, d + o_p, n * (size_t)BPPP_U64_PROOF_SEC1_BYTES, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(commitments33, d + o_c, n * 33, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess && status) e = hipMemcpyAsync(status, d + o_st, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
    }
    if (e != hipSuccess || rc != BPPP_OK) quiesce(c);
    if (e != hipSuccess) { g_last_error = std::string("prove_batch_sec1: ") + hipGetErrorString(e); return BPPP_ERR_HIP; }
    return rc;
}

}  // extern "C"
