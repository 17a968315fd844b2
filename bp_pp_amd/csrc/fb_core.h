// Fixed-base multi-scalar multiplication over the batch-shared window tables: table geometry (windows of two widths sized to the bit),
// the lane loops of the 1- / 8- / 64-lane sums, the constant-address form for secret scalars ("ct_prover"), and the construction of
// the tables at context creation.  Every verifier and prover sum over the generators goes through here.  Split out of verify_core.h
// in round 6.
#pragma once
#include "verify_ws.h"

// The closures of the hot loops (produce / advance: the software pipeline's state lives in their captures) must be inlined whatever the
// inliner's budget says: left as calls, their captures -- the recoded scalar, the window pointer, the pipeline registers -- move to
// scratch memory and EVERY fixed-base kernel slows by 75 % (round 6: an experiment that gave fb_lane_accumulate_seq a second call site
// was enough, the u64 verifier's sums included: profiles/r06/r06_q_fb_blocks_and_lambda_inlining.txt).
#if defined(__clang__) || defined(__GNUC__)
#define BPPP_LAMBDA_INLINE __attribute__((always_inline))
#else
#define BPPP_LAMBDA_INLINE
#endif
namespace bppp {

// ---------------------------------------------------------------- fixed-base MSM over the batch-shared tables
// table[(b * nwin + w) * (2^W - 1) + (d - 1)] = d * 2^(W w) * generator_b, affine (64 B); (0,0) = identity.
// This is `vector_mul(points, scalars)` (util.rs:46-60) for points that are batch constants.
// Window geometry.  W in {4, 8, 16}: unsigned digits, 256/W windows, 2^W - 1 entries per window.
// W = 20: SIGNED digits in [-2^19, 2^19) (k + sum_i 2^(19+20i) has the digit + 2^19 in every 20-bit field), 13 windows,
// 2^19 entries per window (|d| = 1..2^19) and a conditional negation of y -- 13 instead of 16 additions per scalar for
// a 21 GB table; random 64-byte reads from a table of that size still run at ~19 G/s on MI355X (tools/probes/gatherbench.hip),
// above the ~12 G/s the arithmetic can consume.
// (W = 10 is the same signed scheme with a table small enough for the CPU emulation tests: 26 windows of 512 entries.)
// Optional second region (round 5): the generators below `hi_bases` -- g and g_vec, the 17 bases that BOTH fixed-base sums of a u64
// verify run over -- may live in `table_hi` at W_hi = 24 bits (11 additions per scalar instead of 12; 100 GB), and `table` then holds the
// bases hi_bases .. only, counted from 0.  hi_bases = 0: one table for every base, as before.
struct FbTable {
    const apt_packed* table; int W; size_t N;
    const apt_packed* table_hi; int W_hi; int hi_bases;
    int pace;        // 1: the one-lane sums pace their wave priority by progress (fb_pace; VerifyWs::pace >= 2)
};
// Progress-paced wave priority of the ONE-LANE fixed-base sums (see straus_core.h: straus_pace, for why): priority 3 -> 0 as the steps
// that remain drop below a half, a quarter, an eighth of the sum's total.  done / total count table additions of the whole sum.
HD void fb_pace(int done, int total) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int rem = total - done;
    if (2 * rem > total) __builtin_amdgcn_s_setprio(3);
    else if (4 * rem > total) __builtin_amdgcn_s_setprio(2);
    else if (8 * rem > total) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#else
    (void)done; (void)total;
#endif
}
// TEST HOOK, host emulation only (tests/emul, tests/test_ct_trace.py): every fixed-base table entry a sum requests, as the entry's index
// in its table.  The emulator records the sequence while a prover's SECRET sums run; the test requires it to be identical for two
// different secrets in the "ct_prover" forms, and different in the default ones.  Compiled out of the device code and of any host
// build that does not define BPPP_TRACE_TABLE_READS.
#if !defined(__HIPCC__) && defined(BPPP_TRACE_TABLE_READS)
void bppp_trace_table_read(const void* table, size_t index);
#define FB_TRACE(tab, idx) bppp_trace_table_read((const void*)(tab), (size_t)(idx))
#else
#define FB_TRACE(tab, idx) ((void)0)
#endif
// Window code of a table region (FbTable::W, W_hi; the library's "fb_window_bits"): Wb + 100 ka.  A scalar's windows are ka windows of
// Wb + 1 bits first (the low end), then windows of Wb bits, signed digits throughout; ka = 0 is a uniform table of Wb-bit windows (the
// only form of the unsigned widths 4, 8, 16).  With two widths the windows can be sized TO THE BIT: a signed recoding needs 258 bits of
// windows (256 + the carry of the offset + the top digit's sign), so n windows need Wb = floor(258 / n), ka = 258 - n Wb -- e.g. 523 =
// 5 x 24 + 6 x 23 bits: 11 table additions per scalar from 4.3 GB per generator, where 11 uniform windows (24 bits) take 5.9 GB and the
// 1.6 GB of 22-bit windows give 12.
HD int fb_wb(int code) { return code % 100; }
HD int fb_ka(int code) { return code / 100; }
HD bool fb_signed(int code) {
    const int W = fb_wb(code);
    return fb_ka(code) > 0 || W == 20 || W == 10 || W == 22 || W == 18 || W == 19 || W == 24;
}
HD int fb_nwin(int code) {                                       // uniform signed: ceil(257 / W) windows (>= 258 bits for every width in use)
    const int W = fb_wb(code), ka = fb_ka(code);
    if (!fb_signed(code)) return 256 / W;
    return ka ? (258 - ka + W - 1) / W : (257 + W - 1) / W;
}
HD size_t fb_per_narrow(int code) { const int W = fb_wb(code); return fb_signed(code) ? ((size_t)1 << (W - 1)) : (((size_t)1 << W) - 1); }   // entries of a Wb-bit window
HD size_t fb_per_base(int code) { return fb_per_narrow(code) * (size_t)(fb_nwin(code) + fb_ka(code)); }                                       // entries of one generator
HD int fb_pos(int code, int w) { const int ka = fb_ka(code); return fb_wb(code) * w + (w < ka ? w : ka); }                                    // first bit of window w
HD size_t fb_per_win_at(int code, int w) { return fb_per_narrow(code) << (w < fb_ka(code) ? 1 : 0); }
HD size_t fb_win_off(int code, int w) { const int ka = fb_ka(code); return fb_per_narrow(code) * (size_t)(w + (w < ka ? w : ka)); }           // entries of a generator before window w
HD FbTable fb_of(const VerifyWs& ws) { FbTable f = {ws.fb_table, ws.fb_w, ws.N, ws.fb_table_hi, ws.fb_w_hi, ws.fb_hi_bases, ws.pace >= 2 ? 1 : 0}; return f; }
// windows a scalar below 2^bits can reach (0 = full width).  Signed digits: the recoded value is sum d_i 2^(pos i) with d_i in
// [-2^(width_i - 1), 2^(width_i - 1)), and the top digit absorbs a carry of at most one, so the windows up to bit `bits` (inclusive)
// hold everything: ceil((bits + 1) / W) of a uniform table.
HD int fb_windows_for(int bits, int code) {
    const int all = fb_nwin(code);
    if (bits <= 0) return all;
    const int W = fb_wb(code), ka = fb_ka(code);
    if (!fb_signed(code)) { const int need = (bits + W - 1) / W; return need < all ? need : all; }
    int need = (bits + 1 + W) / (W + 1);                    // all of them wide ...
    if (need > ka) need = (bits + 1 - ka + W - 1) / W;      // ... or the ka wide ones and narrow ones for the rest
    return need < all ? need : all;
}
// index (within its run) of the a-th PRESENT term
HD int fb_term_index(int a, int oddsh) {
    if (oddsh < 0) return a;
    const int sh = oddsh & 15, odd = ((oddsh >> 4) & 1) ^ 1, B = 1 << sh;      // bit 4 (BPPP_FB_EVEN): the EVEN blocks instead
    return (((a >> sh) << 1) + odd) * B + (a & (B - 1));
}
// Geometry of a table region, derived ONCE per sum: inside the loops below a digit is a shift and a mask of a scalar that was recoded when
// its first window was reached, and a table address is an increment -- no division, no per-window recoding, no choice by window
// width (round 4's loop made that choice per step: 12.8 % of its dynamic instructions were scalar-unit bookkeeping).
struct FbGeom {
    const apt_packed* table;       // the region's entries: base b, window w at table[(b - base0) per_base + per_win (w + min(w, ka)) ...]
    int base0;
    int code;              // the region's window code (fb_wb: code = W + 100 ka)
    int W, ka, nwin;       // width of the narrow windows; wide (W + 1-bit) windows at the low end; windows of a full-width scalar
    u32 mask, half;        // of a narrow window: 2^W - 1; 2^(W-1) for signed digits (the digit is field - half), 0 for unsigned ones
    size_t per_win;        // entries of a narrow window (a wide one has twice as many)
    size_t per_base;       // entries of one generator
    u32 off[9];            // signed digits: sum_i 2^(top bit of window i) -- k + off carries digit + half in every field
};
HD bool fb_wide(const FbGeom& g, int w) { return w < g.ka; }
HD u32 fb_mask_at(const FbGeom& g, int w) { return fb_wide(g, w) ? ((g.mask << 1) | 1u) : g.mask; }
HD u32 fb_half_at(const FbGeom& g, int w) { return fb_wide(g, w) ? (g.half << 1) : g.half; }
HD void fb_geom_w(FbGeom& g, int code) {
    g.code = code;
    g.W = fb_wb(code);
    g.ka = fb_ka(code);
    g.nwin = fb_nwin(code);
    g.per_win = fb_per_narrow(code);
    g.per_base = fb_per_base(code);
    g.mask = (1u << g.W) - 1u;
    g.half = fb_signed(code) ? (1u << (g.W - 1)) : 0u;
#pragma unroll
    for (int l = 0; l < 9; l++) g.off[l] = 0;
    if (fb_signed(code)) {
#pragma nounroll
        for (int i = 0; i < g.nwin; i++) {
            const int bit = fb_pos(code, i + 1) - 1;         // the window's top bit
#pragma unroll
            for (int l = 0; l < 9; l++) g.off[l] |= (l == (bit >> 5)) ? (1u << (bit & 31)) : 0u;
        }
    }
}
HD void fb_geom(FbGeom& g, const FbTable& f, bool hi) {
    g.table = hi ? f.table_hi : f.table;
    g.base0 = hi ? 0 : f.hi_bases;
    fb_geom_w(g, hi ? f.W_hi : f.W);
}
HD bool fb_in_hi(const FbTable& f, int base) { return base < f.hi_bases; }
HD const apt_packed* fb_window(const FbGeom& g, int base, int w) {
    return g.table + (size_t)(base - g.base0) * g.per_base + g.per_win * (size_t)(w + (w < g.ka ? w : g.ka));
}
HD void fb_recode(u32 kp[9], const u32 k[8], const FbGeom& g) {      // kp = k + off, 9 limbs (< 2^264)
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) kp[i] = addc(k[i], g.off[i], c);
    kp[8] = g.off[8] + c;
}
// the field of window w of the recoded scalar (w differs from lane to lane: selects)
HD u32 fb_field(const u32 k[8], int w, const FbGeom& g) {
    u32 kp[10];
    fb_recode(kp, k, g);
    kp[9] = 0;
    const int bit = g.W * w + (w < g.ka ? w : g.ka), li = bit >> 5, sh = bit & 31;
    u32 lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { lo = (i == li) ? kp[i] : lo; hi = (i == li) ? kp[i + 1] : hi; }
    return (u32)((((u64)hi << 32) | lo) >> sh) & fb_mask_at(g, w);
}
// digit of window w: returns the table index (|d| - 1), whether to skip (d == 0) and whether to negate
HD void fb_digit(const u32 k[8], int W, int w, size_t& idx, bool& skip, bool& neg) {
    FbGeom g;
    fb_geom_w(g, W);
    const int d = (int)fb_field(k, w, g) - (int)fb_half_at(g, w);
    const u32 mag = (u32)(d < 0 ? -d : d);
    idx = mag ? (size_t)(mag - 1) : 0;
    skip = mag == 0;
    neg = d < 0;
}
// one table addition with the complete law (the provers' small sums, commit_value, and the re-do of a sum whose fast form met an
// exceptional addition)
HD void fb_lookup_add(pt& acc, const FbGeom& g, int base, int w, const u32 k[8]) {
    const int d = (int)fb_field(k, w, g) - (int)fb_half_at(g, w);
    const u32 mag = (u32)(d < 0 ? -d : d);
    const apt_packed* tb = fb_window(g, base, w);
    const size_t idx = mag ? (size_t)(mag - 1) : 0;
    apt e;
    bool id;
    FB_TRACE(g.table, (tb - g.table) + idx);
    apt_unpack(e, id, tb[idx]);
    fe ny;
    fe_neg_m<1>(ny, e.y);
    fe_cmov(e.y, d < 0, ny);
    pt_madd(acc, acc, e, (mag == 0) | id);
}
HD void fixed_base_msm(pt& accp, const FbTable& fbt, size_t t, const u32* scal, int first_slot, int first_base, int count, int bits = 0) {
    pt acc = accp;
#pragma nounroll
    for (int j = 0; j < count; j++) {
        FbGeom g;
        fb_geom(g, fbt, fb_in_hi(fbt, first_base + j));
        const int nwin = fb_windows_for(bits, g.code);  // bits > 0: the scalars are below 2^bits -- only the windows they can reach
        u32 k[8];
        ws_ld8(k, scal, fbt.N, t, first_slot + j);
#pragma nounroll
        for (int w = 0; w < nwin; w++) fb_lookup_add(acc, g, first_base + j, w, k);
    }
    accp = acc;
}

// ---- the same MSM split over BPPP_FB_LANES lanes per proof: lane `lane` takes every (base, window) pair whose window
// index is congruent to it, accumulates a partial sum, and the partial sums are tree-added across the lane group
// (wavefront shuffles on the device).  49 bases x 16 windows = 784 independent table additions per proof is where this
// path has intra-proof parallelism; it lifts the kernel from 1 to 4 resident wavefronts per SIMD at 2^16 proofs.
#define BPPP_FB_LANES 8
HD void fixed_base_msm_partial(pt& accp, const FbGeom& g, size_t N, size_t t, int lane, const u32* scal, int first_slot, int first_base,
                               int count, int nl = BPPP_FB_LANES, int bits = 0, int oddsh = -1) {
    const int nwin = fb_windows_for(bits, g.code);
    pt acc;
    pt_set_identity(acc);
    // the (term, window) pairs of the run, window-fastest, dealt round-robin over the lanes (the fast form's dealing: verify_core.h,
    // fb_lane_accumulate_fast)
    const int pairs = count * nwin;
#pragma nounroll
    for (int q = lane; q < pairs; q += nl) {
        const int a = q / nwin, w = q - a * nwin, j = fb_term_index(a, oddsh);
        u32 k[8];
        ws_ld8(k, scal, N, t, first_slot + j);
        fb_lookup_add(acc, g, first_base + j, w, k);
    }
    accp = acc;
}
// ---- fast form of the lane partial sums: XYZZ accumulator (point.h), 8M + 2S per table addition instead of 11M + 2m.  The
// law is incomplete; a lane that hit an exceptional addition reports it (fb_lane_finish_fast returns false) and the whole
// lane group re-does its sums with fixed_base_msm_partial.
// Table reads are random 64-byte gathers from a multi-GB table (HBM + TLB latency of microseconds), so the loop is software
// pipelined two deep: at the top of step i the table entry of step i+1 (address known) and the scalar words of step i+2 are
// requested, then the ~2000-instruction addition of step i runs, then the digit/address of step i+2 is derived.  The vector
// memory counter retires in order, so any load that is WAITED for before the addition would also wait for the table entry;
// fb_order_after() gives the scalar words a (fake) data dependency on the addition's result so that the compiler cannot
// place their use -- and with it the wait -- ahead of the addition.
HD void fb_sched_fence() {      // keeps the requests above the addition in the instruction stream
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
}
HD void fb_order_after(u32 k[8], const ptz& a) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]), "+v"(k[4]), "+v"(k[5]), "+v"(k[6]), "+v"(k[7])
                 : "v"(a.X.v[9]), "v"(a.Y.v[9]), "v"(a.ZZ.v[9]), "v"(a.ZZZ.v[9]));
#else
    (void)k;
    (void)a;
#endif
}
HD u32 funnel_shr(u32 hi, u32 lo, int sh) {      // low word of (hi:lo) >> sh, 0 < sh < 32: one v_alignbit_b32
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, (u32)sh);
#else
    return (u32)((((u64)hi << 32) | lo) >> sh);
#endif
}
struct FbStep {          // where a step's table entry lives, and how to use it
    const apt_packed* ptr;
    bool skip, neg;
};
HD void fb_step_from_field(FbStep& st, const apt_packed* win, u32 field, u32 half) {
    const int d = (int)field - (int)half;
    const u32 mag = (u32)(d < 0 ? -d : d);
    st.skip = mag == 0;
    st.neg = d < 0;
    st.ptr = win + (mag ? mag - 1u : 0u);
}
HD void fb_consume_fast(ptz& acc, bool& empty, const apt_packed& pe, bool skip, bool neg) {
    apt e;
    bool id;
    apt_unpack(e, id, pe);
    fe ny;
    fe_neg_m<1>(ny, e.y);
    fe_cmov(e.y, neg, ny);
    ptz_madd(acc, empty, e, skip | id);
}
// One lane per sum (nl == 1, every batch from 2^17 proofs up): the lane walks a scalar's windows in order, so the recoded scalar is
// a shift register -- recoded when the producer reaches the term, shifted right by W per step.  Term and window are wave-uniform
// here: the window's base address lives in scalar registers and moves by per_win entries per step.
HD void fb_lane_accumulate_seq(ptz& acc, bool& empty, const FbTable& fbt, const FbGeom& g, size_t t, const u32* scal, int first_slot,
                               int first_base, int count, int bits, int oddsh, int pace_done = 0, int pace_total = 0) {
    const int nw = fb_windows_for(bits, g.code);       // windows walked per scalar
    const int steps = count * nw;
    if (steps <= 0) return;
    // the producer hands out window pw of term pa next; past the last term it walks the last term again (requested, never consumed)
    int pa = 0, pw = 0;
    int j = fb_term_index(0, oddsh), jn = fb_term_index(count > 1 ? 1 : 0, oddsh);
    const apt_packed* win = fb_window(g, first_base + j, 0);
    u32 k[8], kp[9];
    ws_ld8(k, scal, fbt.N, t, first_slot + j);
    fb_recode(kp, k, g);
    auto produce = [&](FbStep& st) BPPP_LAMBDA_INLINE {
        if (pw == nw) {          // the scalar words of term jn were requested at the top of this step
            pa = pa + 1 < count ? pa + 1 : count - 1;
            j = jn;
            jn = fb_term_index(pa + 1 < count ? pa + 1 : count - 1, oddsh);
            win = fb_window(g, first_base + j, 0);
            fb_recode(kp, k, g);
            pw = 0;
        }
        const bool wide = pw < g.ka;                       // (wave-uniform: scalar registers)
        const int width = g.W + (wide ? 1 : 0);
        fb_step_from_field(st, win, kp[0] & (wide ? ((g.mask << 1) | 1u) : g.mask), wide ? (g.half << 1) : g.half);
#pragma unroll
        for (int i = 0; i < 8; i++) kp[i] = funnel_shr(kp[i + 1], kp[i], width);
        kp[8] >>= width;
        win += wide ? 2 * g.per_win : g.per_win;
        pw++;
    };
    FbStep cur_st, nxt_st;
    apt_packed cur_e, nxt_e;
    produce(cur_st);
    FB_TRACE(g.table, cur_st.ptr - g.table);
    cur_e = *cur_st.ptr;
    ws_ld8(k, scal, fbt.N, t, first_slot + jn);
    produce(nxt_st);
#pragma nounroll
    for (int i = 0; i < steps; i++) {
        if (pace_total && (i & 7) == 0) fb_pace(pace_done + i, pace_total);      // (wave-uniform; every 8th step)
        FB_TRACE(g.table, nxt_st.ptr - g.table);
        nxt_e = *nxt_st.ptr;                                    // step i+1's entry
        ws_ld8(k, scal, fbt.N, t, first_slot + jn);             // the producer's next scalar (used when step i+2 starts a term)
        fb_sched_fence();
        fb_consume_fast(acc, empty, cur_e, cur_st.skip, cur_st.neg);
        fb_order_after(k, acc);
        cur_e = nxt_e;
        cur_st = nxt_st;
        produce(nxt_st);
    }
}
// nl lanes per sum: lane `lane` takes the (term, window) pairs lane, lane + nl, lane + 2 nl, ... of the run (pairs counted window-fastest),
// so term and window differ from lane to lane: the step's scalar is recoded and its field picked by selects, the pair advances by
// (nl div nw, nl mod nw) with one conditional carry.
HD void fb_lane_accumulate_fast(ptz& acc, bool& empty, const FbTable& fbt, const FbGeom& g, size_t t, int lane, const u32* scal,
                                int first_slot, int first_base, int count, int nl, int bits, int oddsh, int pace_done = 0, int pace_total = 0) {
    if (nl == 1) {
        fb_lane_accumulate_seq(acc, empty, fbt, g, t, scal, first_slot, first_base, count, bits, oddsh, pace_done, pace_total);
        return;
    }
    const int nw = fb_windows_for(bits, g.code);
    const int pairs = count * nw;
    if (lane >= pairs) return;
    const int steps = (pairs - lane + nl - 1) / nl;
    const int da = nl / nw, dw = nl - da * nw;
    int a = lane / nw, w = lane - a * nw;
    auto advance = [&]() BPPP_LAMBDA_INLINE {      // steps past the end re-use the last one (requested, never consumed)
        int na = a + da, nwn = w + dw;
        if (nwn >= nw) { nwn -= nw; na++; }
        const bool in = na < count;
        a = in ? na : a;
        w = in ? nwn : w;
    };
    auto produce = [&](FbStep& st, const u32 k[8]) BPPP_LAMBDA_INLINE {      // k: the scalar of term a
        const int j = fb_term_index(a, oddsh);
        fb_step_from_field(st, fb_window(g, first_base + j, w), fb_field(k, w, g), fb_half_at(g, w));
    };
    u32 k[8];
    FbStep cur_st, nxt_st;
    apt_packed cur_e, nxt_e;
    // prologue: entry of step 0, address of step 1
    ws_ld8(k, scal, fbt.N, t, first_slot + fb_term_index(a, oddsh));
    produce(cur_st, k);
    FB_TRACE(g.table, cur_st.ptr - g.table);
    cur_e = *cur_st.ptr;
    advance();
    ws_ld8(k, scal, fbt.N, t, first_slot + fb_term_index(a, oddsh));
    produce(nxt_st, k);
    advance();
#pragma nounroll
    for (int i = 0; i < steps; i++) {
        FB_TRACE(g.table, nxt_st.ptr - g.table);
        nxt_e = *nxt_st.ptr;                                                    // step i+1's entry
        ws_ld8(k, scal, fbt.N, t, first_slot + fb_term_index(a, oddsh));        // step i+2's scalar
        fb_sched_fence();
        fb_consume_fast(acc, empty, cur_e, cur_st.skip, cur_st.neg);
        fb_order_after(k, acc);
        cur_e = nxt_e;
        cur_st = nxt_st;
        produce(nxt_st, k);
        advance();
    }
}
// The lane sums start from a fixed point T (x from SHA-256 of "bp_pp_amd fixed-base accumulator offset 1") instead of an empty
// accumulator and take it off again at the end with one complete addition: the incomplete law then never sees an empty operand, and
// the four selects per addition that the "first point" case cost are gone.  (Should a sum ever pass through -T or T, ZZ = 0 reports
// it like any other exceptional addition and the complete path re-does the sum.)  A whole wavefront per sum (nl = 64: a handful of
// additions per lane) keeps the empty start: there the extra addition would cost more than the selects.
HD void fb_offset_point(apt& T, bool negated) {
    const u32 X[8] = {0x3003A5ABu, 0x0CC9A3AFu, 0xC7A4AC74u, 0xB36E34E9u, 0xF816F85Eu, 0xC7857C12u, 0x72CF9444u, 0x39DE2EB9u};
    const u32 Y[8] = {0x186C3A6Cu, 0x93FE7D16u, 0x02363020u, 0x24F39B91u, 0x0E9EDE9Fu, 0x61EC1755u, 0x8C1AFEDBu, 0x8F845346u};
    const u32 NY[8] = {0xE793C1C3u, 0x6C0182E8u, 0xFDC9CFDFu, 0xDB0C646Eu, 0xF1612160u, 0x9E13E8AAu, 0x73E50124u, 0x707BACB9u};
    fe_from_w8(T.x, X);
    fe_from_w8(T.y, negated ? NY : Y);
}
HD bool fb_offset_start(int nl) { return nl < 64; }
HD bool fb_lane_finish_fast(pt& part, const ptz& acc, bool empty) {
    const bool exceptional = !empty && fe_is_zero(acc.ZZ);
    ptz_to_pt(part, acc, empty);
    return !exceptional;
}
// The sums every fixed-base kernel computes are described as up to 3 runs of consecutive (scalar slot, base) pairs.
// A run can say two things about its scalars that spare table additions (the provers use both; every verifier sum is full-width):
//   bits  > 0: every scalar of the run is below 2^bits (a hexadecimal digit, a multiplicity, a u64 value): only the windows such a
//              value reaches are looked up (fb_windows_for) -- the others hold the zero digit by construction;
//   oddsh >= 0: only the ODD blocks of 2^oddsh consecutive terms are present (terms (2 b + 1) 2^oddsh + r, r < 2^oddsh): the WNLA
//              prover's R is a sum over the odd halves of the folded vectors, the even ones have scalar zero (wnla.rs:140-150);
//              `count` then counts the terms that ARE present.  oddsh | BPPP_FB_EVEN: the EVEN blocks instead (the u64 prover's next
//              commitment = those + R of the next round: prove_core.h, job_e).
#define BPPP_FB_EVEN 16
#define BPPP_FB_MAX_RUNS 5
struct FbRanges {
    int n;
    int slot[BPPP_FB_MAX_RUNS], base[BPPP_FB_MAX_RUNS], count[BPPP_FB_MAX_RUNS];
    int bits[BPPP_FB_MAX_RUNS] = {0, 0, 0, 0, 0};
    int oddsh[BPPP_FB_MAX_RUNS] = {-1, -1, -1, -1, -1};
};
HD void fb_ranges_one(FbRanges& r, int slot, int base, int count) { r.n = 1; r.slot[0] = slot; r.base[0] = base; r.count[0] = count; r.bits[0] = 0; r.oddsh[0] = -1; }
// A run of consecutive (slot, base) terms in the regions of its table: the part below hi_bases (the wide-window region, if the table has
// one), then the rest.  fn(geometry, slot, base, count).  A run of odd / even blocks (oddsh >= 0: the provers' sums over g_vec or h_vec)
// never straddles the boundary -- hi_bases is 1 + |g_vec| -- and goes by its first base.
template <class F>
HD void fb_run_regions(const FbTable& fbt, const FbGeom& g_lo, const FbGeom& g_hi, int slot, int base, int count, int oddsh, F&& fn) {
    if (fbt.hi_bases > 0 && base < fbt.hi_bases) {
        const int room = fbt.hi_bases - base;
        const int c1 = (oddsh >= 0 || count < room) ? count : room;
        fn(g_hi, slot, base, c1);
        if (c1 < count) fn(g_lo, slot + c1, base + c1, count - c1);
    } else {
        fn(g_lo, slot, base, count);
    }
}
HD void fb_lane_sum_complete(pt& part, const FbTable& fbt, size_t t, int lane, const u32* scal, const FbRanges& rg, int nl = BPPP_FB_LANES) {
    FbGeom g_lo, g_hi;
    fb_geom(g_lo, fbt, false);
    if (fbt.hi_bases > 0) fb_geom(g_hi, fbt, true); else g_hi = g_lo;
    pt acc;
    pt_set_identity(acc);
#pragma nounroll
    for (int r = 0; r < rg.n; r++) {
        fb_run_regions(fbt, g_lo, g_hi, rg.slot[r], rg.base[r], rg.count[r], rg.oddsh[r], [&](const FbGeom& g, int slot, int base, int count) BPPP_LAMBDA_INLINE {
            pt p;
            fixed_base_msm_partial(p, g, fbt.N, t, lane, scal, slot, base, count, nl, rg.bits[r], rg.oddsh[r]);
            pt_add(acc, acc, p);
        });
    }
    part = acc;
}
HD bool fb_lane_sum_fast(pt& part, const FbTable& fbt, size_t t, int lane, const u32* scal, const FbRanges& rg, int nl = BPPP_FB_LANES) {
    FbGeom g_lo, g_hi;
    fb_geom(g_lo, fbt, false);
    if (fbt.hi_bases > 0) fb_geom(g_hi, fbt, true); else g_hi = g_lo;
    ptz acc;
    ptz_init(acc);
    bool empty = true;
    if (fb_offset_start(nl)) {
        apt T;
        fb_offset_point(T, false);
        acc.X = T.x;
        acc.Y = T.y;
        empty = false;
    }
    // (pacing, one-lane sums only: the sum's table additions in all, and those behind it as each run starts)
    int pace_total = 0, pace_done = 0;
    if (fbt.pace && nl == 1) {
#pragma nounroll
        for (int r = 0; r < rg.n; r++)
            fb_run_regions(fbt, g_lo, g_hi, rg.slot[r], rg.base[r], rg.count[r], rg.oddsh[r],
                           [&](const FbGeom& g, int, int, int count) BPPP_LAMBDA_INLINE { pace_total += count * fb_windows_for(rg.bits[r], g.code); });
    }
#pragma nounroll
    for (int r = 0; r < rg.n; r++) {
        fb_run_regions(fbt, g_lo, g_hi, rg.slot[r], rg.base[r], rg.count[r], rg.oddsh[r], [&](const FbGeom& g, int slot, int base, int count) BPPP_LAMBDA_INLINE {
            fb_lane_accumulate_fast(acc, empty, fbt, g, t, lane, scal, slot, base, count, nl, rg.bits[r], rg.oddsh[r], pace_done, pace_total);
            if (pace_total) pace_done += count * fb_windows_for(rg.bits[r], g.code);
        });
    }
    if (pace_total) fb_pace(pace_total, pace_total);
    const bool ok = fb_lane_finish_fast(part, acc, empty);
    if (fb_offset_start(nl)) {
        apt T;
        fb_offset_point(T, true);
        pt_madd(part, part, T, false);
    }
    return ok;
}
// single-thread form of the group sum (host emulation, and device code that runs one thread per proof)
HD void fb_sum_serial(pt& total, const FbTable& fbt, size_t t, const u32* scal, const FbRanges& rg, int nl = BPPP_FB_LANES) {
    pt part;
    bool ok = true;
    pt_set_identity(total);
    for (int lane = 0; lane < nl; lane++) {
        ok &= fb_lane_sum_fast(part, fbt, t, lane, scal, rg, nl);
        pt_add(total, total, part);
    }
    if (ok) return;
    pt_set_identity(total);
    for (int lane = 0; lane < nl; lane++) {
        fb_lane_sum_complete(part, fbt, t, lane, scal, rg, nl);
        pt_add(total, total, part);
    }
}
#if defined(__HIPCC__)
// tree-add the partial sums of the BPPP_FB_LANES consecutive lanes of a group; every lane ends with the total
template <int NL = BPPP_FB_LANES>
__device__ __forceinline__ void lane_group_sum(pt& acc) {
#pragma unroll
    for (int m = 1; m < NL; m <<= 1) {
        pt o;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            o.X.v[i] = __shfl_xor(acc.X.v[i], m, 64);
            o.Y.v[i] = __shfl_xor(acc.Y.v[i], m, 64);
            o.Z.v[i] = __shfl_xor(acc.Z.v[i], m, 64);
        }
        pt_add(acc, acc, o);
    }
}
// the 8-lane group sum the fixed-base kernels run: fast lane sums, group-wide vote, complete-formula re-do if any lane asks
template <int NL = BPPP_FB_LANES>
__device__ __forceinline__ void fb_group_sum(pt& total, const FbTable& fbt, size_t t, int lane, const u32* scal, const FbRanges& rg) {
    int bad = fb_lane_sum_fast(total, fbt, t, lane, scal, rg, NL) ? 0 : 1;
#pragma unroll
    for (int m = 1; m < NL; m <<= 1) bad |= __shfl_xor(bad, m, 64);
    if (bad) fb_lane_sum_complete(total, fbt, t, lane, scal, rg, NL);
    lane_group_sum<NL>(total);
}
#endif

// ---------------------------------------------------------------- fixed-base sums over SECRET scalars: the provers' opt-in "ct_prover" mode
// The fast sums above gather ONE table entry per window at an address the scalar's digit selects, skip zero digits' work by a flag
// and use an incomplete addition law with a fall-back: fine for public scalars (every verifier sum; the WNLA prover's sums, whose
// vectors the argument reveals by design), but for the witness and its blindings the addresses are a memory-access side channel
// (cache / TLB / DRAM-row timing observable by whoever shares the GPU).  k256, which the reference uses, multiplies in constant time
// (reciprocal.rs:88-95,118; circuit.rs:146-151,335-345,469-470).  This form restores that: 4-bit unsigned windows over a small table
// (64 windows x 15 entries x 64 B per generator: 3 MB for the 49 generators), EVERY entry of the window is read and the wanted one
// kept by mask, the zero digit is the all-zero (identity) entry of the same masked select, and the accumulation uses the complete
// RCB16 mixed addition with a masked result -- no secret-dependent address, branch or instruction count.  (The `bits` / `oddsh` hints
// stay in force: that a hexadecimal digit is below 2^4 or that a slot is structurally zero is public.)
HD void fb_lookup_add_ct(pt& acc, const FbTable& fbt, int base, int w, const u32 k[8]) {
    u32 limb = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) limb = (i == (w >> 3)) ? k[i] : limb;
    const u32 d = (limb >> (4 * (w & 7))) & 15u;
    const apt_packed* tb = fbt.table + ((size_t)base * 64 + w) * 15;
    apt_packed sel;
#pragma unroll
    for (int i = 0; i < 8; i++) { sel.x[i] = 0; sel.y[i] = 0; }
#pragma nounroll
    for (u32 e = 1; e <= 15; e++) {
        FB_TRACE(fbt.table, (tb - fbt.table) + (e - 1));
        const apt_packed v = tb[e - 1];
        const u32 m = 0u - (u32)(d == e);
#pragma unroll
        for (int i = 0; i < 8; i++) { sel.x[i] |= v.x[i] & m; sel.y[i] |= v.y[i] & m; }
    }
    apt a;
    bool id;
    apt_unpack(a, id, sel);           // all zero (digit 0, or an identity entry): the addition below is computed and discarded
    pt_madd(acc, acc, a, id);
}
HD void fixed_base_msm_partial_ct(pt& accp, const FbTable& fbt, size_t t, int lane, const u32* scal, int first_slot, int first_base, int count, int nl,
                                  int bits, int oddsh) {
    const int nwin = fb_windows_for(bits, 4);
    pt acc;
    pt_set_identity(acc);
    const int pairs = count * nwin;
#pragma nounroll
    for (int q = lane; q < pairs; q += nl) {
        const int a = q / nwin, w = q - a * nwin, j = fb_term_index(a, oddsh);
        u32 k[8];
        ws_ld8(k, scal, fbt.N, t, first_slot + j);
        fb_lookup_add_ct(acc, fbt, first_base + j, w, k);
    }
    accp = acc;
}
HD void fb_lane_sum_ct(pt& part, const FbTable& fbt, size_t t, int lane, const u32* scal, const FbRanges& rg, int nl = BPPP_FB_LANES) {
    pt acc;
    pt_set_identity(acc);
#pragma nounroll
    for (int r = 0; r < rg.n; r++) {
        pt p;
        fixed_base_msm_partial_ct(p, fbt, t, lane, scal, rg.slot[r], rg.base[r], rg.count[r], nl, rg.bits[r], rg.oddsh[r]);
        pt_add(acc, acc, p);
    }
    part = acc;
}
HD void fb_sum_serial_ct(pt& total, const FbTable& fbt, size_t t, const u32* scal, const FbRanges& rg, int nl = BPPP_FB_LANES) {
    pt part;
    pt_set_identity(total);
    for (int lane = 0; lane < nl; lane++) {
        fb_lane_sum_ct(part, fbt, t, lane, scal, rg, nl);
        pt_add(total, total, part);
    }
}
#if defined(__HIPCC__)
template <int NL = BPPP_FB_LANES>
__device__ __forceinline__ void fb_group_sum_ct(pt& total, const FbTable& fbt, size_t t, int lane, const u32* scal, const FbRanges& rg) {
    fb_lane_sum_ct(total, fbt, t, lane, scal, rg, NL);
    lane_group_sum<NL>(total);
}
#endif


// ---------------------------------------------------------------- fixed-base table construction (context creation)
// Pass 1: thread (b, w, chunk c) writes projective d * 2^(W w) * G_b for d in (c*CH, (c+1)*CH] into X/Y (table slots) and Z (ztmp).
// Pass 2: same thread batch-inverts its Z's (Montgomery trick) and normalises the slots to affine.
#define BPPP_FB_CHUNK 256
struct FbBuild {
    const apt* gens;        // [nbases]
    int nbases, W;          // W: the region's window code (fb_wb)
    apt_packed* table;      // [nbases][fb_per_base(W)], packed canonical affine
    fe *xtmp, *ytmp, *ztmp; // projective coordinates of the entries of THIS pass (pass 1 -> pass 2)
    fe* ptmp;               // prefix products of Z
    int base0, nb;          // the bases built by this pass: base0 .. base0 + nb - 1 (the scratch holds nb bases' worth of entries)
    int tbase0;             // the generator whose entries open `table` (0, or FbTable::hi_bases for the region that holds the rest)
};
HD size_t fb_chunks_per_window(int code) { return (fb_per_win_at(code, 0) + BPPP_FB_CHUNK - 1) / BPPP_FB_CHUNK; }     // of the widest window (a narrow one uses the first half)
HD void fb_build_pass1(const FbBuild& fb, size_t tid) {
    const int nwin = fb_nwin(fb.W);
    const size_t cpw = fb_chunks_per_window(fb.W);
    size_t c = tid % cpw;
    size_t w = (tid / cpw) % nwin;
    size_t b = tid / (cpw * nwin);
    if (b >= (size_t)fb.nb) return;
    const size_t per_win = fb_per_win_at(fb.W, (int)w);
    size_t d0 = c * BPPP_FB_CHUNK;   // entries d0+1 .. min(d0+CH, per_win)
    if (d0 >= per_win) return;
    apt G = fb.gens[fb.base0 + b];
    pt base;
    pt_from_affine(base, G);
    const int pos = fb_pos(fb.W, (int)w);
#pragma nounroll
    for (int i = 0; i < pos; i++) pt_dbl(base, base);
    // start = (c*CH + 1) * base by double-and-add over the (<= 24-bit) multiplier
    u32 m = (u32)(c * BPPP_FB_CHUNK + 1);
    pt cur;
    pt_set_identity(cur);
#pragma nounroll
    for (int bit = 24; bit >= 0; bit--) {
        pt_dbl(cur, cur);
        pt s;
        pt_add(s, cur, base);
        pt_cmov(cur, (m >> bit) & 1, s);
    }
    size_t off = b * fb_per_base(fb.W) + fb_win_off(fb.W, (int)w);
#pragma nounroll
    for (size_t i = 0; i < BPPP_FB_CHUNK && d0 + i < per_win; i++) {
        fb.xtmp[off + d0 + i] = cur.X;
        fb.ytmp[off + d0 + i] = cur.Y;
        fb.ztmp[off + d0 + i] = cur.Z;
        pt_add(cur, cur, base);
    }
}
HD void fb_build_pass2(const FbBuild& fb, size_t tid) {
    const int nwin = fb_nwin(fb.W);
    const size_t cpw = fb_chunks_per_window(fb.W);
    size_t c = tid % cpw;
    size_t w = (tid / cpw) % nwin;
    size_t b = tid / (cpw * nwin);
    if (b >= (size_t)fb.nb) return;
    const size_t per_win = fb_per_win_at(fb.W, (int)w);
    size_t d0 = c * BPPP_FB_CHUNK;
    if (d0 >= per_win) return;
    size_t off = b * fb_per_base(fb.W) + fb_win_off(fb.W, (int)w) + d0;    // within this pass's scratch
    const size_t toff = (size_t)(fb.base0 - fb.tbase0) * fb_per_base(fb.W); // this pass's first table entry
    size_t cnt = per_win - d0 < BPPP_FB_CHUNK ? per_win - d0 : BPPP_FB_CHUNK;
    // identity entries (Z = 0; only when the generator itself is the identity) are skipped in the product
    fe run;
    fe_set_u32(run, 1);
#pragma nounroll
    for (size_t i = 0; i < cnt; i++) {
        fe z = fb.ztmp[off + i];
        fb.ptmp[off + i] = run;
        fe m;
        fe_mul(m, run, z);
        fe_cmov(run, !fe_is_zero(z), m);
    }
    fe inv;
    fe_inv(inv, run);
#pragma nounroll
    for (size_t i = cnt; i-- > 0;) {
        fe z = fb.ztmp[off + i];
        bool id = fe_is_zero(z);
        fe zi, m;
        fe_mul(zi, inv, fb.ptmp[off + i]);
        fe_mul(m, inv, z);
        fe_cmov(inv, !id, m);
        apt xy;
        fe_mul(xy.x, fb.xtmp[off + i], zi);
        fe_mul(xy.y, fb.ytmp[off + i], zi);
        if (id) { fe_set_u32(xy.x, 0); fe_set_u32(xy.y, 0); }
        apt_packed k;
        apt_pack(k, xy);
        fb.table[toff + off + i] = k;
    }
}


}  // namespace bppp
