// Batch u64 range-proof verification, exact per-proof mode: the per-proof work of
// `U64RangeProofProtocol::verify` (u64_proof.rs:42-54) -> `ReciprocalRangeProofProtocol::verify`
// (reciprocal.rs:98-107) -> `ArithmeticCircuit::verify` (circuit.rs:154-256) -> `WeightNormLinearArgument::verify`
// (wnla.rs:75-121), restructured for one lane per proof over a batch that shares one generator set.
//
// What is restructured (same group elements, same transcript bytes, same accept bit -- SURVEY.md 8a closed forms,
// checked numerically against the dense reference-shaped oracle in tests/):
//  * make_circuit + collect_c's dense 16x48 / 17x48 matrices (reciprocal.rs:150-214, circuit.rs:584-653) collapse to
//    closed forms for pn_tau / ps_tau / c; the 256 recomputed inversions become ONE Fn inversion (Montgomery trick).
//  * circuit.rs:206,230-235 (22 scalar multiplications) -> one 17-term fixed-base MSM over batch-shared tables plus
//    one 5-point shared-doubling (Straus) multi-scalar multiplication.
//  * each WNLA round's generator folding (wnla.rs:96-97, 68 scalar multiplications over 4 rounds) is NOT executed:
//    folded generators only matter in the base case (wnla.rs:80-82), where they unroll to a 49-term fixed-base MSM
//    over the ORIGINAL generators.  Only com_ = com + y*X + (y^2-1)*R (wnla.rs:100-102) runs per round, because the
//    next challenge hashes it (wnla.rs:88).
//
// Data layout in HBM (workspace): structure-of-arrays, limb-major -- word (slot*8 + limb) of proof t sits at
// base[(slot*8 + limb) * N + t], so a wavefront's 64 lanes read 256 contiguous bytes per load.
#pragma once
#include "merlin.h"
#include "point.h"

namespace bppp {

// Optional phase stamps (diagnostic builds only: -DBPPP_PHASE_TIMING): lane 0 of the sampled wavefronts records the constant-rate 100 MHz
// counter (s_memrealtime: one time base for all eight XCDs, unlike the per-XCD shader-clock counter of clock64()) at marked points of
// verify_phase1 / verify_round / verify_tables / verify_c0_var into ws.stamps (BPPP_STAMP_WAVES rows of 32 words; every ws.stamp_stride-th
// wavefront of a launch has a row) and, at the first stamp of each kernel, where it runs (HW_ID | XCC_ID << 32, words 24..28);
// tools/probes/phase_probe.py and tools/probes/wave_timeline.py read them back through bppp_debug_read_stamps.
#define BPPP_STAMP_WAVES 4096
#if defined(BPPP_PHASE_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define BPPP_STAMP(t, i) bppp_stamp(ws.stamps, ws.stamp_stride, (t), (i))
__device__ __forceinline__ void bppp_stamp(unsigned long long* stamps, unsigned stride, size_t t, int i) {
    if ((t & 63) != 0 || !stamps) return;
    const size_t w = t >> 6;
    if (w % stride != 0 || w / stride >= BPPP_STAMP_WAVES) return;
    unsigned long long* row = stamps + (w / stride) * 32;
    row[i] = (unsigned long long)wall_clock64();
    const int where = i == 0 ? 24 : i == 9 ? 25 : i == 16 ? 26 : i == 20 ? 27 : i == 22 ? 28 : -1;
    if (where >= 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        row[where] = (unsigned long long)hw | (unsigned long long)xcc << 32;
    }
}
#else
#define BPPP_STAMP(t, i) ((void)0)
#endif

enum : int32_t {
    ST_OK = 0,
    ST_BAD_ENCODING = 1,     // coordinate >= p, point off curve, scalar >= n (k256 deserialisation would have failed)
    ST_DEGENERATE = 2,       // challenge >= n or a zero inverse: the reference panics on unwrap() here
};

#define BPPP_U64_PROOF_BYTES 928
#define BPPP_NG 49            // g, g_vec[16], h_vec[32]
#define BPPP_STRAUS_ENTRIES 9 // 0..8 times the point (signed 4-bit windows)

// C0 MSM scalar slots (sc0): 0 ps_tau(g) | 1..16 pn_tau(g_vec) | 17 tau^-1 (c_s) | 18 -delta (c_o) | 19 tau (c_l) |
//                            20 -tau^2 (c_r) | 21 2 tau^3 (V+r)
// proof point slots (pts): 0 c_l | 1 c_r | 2 c_o | 3 c_s | 4..7 r[0..3] | 8..11 x[0..3] | 12 V+r
// challenge slots (chal): 0 e | 1 rho | 2 lambda | 3 beta | 4 delta | 5 tau | 6..9 y1..y4
struct VerifyWs {
    size_t N;
    const uint8_t* commitments;  // N x 64 (C-ABI layout)
    const uint8_t* proofs;       // N x 928
    uint8_t* accept;             // N
    int32_t* status;             // N
    uint8_t* trace;              // N x 704 or null
    u32* tstate;                 // [52][N] transcript (STROBE) state
    u32* chal;                   // [10*8][N]
    u32* sc0;                    // [22*8][N]
    u32* cvec;                   // [25*8][N]
    u32* pts;                    // [13*16][N]
    u32* lns;                    // [3*8][N]
    u32* acc;                    // [30][N] running commitment, projective limbs
    u32* pfix;                   // [30][N]
    u32* fsc;                    // [49*8][N]
    pt_slot* straus;             // [N][5][9]  (generic WNLA / reciprocal paths)
    apt_packed* atab;            // [13][2][8][N] (entry-major, see atab_of) affine multiples 1..8 of the 13 proof points, and of their GLV images (beta x, y)
    u32* tscr;                   // [BPPP_TSCR_FE * 10][N] scratch of verify_tables: running products of the slope denominators
    u32* zinv;                   // [10][N] or null.  Non-null: the large-batch form with SHARED inversions -- a kernel that needs 1 / v of its
                                 // proof finds it here, put there by fe_batch_inv_lane (one inversion per G proofs) from the v the kernel before
                                 // left: the rounds' Z of C_{k-1}, the table build's running products (k_verify_tables_pass).  Null: every
                                 // lane inverts for itself.
    const apt_packed* fb_table;  // [49][nwin][2^W - 1]
    int fb_w;                    // window bits: 4, 8 or 16
    const apt_packed* fb_table_hi;      // FbTable's second region (0 / null: none)
    int fb_w_hi, fb_hi_bases;
    strobe base;                 // Transcript::new(label)
    // pre-loaded transcripts (the reference's `t: &mut Transcript`, u64_proof.rs:42): serialized STROBE states, 203 bytes each
    // (200 state bytes, pos, pos_begin, cur_flags); n_states = 1 (one state shared by the batch) or N (one per proof); null =
    // every proof starts from `base`.  states_out (optional, N x 203): each proof's transcript as verify leaves it.
    const uint8_t* states;
    size_t n_states;
    uint8_t* states_out;
    int pace;                    // 1: the one-lane sums pace their wave priority by progress (straus_pace; plan_core.h: VerifyPlan::pace)
#if defined(BPPP_PHASE_TIMING)
    unsigned long long* stamps;  // diagnostic builds: BPPP_STAMP's rows (null: none)
    unsigned stamp_stride;
#endif
};
#define BPPP_TRANSCRIPT_STATE_BYTES 203
HD bool strobe_from_bytes(strobe& s, const uint8_t* b) {
#pragma nounroll
    for (int i = 0; i < 25; i++) {
        u64 v = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) v |= (u64)b[8 * i + k] << (8 * k);
        s.st[i] = v;
    }
    s.pos = b[200];
    s.pos_begin = b[201];
    return s.pos < BPPP_STROBE_R && s.pos_begin <= BPPP_STROBE_R;      // merlin keeps pos in [0, R) between operations
}
HD void strobe_to_bytes(uint8_t* b, const strobe& s, u32 cur_flags) {
#pragma nounroll
    for (int i = 0; i < 25; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) b[8 * i + k] = (uint8_t)(s.st[i] >> (8 * k));
    }
    b[200] = (uint8_t)s.pos;
    b[201] = (uint8_t)s.pos_begin;
    b[202] = (uint8_t)cur_flags;
}

// The same pre-loaded transcript plumbing for the generic verifiers (wnla_core.h, recip_core.h, circuit_core.h): where a
// transcript starts (tio_begin) and how it goes back to the caller (tio_export).
struct TranscriptIo {
    const uint8_t* states;   // n_states x 203 or null (= start from the context's Transcript::new(label))
    size_t n_states;         // 1 or N
    uint8_t* states_out;     // N x 203 or null
    int no_ops;              // 1: the protocol performs no transcript operation for this shape (WNLA base case, wnla.rs:80-82):
                             // the caller's transcript comes back exactly as it went in, cur_flags included
};
// Position-group key of instance t's pre-loaded transcript (kernels.h: for_each_position_group): its byte position -- but only if the
// state is one strobe_from_bytes accepts.  A rejected state makes its lane start from `base` instead, i.e. at base.pos: keyed by its
// raw byte 200 it would share a group with valid lanes at that position and, as the group's leader, force base.pos onto them.  Such a
// lane gets a key no valid lane can have (bit 8 set), so it runs alone and per-proof isolation holds.
HD u32 preloaded_position_key(const uint8_t* states, size_t n_states, size_t t) {
    if (!states || n_states == 1) return 0u;
    const uint8_t* b = states + (size_t)BPPP_TRANSCRIPT_STATE_BYTES * t;
    const bool valid = b[200] < BPPP_STROBE_R && b[201] <= BPPP_STROBE_R;
    return valid ? (u32)b[200] : 0x100u;
}
HD void tio_begin(strobe& tr, int32_t& status, const TranscriptIo& io, const strobe& base, size_t t) {
    tr = base;
    if (!io.states) return;
    strobe pre;
    if (strobe_from_bytes(pre, io.states + (size_t)BPPP_TRANSCRIPT_STATE_BYTES * (io.n_states == 1 ? 0 : t))) tr = pre;
    else status |= ST_BAD_ENCODING;
}
// ---------------------------------------------------------------- SoA access
HD void ws_ld8(u32 r[8], const u32* base, size_t N, size_t t, int slot) {
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = base[(size_t)(slot * 8 + i) * N + t];
}
HD void ws_st8(u32* base, size_t N, size_t t, int slot, const u32 r[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) base[(size_t)(slot * 8 + i) * N + t] = r[i];
}
HD void ws_ld_apt(apt& a, const u32* base, size_t N, size_t t, int slot) {   // packed canonical words
    u32 w[8];
    ws_ld8(w, base, N, t, 2 * slot);
    fe_from_w8(a.x, w);
    ws_ld8(w, base, N, t, 2 * slot + 1);
    fe_from_w8(a.y, w);
}
HD void ws_st_apt(u32* base, size_t N, size_t t, int slot, const apt& a) {
    u32 w[8];
    fe_to_w8(w, a.x);
    ws_st8(base, N, t, 2 * slot, w);
    fe_to_w8(w, a.y);
    ws_st8(base, N, t, 2 * slot + 1, w);
}
// projective points travel between kernels as raw limbs (30 words); their magnitudes are the (5, 2, 2) the group law leaves
HD void ws_ld_pt(pt& p, const u32* base, size_t N, size_t t) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        p.X.v[i] = base[(size_t)i * N + t];
        p.Y.v[i] = base[(size_t)(10 + i) * N + t];
        p.Z.v[i] = base[(size_t)(20 + i) * N + t];
    }
    FE_SETMAG(p.X, 5); FE_SETMAG(p.Y, 2); FE_SETMAG(p.Z, 2);
}
HD void ws_st_pt(u32* base, size_t N, size_t t, const pt& p) {
    FE_CHECK(p.X, 5); FE_CHECK(p.Y, 2); FE_CHECK(p.Z, 2);
#pragma unroll
    for (int i = 0; i < 10; i++) {
        base[(size_t)i * N + t] = p.X.v[i];
        base[(size_t)(10 + i) * N + t] = p.Y.v[i];
        base[(size_t)(20 + i) * N + t] = p.Z.v[i];
    }
}
HD void ws_ld_strobe(strobe& s, const u32* base, size_t N, size_t t) {
#pragma unroll
    for (int i = 0; i < 25; i++) s.st[i] = (u64)base[(size_t)(2 * i) * N + t] | ((u64)base[(size_t)(2 * i + 1) * N + t] << 32);
    s.pos = base[(size_t)50 * N + t];
    s.pos_begin = base[(size_t)51 * N + t];
}
HD void ws_st_strobe(u32* base, size_t N, size_t t, const strobe& s) {
#pragma unroll
    for (int i = 0; i < 25; i++) {
        base[(size_t)(2 * i) * N + t] = (u32)s.st[i];
        base[(size_t)(2 * i + 1) * N + t] = (u32)(s.st[i] >> 32);
    }
    base[(size_t)50 * N + t] = s.pos;
    base[(size_t)51 * N + t] = s.pos_begin;
}

// the caller's `&mut Transcript` after a verify: the stored state of instance t (its last operation was a challenge: cur_flags 7);
// an instance flagged BPPP_ST_BAD_ENCODING gets its input state back
HD void ws_st_transcript(u32* base, size_t N, size_t t, const strobe& s) { ws_st_strobe(base, N, t, s); }
HD void ws_ld_transcript(strobe& s, const u32* base, size_t N, size_t t) { ws_ld_strobe(s, base, N, t); }
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void ws_ld_transcript(strobe_lds& s, const u32* base, size_t N, size_t t) {
#pragma unroll
    for (int i = 0; i < 50; i++) s.col[i * BPPP_LDS_STRIDE] = base[(size_t)i * N + t];
    s.pos = base[(size_t)50 * N + t];
    s.pos_begin = base[(size_t)51 * N + t];
}
__device__ __forceinline__ void ws_st_transcript(u32* base, size_t N, size_t t, const strobe_lds& s) {   // same workspace layout as ws_st_strobe
#pragma unroll
    for (int i = 0; i < 50; i++) base[(size_t)i * N + t] = s.col[i * BPPP_LDS_STRIDE];
    base[(size_t)50 * N + t] = s.pos;
    base[(size_t)51 * N + t] = s.pos_begin;
}
#endif
HD void tio_export(const TranscriptIo& io, const strobe& base, const u32* tstate, size_t N, const int32_t* status, size_t t) {
    if (!io.states_out) return;
    uint8_t* out = io.states_out + (size_t)BPPP_TRANSCRIPT_STATE_BYTES * t;
    if ((status[t] & ST_BAD_ENCODING) || io.no_ops) {
        if (io.states) {
            const uint8_t* in = io.states + (size_t)BPPP_TRANSCRIPT_STATE_BYTES * (io.n_states == 1 ? 0 : t);
#pragma nounroll
            for (int i = 0; i < BPPP_TRANSCRIPT_STATE_BYTES; i++) out[i] = in[i];
        } else {
            strobe_to_bytes(out, base, 2);
        }
        return;
    }
    strobe tr;
    ws_ld_strobe(tr, tstate, N, t);
    strobe_to_bytes(out, tr, 7);
}
template <typename S, int L>
HD void app_point(S& t, const char (&label)[L], const apt& a) {  // transcript.rs:6-8
    // SEC1 compressed bytes (tag, then x big-endian) packed little-endian into 9 message words, all in registers
    const bool id = apt_is_identity(a);
    const u32 tag = id ? 0u : (2u + (fe_is_odd(a.y) ? 1u : 0u));
    u32 xw[8], be[8], mw[9];
    fe_to_w8(xw, a.x);
#pragma unroll
    for (int k = 0; k < 8; k++) be[k] = bswap32(xw[7 - k]);      // be[k] = message bytes 1 + 4k .. 4 + 4k, first byte lowest
    mw[0] = tag | (be[0] << 8);
#pragma unroll
    for (int k = 1; k < 8; k++) mw[k] = (be[k - 1] >> 24) | (be[k] << 8);
    mw[8] = be[7] >> 24;
    t_append_words(t, label, mw, 33);
}

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
};
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
HD FbTable fb_of(const VerifyWs& ws) { FbTable f = {ws.fb_table, ws.fb_w, ws.N, ws.fb_table_hi, ws.fb_w_hi, ws.fb_hi_bases}; return f; }
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
                               int first_base, int count, int bits, int oddsh) {
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
    auto produce = [&](FbStep& st) {
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
                                int first_slot, int first_base, int count, int nl, int bits, int oddsh) {
    if (nl == 1) {
        fb_lane_accumulate_seq(acc, empty, fbt, g, t, scal, first_slot, first_base, count, bits, oddsh);
        return;
    }
    const int nw = fb_windows_for(bits, g.code);
    const int pairs = count * nw;
    if (lane >= pairs) return;
    const int steps = (pairs - lane + nl - 1) / nl;
    const int da = nl / nw, dw = nl - da * nw;
    int a = lane / nw, w = lane - a * nw;
    auto advance = [&]() {      // steps past the end re-use the last one (requested, never consumed)
        int na = a + da, nwn = w + dw;
        if (nwn >= nw) { nwn -= nw; na++; }
        const bool in = na < count;
        a = in ? na : a;
        w = in ? nwn : w;
    };
    auto produce = [&](FbStep& st, const u32 k[8]) {      // k: the scalar of term a
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
        fb_run_regions(fbt, g_lo, g_hi, rg.slot[r], rg.base[r], rg.count[r], rg.oddsh[r], [&](const FbGeom& g, int slot, int base, int count) {
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
#pragma nounroll
    for (int r = 0; r < rg.n; r++) {
        fb_run_regions(fbt, g_lo, g_hi, rg.slot[r], rg.base[r], rg.count[r], rg.oddsh[r], [&](const FbGeom& g, int slot, int base, int count) {
            fb_lane_accumulate_fast(acc, empty, fbt, g, t, lane, scal, slot, base, count, nl, rg.bits[r], rg.oddsh[r]);
        });
    }
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

// ---------------------------------------------------------------- variable-base shared-doubling MSM (Straus), signed 4-bit windows
// k = sum_{i<64} (nib_i(k') - 8) 16^i + c 16^64 with k' = k + 0x88..8 (mod 2^256), c = carry out; digits in [-8, 7].
struct straus_scalar { u32 kp[8]; u32 top; };
HD void straus_recode(straus_scalar& r, const sc& k) {
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (u64)k.v[i] + 0x88888888u; r.kp[i] = (u32)c; c >>= 32; }
    r.top = (u32)c;
}
// tbl[e] = e * P, e = 0..8 (P affine, may be the identity sentinel); entries are stored with canonical coordinates
HD void straus_build_table(pt_slot* tbl, const apt& P) {
    pt cur;
    pt_set_identity(cur);
    tbl[0].p = cur;
    pt_from_affine(cur, P);
    tbl[1].p = cur;
    const bool pid = apt_is_identity(P);
#pragma nounroll
    for (int e = 2; e <= 8; e++) {
        pt src = tbl[(e & 1) ? e - 1 : e / 2].p;
        pt d;
        if (e & 1) pt_madd(d, src, P, pid);     // loop counter: wave-uniform branch
        else pt_dbl(d, src);
        pt_normalize(d);
        tbl[e].p = d;
    }
}
// acc = sum_j k_j * P_j using tables tbl[j*9 + e]; scalars recoded in rs[0..m)
HD void straus_msm(pt& out, const pt_slot* tbl, const straus_scalar* rs, int m) {
    pt acc;
    pt_set_identity(acc);
    // top digit (0 or 1) for each scalar
#pragma nounroll
    for (int j = 0; j < m; j++) {
        pt q = tbl[j * BPPP_STRAUS_ENTRIES + (rs[j].top ? 1 : 0)].p;
        pt_add(acc, acc, q);
    }
#pragma nounroll
    for (int i = 63; i >= 0; i--) {
#pragma nounroll
        for (int d = 0; d < 4; d++) pt_dbl(acc, acc);
#pragma nounroll
        for (int j = 0; j < m; j++) {
            u32 limb = 0;
#pragma unroll
            for (int l = 0; l < 8; l++) limb = (l == (i >> 3)) ? rs[j].kp[l] : limb;
            int dg = (int)((limb >> ((i & 7) * 4)) & 15) - 8;
            int mag = dg < 0 ? -dg : dg;
            pt q = tbl[j * BPPP_STRAUS_ENTRIES + mag].p;
            fe ny;
            fe_neg_m<1>(ny, q.Y);
            fe_cmov(q.Y, dg < 0, ny);
            pt_add(acc, acc, q);
        }
    }
    out = acc;
}

// ---------------------------------------------------------------- GLV endomorphism split (secp256k1: lambda*(x, y) = (beta*x, y))
// k = k1 + k2*lambda (mod n) with |k1|, |k2| < 2^128: halves the doublings of every variable-base multiplication.
// Constants: lattice basis of (n, lambda); g1, g2 = round(2^384 * b2 / n), round(2^384 * (-b1) / n)  (derived and checked in
// tests/test_core_emul.py against big-integer arithmetic).
struct glv_split { u32 k1[5], k2[5]; bool neg1, neg2; };
HD void glv_round_shift384(sc& c, const sc& k, const u32 g[8]) {   // c = (k*g + 2^383) >> 384
    u32 t[16];
    mul256(t, k.v, g);
    u32 cy = (t[11] >> 31) & 1u;
#pragma unroll
    for (int i = 0; i < 4; i++) c.v[i] = addc(t[12 + i], 0u, cy);
#pragma unroll
    for (int i = 4; i < 8; i++) c.v[i] = 0;
}
HD bool glv_abs(u32 out[5], const sc& r) {   // r is either small (< 2^129) or n - small; returns true when negated
    bool neg = ((r.v[5] | r.v[6] | r.v[7]) != 0) | (r.v[4] > 1u);
    sc m;
    sc_neg(m, r);
#pragma unroll
    for (int i = 0; i < 5; i++) out[i] = neg ? m.v[i] : r.v[i];
    return neg;
}
HD void glv_decompose(glv_split& out, const sc& k) {
    const u32 G1[8] = {0x45DBB031u, 0xE893209Au, 0x71E8CA7Fu, 0x3DAA8A14u, 0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u};
    const u32 G2[8] = {0x8AC47F71u, 0x1571B4AEu, 0x9DF506C6u, 0x221208ACu, 0x0ABFE4C4u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u};
    const sc MB1 = {{0x0ABFE4C3u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u}};
    const sc MB2 = {{0x3DB1562Cu, 0xD765CDA8u, 0x0774346Du, 0x8A280AC5u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}};
    const sc LAM = {{0x1B23BD72u, 0xDF02967Cu, 0x20816678u, 0x122E22EAu, 0x8812645Au, 0xA5261C02u, 0xC05C30E0u, 0x5363AD4Cu}};
    sc c1, c2, r1, r2, t;
    glv_round_shift384(c1, k, G1);
    glv_round_shift384(c2, k, G2);
    sc_mul(c1, c1, MB1);
    sc_mul(c2, c2, MB2);
    sc_add(r2, c1, c2);
    sc_mul(t, r2, LAM);
    sc_sub(r1, k, t);
    out.neg1 = glv_abs(out.k1, r1);
    out.neg2 = glv_abs(out.k2, r2);
    // signed 4-bit recoding offset (33 nibbles): k' = |k| + 0x8...8; digit_i = nib_i(k') - 8 in [-8, 7]
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) out.k1[i] = addc(out.k1[i], i < 4 ? 0x88888888u : 0x8u, c);
    c = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) out.k2[i] = addc(out.k2[i], i < 4 ? 0x88888888u : 0x8u, c);
}
HD u32 limb5_at(const u32 v[5], int idx) {
    u32 r = 0;
#pragma unroll
    for (int l = 0; l < 5; l++) r = (l == idx) ? v[l] : r;
    return r;
}
// acc = sum_j k_j * P_j with tables tbl[j*9 + e] = e*P_j: 33 windows x (4 doublings + 2m additions); the lambda stream reuses
// P_j's table with X scaled by beta.
HD void straus_msm_glv(pt& out, const pt_slot* tbl, const glv_split* sp, int m) {
    const u32 BETA_W[8] = {0x719501EEu, 0xC1396C28u, 0x12F58995u, 0x9CF04975u, 0xAC3434E9u, 0x6E64479Eu, 0x657C0710u, 0x7AE96A2Bu};
    fe BETA;
    fe_from_w8(BETA, BETA_W);
    pt acc;
    pt_set_identity(acc);
#pragma nounroll
    for (int i = 32; i >= 0; i--) {
        if (i != 32) {
#pragma nounroll
            for (int d = 0; d < 4; d++) pt_dbl(acc, acc);
        }
#pragma nounroll
        for (int j = 0; j < m; j++) {
#pragma nounroll
            for (int h = 0; h < 2; h++) {
                const u32* kp = h ? sp[j].k2 : sp[j].k1;
                bool sneg = h ? sp[j].neg2 : sp[j].neg1;
                int dg = (int)((limb5_at(kp, i >> 3) >> ((i & 7) * 4)) & 15) - 8;
                int mag = dg < 0 ? -dg : dg;
                pt q = tbl[j * BPPP_STRAUS_ENTRIES + mag].p;
                fe bx, ny;
                fe_mul(bx, q.X, BETA);
                fe_cmov(q.X, h != 0, bx);
                fe_neg_m<1>(ny, q.Y);
                fe_cmov(q.Y, (dg < 0) != sneg, ny);
                pt_add(acc, acc, q);
            }
        }
    }
    out = acc;
}

// ---------------------------------------------------------------- the u64 verifier's variable-base path: affine per-proof tables
// All 13 variable-base points of a proof (c_l, c_r, c_o, c_s, r[4], x[4], V + r) are inputs, known before any challenge, so
// their window tables are built once, up front, and brought to AFFINE form with a single field inversion per proof
// (Montgomery's trick over the 91 non-trivial multiples).  The five shared-doubling sums that follow (C0 and the four WNLA
// rounds) then run on a Jacobian accumulator with mixed additions (point.h), and the GLV image tables (beta x, y) are stored
// too, so the inner loop has no beta multiplication.
#define BPPP_VPOINTS 13
#define BPPP_ATAB_PER_PROOF (BPPP_VPOINTS * 16)
// BPPP_VWIN 5: signed 5-bit windows over the 128-bit GLV halves -- 26 windows x 2M mixed additions instead of 33 x 2M (the 4-bit
// recoding needs a 33rd window for its carry), tables 1P..16P per point (4 levels, 4 batched inversions) with the GLV image
// (beta x, y) formed on the fly by one multiplication (the table memory stays 13 x 16 entries).  BPPP_VWIN 4: 1P..8P + stored images.
#ifndef BPPP_VWIN
#define BPPP_VWIN 5
#endif
HD void ws_st_fe(u32* base, size_t N, size_t t, int slot, const fe& a) {
#pragma unroll
    for (int i = 0; i < 10; i++) base[(size_t)(slot * 10 + i) * N + t] = a.v[i];
}
HD void ws_ld_fe(fe& a, const u32* base, size_t N, size_t t, int slot, int mag) {
#pragma unroll
    for (int i = 0; i < 10; i++) a.v[i] = base[(size_t)(slot * 10 + i) * N + t];
    FE_SETMAG(a, mag);
    (void)mag;
}
// out[t] = 1 / in[t] (0 for 0, as fe_inv) for the G elements t = i, i + L, i + 2L, ... (L = ceil(N / G)) that lane i takes: their
// product is inverted once and unwound (3 multiplications per element), so a batch of N pays N / G inversions instead of N.  in and
// out are [10][N] limb arrays and may be the same one (a lane reads its G elements before it writes any; lanes share none).
template <int G>
HD void fe_batch_inv_lane(const u32* in, u32* out, size_t N, size_t i) {
    const size_t L = (N + G - 1) / G;
    fe z[G], pre[G], run, inv, one;
    bool zero[G];
    fe_set_u32(one, 1);
    run = one;
#pragma unroll
    for (int j = 0; j < G; j++) {
        const size_t t = i + (size_t)j * L;
        z[j] = one;
        zero[j] = true;
        if (t < N) {
            ws_ld_fe(z[j], in, N, t, 0, 2);
            zero[j] = fe_is_zero(z[j]);
            if (zero[j]) z[j] = one;
        }
        pre[j] = run;
        fe_mul(run, run, z[j]);
    }
    fe_inv(inv, run);
#pragma unroll
    for (int j = G - 1; j >= 0; j--) {
        const size_t t = i + (size_t)j * L;
        fe o;
        fe_mul(o, inv, pre[j]);
        fe_mul(inv, inv, z[j]);
        if (zero[j]) fe_set_u32(o, 0);
        if (t < N) ws_st_fe(out, N, t, 0, o);
    }
}
HD void glv_beta(fe& b) {
    const u32 BETA_W[8] = {0x719501EEu, 0xC1396C28u, 0x12F58995u, 0x9CF04975u, 0xAC3434E9u, 0x6E64479Eu, 0x657C0710u, 0x7AE96A2Bu};
    fe_from_w8(b, BETA_W);
}
// Layout of the per-proof window tables in HBM.  Entry i of proof t (i = point * 16 + image * 8 + multiple - 1):
//   BPPP_ATAB_SOA 0:  atab[t * 208 + i]   a proof's 13 KB of tables contiguous: the build kernel writes 64-byte records 13 KB apart
//   BPPP_ATAB_SOA 1:  atab[i * N + t]     entry-major: the build kernel's stores coalesce across the wavefront; the sums' gathers are
//                                         64-byte records either way
#ifndef BPPP_ATAB_SOA
#define BPPP_ATAB_SOA 1   // measured on 2^20 proofs: k_verify_tables 9.85 -> 8.82 ms, the five sums unchanged (28.4 / 57.8 ms)
#endif
struct atab_ref {
    apt_packed* p;
    size_t s;
    HD apt_packed& operator[](int i) const { return p[(size_t)i * s]; }
    HD atab_ref operator+(int k) const { atab_ref r = {p + (size_t)k * s, s}; return r; }
};
HD atab_ref atab_of(apt_packed* atab, size_t N, size_t t, int entries_per_instance = 13 * 16) {
#if BPPP_ATAB_SOA
    (void)entries_per_instance;
    atab_ref r = {atab + t, N};
#else
    (void)N;
    atab_ref r = {atab + t * (size_t)entries_per_instance, 1};
#endif
    return r;
}
HD void atab_store(atab_ref tb, int e, const apt& a, const fe& beta, bool identity) {   // e = 1..8 (1..16 with 5-bit windows)
    apt_packed k;
    fe_to_w8(k.x, a.x);
    fe_to_w8(k.y, a.y);
    if (identity) {
#pragma unroll
        for (int i = 0; i < 8; i++) k.x[i] = k.y[i] = 0;
    }
    tb[e - 1] = k;
#if BPPP_VWIN != 5
    apt_packed kb;
    fe bx;
    fe_mul(bx, a.x, beta);
    fe_to_w8(kb.x, bx);
#pragma unroll
    for (int i = 0; i < 8; i++) kb.y[i] = identity ? 0u : k.y[i];
    if (identity) {
#pragma unroll
        for (int i = 0; i < 8; i++) kb.x[i] = 0;
    }
    tb[8 + e - 1] = kb;
#else
    (void)beta;
#endif
}
// Window tables by AFFINE arithmetic, three batched inversions per proof.  The multiples of one point form three levels whose
// slopes only need earlier levels:   2P = 2.P  |  3P = 2P + P, 4P = 2.2P  |  5P = 4P + P, 6P = 2.3P, 7P = 4P + 3P, 8P = 2.4P,
// so all 13 points' level-l slope denominators (13, 26, 52 of them) are inverted together with Montgomery's trick.  An affine
// step costs 1M (prefix) + 2M (unwinding) + 1M + 2S (slope, x, y) against 12M for a complete projective step plus 6M of
// normalisation afterwards, and the only scratch is the running products (91 field elements per proof instead of 364).  The
// unwinding of level l (which produces that level's points) is fused with the forward pass of level l + 1 on the same point,
// so the passes alternate direction over the 13 points: A up, B down, C up, D down.
// No exceptional cases arise: the group has prime order n > 8, so for a point P != O none of P .. 8P is O, 2y != 0, and the
// additions jP + P (j = 2, 4) and 4P + 3P never meet equal x.  P = O (the (0, 0) sentinel, also what a malformed proof's
// points are replaced by) gives zero denominators: they are replaced by 1 and every multiple is stored as O.
#define BPPP_TSCR_FE (5 * BPPP_VPOINTS)   // running products per proof (BPPP_TSCR_PER_POINT below): 65 field elements, was 182 with one per denominator
struct aff_src { fe x, y; };
HD void aff_ld(aff_src& r, atab_ref tb, int e) {   // multiple e (1..8) of the point whose table is tb
    const apt_packed k = tb[e - 1];
    fe_from_w8(r.x, k.x);
    fe_from_w8(r.y, k.y);
}
HD void aff_den_dbl(fe& d, const aff_src& a, bool pid, const fe& one) { fe_add(d, a.y, a.y); fe_cmov(d, pid, one); }
HD void aff_den_add(fe& d, const aff_src& a, const aff_src& b, bool pid, const fe& one) { fe_sub_m<1>(d, a.x, b.x); fe_cmov(d, pid, one); }   // a + b
// 2a given 1 / (2 y_a)
HD void aff_dbl(apt& r, const aff_src& a, const fe& dinv) {
    fe num, lam, t;
    fe_sqr(num, a.x);
    fe_mul_small(num, num, 3);
    fe_mul(lam, num, dinv);
    fe_sqr(r.x, lam);
    fe_add(t, a.x, a.x);
    fe_sub_m<2>(r.x, r.x, t);            // magnitude 4
    fe_sub_m<4>(t, a.x, r.x);            // 6
    fe_mul(t, lam, t);
    fe_sub_m<1>(r.y, t, a.y);            // 3
}
// a + b given 1 / (x_a - x_b)
HD void aff_add(apt& r, const aff_src& a, const aff_src& b, const fe& dinv) {
    fe num, lam, t;
    fe_sub_m<1>(num, a.y, b.y);
    fe_mul(lam, num, dinv);
    fe_sqr(r.x, lam);
    fe_sub_m<1>(r.x, r.x, a.x);          // 3
    fe_sub_m<1>(r.x, r.x, b.x);          // 5
    fe_sub_m<5>(t, a.x, r.x);            // 7
    fe_mul(t, lam, t);
    fe_sub_m<1>(r.y, t, a.y);            // 3
}
HD void aff_take(aff_src& r, const apt& a) {   // a freshly computed point as the operand of the next level (magnitudes -> 1)
    fe_mul_small(r.x, a.x, 1);
    fe_mul_small(r.y, a.y, 1);
}
// one Montgomery-trick step forward: store the running product, multiply the denominator in
HD void aff_push(u32* tscr, size_t N, size_t t, int slot, fe& run, const fe& den) {
    ws_st_fe(tscr, N, t, slot, run);
    fe_mul(run, run, den);
}
// ... and backward: dinv = 1 / den, inv loses den
HD void aff_pop(fe& dinv, const u32* tscr, size_t N, size_t t, int slot, fe& inv, const fe& den) {
    fe pre;
    ws_ld_fe(pre, tscr, N, t, slot, 1);
    fe_mul(dinv, inv, pre);
    fe_mul(inv, inv, den);
}
// ---- blocks of four denominators (levels 3 and 4): ONE running product per block instead of one per denominator.  The running
// products are the table builder's scratch traffic (written in one pass, read back in the next, through HBM: at one lane per proof
// nothing of that size stays on chip), so a block costs a quarter of the stores and loads for three more multiplications when it is
// unwound (the block product is re-formed from the denominators, which the unwinding pass has in registers anyway).
HD void aff_push_block(u32* tscr, size_t N, size_t t, int slot, fe& run, const fe& block_product) {
    ws_st_fe(tscr, N, t, slot, run);
    fe_mul(run, run, block_product);
}
// di[k] = 1 / d[k] for the block pushed at `slot`; inv (the inverse of everything not yet unwound) loses the block
HD void aff_pop_block(fe di[4], const fe d[4], const u32* tscr, size_t N, size_t t, int slot, fe& inv) {
    fe t01, t23, B, pre, q;
    fe_mul(t01, d[0], d[1]);
    fe_mul(t23, d[2], d[3]);
    fe_mul(B, t01, t23);
    ws_ld_fe(pre, tscr, N, t, slot, 1);
    fe_mul(q, inv, pre);                 // 1 / (d0 d1 d2 d3)
    fe_mul(inv, inv, B);
    fe_mul(B, q, t23);                   // 1 / (d0 d1)
    fe_mul(q, q, t01);                   // 1 / (d2 d3)
    fe_mul(di[0], B, d[1]);
    fe_mul(di[1], B, d[0]);
    fe_mul(di[2], q, d[3]);
    fe_mul(di[3], q, d[2]);
}
// Window tables of NP points per instance (the u64 verifier's 13 proof points; the generic WNLA verifier's 2 x rounds round points):
// pts = the points in packed affine words [NP * 16][N], tscr = BPPP_TSCR_PER_POINT NP running products [.. * 10][N], tab = the
// instance's table view.
#define BPPP_TSCR_PER_POINT 5    // level 1: 1 (slots 0 .. NP, re-used by level 3: 1 block) | level 2: 2 | level 4: 2 blocks
// The build in five passes with an inversion of `run` between them: the state that crosses a boundary is `run` going in and its inverse
// coming out (tables, points and running products live in the workspace), so the passes are also kernels of their own with the
// inversions shared between proofs (k_verify_tables_pass, fe_batch_inv_lane).
HD void affine_tables_pass_a(const atab_ref tab, u32* tscr, const u32* pts, size_t N, size_t t, const int NP, fe& run) {
    const int L2 = NP, L4 = 3 * NP;          // running products: levels 1 and 3 share slots 0 .. NP, level 2 lives in NP .. 3 NP, level 4 in 3 NP .. 5 NP
    (void)L2; (void)L4;
    fe beta, one, d, dinv;
    fe_set_u32(one, 1);
    glv_beta(beta);
    (void)d; (void)dinv;
    // ---- pass A (up): entry 1 of every table; level-1 denominators 2 y_P
    fe_set_u32(run, 1);
#pragma nounroll
    for (int p = 0; p < NP; p++) {
        apt P;
        ws_ld_apt(P, pts, N, t, p);
        const bool pid = apt_is_identity(P);
        atab_store(tab + p * 16, 1, P, beta, pid);
        aff_src a = {P.x, P.y};
        aff_den_dbl(d, a, pid, one);
        aff_push(tscr, N, t, p, run, d);
    }
}
HD void affine_tables_pass_b(const atab_ref tab, u32* tscr, const u32* pts, size_t N, size_t t, const int NP, fe inv, fe& run) {
    const int L2 = NP, L4 = 3 * NP;          // running products: levels 1 and 3 share slots 0 .. NP, level 2 lives in NP .. 3 NP, level 4 in 3 NP .. 5 NP
    (void)L2; (void)L4;
    fe beta, one, d, dinv;
    fe_set_u32(one, 1);
    glv_beta(beta);
    (void)d; (void)dinv;
    // ---- pass B (down): 2P; level-2 denominators x_2P - x_P (3P = 2P + P), 2 y_2P (4P)
    fe_set_u32(run, 1);
#pragma nounroll
    for (int p = NP - 1; p >= 0; p--) {
        apt P;
        ws_ld_apt(P, pts, N, t, p);
        const bool pid = apt_is_identity(P);
        aff_src a = {P.x, P.y};
        aff_den_dbl(d, a, pid, one);
        aff_pop(dinv, tscr, N, t, p, inv, d);
        apt P2;
        aff_dbl(P2, a, dinv);
        atab_store(tab + p * 16, 2, P2, beta, pid);
        aff_src a2;
        aff_take(a2, P2);
        const int q = L2 + (NP - 1 - p) * 2;
        aff_den_add(d, a2, a, pid, one);
        aff_push(tscr, N, t, q, run, d);
        aff_den_dbl(d, a2, pid, one);
        aff_push(tscr, N, t, q + 1, run, d);
    }
}
HD void affine_tables_pass_c(const atab_ref tab, u32* tscr, const u32* pts, size_t N, size_t t, const int NP, fe inv, fe& run) {
    const int L2 = NP, L4 = 3 * NP;          // running products: levels 1 and 3 share slots 0 .. NP, level 2 lives in NP .. 3 NP, level 4 in 3 NP .. 5 NP
    (void)L2; (void)L4;
    fe beta, one, d, dinv;
    fe_set_u32(one, 1);
    glv_beta(beta);
    (void)d; (void)dinv;
    // ---- pass C (up): 4P, 3P; level-3 denominators x_4P - x_P (5P), 2 y_3P (6P), x_4P - x_3P (7P), 2 y_4P (8P): one block per point
    fe_set_u32(run, 1);
#pragma nounroll
    for (int p = 0; p < NP; p++) {
        const atab_ref tb = tab + p * 16;
        aff_src a, a2;
        aff_ld(a, tb, 1);
        aff_ld(a2, tb, 2);
        const bool pid = fe_is_zero(a.x) & fe_is_zero(a.y);
        const int q = L2 + (NP - 1 - p) * 2;
        apt P3, P4;
        aff_den_dbl(d, a2, pid, one);
        aff_pop(dinv, tscr, N, t, q + 1, inv, d);
        aff_dbl(P4, a2, dinv);
        aff_den_add(d, a2, a, pid, one);
        aff_pop(dinv, tscr, N, t, q, inv, d);
        aff_add(P3, a2, a, dinv);
        atab_store(tb, 3, P3, beta, pid);
        atab_store(tb, 4, P4, beta, pid);
        aff_src a3, a4;
        aff_take(a3, P3);
        aff_take(a4, P4);
        fe bp;
        aff_den_add(bp, a4, a, pid, one);
        aff_den_dbl(d, a3, pid, one);
        fe_mul(bp, bp, d);
        aff_den_add(d, a4, a3, pid, one);
        fe_mul(bp, bp, d);
        aff_den_dbl(d, a4, pid, one);
        fe_mul(bp, bp, d);
        aff_push_block(tscr, N, t, p, run, bp);
    }
}
HD void affine_tables_pass_d(const atab_ref tab, u32* tscr, const u32* pts, size_t N, size_t t, const int NP, fe inv, fe& run) {
    const int L2 = NP, L4 = 3 * NP;          // running products: levels 1 and 3 share slots 0 .. NP, level 2 lives in NP .. 3 NP, level 4 in 3 NP .. 5 NP
    (void)L2; (void)L4;
    fe beta, one, d, dinv;
    fe_set_u32(one, 1);
    glv_beta(beta);
    (void)d; (void)dinv;
    // ---- pass D (down): 5P, 6P, 7P, 8P  [5-bit windows: + level-4 denominators for 9P .. 16P, two blocks per point]
    fe_set_u32(run, 1);
#pragma nounroll
    for (int p = NP - 1; p >= 0; p--) {
        const atab_ref tb = tab + p * 16;
        aff_src a, a3, a4;
        aff_ld(a, tb, 1);
        aff_ld(a3, tb, 3);
        aff_ld(a4, tb, 4);
        const bool pid = fe_is_zero(a.x) & fe_is_zero(a.y);
        fe dd[4], di[4];
        aff_den_add(dd[0], a4, a, pid, one);      // 5P = 4P + P
        aff_den_dbl(dd[1], a3, pid, one);         // 6P = 2 . 3P
        aff_den_add(dd[2], a4, a3, pid, one);     // 7P = 4P + 3P
        aff_den_dbl(dd[3], a4, pid, one);         // 8P = 2 . 4P
        aff_pop_block(di, dd, tscr, N, t, p, inv);
        apt R;
#if BPPP_VWIN == 5
        // level 4: 8P against P, 3P, 5P, 7P (9P, 11P, 13P, 15P: block "odd") and the doublings of 5P .. 8P (10P .. 16P: block "even")
        fe x8, bo, be;
        aff_src ax;
#endif
        aff_dbl(R, a4, di[3]);
        atab_store(tb, 8, R, beta, pid);
#if BPPP_VWIN == 5
        aff_take(ax, R);
        x8 = ax.x;
        aff_den_dbl(be, ax, pid, one);                                          // 16P = 2 . 8P
        fe_sub_m<1>(bo, x8, a.x);  fe_cmov(bo, pid, one);                       //  9P = 8P + P
        fe_sub_m<1>(d, x8, a3.x);  fe_cmov(d, pid, one);  fe_mul(bo, bo, d);    // 11P = 8P + 3P
#endif
        aff_add(R, a4, a3, di[2]);
        atab_store(tb, 7, R, beta, pid);
#if BPPP_VWIN == 5
        aff_take(ax, R);
        fe_sub_m<1>(d, x8, ax.x);  fe_cmov(d, pid, one);  fe_mul(bo, bo, d);    // 15P = 8P + 7P
        aff_den_dbl(d, ax, pid, one);                     fe_mul(be, be, d);    // 14P = 2 . 7P
#endif
        aff_dbl(R, a3, di[1]);
        atab_store(tb, 6, R, beta, pid);
#if BPPP_VWIN == 5
        aff_take(ax, R);
        aff_den_dbl(d, ax, pid, one);                     fe_mul(be, be, d);    // 12P = 2 . 6P
#endif
        aff_add(R, a4, a, di[0]);
        atab_store(tb, 5, R, beta, pid);
#if BPPP_VWIN == 5
        aff_take(ax, R);
        fe_sub_m<1>(d, x8, ax.x);  fe_cmov(d, pid, one);  fe_mul(bo, bo, d);    // 13P = 8P + 5P
        aff_den_dbl(d, ax, pid, one);                     fe_mul(be, be, d);    // 10P = 2 . 5P
        aff_push_block(tscr, N, t, L4 + 2 * p, run, bo);
        aff_push_block(tscr, N, t, L4 + 2 * p + 1, run, be);
#endif
    }
}
#if BPPP_VWIN == 5
HD void affine_tables_pass_e(const atab_ref tab, u32* tscr, const u32* pts, size_t N, size_t t, const int NP, fe inv) {
    const int L2 = NP, L4 = 3 * NP;          // running products: levels 1 and 3 share slots 0 .. NP, level 2 lives in NP .. 3 NP, level 4 in 3 NP .. 5 NP
    (void)L2; (void)L4;
    fe beta, one, d, dinv;
    fe_set_u32(one, 1);
    glv_beta(beta);
    (void)d; (void)dinv;
    // ---- pass E (up): 9P .. 16P; per point the even block (pushed last) unwinds first
#pragma nounroll
    for (int p = 0; p < NP; p++) {
        const atab_ref tb = tab + p * 16;
        aff_src a5, a6, a7, a8;
        aff_ld(a5, tb, 5);
        aff_ld(a8, tb, 8);
        const bool pid = fe_is_zero(a8.x) & fe_is_zero(a8.y);       // P = O <=> every stored multiple is the (0, 0) sentinel
        fe dd[4], di[4];
        apt R;
        aff_ld(a6, tb, 6);
        aff_ld(a7, tb, 7);
        aff_den_dbl(dd[0], a5, pid, one);
        aff_den_dbl(dd[1], a6, pid, one);
        aff_den_dbl(dd[2], a7, pid, one);
        aff_den_dbl(dd[3], a8, pid, one);
        aff_pop_block(di, dd, tscr, N, t, L4 + 2 * p + 1, inv);
        aff_dbl(R, a5, di[0]);  atab_store(tb, 10, R, beta, pid);
        aff_dbl(R, a6, di[1]);  atab_store(tb, 12, R, beta, pid);
        aff_dbl(R, a7, di[2]);  atab_store(tb, 14, R, beta, pid);
        aff_dbl(R, a8, di[3]);  atab_store(tb, 16, R, beta, pid);
        aff_src a, a3;                                              // a6 is done with: its registers serve P and 3P
        aff_ld(a, tb, 1);
        aff_ld(a3, tb, 3);
        aff_den_add(dd[0], a8, a, pid, one);
        aff_den_add(dd[1], a8, a3, pid, one);
        aff_den_add(dd[2], a8, a5, pid, one);
        aff_den_add(dd[3], a8, a7, pid, one);
        aff_pop_block(di, dd, tscr, N, t, L4 + 2 * p, inv);
        aff_add(R, a8, a, di[0]);   atab_store(tb, 9, R, beta, pid);
        aff_add(R, a8, a3, di[1]);  atab_store(tb, 11, R, beta, pid);
        aff_add(R, a8, a5, di[2]);  atab_store(tb, 13, R, beta, pid);
        aff_add(R, a8, a7, di[3]);  atab_store(tb, 15, R, beta, pid);
    }
}
#endif
HD void affine_tables_build(const atab_ref tab, u32* tscr, const u32* pts, size_t N, size_t t, const int NP) {
    fe run, inv;
    affine_tables_pass_a(tab, tscr, pts, N, t, NP, run);
    fe_inv(inv, run);
    affine_tables_pass_b(tab, tscr, pts, N, t, NP, inv, run);
    fe_inv(inv, run);
    affine_tables_pass_c(tab, tscr, pts, N, t, NP, inv, run);
    fe_inv(inv, run);
    affine_tables_pass_d(tab, tscr, pts, N, t, NP, inv, run);
#if BPPP_VWIN == 5
    fe_inv(inv, run);
    affine_tables_pass_e(tab, tscr, pts, N, t, NP, inv);
#endif
}
// pass = 0 .. 4 of the u64 verifier's 13 tables with the inversions shared: the running product goes out through ws.zinv, its inverse
// (fe_batch_inv_lane, in place) comes back through it
template <int PASS>
HD void verify_tables_pass(const VerifyWs& ws, size_t t) {
    const atab_ref tab = atab_of(ws.atab, ws.N, t);
    fe run, inv;
    if constexpr (PASS > 0) ws_ld_fe(inv, ws.zinv, ws.N, t, 0, 1);
    if constexpr (PASS == 0) affine_tables_pass_a(tab, ws.tscr, ws.pts, ws.N, t, BPPP_VPOINTS, run);
    if constexpr (PASS == 1) affine_tables_pass_b(tab, ws.tscr, ws.pts, ws.N, t, BPPP_VPOINTS, inv, run);
    if constexpr (PASS == 2) affine_tables_pass_c(tab, ws.tscr, ws.pts, ws.N, t, BPPP_VPOINTS, inv, run);
    if constexpr (PASS == 3) affine_tables_pass_d(tab, ws.tscr, ws.pts, ws.N, t, BPPP_VPOINTS, inv, run);
#if BPPP_VWIN == 5
    if constexpr (PASS == 4) affine_tables_pass_e(tab, ws.tscr, ws.pts, ws.N, t, BPPP_VPOINTS, inv);
#endif
    if constexpr (PASS < 4) ws_st_fe(ws.zinv, ws.N, t, 0, run);
}
HD void verify_tables(const VerifyWs& ws, size_t t) {
    BPPP_STAMP(t, 16);
    affine_tables_build(atab_of(ws.atab, ws.N, t), ws.tscr, ws.pts, ws.N, t, BPPP_VPOINTS);
    BPPP_STAMP(t, 19);
}
// The same tables straight from the caller's BYTES, for batches whose one-lane kernels are a lone wavefront per SIMD (2^15, 2^16 proofs):
// the kernel then runs on the helper stream BESIDE phase 1 instead of after it, and every SIMD has two wavefronts to interleave.  It
// decodes the 13 points exactly as verify_phase1_on does -- V + proof.r to affine, ALL points zero if any field of the proof is
// malformed -- into a private copy (rows 17..42 of the final-scalar buffer, which nothing touches before k_verify_final_scalars; phase 1
// parks its reciprocals in rows 0..15), so the tables are bit for bit those of the serial order.
HD void verify_tables_own(const VerifyWs& ws, size_t t) {
    const size_t N = ws.N;
    u32* tp = ws.fsc + (size_t)(17 * 8) * N;
    const uint8_t* pv = ws.commitments + 64 * t;
    const uint8_t* pp = ws.proofs + (size_t)BPPP_U64_PROOF_BYTES * t;
    apt V, Pr;
    bool ok = apt_from_xy64(V, pv);
#pragma nounroll
    for (int i = 0; i < 12; i++) {
        apt Q;
        ok &= apt_from_xy64(Q, pp + 64 * i);
        ws_st_apt(tp, N, t, i, Q);
    }
    ok &= apt_from_xy64(Pr, pp + 64 * 12);
    sc l0;
    ok &= sc_from_be(l0, pp + 832);
    ok &= sc_from_be(l0, pp + 864);
    ok &= sc_from_be(l0, pp + 896);
    if (!ok) {
        apt zero;
        fe_set_u32(zero.x, 0);
        fe_set_u32(zero.y, 0);
        V = zero;
        Pr = zero;
#pragma nounroll
        for (int i = 0; i < 12; i++) ws_st_apt(tp, N, t, i, zero);
    }
    apt Vr;
    {
        pt s;
        pt_from_affine(s, V);
        pt_madd(s, s, Pr, apt_is_identity(Pr));
        pt_to_affine(Vr, s);
    }
    ws_st_apt(tp, N, t, 12, Vr);
    affine_tables_build(atab_of(ws.atab, ws.N, t), ws.tscr, tp, N, t, BPPP_VPOINTS);
}
#if BPPP_VWIN == 5
// The window table of ONE point by ONE lane -- for calls so small that the chip is empty and what counts is the length of the dependent
// chain a proof has to wait for (a lane per table instead of a lane per proof: k_verify_tables_split).  pre_doublings > 0 first replaces
// P by 2^pre_doublings P: the table of a LATER part of a 26-window stream (part j starts at window split_begin(parts, j): 65, or 35 / 70 / 100, doublings),
// so that a sum can walk the parts of every stream on separate lanes (straus_affine_split).  The multiples 2P .. 16P as a Jacobian chain (one doubling, 14
// mixed additions: kP + P is never exceptional for 2 <= k <= 15 in a group of prime order), one inversion for their 15 Z's.
// A 26-window stream in `parts` parts (2 or 4): part j covers windows split_begin(parts, j) .. split_begin(parts, j + 1) - 1 over the
// table of 2^(5 split_begin(parts, j)) P -- 13 + 13 windows (tables of P, 2^65 P) or 7 + 7 + 6 + 6 (P, 2^35 P, 2^70 P, 2^100 P).
#define BPPP_SPLIT_PARTS_MAX 4
HD int split_begin(int parts, int part) {   // 26 = BPPP_VWINDOWS (defined below)
    if (parts == 1) return part == 0 ? 0 : 26;
    if (parts == 2) return part == 0 ? 0 : part == 1 ? 13 : 26;
    return part == 0 ? 0 : part == 1 ? 7 : part == 2 ? 14 : part == 3 ? 20 : 26;
}
HD void affine_table_one(atab_ref tb, const apt& Pin, int pre_doublings) {
    fe beta;
    glv_beta(beta);
    apt P = Pin;
    const bool pid = apt_is_identity(Pin);
    if (pre_doublings) {
        ptj a;
        bool e0 = true;
        ptj_init(a);
        ptj_madd(a, e0, P, false);
#pragma nounroll
        for (int d = 0; d < pre_doublings; d++) ptj_dbl(a);
        fe zi, zi2;
        fe_inv(zi, a.Z);                          // the identity's Z is 0 and stays 0: every entry is stored as the identity below
        fe_sqr(zi2, zi);
        fe_mul(P.x, a.X, zi2);
        fe_mul(zi2, zi2, zi);
        fe_mul(P.y, a.Y, zi2);
    }
    atab_store(tb, 1, P, beta, pid);
    ptj T;
    bool empty = true;
    ptj_init(T);
    ptj_madd(T, empty, P, false);
    ptj_dbl(T);
    fe jx[15], jy[15], jz[15], pre[15];
#pragma nounroll
    for (int k = 0; k < 15; k++) {                // entry k holds (k + 2) P
        if (k) ptj_madd(T, empty, P, false);
        jx[k] = T.X; jy[k] = T.Y; jz[k] = T.Z;
        if (k) fe_mul(pre[k], pre[k - 1], T.Z);
        else fe_mul_small(pre[0], T.Z, 1);
    }
    fe inv;
    fe_inv(inv, pre[14]);
#pragma nounroll
    for (int k = 14; k >= 0; k--) {
        fe zi, zi2;
        if (k) { fe_mul(zi, inv, pre[k - 1]); fe_mul(inv, inv, jz[k]); }
        else zi = inv;
        apt R;
        fe_sqr(zi2, zi);
        fe_mul(R.x, jx[k], zi2);
        fe_mul(zi2, zi2, zi);
        fe_mul(R.y, jy[k], zi2);
        atab_store(tb, k + 2, R, beta, pid);
    }
}
// Point p of proof t's window tables straight from the caller's bytes -- what verify_phase1 parks in ws.pts: the 12 proof points it
// decodes and circuit_commitment = V + proof.r (reciprocal.rs:104), all of them the identity when anything in the proof is malformed
// -- so that the table kernel of a small call can run beside phase 1 instead of after it.
HD void verify_table_source(apt& P, const VerifyWs& ws, size_t t, int p) {
    const uint8_t* pv = ws.commitments + 64 * t;
    const uint8_t* pp = ws.proofs + (size_t)BPPP_U64_PROOF_BYTES * t;
    apt V, Pr, Q;
    fe_set_u32(P.x, 0);
    fe_set_u32(P.y, 0);
    bool ok = apt_from_xy64(V, pv);
#pragma nounroll
    for (int i = 0; i < 12; i++) {
        ok &= apt_from_xy64(Q, pp + 64 * i);
        if (i == p) P = Q;
    }
    ok &= apt_from_xy64(Pr, pp + 64 * 12);
    sc k;
    ok &= sc_from_be(k, pp + 832);
    ok &= sc_from_be(k, pp + 864);
    ok &= sc_from_be(k, pp + 896);
    if (p == 12 && ok) {
        pt s;
        pt_from_affine(s, V);
        pt_madd(s, s, Pr, apt_is_identity(Pr));
        pt_to_affine(P, s);
    }
    if (!ok) { fe_set_u32(P.x, 0); fe_set_u32(P.y, 0); }
}
// lane (point p, part h) of proof t: table slot h BPPP_VPOINTS + p
HD void verify_table_one(const VerifyWs& ws, size_t t, int p, int h, int parts, bool from_bytes = false) {
    apt P;
    if (from_bytes) verify_table_source(P, ws, t, p);
    else ws_ld_apt(P, ws.pts, ws.N, t, p);
    affine_table_one(atab_of(ws.atab, ws.N, t) + (h * BPPP_VPOINTS + p) * 16, P, 5 * split_begin(parts, h));
}
#endif
#if BPPP_VWIN == 5
// The 2M GLV half-scalars of an M-point sum, kept in registers, recoded for signed 5-bit windows:
//   w = |k| + OFF5,  OFF5 = sum_{i < 26} 16 * 32^i   (|k| < 2^128, so w < 2^130: 26 digits),  digit_i = ((w >> 5 i) & 31) - 16 in [-16, 15].
// glv_decompose hands over |k| + 0x8...8 (the 4-bit offset of the generic path); the difference of the two offsets is added here.
template <int M>
struct glv_words {
    u32 w[2 * M][5];
    bool neg[2 * M];
};
#define BPPP_VWINDOWS 26
HD void glv_recode5(u32 out[5], const u32 k4[5]) {
    const u32 D[5] = {0x987FB988u, 0x7FB987FBu, 0xB987FB98u, 0x87FB987Fu, 0xFFFFFFF9u};   // OFF5 - OFF4 mod 2^160
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) out[i] = addc(k4[i], D[i], c);
}
template <int M>
HD void glv_words_set(glv_words<M>& g, int j, const glv_split& sp) {
    glv_recode5(g.w[2 * j], sp.k1);
    glv_recode5(g.w[2 * j + 1], sp.k2);
    g.neg[2 * j] = sp.neg1;
    g.neg[2 * j + 1] = sp.neg2;
}
// the 2M digits of window i, 5 bits each, packed into one 64-bit word (2M <= 10): the window index is uniform over the wavefront,
// so this is a handful of selects per stream, once per window instead of once per addition
template <int M>
HD u64 glv_window_digits(const glv_words<M>& g, int i) {
    const int b = 5 * i, l = b >> 5, sh = b & 31;
    u64 pk = 0;
#pragma unroll
    for (int st = 0; st < 2 * M; st++) {
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            lo = (q == l) ? g.w[st][q] : lo;
            hi = (q == l + 1) ? g.w[st][q] : hi;
        }
        const u32 v = (u32)(((((u64)hi) << 32) | lo) >> sh) & 31u;
        pk |= (u64)v << (5 * st);
    }
    return pk;
}
template <int M>
HD void glv_digit_of(const glv_words<M>& g, u64 pk, int r, int& mag, bool& neg) {
    bool sneg = false;
#pragma unroll
    for (int st = 0; st < 2 * M; st++) sneg = (st == r) ? g.neg[st] : sneg;
    const int dg = (int)((pk >> (5 * r)) & 31u) - 16;
    mag = dg < 0 ? -dg : dg;
    neg = (dg < 0) != sneg;
}
// Progress-paced wave priority (VerifyWs::pace).  The SIMD's instruction arbiter serves the OLDER of two wavefronts first, so when a launch
// fills the chip exactly once (2^17 proofs: two wavefronts per SIMD, all started together) one wavefront of each pair runs almost
// as if alone and its partner mostly waits, then finishes alone at a lone wavefront's poor issue rate: 41 % of the SIMD-time of
// k_verify_round at 2^17 proofs has ONE wavefront resident (profiles/r06/r06_a_wave_timeline.txt).  With pacing on, a wavefront lowers
// its own priority (s_setprio 3 .. 0) as it advances through the windows of its sum, in spans that halve towards the end: whichever
// of the pair is behind is served first, the two reach the end within a few windows of each other, and the lone tail shrinks to that.
// window: 25 (first) .. 0 (last).  Wave-uniform; a handful of scalar instructions per window.
HD void straus_pace(bool pace, int window) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (!pace) return;
    if (window >= 13) __builtin_amdgcn_s_setprio(3);
    else if (window >= 6) __builtin_amdgcn_s_setprio(2);
    else if (window >= 3) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#else
    (void)pace; (void)window;
#endif
}
// sum_j k_j P_j over the affine tables; pidx[j] = table (proof point slot) of P_j.  26 windows x (5 doublings + 2M mixed
// additions); stream 2j is k1 of P_j, stream 2j + 1 its GLV partner (the entry's x times beta: the stream index is uniform over
// the wavefront, so that multiplication is behind a real branch).  The table entry of the next addition is requested before the
// current one starts.  Returns false when an exceptional addition was met (re-do with straus_affine_complete).
template <int M>
HD bool straus_affine_fast(pt& out, atab_ref tab, const int* pidx, const glv_words<M>& g, bool pace = false) {
    const int total = BPPP_VWINDOWS * 2 * M;
    fe beta;
    glv_beta(beta);
    straus_pace(pace, BPPP_VWINDOWS - 1);
    ptj acc;
    ptj_init(acc);
    bool empty = true;
    apt_packed cur_e, nxt_e;
    int cur_mag, nxt_mag;
    bool cur_neg, nxt_neg;
    u64 pk_cur = glv_window_digits<M>(g, BPPP_VWINDOWS - 1), pk_nxt = glv_window_digits<M>(g, BPPP_VWINDOWS - 2);
    glv_digit_of<M>(g, pk_cur, 0, cur_mag, cur_neg);
    cur_e = tab[pidx[0] * 16 + (cur_mag ? cur_mag - 1 : 0)];
    int r = 0, i = BPPP_VWINDOWS - 1;
#pragma nounroll
    for (int s = 0; s < total; s++) {
        // successor step (clamped at the end: requested, never consumed)
        int rn = r + 1, in = i;
        if (rn == 2 * M) { rn = 0; in = i - 1; }
        if (in < 0) { rn = r; in = i; }
        glv_digit_of<M>(g, in == i ? pk_cur : pk_nxt, rn, nxt_mag, nxt_neg);
        int pn = 0;
#pragma unroll
        for (int j = 0; j < M; j++) pn = (j == (rn >> 1)) ? pidx[j] : pn;
        nxt_e = tab[pn * 16 + (nxt_mag ? nxt_mag - 1 : 0)];
        if (r == 0 && s != 0) {
            straus_pace(pace, i);
#pragma nounroll
            for (int d = 0; d < 5; d++) ptj_dbl(acc);
        }
        apt e;
        bool id;
        apt_unpack(e, id, cur_e);
        if (r & 1) fe_mul(e.x, e.x, beta);            // wave-uniform: the GLV image (beta x, y)
        fe ny;
        fe_neg_m<1>(ny, e.y);
        fe_cmov(e.y, cur_neg, ny);
        ptj_madd(acc, empty, e, (cur_mag == 0) | id);
        cur_e = nxt_e;
        cur_mag = nxt_mag;
        cur_neg = nxt_neg;
        if (in != i) { pk_cur = pk_nxt; pk_nxt = glv_window_digits<M>(g, in > 0 ? in - 1 : 0); }
        r = rn;
        i = in;
    }
    const bool exceptional = !empty && fe_is_zero(acc.Z);
    ptj_to_pt(out, acc, empty);
    return !exceptional;
}
template <int M>
HD void straus_affine_complete(pt& out, atab_ref tab, const int* pidx, const glv_words<M>& g) {
    fe beta;
    glv_beta(beta);
    pt acc;
    pt_set_identity(acc);
#pragma nounroll
    for (int i = BPPP_VWINDOWS - 1; i >= 0; i--) {
        if (i != BPPP_VWINDOWS - 1) {
#pragma nounroll
            for (int d = 0; d < 5; d++) pt_dbl(acc, acc);
        }
        const u64 pk = glv_window_digits<M>(g, i);
#pragma nounroll
        for (int r = 0; r < 2 * M; r++) {
            int mag, pn = 0;
            bool neg, id;
            glv_digit_of<M>(g, pk, r, mag, neg);
#pragma unroll
            for (int j = 0; j < M; j++) pn = (j == (r >> 1)) ? pidx[j] : pn;
            apt e;
            apt_unpack(e, id, tab[pn * 16 + (mag ? mag - 1 : 0)]);
            if (r & 1) fe_mul(e.x, e.x, beta);
            fe ny;
            fe_neg_m<1>(ny, e.y);
            fe_cmov(e.y, neg, ny);
            pt_madd(acc, acc, e, (mag == 0) | id);
        }
    }
    out = acc;
}
#else
// The 2M GLV half-scalars of an M-point sum, kept in registers; digits are picked with select chains (no dynamic indexing).
template <int M>
struct glv_words {
    u32 w[2 * M][5];
    bool neg[2 * M];
};
template <int M>
HD void glv_words_set(glv_words<M>& g, int j, const glv_split& sp) {
#pragma unroll
    for (int l = 0; l < 5; l++) { g.w[2 * j][l] = sp.k1[l]; g.w[2 * j + 1][l] = sp.k2[l]; }
    g.neg[2 * j] = sp.neg1;
    g.neg[2 * j + 1] = sp.neg2;
}
// signed digit of stream r at window i: magnitude 0..8 and whether the table entry is negated
template <int M>
HD void glv_digit(const glv_words<M>& g, int r, int i, int& mag, bool& neg) {
    u32 word = 0;
    bool sneg = false;
#pragma unroll
    for (int st = 0; st < 2 * M; st++) {
#pragma unroll
        for (int l = 0; l < 5; l++) word = (st == r && l == (i >> 3)) ? g.w[st][l] : word;
        sneg = (st == r) ? g.neg[st] : sneg;
    }
    const int dg = (int)((word >> ((i & 7) * 4)) & 15) - 8;
    mag = dg < 0 ? -dg : dg;
    neg = (dg < 0) != sneg;
}
// sum_j k_j P_j over the affine tables; pidx[j] = table (proof point slot) of P_j.  33 windows x (4 doublings + 2M mixed
// additions); the table entry of the next addition is requested before the current one starts.  Returns false when an
// exceptional addition was met (re-do with straus_affine_complete).
template <int M>
HD bool straus_affine_fast(pt& out, atab_ref tab, const int* pidx, const glv_words<M>& g) {
    const int total = 33 * 2 * M;
    ptj acc;
    ptj_init(acc);
    bool empty = true;
    apt_packed cur_e, nxt_e;
    int cur_mag, nxt_mag;
    bool cur_neg, nxt_neg;
    glv_digit<M>(g, 0, 32, cur_mag, cur_neg);
    cur_e = tab[pidx[0] * 16 + (cur_mag ? cur_mag - 1 : 0)];
    int r = 0, i = 32;
#pragma nounroll
    for (int s = 0; s < total; s++) {
        // successor step (clamped at the end: requested, never consumed)
        int rn = r + 1, in = i;
        if (rn == 2 * M) { rn = 0; in = i - 1; }
        if (in < 0) { rn = r; in = i; }
        glv_digit<M>(g, rn, in, nxt_mag, nxt_neg);
        int pn = 0;
#pragma unroll
        for (int j = 0; j < M; j++) pn = (j == (rn >> 1)) ? pidx[j] : pn;
        nxt_e = tab[pn * 16 + (rn & 1) * 8 + (nxt_mag ? nxt_mag - 1 : 0)];
        if (r == 0 && s != 0) {
#pragma nounroll
            for (int d = 0; d < 4; d++) ptj_dbl(acc);
        }
        apt e;
        bool id;
        apt_unpack(e, id, cur_e);
        fe ny;
        fe_neg_m<1>(ny, e.y);
        fe_cmov(e.y, cur_neg, ny);
        ptj_madd(acc, empty, e, (cur_mag == 0) | id);
        cur_e = nxt_e;
        cur_mag = nxt_mag;
        cur_neg = nxt_neg;
        r = rn;
        i = in;
    }
    const bool exceptional = !empty && fe_is_zero(acc.Z);
    ptj_to_pt(out, acc, empty);
    return !exceptional;
}
template <int M>
HD void straus_affine_complete(pt& out, atab_ref tab, const int* pidx, const glv_words<M>& g) {
    pt acc;
    pt_set_identity(acc);
#pragma nounroll
    for (int i = 32; i >= 0; i--) {
        if (i != 32) {
#pragma nounroll
            for (int d = 0; d < 4; d++) pt_dbl(acc, acc);
        }
#pragma nounroll
        for (int r = 0; r < 2 * M; r++) {
            int mag, pn = 0;
            bool neg, id;
            glv_digit<M>(g, r, i, mag, neg);
#pragma unroll
            for (int j = 0; j < M; j++) pn = (j == (r >> 1)) ? pidx[j] : pn;
            apt e;
            apt_unpack(e, id, tab[pn * 16 + (r & 1) * 8 + (mag ? mag - 1 : 0)]);
            fe ny;
            fe_neg_m<1>(ny, e.y);
            fe_cmov(e.y, neg, ny);
            pt_madd(acc, acc, e, (mag == 0) | id);
        }
    }
    out = acc;
}
#endif   // BPPP_VWIN
template <int M>
HD void straus_affine(pt& out, atab_ref tab, const int* pidx, const glv_words<M>& g, bool pace = false) {
    if (!straus_affine_fast<M>(out, tab, pidx, g, pace)) {
        // the out-of-line call takes addresses: hand it copies, so the hot loop's scalars and accumulator stay in registers
        glv_words<M> gc = g;
        int pc[M];
#pragma unroll
        for (int j = 0; j < M; j++) pc[j] = pidx[j];
        pt o;
        straus_affine_complete<M>(o, tab, pc, gc);
        out = o;
    }
}

#if BPPP_VWIN == 5
// lane q's share of the split sum (below): part h = q / 2M (windows split_begin(parts, h) .. split_begin(parts, h + 1) - 1) of stream
// r = q % 2M over the table of 2^(5 split_begin(parts, h)) P (slot pidx + BPPP_VPOINTS h: verify_table_one); q >= 2M parts: nothing.
// False on an exceptional addition.
template <int M>
HD bool straus_split_lane(pt& part, atab_ref tab, const int* pidx, const glv_words<M>& g, int q, int parts) {
    fe beta;
    glv_beta(beta);
    const bool have = q < 2 * M * parts;
    const int h = have ? q / (2 * M) : 0, r = have ? q - 2 * M * h : 0;
    u32 w[5];
    bool sneg = false;
    int pn = 0;
#pragma unroll
    for (int l = 0; l < 5; l++) w[l] = 0;
#pragma unroll
    for (int st = 0; st < 2 * M; st++) {
#pragma unroll
        for (int l = 0; l < 5; l++) w[l] = (st == r) ? g.w[st][l] : w[l];
        sneg = (st == r) ? g.neg[st] : sneg;
        pn = (st == r) ? pidx[st >> 1] : pn;
    }
    const bool img = (r & 1) != 0;
    const int base = (pn + BPPP_VPOINTS * h) * 16, w0 = split_begin(parts, h), nw = split_begin(parts, h + 1) - w0;   // 13, or 7 / 6, windows
    auto digit = [&](int i, int& mag, bool& neg) {
        const int b = 5 * i, l = b >> 5, sh = b & 31;
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) { lo = (k == l) ? w[k] : lo; hi = (k == l + 1) ? w[k] : hi; }
        const int dg = (int)((u32)(((((u64)hi) << 32) | lo) >> sh) & 31u) - 16;
        mag = dg < 0 ? -dg : dg;
        neg = (dg < 0) != sneg;
    };
    ptj acc;
    ptj_init(acc);
    bool empty = true;
    int cur_mag, nxt_mag;
    bool cur_neg, nxt_neg;
    apt_packed cur_e, nxt_e;
    digit(w0 + nw - 1, cur_mag, cur_neg);
    cur_e = tab[base + (cur_mag ? cur_mag - 1 : 0)];
#pragma nounroll
    for (int i = nw - 1; i >= 0; i--) {
        digit(w0 + (i > 0 ? i - 1 : 0), nxt_mag, nxt_neg);       // the next window's entry is requested before this window's doublings
        nxt_e = tab[base + (nxt_mag ? nxt_mag - 1 : 0)];
        if (i != nw - 1) {
#pragma nounroll
            for (int d = 0; d < 5; d++) ptj_dbl(acc);
        }
        apt e;
        bool id;
        apt_unpack(e, id, cur_e);
        fe bx, ny;
        fe_mul(bx, e.x, beta);
        fe_cmov(e.x, img, bx);
        fe_neg_m<1>(ny, e.y);
        fe_cmov(e.y, cur_neg, ny);
        ptj_madd(acc, empty, e, (cur_mag == 0) | id | !have);
        cur_e = nxt_e;
        cur_mag = nxt_mag;
        cur_neg = nxt_neg;
    }
    const bool exceptional = !empty && fe_is_zero(acc.Z);
    ptj_to_pt(part, acc, empty);
    return !exceptional;
}
#endif
#if defined(__HIPCC__) && BPPP_VWIN == 5
// The same M-point sum spread over a GROUP OF FOUR LANES: lane q takes the GLV streams q, q + 4, q + 8 (< 2M; stream r is point
// r >> 1, its image if r & 1), i.e. 26 windows x (5 doublings + 1 .. 3 mixed additions) per lane instead of 26 x (5 + 2M), then a
// two-step shuffle tree.  For batches so small that the chip is mostly empty (one lane per proof leaves SIMDs without a wavefront)
// this shortens the dependent chain a call has to wait for; the doublings are repeated on every lane, so it is not used once one lane
// per proof fills the SIMDs.  All four lanes of a group must be active and hold the same g / pidx; every lane ends with the total.
template <int M, int G = 4>
__device__ __forceinline__ void straus_affine_g4(pt& out, atab_ref tab, const int* pidx, const glv_words<M>& g, int q) {
    // (G = 2: groups of two lanes, streams q, q + 2, ... -- for batches that fill half of the wavefront slots with one lane per proof)
    constexpr int NS = (2 * M + G - 1) / G;      // streams per lane (the last one may be missing on the last lanes of the group)
    fe beta;
    glv_beta(beta);
    u32 w[NS][5];
    bool sneg[NS], img[NS], have[NS];
    int pn[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
        const int r = q + G * j;                 // my j-th stream
        have[j] = r < 2 * M;
        sneg[j] = false;
        pn[j] = 0;
#pragma unroll
        for (int l = 0; l < 5; l++) w[j][l] = 0;
#pragma unroll
        for (int st = 0; st < 2 * M; st++) {
#pragma unroll
            for (int l = 0; l < 5; l++) w[j][l] = (st == r) ? g.w[st][l] : w[j][l];
            sneg[j] = (st == r) ? g.neg[st] : sneg[j];
            pn[j] = (st == r) ? pidx[st >> 1] : pn[j];
        }
        img[j] = (r & 1) != 0;
    }
    auto digit = [&](const u32 (&ww)[5], bool sn, int i, int& mag, bool& neg) {
        const int b = 5 * i, l = b >> 5, sh = b & 31;
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) { lo = (k == l) ? ww[k] : lo; hi = (k == l + 1) ? ww[k] : hi; }
        const int dg = (int)((u32)(((((u64)hi) << 32) | lo) >> sh) & 31u) - 16;
        mag = dg < 0 ? -dg : dg;
        neg = (dg < 0) != sn;
    };
    ptj acc;
    ptj_init(acc);
    bool empty = true;
    int cur_mag[NS], nxt_mag[NS];
    bool cur_neg[NS], nxt_neg[NS];
    apt_packed cur_e[NS], nxt_e[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
        digit(w[j], sneg[j], BPPP_VWINDOWS - 1, cur_mag[j], cur_neg[j]);
        cur_e[j] = tab[pn[j] * 16 + (cur_mag[j] ? cur_mag[j] - 1 : 0)];
    }
#pragma nounroll
    for (int i = BPPP_VWINDOWS - 1; i >= 0; i--) {
#pragma unroll
        for (int j = 0; j < NS; j++) {           // the next window's entries are requested before this window's doublings
            digit(w[j], sneg[j], i > 0 ? i - 1 : 0, nxt_mag[j], nxt_neg[j]);
            nxt_e[j] = tab[pn[j] * 16 + (nxt_mag[j] ? nxt_mag[j] - 1 : 0)];
        }
        if (i != BPPP_VWINDOWS - 1) {
#pragma nounroll
            for (int d = 0; d < 5; d++) ptj_dbl(acc);
        }
#pragma unroll
        for (int j = 0; j < NS; j++) {
            apt e;
            bool id;
            apt_unpack(e, id, cur_e[j]);
            fe bx, ny;
            fe_mul(bx, e.x, beta);
            fe_cmov(e.x, img[j], bx);
            fe_neg_m<1>(ny, e.y);
            fe_cmov(e.y, cur_neg[j], ny);
            ptj_madd(acc, empty, e, (cur_mag[j] == 0) | id | !have[j]);
            cur_e[j] = nxt_e[j];
            cur_mag[j] = nxt_mag[j];
            cur_neg[j] = nxt_neg[j];
        }
    }
    int bad = (!empty && fe_is_zero(acc.Z)) ? 1 : 0;
#pragma unroll
    for (int m = 1; m < G; m <<= 1) bad |= __shfl_xor(bad, m, 64);
    if (bad) {                      // an exceptional addition somewhere in the group: every lane re-does the whole sum completely
        straus_affine_complete<M>(out, tab, pidx, g);
        return;
    }
    pt part;
    ptj_to_pt(part, acc, empty);
    lane_group_sum<G>(part);
    out = part;
}
// The same sum with every stream cut in PARTS (2 or 4): lane q < 2M PARTS of a group of G walks part h = q / 2M of stream r = q % 2M
// over the table of 2^(5 split_begin(PARTS, h)) P (slot pidx + BPPP_VPOINTS h: verify_table_one) -- 12 x 5 doublings + 13 mixed additions
// (two parts) or at most 6 x 5 + 7 (four) per lane instead of 25 x 5 + 26 ... 78, then a log2(G)-step shuffle tree.  For calls that leave the chip empty (a handful of proofs):
// the length of the chain is all that counts there.  The other lanes of the group hold no stream and add the identity.  All G lanes of
// a group must be active and hold the same g / pidx; every lane ends with the total.
template <int M, int G, int PARTS>
__device__ __forceinline__ void straus_affine_split(pt& out, atab_ref tab, const int* pidx, const glv_words<M>& g, int q) {
    static_assert(2 * M * PARTS <= G, "a lane per part of a stream");
    pt part;
    int bad = straus_split_lane<M>(part, tab, pidx, g, q, PARTS) ? 0 : 1;
#pragma unroll
    for (int m = 1; m < G; m <<= 1) bad |= __shfl_xor(bad, m, 64);
    if (bad) {                      // an exceptional addition somewhere in the group: every lane re-does the whole sum completely
        straus_affine_complete<M>(out, tab, pidx, g);
        return;
    }
    lane_group_sum<G>(part);
    out = part;
}
#endif

// ---------------------------------------------------------------- phase 1: decode, transcript up to tau, scalar derivation
// reciprocal.rs:98-104 + circuit.rs:155-228 (closed forms of SURVEY.md 8a)
// TR = strobe (sponge state in registers: host emulation, small batches) or strobe_lds (state in the workgroup's LDS: k_verify_phase1).
// `tr` arrives holding the transcript every proof starts from (ws.base); status_in carries flags the caller already raised.
#if defined(__HIPCC__)
__device__ __forceinline__ void sc_group_sum16(sc& a, int group = 16);      // the sum over a group of lanes, on every lane (shuffles; below)
#endif
// a^e for 1 <= e <= 31, the same ten multiplications whatever e is (the lane forms: a lane's power of mu, lambda, ... directly)
HD void sc_pow_u5(sc& r, const sc& a, unsigned e) {      // a^e for 1 <= e <= 31, the same ten multiplications whatever e is
    sc acc, tmp;
    sc_set_u32(acc, 1);
#pragma unroll
    for (int bit = 4; bit >= 0; bit--) {
        sc_mul(acc, acc, acc);
        sc_mul(tmp, acc, a);
        const bool take = ((e >> bit) & 1u) != 0;
#pragma unroll
        for (int i = 0; i < 8; i++) acc.v[i] = take ? tmp.v[i] : acc.v[i];
    }
    r = acc;
}
// lane >= 0: one of SIXTEEN lanes that run phase 1 for proof t together (small calls: k_verify_phase1_g16).  Decode, transcript and
// challenges are done by all sixteen alike (identical values, identical stores); the scalar section -- a chain of ~245 dependent
// multiplications and 36 workspace round trips on one lane -- is spread: lane j inverts e + j itself (and every lane mu tau), takes term
// j of the 16-term loop with its powers by sc_pow_u5, and the three sums meet by shuffles.  Every lane of a group must be active.
template <typename TR>
HD void verify_phase1_on(const VerifyWs& ws, size_t t, TR& tr, int32_t status_in, int lane = -1) {
    const size_t N = ws.N;
    int32_t status = status_in;
    BPPP_STAMP(t, 0);
    const uint8_t* pv = ws.commitments + 64 * t;
    const uint8_t* pp = ws.proofs + (size_t)BPPP_U64_PROOF_BYTES * t;
    // The 13 proof points are decoded one at a time and go straight to the workspace (a 13-point array would live in scratch
    // memory); the transcript below reloads each one right before it is hashed.  Only V and proof.r (needed for V + r) stay in
    // registers.
    apt V, Pr;
    bool ok = apt_from_xy64(V, pv);
#pragma nounroll
    for (int i = 0; i < 12; i++) {
        apt Q;
        ok &= apt_from_xy64(Q, pp + 64 * i);
        ws_st_apt(ws.pts, N, t, i, Q);
    }
    ok &= apt_from_xy64(Pr, pp + 64 * 12);
    sc l0, l1, n0;
    ok &= sc_from_be(l0, pp + 832);
    ok &= sc_from_be(l1, pp + 864);
    ok &= sc_from_be(n0, pp + 896);
    if (!ok) {
        // keep control flow uniform: run on harmless values, the status forces accept = 0 at the end
        status |= ST_BAD_ENCODING;
        apt zero;
        fe_set_u32(zero.x, 0);
        fe_set_u32(zero.y, 0);
        V = zero;
        Pr = zero;
#pragma nounroll
        for (int i = 0; i < 12; i++) ws_st_apt(ws.pts, N, t, i, zero);
        sc_set_u32(l0, 0); sc_set_u32(l1, 0); sc_set_u32(n0, 0);
    }
    sc e, rho, lambda, beta, delta, tau;
    BPPP_STAMP(t, 1);
    app_point(tr, "reciprocal_commitment", V);                          // reciprocal.rs:99
    bool cok = t_get_challenge(tr, "reciprocal_challenge", e);          // reciprocal.rs:100
    // circuit_commitment = commitment + proof.r                         (reciprocal.rs:104)
    apt Vr;
    {
        pt s;
        pt_from_affine(s, V);
        pt_madd(s, s, Pr, apt_is_identity(Pr));
        pt_to_affine(Vr, s);
    }
    ws_st_apt(ws.pts, N, t, 12, Vr);
    BPPP_STAMP(t, 2);
    {
        apt Q;
        ws_ld_apt(Q, ws.pts, N, t, 0);
        app_point(tr, "commitment_cl", Q);                               // circuit.rs:155-159
        ws_ld_apt(Q, ws.pts, N, t, 1);
        app_point(tr, "commitment_cr", Q);
        ws_ld_apt(Q, ws.pts, N, t, 2);
        app_point(tr, "commitment_co", Q);
    }
    app_point(tr, "commitment_v", Vr);
    BPPP_STAMP(t, 3);
    cok &= t_get_challenge(tr, "circuit_rho", rho);                      // circuit.rs:161-164
    cok &= t_get_challenge(tr, "circuit_lambda", lambda);
    cok &= t_get_challenge(tr, "circuit_beta", beta);
    cok &= t_get_challenge(tr, "circuit_delta", delta);
    BPPP_STAMP(t, 4);
    {
        apt Q;
        ws_ld_apt(Q, ws.pts, N, t, 3);
        app_point(tr, "commitment_cs", Q);                               // circuit.rs:189
    }
    cok &= t_get_challenge(tr, "circuit_tau", tau);                      // circuit.rs:191
    if (!cok) {
        status |= ST_DEGENERATE;
        sc_set_u32(e, 1); sc_set_u32(rho, 1); sc_set_u32(lambda, 1); sc_set_u32(beta, 1); sc_set_u32(delta, 1); sc_set_u32(tau, 1);
    }
    ws_st_transcript(ws.tstate, N, t, tr);
    ws_st8(ws.chal, N, t, 0, e.v); ws_st8(ws.chal, N, t, 1, rho.v); ws_st8(ws.chal, N, t, 2, lambda.v);
    ws_st8(ws.chal, N, t, 3, beta.v); ws_st8(ws.chal, N, t, 4, delta.v); ws_st8(ws.chal, N, t, 5, tau.v);
    ws_st8(ws.lns, N, t, 0, l0.v); ws_st8(ws.lns, N, t, 1, l1.v); ws_st8(ws.lns, N, t, 2, n0.v);

    // ---- scalars.  One Fn inversion for {mu, tau, e+0..e+15} (the reference: util.rs:119, circuit.rs:192, reciprocal.rs:181 x256)
    sc mu;
    BPPP_STAMP(t, 5);
    sc_mul(mu, rho, rho);                                                // circuit.rs:166
    // Montgomery's trick over the 18 values a_0 = mu, a_1 = tau, a_{2+j} = e + j.  The values are recomputed where needed and the
    // running products / the inverses travel through workspace slots that are free at this point (cvec: products, fsc:
    // inverses) instead of two 18-element arrays in scratch memory.
    auto batch_value = [&](int i, sc& v) {
        if (i == 0) v = mu;
        else if (i == 1) v = tau;
        else {
            sc js;
            sc_set_u32(js, (u32)(i - 2));
            sc_add(v, e, js);
        }
    };
    bool zero_inv = sc_is_zero(delta);                                   // circuit.rs:196 unwraps delta^-1 although u64 never uses it
    sc one;
    sc_set_u32(one, 1);
    sc mu_inv, tau_inv, tau2, tau3, S, t1, t2, musum, ps;
    sc_mul(tau2, tau, tau);
    sc_mul(tau3, tau2, tau);
#if defined(__HIP_DEVICE_COMPILE__)
    if (lane >= 0) {
        // inverses: (mu tau)^-1 on every lane, (e + lane)^-1 on its lane; a zero is replaced by one, as in the one-lane form below
        sc m1 = mu, tq = tau, ej, js, inv;
        bool z = sc_is_zero(mu);
        zero_inv |= z;
        if (z) m1 = one;
        z = sc_is_zero(tau);
        zero_inv |= z;
        if (z) tq = one;
        sc_mul(t1, m1, tq);
        sc_inv(inv, t1);
        sc_mul(mu_inv, inv, tq);
        sc_mul(tau_inv, inv, m1);
        sc_set_u32(js, (u32)lane);
        sc_add(ej, e, js);
        z = sc_is_zero(ej);
        if (z) ej = one;
        int zany = z ? 1 : 0;
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) zany |= __shfl_xor(zany, m, 64);
        zero_inv |= zany != 0;
        if (zero_inv) status |= ST_DEGENERATE;
        sc einv;
        sc_inv(einv, ej);
        ws_st8(ws.fsc, N, t, lane, einv.v);                  // (e + j)^-1 at fsc slot j
        BPPP_STAMP(t, 6);
        // this lane's powers; S = sum lambda^i, musum = sum mu^i over the group
        sc lp, mp, mip;
        sc_pow_u5(lp, lambda, (unsigned)lane + 1);
        sc_pow_u5(mp, mu, (unsigned)lane + 1);
        sc_pow_u5(mip, mu_inv, (unsigned)lane + 1);
        S = lp;
        musum = mp;
        sc_group_sum16(S);
        sc_group_sum16(musum);
        sc tau_e, two_tau2_S, p16, pn;
        sc_mul(tau_e, tau, e);
        sc_mul(two_tau2_S, tau2, S);
        sc_add(two_tau2_S, two_tau2_S, two_tau2_S);
        // term j = lane of the loop below
        sc_set_u64(p16, (u64)1 << (4 * lane));
        sc_mul(t1, tau2, p16);
        sc_sub(t2, S, lp);
        sc_mul(t2, t2, tau);
        sc_add(t1, t1, t2);
        sc_mul(pn, t1, mip);
        sc_add(pn, pn, tau_e);
        ws_st8(ws.sc0, N, t, 1 + lane, pn.v);
        sc_mul(ps, pn, pn);
        sc_mul(ps, ps, mp);
        sc_group_sum16(ps);
        sc_mul(t1, two_tau2_S, einv);
        sc_sub(t1, t1, lp);
        ws_st8(ws.cvec, N, t, 9 + lane, t1.v);
    } else
#endif
    {
        sc run = one;
#pragma nounroll
        for (int i = 0; i < 18; i++) {
            sc v;
            batch_value(i, v);
            const bool z = sc_is_zero(v);
            zero_inv |= z;
            if (z) v = one;
            ws_st8(ws.cvec, N, t, i, run.v);      // product of a_0 .. a_{i-1}
            sc_mul(run, run, v);
        }
        if (zero_inv) status |= ST_DEGENERATE;
        sc inv;
        sc_inv(inv, run);
#pragma nounroll
        for (int i = 17; i >= 0; i--) {
            sc v, pre, ai;
            batch_value(i, v);
            if (sc_is_zero(v)) v = one;
            ws_ld8(pre.v, ws.cvec, N, t, i);
            sc_mul(ai, inv, pre);                 // a_i^-1
            sc_mul(inv, inv, v);
            if (i >= 2) ws_st8(ws.fsc, N, t, i - 2, ai.v);      // (e + j)^-1 at fsc slot j
            else if (i == 1) tau_inv = ai;
            else mu_inv = ai;
        }
        BPPP_STAMP(t, 6);
        // S = sum_{i=1..16} lambda^i ; musum = sum_{i=1..16} mu^i
        sc lp = lambda, mp = mu;
        S = lambda;
        musum = mu;
#pragma nounroll
        for (int i = 1; i < 16; i++) {
            sc_mul(lp, lp, lambda);
            sc_add(S, S, lp);
            sc_mul(mp, mp, mu);
            sc_add(musum, musum, mp);
        }
        sc tau_e, two_tau2_S;
        sc_mul(tau_e, tau, e);
        sc_mul(two_tau2_S, tau2, S);
        sc_add(two_tau2_S, two_tau2_S, two_tau2_S);
        sc_set_u32(ps, 0);
        sc mip = mu_inv;   // mu^-(j+1)
        lp = lambda;       // lambda^(j+1)
        mp = mu;           // mu^(j+1)
#pragma nounroll
        for (int j = 0; j < 16; j++) {
            // pn_tau[j] = mu^-(j+1) (tau^2 16^j + tau (S - lambda^(j+1))) + tau e          (circuit.rs:198-200)
            sc p16, pn;
            sc_set_u64(p16, (u64)1 << (4 * j));
            sc_mul(t1, tau2, p16);
            sc_sub(t2, S, lp);
            sc_mul(t2, t2, tau);
            sc_add(t1, t1, t2);
            sc_mul(pn, t1, mip);
            sc_add(pn, pn, tau_e);
            ws_st8(ws.sc0, N, t, 1 + j, pn.v);
            // ps_tau += mu^(j+1) pn^2                                                       (circuit.rs:202)
            sc_mul(t1, pn, pn);
            sc_mul(t1, t1, mp);
            sc_add(ps, ps, t1);
            // cl_tau[j] = 2 tau^2 S (e+j)^-1 - lambda^(j+1)                                 (circuit.rs:222-226)
            sc einv;
            ws_ld8(einv.v, ws.fsc, N, t, j);
            sc_mul(t1, two_tau2_S, einv);
            sc_sub(t1, t1, lp);
            ws_st8(ws.cvec, N, t, 9 + j, t1.v);
            sc_mul(mip, mip, mu_inv);
            sc_mul(lp, lp, lambda);
            sc_mul(mp, mp, mu);
        }
    }
    // ps_tau -= 2 tau^3 sum mu^i   (a_l = 0, a_m = 1s; circuit.rs:203-204)
    sc two_tau3;
    sc_add(two_tau3, tau3, tau3);
    sc_mul(t1, two_tau3, musum);
    sc_sub(ps, ps, t1);
    ws_st8(ws.sc0, N, t, 0, ps.v);
    ws_st8(ws.sc0, N, t, 17, tau_inv.v);
    sc_neg(t1, delta);
    ws_st8(ws.sc0, N, t, 18, t1.v);
    ws_st8(ws.sc0, N, t, 19, tau.v);
    sc_neg(t1, tau2);
    ws_st8(ws.sc0, N, t, 20, t1.v);
    ws_st8(ws.sc0, N, t, 21, two_tau3.v);   // v_ = 2 (V + r), times tau^3 (circuit.rs:182-187,235)
    BPPP_STAMP(t, 7);
    // cr_tau = [1, beta/tau, beta tau, ..., beta tau^7]                                  (circuit.rs:208-218)
    ws_st8(ws.cvec, N, t, 0, one.v);
    sc_mul(t1, beta, tau_inv);
    ws_st8(ws.cvec, N, t, 1, t1.v);
    sc bt = beta;
#pragma nounroll
    for (int i = 2; i < 9; i++) {
        sc_mul(bt, bt, tau);
        ws_st8(ws.cvec, N, t, i, bt.v);
    }
    BPPP_STAMP(t, 8);
    ws.status[t] = status;
    if (ws.trace) {
        uint8_t* tb = ws.trace + 704 * t;
        sc_to_be(tb, e); sc_to_be(tb + 32, rho); sc_to_be(tb + 64, lambda); sc_to_be(tb + 96, beta);
        sc_to_be(tb + 128, delta); sc_to_be(tb + 160, tau);
        apt_to_xy64(tb + 320, Vr);
    }
}
// the transcript a proof starts from: the caller's pre-loaded state if there is one (and merlin could be in it), else ws.base
HD void phase1_start_state(strobe& tr, int32_t& status, const VerifyWs& ws, size_t t) {
    tr = ws.base;
    if (ws.states) {
        strobe pre;
        const bool sok = strobe_from_bytes(pre, ws.states + (size_t)BPPP_TRANSCRIPT_STATE_BYTES * (ws.n_states == 1 ? 0 : t));
        if (sok) tr = pre;
        else status |= ST_BAD_ENCODING;      // not a state merlin could be in: flag the proof, run on the shared base
    }
}
HD void verify_phase1(const VerifyWs& ws, size_t t) {          // sponge state in registers
    int32_t status = ST_OK;
    strobe tr;
    phase1_start_state(tr, status, ws, t);
    verify_phase1_on(ws, t, tr, status);
}
#if defined(__HIP_DEVICE_COMPILE__)
// sponge state in LDS: lds_col = this lane's column of the workgroup's [50][64]-word block (merlin.h: strobe_lds)
__device__ __forceinline__ void verify_phase1_lds(const VerifyWs& ws, size_t t, u32* lds_col) {
    int32_t status = ST_OK;
    strobe_lds tl;
    tl.col = lds_col;
    {
        strobe tr;
        phase1_start_state(tr, status, ws, t);
        strobe_lds_load(tl, tr);
    }
    verify_phase1_on(ws, t, tl, status);
}
#endif

// ---------------------------------------------------------------- phase 2b: C0 fixed-base part: ps_tau*g + <g_vec, pn_tau>  (circuit.rs:206) -> pfix
// lane-group form: every lane of the proof's group computes a partial sum; the group total is stored by _store.
HD void verify_c0_fixed_ranges(FbRanges& rg) { fb_ranges_one(rg, 0, 0, 17); }
HD void verify_c0_fixed_store(const VerifyWs& ws, size_t t, const pt& total) { ws_st_pt(ws.pfix, ws.N, t, total); }   // added in round 1
// single-thread form (host emulation in tests/emul, thread order = lane order)
HD void verify_c0_fixed(const VerifyWs& ws, size_t t) {
    FbRanges rg;
    verify_c0_fixed_ranges(rg);
    pt acc;
    fb_sum_serial(acc, fb_of(ws), t, ws.sc0, rg);
    verify_c0_fixed_store(ws, t, acc);
}
// ---------------------------------------------------------------- phase 2a: C0 variable-base part (circuit.rs:230-235)
HD void verify_c0_var(const VerifyWs& ws, size_t t, int group_lane = -1, int group_size = 4) {
    const size_t N = ws.N;
    const int pslot[5] = {3, 2, 0, 1, 12};  // c_s, c_o, c_l, c_r, V+r  <->  sc0 slots 17..21
    glv_words<5> g;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        sc k;
        ws_ld8(k.v, ws.sc0, N, t, 17 + j);
        glv_split sp;
        glv_decompose(sp, k);
        glv_words_set<5>(g, j, sp);
    }
    BPPP_STAMP(t, 20);
    pt acc;
#if defined(__HIP_DEVICE_COMPILE__) && BPPP_VWIN == 5
    if (group_lane >= 0 && group_size == 64) straus_affine_split<5, 64, 4>(acc, atab_of(ws.atab, ws.N, t), pslot, g, group_lane);
    else if (group_lane >= 0 && group_size == 32) straus_affine_split<5, 32, 2>(acc, atab_of(ws.atab, ws.N, t), pslot, g, group_lane);
    else if (group_lane >= 0) straus_affine_g4<5>(acc, atab_of(ws.atab, ws.N, t), pslot, g, group_lane);
    else
#endif
        straus_affine<5>(acc, atab_of(ws.atab, ws.N, t), pslot, g, ws.pace != 0);
    (void)group_lane; (void)group_size;
    BPPP_STAMP(t, 21);
    ws_st_pt(ws.acc, N, t, acc);   // the fixed-base part (pfix) is added at the top of round 1
}
// C0 = variable-base part + fixed-base part, ahead of round 1 when the rounds find 1 / Z ready (ws.zinv): the shared inversion needs C0's Z
HD void verify_c0_join(const VerifyWs& ws, size_t t) {
    pt C, F;
    ws_ld_pt(C, ws.acc, ws.N, t);
    ws_ld_pt(F, ws.pfix, ws.N, t);
    pt_add(C, C, F);
    ws_st_pt(ws.acc, ws.N, t, C);
}
// ---------------------------------------------------------------- phase 3 (k = 1..4): one WNLA round (wnla.rs:84-102)
// group_lane >= 0: this lane is one of four consecutive lanes that all run the round for proof t (identical work and identical
// stores, except the sum, which they share: straus_affine_g4) -- the small-batch kernels; -1: one lane per proof
// TR = strobe (sponge state in registers) or strobe_lds (in the workgroup's LDS: k_verify_round); tr arrives unloaded
// part: 0 the whole round; 1 its HEAD only (C_{k-1} to affine, transcript, challenge -- leaves the affine C_{k-1} in ws.acc and y_k in
// ws.chal); 2 its TAIL only (the two-point sum and C_k).  In two kernels the last round's tail runs beside the final fixed-base sum,
// which needs nothing but the challenges (bppp_u64.hip: batches whose one-lane kernels are a lone wavefront per SIMD).
template <typename TR>
HD void verify_round_on(const VerifyWs& ws, size_t t, int k, TR& tr, int group_lane = -1, int group_size = 4, int part = 0) {
    const size_t N = ws.N;
    pt C;
    ws_ld_pt(C, ws.acc, N, t);
    apt Ca;
    sc y;
    if (part == 2) {        // the head left C_{k-1} affine (Z = 1, or the identity) and y_k
        Ca.x = C.X; Ca.y = C.Y;
        if (fe_is_zero(C.Z)) { fe_set_u32(Ca.x, 0); fe_set_u32(Ca.y, 0); }
        ws_ld8(y.v, ws.chal, N, t, 5 + k);
    } else {
        if (k == 1 && !ws.zinv) {   // C0 = variable-base part (acc) + fixed-base part (pfix); the two kernels run concurrently on two streams
            pt F;
            ws_ld_pt(F, ws.pfix, N, t);
            pt_add(C, C, F);
        }
        BPPP_STAMP(t, 9);
        if (ws.zinv) {              // shared inversions: C0 was joined by verify_c0_join, 1 / Z is there (pt_to_affine's two products remain)
            fe zi;
            ws_ld_fe(zi, ws.zinv, N, t, 0, 1);
            fe_mul(Ca.x, C.X, zi);
            fe_mul(Ca.y, C.Y, zi);
        } else pt_to_affine(Ca, C);
        BPPP_STAMP(t, 10);
        ws_ld_transcript(tr, ws.tstate, N, t);
        app_point(tr, "wnla_com", Ca);                                       // wnla.rs:88-92
        {   // the round's proof points are only hashed here (the sum below reads their window tables): loaded one at a time, right
            // before their append, so that nothing but the sponge state and C is live across the permutations
            apt Q;
            ws_ld_apt(Q, ws.pts, N, t, 8 + (4 - k));   // proof.x.last()
            app_point(tr, "wnla_x", Q);
            ws_ld_apt(Q, ws.pts, N, t, 4 + (4 - k));   // proof.r.last()
            app_point(tr, "wnla_r", Q);
        }
        t_append_u64(tr, "l.sz", (u64)(32 >> (k - 1)));
        t_append_u64(tr, "n.sz", (u64)(16 >> (k - 1)));
        bool cok = t_get_challenge(tr, "wnla_challenge", y);                 // wnla.rs:94
        if (!cok) {
            ws.status[t] |= ST_DEGENERATE;
            sc_set_u32(y, 1);
        }
        BPPP_STAMP(t, 11);
        ws_st_transcript(ws.tstate, N, t, tr);
        ws_st8(ws.chal, N, t, 5 + k, y.v);
        if (ws.trace) {
            uint8_t* tb = ws.trace + 704 * t;
            sc_to_be(tb + 32 * (5 + k), y);
            apt_to_xy64(tb + 320 + 64 * k, Ca);
        }
        if (part == 1) {
            pt Cs;
            pt_from_affine(Cs, Ca);
            ws_st_pt(ws.acc, N, t, Cs);
            return;
        }
    }
    // com_ = com + y X + (y^2 - 1) R                                     (wnla.rs:100-102)
    sc y2m1, one;
    sc_set_u32(one, 1);
    sc_mul(y2m1, y, y);
    sc_sub(y2m1, y2m1, one);
    const int pslot[2] = {8 + (4 - k), 4 + (4 - k)};
    glv_words<2> g;
    glv_split sp;
    glv_decompose(sp, y);
    glv_words_set<2>(g, 0, sp);
    glv_decompose(sp, y2m1);
    glv_words_set<2>(g, 1, sp);
    BPPP_STAMP(t, 12);
    pt acc;
#if defined(__HIP_DEVICE_COMPILE__) && BPPP_VWIN == 5
    if (group_lane >= 0 && group_size == 16) straus_affine_split<2, 16, 4>(acc, atab_of(ws.atab, ws.N, t), pslot, g, group_lane);
    else if (group_lane >= 0 && group_size == 8) straus_affine_split<2, 8, 2>(acc, atab_of(ws.atab, ws.N, t), pslot, g, group_lane);
    else if (group_lane >= 0 && group_size == 4) straus_affine_g4<2, 4>(acc, atab_of(ws.atab, ws.N, t), pslot, g, group_lane);
    else if (group_lane >= 0) straus_affine_g4<2, 2>(acc, atab_of(ws.atab, ws.N, t), pslot, g, group_lane);
    else
#endif
        straus_affine<2>(acc, atab_of(ws.atab, ws.N, t), pslot, g, ws.pace != 0);
    (void)group_lane; (void)group_size;
    BPPP_STAMP(t, 13);
    pt_madd(acc, acc, Ca, apt_is_identity(Ca));
    ws_st_pt(ws.acc, N, t, acc);
}
HD void verify_round(const VerifyWs& ws, size_t t, int k, int group_lane = -1, int group_size = 4, int part = 0) {
    strobe tr;
    verify_round_on(ws, t, k, tr, group_lane, group_size, part);
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void verify_round_lds(const VerifyWs& ws, size_t t, int k, u32* lds_col) {
    strobe_lds tl;
    tl.col = lds_col;
    verify_round_on(ws, t, k, tl);
}
#endif
// ---------------------------------------------------------------- phase 4: base case (wnla.rs:80-82 with :66-72), generators unrolled
// (Round 5 measured a form that keeps ch / cg as four register-resident quarter tables -- no store-to-load round trips through the
// workspace, the cause of this kernel's 40 % memory wait: 2.64 ms against 1.9.  The 128 VGPRs of tables cost the kernel the occupancy
// that hides its latency today.  Not kept.)
HD void verify_final_scalars(const VerifyWs& ws, size_t t) {
    const size_t N = ws.N;
    BPPP_STAMP(t, 22);
    sc rho, y[4], rk[4], l0, l1, n0;
    ws_ld8(rho.v, ws.chal, N, t, 1);
#pragma nounroll
    for (int k = 0; k < 4; k++) ws_ld8(y[k].v, ws.chal, N, t, 6 + k);
    ws_ld8(l0.v, ws.lns, N, t, 0);
    ws_ld8(l1.v, ws.lns, N, t, 1);
    ws_ld8(n0.v, ws.lns, N, t, 2);
    // rho_1 = rho, rho_{k+1} = mu_k, mu_{k+1} = mu_k^2 (wnla.rs:109-110): rho_k = rho^(2^(k-1)); final mu = rho^32
    rk[0] = rho;
#pragma nounroll
    for (int k = 1; k < 4; k++) sc_mul(rk[k], rk[k - 1], rk[k - 1]);
    sc mu5;
    sc_mul(mu5, rk[3], rk[3]);
    sc_mul(mu5, mu5, mu5);
    // ch[b] = prod_{k: bit k of b} y_{k+1}        (h_vec / c folding, wnla.rs:96,98 unrolled)
    // cg[b] = prod_k (bit k of b ? y_{k+1} : rho_{k+1})   (g_vec folding, wnla.rs:97 unrolled)
    // Built in place in the output slots (cg[b] at fsc slot 1 + b, ch[b] at slot 17 + b) instead of two 16-element arrays in
    // scratch memory; the final products overwrite them.
    sc one;
    sc_set_u32(one, 1);
    ws_st8(ws.fsc, N, t, 17, one.v);
    ws_st8(ws.fsc, N, t, 1, one.v);
#pragma nounroll
    for (int k = 0; k < 4; k++) {
        const int half = 1 << k;
#pragma nounroll
        for (int b = 0; b < half; b++) {
            sc chb, cgb, tmp;
            ws_ld8(chb.v, ws.fsc, N, t, 17 + b);
            ws_ld8(cgb.v, ws.fsc, N, t, 1 + b);
            sc_mul(tmp, chb, y[k]);
            ws_st8(ws.fsc, N, t, 17 + b + half, tmp.v);
            sc_mul(tmp, cgb, y[k]);
            ws_st8(ws.fsc, N, t, 1 + b + half, tmp.v);
            sc_mul(tmp, cgb, rk[k]);
            ws_st8(ws.fsc, N, t, 1 + b, tmp.v);
        }
    }
    // c'_0, c'_1 = folded c (c[25..31] = 0)
    sc c0f, c1f, tmp, cv, chv;
    sc_set_u32(c0f, 0);
    sc_set_u32(c1f, 0);
#pragma nounroll
    for (int i = 0; i < 25; i++) {
        ws_ld8(cv.v, ws.cvec, N, t, i);
        ws_ld8(chv.v, ws.fsc, N, t, 17 + (i & 15));
        sc_mul(tmp, cv, chv);
        if (i < 16) sc_add(c0f, c0f, tmp);
        else sc_add(c1f, c1f, tmp);
    }
    // v = <c', l> + n0^2 mu'   (wnla.rs:67 with weight_vector_mul exponent 1, util.rs:28-44)
    sc v, w;
    sc_mul(v, c0f, l0);
    sc_mul(w, c1f, l1);
    sc_add(v, v, w);
    sc_mul(w, n0, n0);
    sc_mul(w, w, mu5);
    sc_add(v, v, w);
    ws_st8(ws.fsc, N, t, 0, v.v);
#pragma nounroll
    for (int i = 0; i < 16; i++) {
        sc cgv;
        ws_ld8(cgv.v, ws.fsc, N, t, 1 + i);
        sc_mul(tmp, n0, cgv);
        ws_st8(ws.fsc, N, t, 1 + i, tmp.v);
        ws_ld8(chv.v, ws.fsc, N, t, 17 + i);
        sc_mul(tmp, l0, chv);
        ws_st8(ws.fsc, N, t, 17 + i, tmp.v);
        sc_mul(tmp, l1, chv);
        ws_st8(ws.fsc, N, t, 33 + i, tmp.v);
    }
}
#if defined(__HIPCC__)
// The same scalars by SIXTEEN lanes per proof, for calls that leave the chip empty (bppp_u64.hip: the small-call path): lane b forms
// ch[b] and cg[b] -- four conditional multiplications each, in registers -- and its three output scalars; the two folded c values are
// a sum over the group (shuffles).  The one-lane form above walks 118 multiplications whose operands travel through the workspace
// (a store-to-load round trip per step): 145 us for a lone proof, a seventh of it here.  Every lane of a group must be active.
__device__ __forceinline__ void sc_group_sum16(sc& a, int group) {      // group: 16 or a smaller power of two (declared above, default 16)
#pragma unroll
    for (int m = 1; m < group; m <<= 1) {
        sc o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.v[i] = __shfl_xor(a.v[i], m, 64);
        sc_add(a, a, o);
    }
}
__device__ __forceinline__ void verify_final_scalars_lane(const VerifyWs& ws, size_t t, int b) {
    const size_t N = ws.N;
    sc rho, y[4], rk[4], l0, l1, n0, mu5, tmp;
    ws_ld8(rho.v, ws.chal, N, t, 1);
#pragma unroll
    for (int k = 0; k < 4; k++) ws_ld8(y[k].v, ws.chal, N, t, 6 + k);
    ws_ld8(l0.v, ws.lns, N, t, 0);
    ws_ld8(l1.v, ws.lns, N, t, 1);
    ws_ld8(n0.v, ws.lns, N, t, 2);
    rk[0] = rho;
#pragma unroll
    for (int k = 1; k < 4; k++) sc_mul(rk[k], rk[k - 1], rk[k - 1]);
    sc_mul(mu5, rk[3], rk[3]);
    sc_mul(mu5, mu5, mu5);
    sc ch, cg;
    sc_set_u32(ch, 1);
    sc_set_u32(cg, 1);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const bool bit = ((b >> k) & 1) != 0;
        sc_mul(tmp, ch, y[k]);
#pragma unroll
        for (int i = 0; i < 8; i++) ch.v[i] = bit ? tmp.v[i] : ch.v[i];
        sc f;
#pragma unroll
        for (int i = 0; i < 8; i++) f.v[i] = bit ? y[k].v[i] : rk[k].v[i];
        sc_mul(cg, cg, f);
    }
    // c'_0 = sum_{i < 16} c[i] ch[i], c'_1 = sum_{16 <= i < 25} c[i] ch[i - 16]
    sc c0f, c1f, cv;
    ws_ld8(cv.v, ws.cvec, N, t, b);
    sc_mul(c0f, cv, ch);
    ws_ld8(cv.v, ws.cvec, N, t, b < 9 ? 16 + b : 16);
    sc_mul(c1f, cv, ch);
    if (b >= 9) sc_set_u32(c1f, 0);
    sc_group_sum16(c0f);
    sc_group_sum16(c1f);
    if (b == 0) {
        sc v, w;
        sc_mul(v, c0f, l0);
        sc_mul(w, c1f, l1);
        sc_add(v, v, w);
        sc_mul(w, n0, n0);
        sc_mul(w, w, mu5);
        sc_add(v, v, w);
        ws_st8(ws.fsc, N, t, 0, v.v);
    }
    sc_mul(tmp, n0, cg);
    ws_st8(ws.fsc, N, t, 1 + b, tmp.v);
    sc_mul(tmp, l0, ch);
    ws_st8(ws.fsc, N, t, 17 + b, tmp.v);
    sc_mul(tmp, l1, ch);
    ws_st8(ws.fsc, N, t, 33 + b, tmp.v);
}
#endif
HD void verify_final_check_ranges(FbRanges& rg) { fb_ranges_one(rg, 0, 0, BPPP_NG); }
HD void verify_final_check_store(const VerifyWs& ws, size_t t, const pt& rhs) { ws_st_pt(ws.pfix, ws.N, t, rhs); }
// accept bit: C4 == rhs as projective classes (wnla.rs:81), and no status flag
HD void verify_accept(const VerifyWs& ws, size_t t) {
    const size_t N = ws.N;
    pt C, rhs;
    ws_ld_pt(C, ws.acc, N, t);
    ws_ld_pt(rhs, ws.pfix, N, t);
    bool eq = pt_eq(C, rhs);                                             // wnla.rs:81
    ws.accept[t] = (eq && ws.status[t] == ST_OK) ? 1 : 0;
    if (ws.trace) {
        apt Ca;
        pt_to_affine(Ca, C);
        apt_to_xy64(ws.trace + 704 * t + 320 + 64 * 5, Ca);
    }
}
// The caller's `&mut Transcript` after verify (Ownership): the state after the last challenge (wnla.rs:94 in round 4,
// a PRF operation: cur_flags = I|A|C = 7).  A proof whose inputs k256 would have refused to deserialize never reaches the
// reference's verify, so its transcript comes back untouched.
HD void verify_export_state(const VerifyWs& ws, size_t t) {
    if (!ws.states_out) return;
    uint8_t* out = ws.states_out + (size_t)BPPP_TRANSCRIPT_STATE_BYTES * t;
    if (ws.status[t] & ST_BAD_ENCODING) {
        if (ws.states) {
            const uint8_t* in = ws.states + (size_t)BPPP_TRANSCRIPT_STATE_BYTES * (ws.n_states == 1 ? 0 : t);
#pragma nounroll
            for (int i = 0; i < BPPP_TRANSCRIPT_STATE_BYTES; i++) out[i] = in[i];
        } else {
            strobe_to_bytes(out, ws.base, 2);      // Transcript::new ends with an AD operation (the dom-sep message)
        }
        return;
    }
    strobe tr;
    ws_ld_strobe(tr, ws.tstate, ws.N, t);
    strobe_to_bytes(out, tr, 7);
}
HD void verify_final_check(const VerifyWs& ws, size_t t) {
    FbRanges rg;
    verify_final_check_ranges(rg);
    pt acc;
    fb_sum_serial(acc, fb_of(ws), t, ws.fsc, rg);
    verify_final_check_store(ws, t, acc);
    verify_accept(ws, t);
}

// ---------------------------------------------------------------- wire format: SEC1 compressed points
// The reference's SerializableProof (reciprocal.rs:37-41, circuit.rs:37-46, wnla.rs:33-38) holds k256 `AffinePoint`s, whose
// byte form is 33-byte SEC1 compressed (02|03 || x; the identity is 33 zero bytes), and 32-byte big-endian scalars: a u64
// proof is 13*33 + 3*32 = 525 bytes, its commitment 33.  One lane per point recovers y = sqrt(x^3 + 7) (p = 3 mod 4: one
// exponentiation) and writes the 64-byte x||y form the verify pipeline reads.  An undecodable point (bad tag, x >= p,
// x^3 + 7 a non-residue -- k256's from_bytes fails) becomes (1, 0), which is never on the curve, so verify_phase1 flags
// BPPP_ST_BAD_ENCODING for that proof.
#define BPPP_U64_PROOF_SEC1_BYTES 525
HD void sec1_decompress_to_xy64(uint8_t* out64, const uint8_t* in33) {
    u32 nz = 0;
#pragma nounroll
    for (int i = 0; i < 33; i++) nz |= in33[i];
    fe x, rhs, y, y2, seven;
    bool ok = fe_from_be(x, in33 + 1);
    const uint8_t tag = in33[0];
    ok &= (tag == 2) | (tag == 3);
    fe_sqr(rhs, x);
    fe_mul(rhs, rhs, x);
    fe_set_u32(seven, 7);
    fe_add(rhs, rhs, seven);
    fe_sqrt_candidate(y, rhs);
    fe_sqr(y2, y);
    ok &= fe_eq(y2, rhs);
    fe ny;
    fe_neg_m<1>(ny, y);
    fe_cmov(y, fe_is_odd(y) != ((tag & 1) != 0), ny);
    // an undecodable point must never alias the identity (0, 0): it becomes (1, 0), which is off the curve for every tag and x
    // (0 != 1 + 7), so verify_phase1 / apt_from_xy64 flag the proof.  (x, 0) would not do: x = 0 mod p -- 02||00..00, a bad tag
    // over x = 0, 02||p -- would come out as 64 zero bytes, the identity's encoding.
    fe zero, one;
    fe_set_u32(zero, 0);
    fe_set_u32(one, 1);
    fe_cmov(x, !ok, one);
    fe_cmov(y, !ok, zero);
    const bool identity = nz == 0;
    fe_cmov(x, identity, zero);
    fe_cmov(y, identity, zero);
    fe_to_be(out64, x);
    fe_to_be(out64 + 32, y);
}
// lane j of proof t: j = 0 commitment, 1..13 proof points, 14 copies the three scalars
HD void sec1_expand_lane(uint8_t* commitments64, uint8_t* proofs928, const uint8_t* commitments33, const uint8_t* proofs525,
                         size_t t, int j) {
    if (j == 0) sec1_decompress_to_xy64(commitments64 + 64 * t, commitments33 + 33 * t);
    else if (j <= 13) sec1_decompress_to_xy64(proofs928 + (size_t)BPPP_U64_PROOF_BYTES * t + 64 * (j - 1),
                                              proofs525 + (size_t)BPPP_U64_PROOF_SEC1_BYTES * t + 33 * (j - 1));
    else if (j == 14) {
#pragma nounroll
        for (int i = 0; i < 96; i++)
            proofs928[(size_t)BPPP_U64_PROOF_BYTES * t + 832 + i] = proofs525[(size_t)BPPP_U64_PROOF_SEC1_BYTES * t + 429 + i];
    }
}

// the other direction (the prover's output as the crate serialises it: GroupEncoding / serde of SerializableProof, wnla.rs:33-61,
// circuit.rs:36-76): lane j of proof t: j = 0 commitment, 1..13 proof points, 14 copies the three scalars.  The 64-byte points are the
// library's own output (canonical, on the curve): tag 02 / 03 by the parity of y, the identity (64 zero bytes) -> 33 zero bytes.
HD void sec1_compress_lane(uint8_t* commitments33, uint8_t* proofs525, const uint8_t* commitments64, const uint8_t* proofs928, size_t t, int j) {
    if (j <= 13) {
        const uint8_t* in = j == 0 ? commitments64 + 64 * t : proofs928 + (size_t)BPPP_U64_PROOF_BYTES * t + 64 * (j - 1);
        uint8_t* out = j == 0 ? commitments33 + 33 * t : proofs525 + (size_t)BPPP_U64_PROOF_SEC1_BYTES * t + 33 * (j - 1);
        uint8_t any = 0;
#pragma nounroll
        for (int i = 0; i < 64; i++) any |= in[i];
        out[0] = any ? (uint8_t)(2 + (in[63] & 1)) : 0;
#pragma nounroll
        for (int i = 0; i < 32; i++) out[1 + i] = in[i];
    } else if (j == 14) {
#pragma nounroll
        for (int i = 0; i < 96; i++)
            proofs525[(size_t)BPPP_U64_PROOF_SEC1_BYTES * t + 429 + i] = proofs928[(size_t)BPPP_U64_PROOF_BYTES * t + 832 + i];
    }
}

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
