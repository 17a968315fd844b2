// The u64 verifier's per-proof workspace in HBM (VerifyWs: structure-of-arrays, limb-major), its accessors, the serialized transcript
// state of the C ABI, and the optional phase stamps of the diagnostic builds.  Split out of verify_core.h in round 6; shared by the
// fixed-base unit (fb_core.h), the variable-base unit (straus_core.h) and the protocol phases (verify_core.h).
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


}  // namespace bppp
