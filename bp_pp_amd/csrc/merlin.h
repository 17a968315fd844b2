// Merlin v1.0 transcripts (STROBE-128 over Keccak-f[1600]) for gfx950, one transcript per lane.
//
// Replaces merlin 3.0.0 as used by the reference: `Transcript::new` (benches/range_proof.rs:32),
// `append_message` (transcript.rs:7), `append_u64` (wnla.rs:91-92), `challenge_bytes` (transcript.rs:12).
// Byte-exact Fiat-Shamir challenges are a hard parity requirement: every later point depends on them.
//
// The sponge state is 25 x u64 per lane.  All lanes of a batch run the same transcript schedule, so the
// byte position `pos` is wave-uniform and the dynamically indexed state word is the same register slot
// (or scratch dword) for every lane.
#pragma once
#include "field.h"

namespace bppp {

struct strobe {
    u64 st[25];
    u32 pos, pos_begin;
};

#define BPPP_STROBE_R 166

// 64-bit rotate by a compile-time amount.  On gfx950 a 64-bit shift is a quarter-rate instruction and hipcc turns the
// shift-or idiom into two of them plus two ORs; two v_alignbit_b32 (full rate) do the same job.
HD u64 rotl64(u64 v, int r) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 lo = (u32)v, hi = (u32)(v >> 32);
    u32 rlo, rhi;
    const int s = r & 31;
    if (s == 0) { rlo = lo; rhi = hi; }
    else {
        rhi = __builtin_amdgcn_alignbit(hi, lo, 32 - s);      // ({hi, lo} >> (32 - s)) low word = bits of the left-rotated high word
        rlo = __builtin_amdgcn_alignbit(lo, hi, 32 - s);
    }
    return (r & 32) ? (((u64)rlo << 32) | rhi) : (((u64)rhi << 32) | rlo);
#else
    return (v << r) | (v >> (64 - r));
#endif
}
// Keccak-f[1600] round constants without a table in memory: a round constant only has bits at positions 2^j - 1 (j = 0..6);
// the 24 seven-bit patterns are packed 8 per 64-bit literal and expanded with a few scalar operations.
HD u64 keccak_rc(int rnd) {
    // pattern bit j of round i = bit (2^j - 1) of RC[i]
    const u64 K[3] = {0x00ABE509FE178D01ULL, 0x00A7767BF4CD460EULL, 0x00E886C79CC5A452ULL};   // rounds 0-7, 8-15, 16-23
    u64 word = 0;
#pragma unroll
    for (int q = 0; q < 3; q++) word = (q == (rnd >> 3)) ? K[q] : word;
    const u32 c = (u32)(word >> (7 * (rnd & 7))) & 0x7Fu;
    const u32 lo = (c & 1u) | ((c >> 1 & 1u) << 1) | ((c >> 2 & 1u) << 3) | ((c >> 3 & 1u) << 7) | ((c >> 4 & 1u) << 15) | ((c >> 5 & 1u) << 31);
    const u32 hi = (c >> 6 & 1u) << 31;
    return ((u64)hi << 32) | lo;
}

HD void keccak_f1600(u64 a[25]) {
    u64 a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a04 = a[4], a05 = a[5], a06 = a[6], a07 = a[7], a08 = a[8], a09 = a[9],
        a10 = a[10], a11 = a[11], a12 = a[12], a13 = a[13], a14 = a[14], a15 = a[15], a16 = a[16], a17 = a[17], a18 = a[18],
        a19 = a[19], a20 = a[20], a21 = a[21], a22 = a[22], a23 = a[23], a24 = a[24];
#pragma nounroll
    for (int rnd = 0; rnd < 24; rnd++) {
        // theta
        u64 c0 = a00 ^ a05 ^ a10 ^ a15 ^ a20, c1 = a01 ^ a06 ^ a11 ^ a16 ^ a21, c2 = a02 ^ a07 ^ a12 ^ a17 ^ a22,
            c3 = a03 ^ a08 ^ a13 ^ a18 ^ a23, c4 = a04 ^ a09 ^ a14 ^ a19 ^ a24;
        u64 d0 = c4 ^ rotl64(c1, 1), d1 = c0 ^ rotl64(c2, 1), d2 = c1 ^ rotl64(c3, 1), d3 = c2 ^ rotl64(c4, 1), d4 = c3 ^ rotl64(c0, 1);
        a00 ^= d0; a05 ^= d0; a10 ^= d0; a15 ^= d0; a20 ^= d0;
        a01 ^= d1; a06 ^= d1; a11 ^= d1; a16 ^= d1; a21 ^= d1;
        a02 ^= d2; a07 ^= d2; a12 ^= d2; a17 ^= d2; a22 ^= d2;
        a03 ^= d3; a08 ^= d3; a13 ^= d3; a18 ^= d3; a23 ^= d3;
        a04 ^= d4; a09 ^= d4; a14 ^= d4; a19 ^= d4; a24 ^= d4;
        // rho + pi: B[y, 2x+3y] = rot(A[x, y], r[x, y]); lanes are a[x + 5y]
        u64 b00 = a00;
        u64 b10 = rotl64(a01, 1), b20 = rotl64(a02, 62), b05 = rotl64(a03, 28), b15 = rotl64(a04, 27);
        u64 b16 = rotl64(a05, 36), b01 = rotl64(a06, 44), b11 = rotl64(a07, 6), b21 = rotl64(a08, 55), b06 = rotl64(a09, 20);
        u64 b07 = rotl64(a10, 3), b17 = rotl64(a11, 10), b02 = rotl64(a12, 43), b12 = rotl64(a13, 25), b22 = rotl64(a14, 39);
        u64 b23 = rotl64(a15, 41), b08 = rotl64(a16, 45), b18 = rotl64(a17, 15), b03 = rotl64(a18, 21), b13 = rotl64(a19, 8);
        u64 b14 = rotl64(a20, 18), b24 = rotl64(a21, 2), b09 = rotl64(a22, 61), b19 = rotl64(a23, 56), b04 = rotl64(a24, 14);
        // chi
        a00 = b00 ^ (~b01 & b02); a01 = b01 ^ (~b02 & b03); a02 = b02 ^ (~b03 & b04); a03 = b03 ^ (~b04 & b00); a04 = b04 ^ (~b00 & b01);
        a05 = b05 ^ (~b06 & b07); a06 = b06 ^ (~b07 & b08); a07 = b07 ^ (~b08 & b09); a08 = b08 ^ (~b09 & b05); a09 = b09 ^ (~b05 & b06);
        a10 = b10 ^ (~b11 & b12); a11 = b11 ^ (~b12 & b13); a12 = b12 ^ (~b13 & b14); a13 = b13 ^ (~b14 & b10); a14 = b14 ^ (~b10 & b11);
        a15 = b15 ^ (~b16 & b17); a16 = b16 ^ (~b17 & b18); a17 = b17 ^ (~b18 & b19); a18 = b18 ^ (~b19 & b15); a19 = b19 ^ (~b15 & b16);
        a20 = b20 ^ (~b21 & b22); a21 = b21 ^ (~b22 & b23); a22 = b22 ^ (~b23 & b24); a23 = b23 ^ (~b24 & b20); a24 = b24 ^ (~b20 & b21);
        // iota
        a00 ^= keccak_rc(rnd);
    }
    a[0] = a00; a[1] = a01; a[2] = a02; a[3] = a03; a[4] = a04; a[5] = a05; a[6] = a06; a[7] = a07; a[8] = a08; a[9] = a09;
    a[10] = a10; a[11] = a11; a[12] = a12; a[13] = a13; a[14] = a14; a[15] = a15; a[16] = a16; a[17] = a17; a[18] = a18; a[19] = a19;
    a[20] = a20; a[21] = a21; a[22] = a22; a[23] = a23; a[24] = a24;
}

// ---- sponge state access.  The byte position is a runtime value, but every lane of a wavefront runs the same transcript
// schedule, so it is wave-uniform: it is read into a scalar register and the state word is picked with a scalar switch, which
// keeps the 25 state words in VGPRs (a dynamically indexed `st[pos >> 3]` would put the state in scratch memory and make every
// absorbed byte a ~500-900 cycle store-load round trip; measured with tools/probes/phase_probe.py: 58 k cycles per challenge).
// Bytes are absorbed in chunks of up to 4 (one switch per chunk), and a 32-byte challenge is read from the first four words.
HD u32 st_uniform(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (u32)__builtin_amdgcn_readfirstlane((int)x);
#else
    return x;
#endif
}
#define BPPP_ST_CASES(OP) \
    OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15) OP(16) OP(17) OP(18) OP(19) \
    OP(20) OP(21) OP(22) OP(23)
// st[w] ^= v0, st[w + 1] ^= v1   (w <= 23 for every byte position below the rate + 2)
HD void st_xor2(strobe& s, u32 w, u64 v0, u64 v1) {
    switch (st_uniform(w)) {
#define BPPP_OP(i) case i: s.st[i] ^= v0; s.st[i + 1] ^= v1; break;
        BPPP_ST_CASES(BPPP_OP)
#undef BPPP_OP
        default: s.st[24] ^= v0; break;
    }
}
HD u64 st_get_word(const strobe& s, u32 w) {
    u64 r = s.st[24];
    switch (st_uniform(w)) {
#define BPPP_OP(i) case i: r = s.st[i]; break;
        BPPP_ST_CASES(BPPP_OP)
#undef BPPP_OP
        default: break;
    }
    return r;
}
HD void st_and_word(strobe& s, u32 w, u64 m) {
    switch (st_uniform(w)) {
#define BPPP_OP(i) case i: s.st[i] &= m; break;
        BPPP_ST_CASES(BPPP_OP)
#undef BPPP_OP
        default: s.st[24] &= m; break;
    }
}
// XOR up to 4 bytes (`word`, little-endian, unused high bytes zero) into the state at byte position pos (pos + nb <= 168)
HD void st_xor_bytes(strobe& s, u32 pos, u32 word) {
    const u32 sh = 8 * (pos & 7);
    const u64 v0 = (u64)word << sh;
    const u64 v1 = sh > 32 ? (u64)word >> (64 - sh) : 0;
    st_xor2(s, pos >> 3, v0, v1);
}
HD void strobe_run_f(strobe& s) {
    st_xor_bytes(s, s.pos, (s.pos_begin & 0xFFu) | (0x04u << 8));
    s.st[(BPPP_STROBE_R + 1) >> 3] ^= (u64)0x80 << (8 * ((BPPP_STROBE_R + 1) & 7));
    keccak_f1600(s.st);
    s.pos = 0;
    s.pos_begin = 0;
}
// bytes 0..31 of the state as four words, read and zeroed (the PRF output right after a forced permutation)
HD u64 st_take_word(strobe& s, int k) {
    const u64 v = s.st[k];
    s.st[k] = 0;
    return v;
}
#define BPPP_LDS_STRIDE 64
#if defined(__HIP_DEVICE_COMPILE__)
// ---- the same sponge with its 200 state bytes in LDS (device only; k_verify_phase1).  A kernel that hashes a lot -- phase 1 is
// ~15 permutations and ~1.3 KB of absorbed bytes per proof -- keeps the 50 state words live in VGPRs across code that also wants
// registers for field arithmetic; at the 256-register cap of a two-wave kernel they end up in scratch (HBM-backed).  Here the state
// lives in the workgroup's LDS, word-major: 32-bit word i of lane l at col[i * 64] with col = block + l, so a wavefront's access to
// one word is 64 consecutive banks (no conflicts), the byte position being wave-uniform.  Absorbing is a read-modify-write of one or
// two words; the permutation loads the 50 words, runs in registers, stores them back, out of line (one copy of Keccak-f per kernel
// and no register pressure on its callers).
struct strobe_lds {
    u32* col;
    u32 pos, pos_begin;
};
__device__ __noinline__ inline void keccak_f1600_lds(u32* col) {
    u64 a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = (u64)col[(2 * i) * BPPP_LDS_STRIDE] | ((u64)col[(2 * i + 1) * BPPP_LDS_STRIDE] << 32);
    keccak_f1600(a);
#pragma unroll
    for (int i = 0; i < 25; i++) { col[(2 * i) * BPPP_LDS_STRIDE] = (u32)a[i]; col[(2 * i + 1) * BPPP_LDS_STRIDE] = (u32)(a[i] >> 32); }
}
__device__ __forceinline__ void st_xor_bytes(strobe_lds& s, u32 pos, u32 word) {      // pos + 4 <= 172 < 200: both words exist
    const u32 w = st_uniform(pos >> 2), sh = 8 * (st_uniform(pos) & 3);
    s.col[w * BPPP_LDS_STRIDE] ^= word << sh;
    if (sh) s.col[(w + 1) * BPPP_LDS_STRIDE] ^= word >> (32 - sh);
}
__device__ __forceinline__ void strobe_run_f(strobe_lds& s) {
    st_xor_bytes(s, s.pos, (s.pos_begin & 0xFFu) | (0x04u << 8));
    s.col[((BPPP_STROBE_R + 1) >> 2) * BPPP_LDS_STRIDE] ^= 0x80u << (8 * ((BPPP_STROBE_R + 1) & 3));
    keccak_f1600_lds(s.col);
    s.pos = 0;
    s.pos_begin = 0;
}
__device__ __forceinline__ u64 st_take_word(strobe_lds& s, int k) {
    const u64 v = (u64)s.col[(2 * k) * BPPP_LDS_STRIDE] | ((u64)s.col[(2 * k + 1) * BPPP_LDS_STRIDE] << 32);
    s.col[(2 * k) * BPPP_LDS_STRIDE] = 0;
    s.col[(2 * k + 1) * BPPP_LDS_STRIDE] = 0;
    return v;
}
__device__ __forceinline__ u32 strobe_squeeze_byte(strobe_lds& s) {   // read and zero
    const u32 w = st_uniform(s.pos >> 2), sh = 8 * (st_uniform(s.pos) & 3);
    const u32 v = s.col[w * BPPP_LDS_STRIDE];
    s.col[w * BPPP_LDS_STRIDE] = v & ~(0xFFu << sh);
    s.pos++;
    if (s.pos == BPPP_STROBE_R) strobe_run_f(s);
    return (v >> sh) & 0xFFu;
}
__device__ __forceinline__ void strobe_lds_load(strobe_lds& d, const strobe& s) {
#pragma unroll
    for (int i = 0; i < 25; i++) { d.col[(2 * i) * BPPP_LDS_STRIDE] = (u32)s.st[i]; d.col[(2 * i + 1) * BPPP_LDS_STRIDE] = (u32)(s.st[i] >> 32); }
    d.pos = s.pos;
    d.pos_begin = s.pos_begin;
}
#endif
// absorb nb <= 4 bytes of `word`; the sponge permutation appears once here, whatever the chunk straddles
template <typename S>
HD void strobe_absorb_chunk(S& s, u32 word, u32 nb) {
#pragma nounroll
    while (nb) {
        s.pos = st_uniform(s.pos);
        u32 take = BPPP_STROBE_R - s.pos;
        take = take < nb ? take : nb;
        const u32 mask = take >= 4 ? 0xFFFFFFFFu : ((1u << (8 * take)) - 1u);
        st_xor_bytes(s, s.pos, word & mask);
        s.pos += take;
        nb -= take;
        word = take >= 4 ? 0u : (word >> (8 * take));
        if (s.pos == BPPP_STROBE_R) strobe_run_f(s);
    }
}
HD void strobe_absorb(strobe& s, const uint8_t* d, u32 n) {
#pragma nounroll
    for (u32 i = 0; i < n; i++) strobe_absorb_chunk(s, d[i], 1);
}
HD u32 strobe_squeeze_byte(strobe& s) {   // read and zero
    const u32 sh = 8 * (s.pos & 7);
    const u32 b = (u32)(st_get_word(s, s.pos >> 3) >> sh) & 0xFFu;
    st_and_word(s, s.pos >> 3, ~((u64)0xFF << sh));
    s.pos++;
    if (s.pos == BPPP_STROBE_R) strobe_run_f(s);
    return b;
}
template <typename S>
HD void strobe_squeeze(S& s, uint8_t* d, u32 n) {
#pragma nounroll
    for (u32 i = 0; i < n; i++) d[i] = (uint8_t)strobe_squeeze_byte(s);
}
HD void strobe_begin_op(strobe& s, u32 flags, bool more) {
    if (more) return;
    const u32 old_begin = s.pos_begin;
    s.pos_begin = s.pos + 1;
    strobe_absorb_chunk(s, (old_begin & 0xFFu) | (flags << 8), 2);
    if ((flags & (4 | 32)) && s.pos != 0) strobe_run_f(s);
}
HD void strobe_meta_ad(strobe& s, const uint8_t* d, u32 n, bool more) { strobe_begin_op(s, 16 | 2, more); strobe_absorb(s, d, n); }
HD void strobe_ad(strobe& s, const uint8_t* d, u32 n, bool more) { strobe_begin_op(s, 2, more); strobe_absorb(s, d, n); }
HD void strobe_prf(strobe& s, uint8_t* d, u32 n) { strobe_begin_op(s, 1 | 2 | 4, false); strobe_squeeze(s, d, n); }

HD void strobe_init(strobe& s, const uint8_t* proto, u32 n) {
    for (int i = 0; i < 25; i++) s.st[i] = 0;
    const uint8_t hdr[18] = {1, BPPP_STROBE_R + 2, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    for (u32 i = 0; i < 18; i++) st_xor_bytes(s, i, hdr[i]);
    keccak_f1600(s.st);
    s.pos = 0;
    s.pos_begin = 0;
    strobe_meta_ad(s, proto, n, false);
}

// ---- merlin::Transcript
// label bytes as compile-time packed words (little-endian within a word), picked with a select chain: no memory access
template <int L>
HD u32 label_word(const char (&label)[L], u32 c) {
    u32 r = 0;
#pragma unroll
    for (int k = 0; k < (L - 1 + 3) / 4; k++) {
        u32 wk = 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (4 * k + j < L - 1) wk |= (u32)(uint8_t)label[4 * k + j] << (8 * j);
        r = (c == (u32)k) ? wk : r;
    }
    return r;
}
// One transcript operation = meta-AD(label) || meta-AD(length, continued) || AD(message) or PRF: header, label, length and
// message go through ONE loop of 4-byte chunks, so the (inlined) sponge permutation appears once per operation.
// `msg_word(c)` supplies message bytes 4c .. 4c+3 (little-endian, bytes past the end zero); kind 0 = append_message,
// kind 1 = challenge_bytes (ends after the PRF header and the forced permutation; the caller squeezes).
template <typename S, int L, typename F>
HD void t_op_absorb(S& t, const char (&label)[L], u32 nbytes, int kind, F msg_word) {
    const u32 LL = (u32)(L - 1), NLC = (LL + 3) / 4, NMC = kind == 0 ? (nbytes + 3) / 4 : 0;
    const u32 total = 1 + NLC + 1 + 1 + NMC;
#pragma nounroll
    for (u32 i = 0; i < total; i++) {
        u32 word, nb;
        if (i == 0 || i == NLC + 2) {                 // begin_op: previous pos_begin, then the flags byte
            const u32 flags = i == 0 ? (16u | 2u) : (kind == 0 ? 2u : (1u | 2u | 4u));
            word = (t.pos_begin & 0xFFu) | (flags << 8);
            nb = 2;
            t.pos_begin = t.pos + 1;
        } else if (i <= NLC) {
            const u32 c = i - 1;
            word = label_word(label, c);
            nb = LL - 4 * c < 4 ? LL - 4 * c : 4;
        } else if (i == NLC + 1) {
            word = nbytes;                            // 4-byte little-endian length: meta-AD continued, no header
            nb = 4;
        } else {
            const u32 c = i - NLC - 3;
            word = msg_word(c);
            nb = nbytes - 4 * c < 4 ? nbytes - 4 * c : 4;
        }
        strobe_absorb_chunk(t, word, nb);
    }
    if (kind != 0 && t.pos != 0) strobe_run_f(t);     // PRF carries the C flag: force F before squeezing
}
// append_message with the message bytes in memory (host-side transcript construction, known-answer tests)
template <int L>
HD void t_append(strobe& t, const char (&label)[L], const uint8_t* m, u32 n) {
    t_op_absorb(t, label, n, 0, [&](u32 c) -> u32 {
        u32 w = 0;
        for (u32 j = 0; j < 4 && 4 * c + j < n; j++) w |= (u32)m[4 * c + j] << (8 * j);
        return w;
    });
}
// append_message with the message in registers: message byte i = byte (i & 3) of mw[i >> 2], bytes past the end zero
template <typename S, int L, int NW>
HD void t_append_words(S& t, const char (&label)[L], const u32 (&mw)[NW], u32 nbytes) {
    t_op_absorb(t, label, nbytes, 0, [&](u32 c) -> u32 {
        u32 wsel = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) wsel = (c == (u32)k) ? mw[k] : wsel;
        return wsel;
    });
}
HD void t_new(strobe& t, const uint8_t* label, u32 n) {
    const uint8_t proto[11] = {'M', 'e', 'r', 'l', 'i', 'n', ' ', 'v', '1', '.', '0'};
    strobe_init(t, proto, 11);
    t_append(t, "dom-sep", label, n);
}
template <typename S, int L>
HD void t_append_u64(S& t, const char (&label)[L], u64 x) {
    const u32 mw[2] = {(u32)x, (u32)(x >> 32)};
    t_append_words(t, label, mw, 8);
}
template <int L>
HD void t_challenge_bytes(strobe& t, const char (&label)[L], uint8_t* out, u32 n) {
    t_op_absorb(t, label, n, 1, [](u32) -> u32 { return 0; });
    strobe_squeeze(t, out, n);
}
// transcript.rs:10-14: 32 PRF bytes, big-endian, Scalar::from_repr(..).unwrap().  Returns false where the reference
// would panic (value >= n, probability ~2^-128); the caller records a DEGENERATE status for that proof.
template <typename S, int L>
HD bool t_get_challenge(S& t, const char (&label)[L], sc& out) {
    t_op_absorb(t, label, 32, 1, [](u32) -> u32 { return 0; });
    if (t.pos == 0) {
        // the usual case (the forced permutation leaves pos = 0): PRF bytes 0..31 are state words 0..3, read then zeroed;
        // byte i of the output is byte (31 - i) of the little-endian scalar
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u64 wk = st_take_word(t, k);
            out.v[7 - 2 * k] = bswap32((u32)wk);
            out.v[6 - 2 * k] = bswap32((u32)(wk >> 32));
        }
        t.pos = 32;
    } else {
        uint8_t b[32];
        strobe_squeeze(t, b, 32);
        be32_to_limbs(out.v, b);
    }
    return sc_is_canonical(out);
}

}  // namespace bppp
