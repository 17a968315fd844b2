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

HD u64 rotl64(u64 v, int r) { return (v << r) | (v >> (64 - r)); }

HD void keccak_f1600(u64 a[25]) {
    const u64 RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL,
        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL,
        0x0000000080008009ULL, 0x000000008000000AULL, 0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL,
        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
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
        a00 ^= RC[rnd];
    }
    a[0] = a00; a[1] = a01; a[2] = a02; a[3] = a03; a[4] = a04; a[5] = a05; a[6] = a06; a[7] = a07; a[8] = a08; a[9] = a09;
    a[10] = a10; a[11] = a11; a[12] = a12; a[13] = a13; a[14] = a14; a[15] = a15; a[16] = a16; a[17] = a17; a[18] = a18; a[19] = a19;
    a[20] = a20; a[21] = a21; a[22] = a22; a[23] = a23; a[24] = a24;
}

HD void st_xor_byte(strobe& s, u32 pos, uint8_t b) { s.st[pos >> 3] ^= (u64)b << (8 * (pos & 7)); }
HD uint8_t st_take_byte(strobe& s, u32 pos) {  // read and zero (squeeze)
    u32 sh = 8 * (pos & 7);
    uint8_t b = (uint8_t)(s.st[pos >> 3] >> sh);
    s.st[pos >> 3] &= ~((u64)0xFF << sh);
    return b;
}
HD void strobe_run_f(strobe& s) {
    st_xor_byte(s, s.pos, (uint8_t)s.pos_begin);
    st_xor_byte(s, s.pos + 1, 0x04);
    st_xor_byte(s, BPPP_STROBE_R + 1, 0x80);
    keccak_f1600(s.st);
    s.pos = 0;
    s.pos_begin = 0;
}
HD void strobe_absorb(strobe& s, const uint8_t* d, u32 n) {
#pragma nounroll
    for (u32 i = 0; i < n; i++) {
        st_xor_byte(s, s.pos, d[i]);
        s.pos++;
        if (s.pos == BPPP_STROBE_R) strobe_run_f(s);
    }
}
HD void strobe_squeeze(strobe& s, uint8_t* d, u32 n) {
#pragma nounroll
    for (u32 i = 0; i < n; i++) {
        d[i] = st_take_byte(s, s.pos);
        s.pos++;
        if (s.pos == BPPP_STROBE_R) strobe_run_f(s);
    }
}
HD void strobe_begin_op(strobe& s, uint8_t flags, bool more) {
    if (more) return;
    uint8_t hdr[2] = {(uint8_t)s.pos_begin, flags};
    s.pos_begin = s.pos + 1;
    strobe_absorb(s, hdr, 2);
    if ((flags & (4 | 32)) && s.pos != 0) strobe_run_f(s);
}
HD void strobe_meta_ad(strobe& s, const uint8_t* d, u32 n, bool more) { strobe_begin_op(s, 16 | 2, more); strobe_absorb(s, d, n); }
HD void strobe_ad(strobe& s, const uint8_t* d, u32 n, bool more) { strobe_begin_op(s, 2, more); strobe_absorb(s, d, n); }
HD void strobe_prf(strobe& s, uint8_t* d, u32 n) { strobe_begin_op(s, 1 | 2 | 4, false); strobe_squeeze(s, d, n); }

HD void strobe_init(strobe& s, const uint8_t* proto, u32 n) {
    for (int i = 0; i < 25; i++) s.st[i] = 0;
    const uint8_t hdr[18] = {1, BPPP_STROBE_R + 2, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    for (u32 i = 0; i < 18; i++) st_xor_byte(s, i, hdr[i]);
    keccak_f1600(s.st);
    s.pos = 0;
    s.pos_begin = 0;
    strobe_meta_ad(s, proto, n, false);
}

// ---- merlin::Transcript
template <int L>
HD void t_append(strobe& t, const char (&label)[L], const uint8_t* m, u32 n) {
    uint8_t le[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    strobe_meta_ad(t, (const uint8_t*)label, L - 1, false);
    strobe_meta_ad(t, le, 4, true);
    strobe_ad(t, m, n, false);
}
HD void t_new(strobe& t, const uint8_t* label, u32 n) {
    const uint8_t proto[11] = {'M', 'e', 'r', 'l', 'i', 'n', ' ', 'v', '1', '.', '0'};
    strobe_init(t, proto, 11);
    t_append(t, "dom-sep", label, n);
}
template <int L>
HD void t_append_u64(strobe& t, const char (&label)[L], u64 x) {
    uint8_t le[8];
#pragma unroll
    for (int i = 0; i < 8; i++) le[i] = (uint8_t)(x >> (8 * i));
    t_append(t, label, le, 8);
}
template <int L>
HD void t_challenge_bytes(strobe& t, const char (&label)[L], uint8_t* out, u32 n) {
    uint8_t le[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    strobe_meta_ad(t, (const uint8_t*)label, L - 1, false);
    strobe_meta_ad(t, le, 4, true);
    strobe_prf(t, out, n);
}
// transcript.rs:10-14: 32 PRF bytes, big-endian, Scalar::from_repr(..).unwrap().  Returns false where the reference
// would panic (value >= n, probability ~2^-128); the caller records a DEGENERATE status for that proof.
template <int L>
HD bool t_get_challenge(strobe& t, const char (&label)[L], sc& out) {
    uint8_t b[32];
    t_challenge_bytes(t, label, b, 32);
    return sc_from_be(out, b);
}

}  // namespace bppp
