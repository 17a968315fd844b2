// secp256k1 group operations for gfx950: homogeneous projective coordinates with the COMPLETE a = 0
// formulas of Renes-Costello-Batina 2016 (algorithms 7, 8, 9; b3 = 3*7 = 21).
//
// Replaces k256 0.13.3 `ProjectivePoint::{add, sub, double, eq, to_affine, to_bytes}` as used by the
// reference (util.rs:46-85, wnla.rs:66-102, circuit.rs:182-235, transcript.rs:6-8).  Complete formulas are
// used on purpose: a verifier sees adversarial inputs (identity operands, P + P, P + (-P)) and the reference's
// library handles all of them; they are also branch-free, so the 64 lanes of a wavefront never diverge.
#pragma once
#include "field.h"

namespace bppp {

struct pt { fe X, Y, Z; };   // (X : Y : Z), identity = (0 : 1 : 0); coordinate magnitudes <= (5, 2, 2) between operations
struct apt { fe x, y; };     // affine, magnitude 1; (0, 0) is the identity sentinel (not on the curve: 0 != 7)
// HBM formats: packed canonical words (8 x u32 per coordinate) for affine points; 128-byte slots for projective table entries
struct apt_packed { u32 x[8], y[8]; };                                   // 64 B
struct __attribute__((aligned(16))) pt_slot { pt p; u32 pad[2]; };        // 30 limbs + pad = 128 B (device build)

HD void pt_set_identity(pt& r) {
    fe_set_u32(r.X, 0);
    fe_set_u32(r.Y, 1);
    fe_set_u32(r.Z, 0);
}
HD bool pt_is_identity(const pt& p) { return fe_is_zero(p.Z); }
HD bool apt_is_identity(const apt& a) { return fe_is_zero(a.x) & fe_is_zero(a.y); }
HD void pt_from_affine(pt& r, const apt& a) {
    bool id = apt_is_identity(a);
    r.X = a.x;
    r.Y = a.y;
    fe_set_u32(r.Z, id ? 0u : 1u);
    if (id) fe_set_u32(r.Y, 1);
}
HD void pt_cmov(pt& r, bool take, const pt& b) {
    fe_cmov(r.X, take, b.X);
    fe_cmov(r.Y, take, b.Y);
    fe_cmov(r.Z, take, b.Z);
}
HD void apt_unpack(apt& a, bool& is_identity, const apt_packed& k) {
    u32 z = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) z |= k.x[i] | k.y[i];
    is_identity = z == 0;
    fe_from_w8(a.x, k.x);
    fe_from_w8(a.y, k.y);
}
HD void apt_pack(apt_packed& k, const apt& a) {
    fe_to_w8(k.x, a.x);
    fe_to_w8(k.y, a.y);
}
// canonical coordinates (magnitude 1): what table entries are stored as, so look-ups can negate / scale lazily
HD void pt_normalize(pt& p) {
    fe_normalize(p.X);
    fe_normalize(p.Y);
    fe_normalize(p.Z);
}

// RCB16 algorithm 7: complete addition, 12M + 2 m(b3)
HD void pt_add(pt& r, const pt& p, const pt& q) {
    fe t0, t1, t2, t3, t4, X3, Y3, Z3;
    fe_mul(t0, p.X, q.X);
    fe_mul(t1, p.Y, q.Y);
    fe_mul(t2, p.Z, q.Z);
    fe_add(t3, p.X, p.Y);
    fe_add(t4, q.X, q.Y);
    fe_mul(t3, t3, t4);
    fe_add(t4, t0, t1);
    fe_sub(t3, t3, t4);
    fe_add(t4, p.Y, p.Z);
    fe_add(X3, q.Y, q.Z);
    fe_mul(t4, t4, X3);
    fe_add(X3, t1, t2);
    fe_sub(t4, t4, X3);
    fe_add(X3, p.X, p.Z);
    fe_add(Y3, q.X, q.Z);
    fe_mul(X3, X3, Y3);
    fe_add(Y3, t0, t2);
    fe_sub(Y3, X3, Y3);
    fe_add(X3, t0, t0);
    fe_add(t0, X3, t0);
    fe_mul_small(t2, t2, 21);
    fe_add(Z3, t1, t2);
    fe_sub(t1, t1, t2);
    fe_mul_small(Y3, Y3, 21);
    fe_mul(X3, t4, Y3);
    fe_mul(t2, t3, t1);
    fe_sub(X3, t2, X3);
    fe_mul(Y3, Y3, t0);
    fe_mul(t1, t1, Z3);
    fe_add(Y3, t1, Y3);
    fe_mul(t0, t0, t3);
    fe_mul(Z3, Z3, t4);
    fe_add(Z3, Z3, t0);
    r.X = X3; r.Y = Y3; r.Z = Z3;
}
// RCB16 algorithm 8: complete mixed addition (q affine, q != identity), 11M + 2 m(b3)
HD void pt_madd_nonid(pt& r, const pt& p, const apt& q) {
    fe t0, t1, t2, t3, t4, X3, Y3, Z3;
    fe_mul(t0, p.X, q.x);
    fe_mul(t1, p.Y, q.y);
    fe_add(t3, q.x, q.y);
    fe_add(t4, p.X, p.Y);
    fe_mul(t3, t3, t4);
    fe_add(t4, t0, t1);
    fe_sub(t3, t3, t4);
    fe_mul(t4, q.y, p.Z);
    fe_add(t4, t4, p.Y);
    fe_mul(Y3, q.x, p.Z);
    fe_add(Y3, Y3, p.X);
    fe_add(X3, t0, t0);
    fe_add(t0, X3, t0);
    fe_mul_small(t2, p.Z, 21);
    fe_add(Z3, t1, t2);
    fe_sub(t1, t1, t2);
    fe_mul_small(Y3, Y3, 21);
    fe_mul(X3, t4, Y3);
    fe_mul(t2, t3, t1);
    fe_sub(X3, t2, X3);
    fe_mul(Y3, Y3, t0);
    fe_mul(t1, t1, Z3);
    fe_add(Y3, t1, Y3);
    fe_mul(t0, t0, t3);
    fe_mul(Z3, Z3, t4);
    fe_add(Z3, Z3, t0);
    r.X = X3; r.Y = Y3; r.Z = Z3;
}
// mixed addition with a skip flag (result = p when skipped); the caller passes skip = true for the identity sentinel
HD void pt_madd(pt& r, const pt& p, const apt& q, bool skip) {
    pt s;
    pt_madd_nonid(s, p, q);
    r = p;
    pt_cmov(r, !skip, s);
}
// RCB16 algorithm 9: complete doubling, 6M + 2S + 1 m(b3)
HD void pt_dbl(pt& r, const pt& p) {
    fe t0, t1, t2, X3, Y3, Z3;
    fe_sqr(t0, p.Y);
    fe_add(Z3, t0, t0);
    fe_add(Z3, Z3, Z3);
    fe_add(Z3, Z3, Z3);
    fe_mul(t1, p.Y, p.Z);
    fe_sqr(t2, p.Z);
    fe_mul_small(t2, t2, 21);
    fe_mul(X3, t2, Z3);
    fe_add(Y3, t0, t2);
    fe_mul(Z3, t1, Z3);
    fe_add(t1, t2, t2);
    fe_add(t2, t1, t2);
    fe_sub(t0, t0, t2);
    fe_mul(Y3, t0, Y3);
    fe_add(Y3, X3, Y3);
    fe_mul(t1, p.X, p.Y);
    fe_mul(X3, t0, t1);
    fe_add(X3, X3, X3);
    r.X = X3; r.Y = Y3; r.Z = Z3;
}
// ---- XYZZ accumulator for sums of MANY affine table points (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): mixed addition 8M + 2S.
// The formulas are INCOMPLETE (acc = +-q is not handled), so they are used with deferred detection: an exceptional addition
// has P = x2 ZZ1 - X1 = 0, which makes ZZ3 = ZZ1 P^2 = 0, and ZZ then stays 0 through every later addition.  A lane therefore
// tests ZZ once, after its whole sum; if it is 0 although points were added, the proof is re-done with the complete formulas
// (fb_core.h: fb_lane_finish_fast reports it; k_verify_fixed.hip: the k_verify_final_check_flagged* kernels redo the flagged proofs).  For independent generators this never happens; it does
// for degenerate generator sets (repeated or related generators), which stay correct through the fallback.
struct ptz { fe X, Y, ZZ, ZZZ; };
// true if the flag is set on ANY active lane of the wavefront (a wave-uniform value: branching on it never diverges)
HD bool wave_any(bool x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(x) != 0;
#else
    return x;
#endif
}
HD void ptz_madd(ptz& a, bool& empty, const apt& q, bool skip) {
    fe U2, S2, P, R, PP, PPP, Q, X3, Y3, ZZ3, ZZZ3, t;
    fe_mul(U2, q.x, a.ZZ);
    fe_mul(S2, q.y, a.ZZZ);
    fe_sub_m<6>(P, U2, a.X);
    fe_sub_m<3>(R, S2, a.Y);
    fe_sqr(PP, P);
    fe_mul(PPP, P, PP);
    fe_mul(Q, a.X, PP);
    fe_sqr(X3, R);
    fe_sub_m<1>(X3, X3, PPP);
    fe_add(t, Q, Q);
    fe_sub_m<2>(X3, X3, t);            // magnitude 6
    fe_sub_m<6>(t, Q, X3);             // magnitude 8
    fe nY;
    fe_neg_m<3>(nY, a.Y);              // magnitude 4
    fe_mul2_add(Y3, R, t, nY, PPP);    // Y3 = R (Q - X3) - Y1 PPP, one reduction (5 * 8 + 4 * 1 <= 64); magnitude 1
    fe_mul(ZZ3, a.ZZ, PP);
    fe_mul(ZZZ3, a.ZZZ, PPP);
    // first real point: the sum IS q
    fe one;
    fe_set_u32(one, 1);
    fe_cmov(X3, empty, q.x); fe_cmov(Y3, empty, q.y); fe_cmov(ZZ3, empty, one); fe_cmov(ZZZ3, empty, one);
    // a skipped addition (zero digit, identity entry) keeps the old sum.  With table windows of 16-22 bits that is one step in 10^5: the
    // four selects sit behind a wave-uniform branch instead of in every addition's instruction stream
    if (wave_any(skip)) { fe_cmov(X3, skip, a.X); fe_cmov(Y3, skip, a.Y); fe_cmov(ZZ3, skip, a.ZZ); fe_cmov(ZZZ3, skip, a.ZZZ); }
    a.X = X3; a.Y = Y3; a.ZZ = ZZ3; a.ZZZ = ZZZ3;
    empty = empty & skip;
}
HD void ptz_init(ptz& a) {
    fe_set_u32(a.X, 0); fe_set_u32(a.Y, 0); fe_set_u32(a.ZZ, 1); fe_set_u32(a.ZZZ, 1);
}
// -> homogeneous projective (X ZZZ : Y ZZ : ZZ ZZZ); `empty` -> identity
HD void ptz_to_pt(pt& r, const ptz& a, bool empty) {
    fe_mul(r.X, a.X, a.ZZZ);
    fe_mul(r.Y, a.Y, a.ZZ);
    fe_mul(r.Z, a.ZZ, a.ZZZ);
    pt id;
    pt_set_identity(id);
    pt_cmov(r, empty, id);
}
// ---- Jacobian accumulator (x = X/Z^2, y = Y/Z^3) for the shared-doubling (Straus) loops over AFFINE per-proof tables:
// doubling 2M + 5S, mixed addition 8M + 3S (the complete projective law costs 6M + 2S + m and 12M).  Incomplete in the same
// way as the XYZZ law above and used with the same deferred detection: an exceptional addition has H = 0, so Z3 = Z1 H = 0,
// and Z stays 0 through every later doubling (Z3 = 2 Y1 Z1) and addition.  Coordinate magnitudes stay <= (6, 3, 2).
struct ptj { fe X, Y, Z; };
HD void ptj_init(ptj& a) { fe_set_u32(a.X, 0); fe_set_u32(a.Y, 0); fe_set_u32(a.Z, 0); }
HD void ptj_dbl(ptj& a) {   // dbl-2009-l (a = 0)
    fe A, B, C, D, E, F, t;
    fe_sqr(A, a.X);
    fe_sqr(B, a.Y);
    fe_sqr(C, B);
    fe_add(t, a.X, B);                 // <= 7
    fe_sqr(t, t);
    fe_sub_m<1>(t, t, A);              // 3
    fe_sub_m<1>(t, t, C);              // 5
    fe_mul_small(D, t, 2);             // 1
    fe_add(E, A, A);
    fe_add(E, E, A);                   // 3
    fe_sqr(F, E);
    fe_mul(t, a.Y, a.Z);
    fe_add(a.Z, t, t);                 // Z3 = 2 Y1 Z1, magnitude 2
    fe_add(t, D, D);                   // 2
    fe_sub_m<2>(a.X, F, t);            // X3 = F - 2D, magnitude 4
    fe_sub_m<4>(t, D, a.X);            // 6
    fe_mul(t, E, t);
    fe_mul_small(C, C, 8);
    fe_sub_m<1>(a.Y, t, C);            // Y3 = E (D - X3) - 8C, magnitude 3
}
HD void ptj_madd(ptj& a, bool& empty, const apt& q, bool skip) {
    fe Z2, U2, S2, H, R, HH, HHH, V, X3, Y3, Z3, t;
    fe_sqr(Z2, a.Z);
    fe_mul(U2, q.x, Z2);
    fe_mul(t, a.Z, Z2);
    fe_mul(S2, q.y, t);
    fe_sub_m<6>(H, U2, a.X);           // 8
    fe_sub_m<3>(R, S2, a.Y);           // 5
    fe_sqr(HH, H);
    fe_mul(HHH, H, HH);
    fe_mul(V, a.X, HH);
    fe_sqr(X3, R);
    fe_sub_m<1>(X3, X3, HHH);          // 3
    fe_add(t, V, V);
    fe_sub_m<2>(X3, X3, t);            // 6
    fe_sub_m<6>(t, V, X3);             // 8
    fe nY;
    fe_neg_m<3>(nY, a.Y);              // 4
    fe_mul2_add(Y3, R, t, nY, HHH);    // Y3 = R (V - X3) - Y1 HHH, one reduction; magnitude 1
    fe_mul(Z3, a.Z, H);
    fe one;
    fe_set_u32(one, 1);
    fe_cmov(X3, empty, q.x); fe_cmov(Y3, empty, q.y); fe_cmov(Z3, empty, one);
    fe_cmov(a.X, !skip, X3); fe_cmov(a.Y, !skip, Y3); fe_cmov(a.Z, !skip, Z3);
    empty = empty & skip;
}
// -> homogeneous projective (X Z : Y : Z^3); `empty` -> identity
HD void ptj_to_pt(pt& r, const ptj& a, bool empty) {
    fe z2;
    fe_sqr(z2, a.Z);
    fe_mul(r.X, a.X, a.Z);
    fe_mul(r.Z, z2, a.Z);
    fe_mul_small(r.Y, a.Y, 1);         // magnitude 3 -> 1 (what the projective law expects of its inputs)
    pt id;
    pt_set_identity(id);
    pt_cmov(r, empty, id);
}
// projective-class equality (k256 `ProjectivePoint::eq`, used at wnla.rs:81)
HD bool pt_eq(const pt& a, const pt& b) {
    fe l, r;
    fe_mul(l, a.X, b.Z);
    fe_mul(r, b.X, a.Z);
    bool ok = fe_eq(l, r);
    fe_mul(l, a.Y, b.Z);
    fe_mul(r, b.Y, a.Z);
    return ok & fe_eq(l, r);
}
// k256 `to_affine`: one field inversion; identity -> (0, 0)
HD void pt_to_affine(apt& r, const pt& p) {
    fe zi;
    fe_inv(zi, p.Z);  // 0 -> 0
    fe_mul(r.x, p.X, zi);
    fe_mul(r.y, p.Y, zi);
}
// y^2 == x^3 + 7 (identity sentinel accepted)
HD bool apt_on_curve(const apt& a) {
    fe y2, x3, seven;
    fe_sqr(y2, a.y);
    fe_sqr(x3, a.x);
    fe_mul(x3, x3, a.x);
    fe_set_u32(seven, 7);
    fe_add(x3, x3, seven);
    return fe_eq(y2, x3) | apt_is_identity(a);
}
// k256 GroupEncoding::to_bytes: 33-byte SEC1 compressed; identity -> 33 zero bytes (transcript.rs:7)
HD void apt_to_sec1(uint8_t out[33], const apt& a) {
    bool id = apt_is_identity(a);
    out[0] = id ? 0 : (uint8_t)(2 + (fe_is_odd(a.y) ? 1 : 0));
    fe_to_be(out + 1, a.x);
}
// C-ABI point: 64 B affine big-endian x||y, identity = 64 zero bytes.  false if a coordinate is >= p or the
// point is off the curve (k256 would never have produced such an AffinePoint).
HD bool apt_from_xy64(apt& r, const uint8_t* b) {
    bool ok = fe_from_be(r.x, b);
    ok &= fe_from_be(r.y, b + 32);
    ok &= apt_on_curve(r);
    return ok;
}
HD void apt_to_xy64(uint8_t* b, const apt& a) {
    fe_to_be(b, a.x);
    fe_to_be(b + 32, a.y);
}

}  // namespace bppp
