//! Byte conventions of the C ABI (include/bppp.h) for k256 0.13.3 values.
//!   point   64 B  affine big-endian x || y, identity = 64 zero bytes
//!   scalar  32 B  big-endian canonical (`Scalar::to_bytes` / `from_repr`)
//!   u64 proof 928 B = c_l, c_r, c_o, c_s, r[0..4], x[0..4], proof.r (13 points), l[0], l[1], n[0] (3 scalars)
use bp_pp::circuit;
use bp_pp::range_proof::reciprocal;
use k256::elliptic_curve::sec1::{FromEncodedPoint, ToEncodedPoint};
use k256::elliptic_curve::PrimeField;
use k256::{AffinePoint, EncodedPoint, FieldBytes, ProjectivePoint, Scalar};

pub const POINT_BYTES: usize = 64;
pub const U64_PROOF_BYTES: usize = 928;

pub fn put_point(dst: &mut Vec<u8>, p: &ProjectivePoint) {
    let e = p.to_affine().to_encoded_point(false); // 0x04 || x || y, or the 1-byte identity
    if e.is_identity() {
        dst.extend_from_slice(&[0u8; 64]);
    } else {
        dst.extend_from_slice(&e.as_bytes()[1..65]);
    }
}

pub fn put_scalar(dst: &mut Vec<u8>, s: &Scalar) {
    dst.extend_from_slice(s.to_bytes().as_slice());
}

/// 64 zero bytes -> IDENTITY; anything else must be a curve point (the library only emits valid ones).
pub fn get_point(b: &[u8]) -> Option<ProjectivePoint> {
    assert_eq!(b.len(), 64);
    if b.iter().all(|v| *v == 0) {
        return Some(ProjectivePoint::IDENTITY);
    }
    let mut tagged = [0u8; 65];
    tagged[0] = 4;
    tagged[1..].copy_from_slice(b);
    let e = EncodedPoint::from_bytes(tagged).ok()?;
    Option::<AffinePoint>::from(AffinePoint::from_encoded_point(&e)).map(ProjectivePoint::from)
}

pub fn get_scalar(b: &[u8]) -> Option<Scalar> {
    Option::<Scalar>::from(Scalar::from_repr(*FieldBytes::from_slice(b)))
}

/// `reciprocal::Proof` of the u64 shape -> 928 bytes; None for any other shape (r / x not 4 points, l not 2, n not 1 scalars),
/// which the 928-byte form cannot carry: such a proof goes to the generic entry point or to the crate's CPU verifier.
pub fn put_u64_proof(dst: &mut Vec<u8>, pr: &reciprocal::Proof) -> Option<()> {
    let c = &pr.circuit_proof;
    if c.r.len() != 4 || c.x.len() != 4 || c.l.len() != 2 || c.n.len() != 1 {
        return None;
    }
    for q in [&c.c_l, &c.c_r, &c.c_o, &c.c_s] {
        put_point(dst, q);
    }
    c.r.iter().for_each(|q| put_point(dst, q));
    c.x.iter().for_each(|q| put_point(dst, q));
    put_point(dst, &pr.r);
    c.l.iter().for_each(|s| put_scalar(dst, s));
    put_scalar(dst, &c.n[0]);
    Some(())
}

pub fn get_u64_proof(b: &[u8]) -> Option<reciprocal::Proof> {
    assert_eq!(b.len(), U64_PROOF_BYTES);
    let pt = |i: usize| get_point(&b[64 * i..64 * i + 64]);
    let sc = |i: usize| get_scalar(&b[832 + 32 * i..864 + 32 * i]);
    Some(reciprocal::Proof {
        circuit_proof: circuit::Proof {
            c_l: pt(0)?,
            c_r: pt(1)?,
            c_o: pt(2)?,
            c_s: pt(3)?,
            r: (4..8).map(pt).collect::<Option<Vec<_>>>()?,
            x: (8..12).map(pt).collect::<Option<Vec<_>>>()?,
            l: vec![sc(0)?, sc(1)?],
            n: vec![sc(2)?],
        },
        r: pt(12)?,
    })
}
