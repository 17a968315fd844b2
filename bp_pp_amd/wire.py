"""Wire formats of a u64 range proof, host side.

Three byte-level forms of the same content:
  * ABI form (include/bppp.h): 928 B = 13 x (x||y, 64 B) + 3 x 32 B
  * SEC1 form: 525 B = 13 x 33 B compressed points + 3 x 32 B -- the bytes k256 puts into the reference's
    `SerializableProof` (reciprocal.rs:37-41, circuit.rs:37-46, wnla.rs:33-38)
  * JSON: the `serde_json` shape of `reciprocal::SerializableProof`:
        {"circuit_proof": {"c_l", "c_r", "c_o", "c_s", "r": [...], "x": [...], "l": [...], "n": [...]}, "r"}
    with every point / scalar as a hex string.  k256's serde goes through `serdect` (hex for human-readable formats);
    whether it emits upper- or lower-case hex cannot be checked here (no Rust toolchain) -- this module WRITES upper case
    and READS either.
The identity point has two byte forms in k256 0.13.3 and they must not be confused:
  * `GroupEncoding::to_bytes` (what `transcript::app_point` hashes, transcript.rs:7, and this ABI's 33-byte SEC1 form) is a
    fixed 33-byte array: the identity is 33 zero bytes, and `from_bytes` accepts exactly that;
  * serde of an `AffinePoint` goes through `to_encoded_point(true)`, whose identity is the ONE byte 0x00: the JSON string is
    "00", and a 66-character all-zero string would be refused by the Rust side.
The JSON helpers below therefore write "00" for the identity and read "00" (and, leniently, 66 zeros) back to it.
Only integer arithmetic on the curve equation is needed (square root for decompression), no group law.
"""
from __future__ import annotations

import json
from typing import Dict, List

P = 2**256 - 2**32 - 977
POINT_FIELDS = ["c_l", "c_r", "c_o", "c_s"]


def compress_point(xy64: bytes) -> bytes:
    if xy64 == bytes(64):
        return bytes(33)
    y = int.from_bytes(xy64[32:], "big")
    return bytes([2 + (y & 1)]) + xy64[:32]


def decompress_point(sec1: bytes) -> bytes:
    if sec1 == bytes(33):
        return bytes(64)
    if len(sec1) != 33 or sec1[0] not in (2, 3):
        raise ValueError("bad SEC1 compressed point")
    x = int.from_bytes(sec1[1:], "big")
    if x >= P:
        raise ValueError("x out of range")
    rhs = (x * x * x + 7) % P
    y = pow(rhs, (P + 1) // 4, P)
    if y * y % P != rhs:
        raise ValueError("not on the curve")
    if (y & 1) != (sec1[0] & 1):
        y = P - y
    return sec1[1:] + y.to_bytes(32, "big")


def point_to_hex(sec1: bytes) -> str:
    """33-byte form -> the hex string serde emits for an `AffinePoint` (identity: "00")."""
    return "00" if sec1 == bytes(33) else sec1.hex().upper()


def hex_to_point(h: str) -> bytes:
    """Inverse of point_to_hex: "00" (the SEC1 identity) -> 33 zero bytes, else the 33 compressed bytes."""
    b = bytes.fromhex(h)
    if b == b"\x00":
        return bytes(33)
    if len(b) != 33:
        raise ValueError("a serialized point is 33 bytes (66 hex characters) or \"00\" for the identity")
    return b


def abi_to_sec1(proof928: bytes) -> bytes:
    assert len(proof928) == 928
    return b"".join(compress_point(proof928[64 * i:64 * i + 64]) for i in range(13)) + proof928[832:]


def sec1_to_abi(proof525: bytes) -> bytes:
    assert len(proof525) == 525
    return b"".join(decompress_point(proof525[33 * i:33 * i + 33]) for i in range(13)) + proof525[429:]


def sec1_to_json(proof525: bytes) -> str:
    pts = [point_to_hex(proof525[33 * i:33 * i + 33]) for i in range(13)]
    sc = [proof525[429 + 32 * i:429 + 32 * i + 32].hex().upper() for i in range(3)]
    doc = {"circuit_proof": {"c_l": pts[0], "c_r": pts[1], "c_o": pts[2], "c_s": pts[3], "r": pts[4:8], "x": pts[8:12],
                             "l": sc[0:2], "n": sc[2:3]}, "r": pts[12]}
    return json.dumps(doc, indent=2)


def json_to_sec1(text: str) -> bytes:
    doc = json.loads(text)
    cp = doc["circuit_proof"]
    if len(cp["r"]) != 4 or len(cp["x"]) != 4 or len(cp["l"]) != 2 or len(cp["n"]) != 1:
        raise ValueError("not a u64 range proof shape (r, x: 4 points; l: 2 scalars; n: 1 scalar)")
    pts: List[str] = [cp[k] for k in POINT_FIELDS] + list(cp["r"]) + list(cp["x"]) + [doc["r"]]
    out = b"".join(hex_to_point(h) for h in pts) + b"".join(bytes.fromhex(h) for h in list(cp["l"]) + list(cp["n"]))
    if len(out) != 525:
        raise ValueError("bad field length")
    return out


# ---------------------------------------------------------------- generic proofs (any shape)
# The generic C-ABI proof layouts (include/bppp.h) <-> the `serde_json` shapes of the reference's serializable mirrors:
#   wnla::SerializableProof      {"r": [...], "x": [...], "l": [...], "n": [...]}                      (wnla.rs:33-38)
#   circuit::SerializableProof   {"c_l", "c_r", "c_o", "c_s", "r": [...], "x": [...], "l": [...], "n": [...]}   (circuit.rs:37-46)
#   reciprocal::SerializableProof {"circuit_proof": {...}, "r"}                                       (reciprocal.rs:37-41)
# with points as SEC1-compressed hex and scalars as 32-byte big-endian hex (same caveat on hex case as above).
def _pts(buf: bytes, off: int, count: int):
    return [point_to_hex(compress_point(buf[off + 64 * i:off + 64 * i + 64])) for i in range(count)], off + 64 * count


def _scs(buf: bytes, off: int, count: int):
    return [buf[off + 32 * i:off + 32 * i + 32].hex().upper() for i in range(count)], off + 32 * count


def wnla_proof_to_doc(proof_r: bytes, proof_x: bytes, proof_l: bytes, proof_n: bytes) -> Dict:
    """The four arrays of bppp_wnla_{prove,verify}_batch for ONE instance -> wnla::SerializableProof document."""
    r, _ = _pts(proof_r, 0, len(proof_r) // 64)
    x, _ = _pts(proof_x, 0, len(proof_x) // 64)
    l, _ = _scs(proof_l, 0, len(proof_l) // 32)
    n, _ = _scs(proof_n, 0, len(proof_n) // 32)
    return {"r": r, "x": x, "l": l, "n": n}


def circuit_proof_to_doc(proof: bytes, rounds: int, nl: int, nn: int, reciprocal: bool = False) -> Dict:
    """One proof in the layout of bppp_circuit_* (or, with reciprocal=True, bppp_reciprocal_*) -> serializable document."""
    want = 64 * ((5 if reciprocal else 4) + 2 * rounds) + 32 * (nl + nn)
    if len(proof) != want:
        raise ValueError(f"proof has {len(proof)} bytes, the shape needs {want}")
    head, off = _pts(proof, 0, 4)
    r, off = _pts(proof, off, rounds)
    x, off = _pts(proof, off, rounds)
    rr = None
    if reciprocal:
        (rr,), off = _pts(proof, off, 1)
    l, off = _scs(proof, off, nl)
    n, off = _scs(proof, off, nn)
    cp = dict(zip(POINT_FIELDS, head), r=r, x=x, l=l, n=n)
    return {"circuit_proof": cp, "r": rr} if reciprocal else cp


def doc_to_circuit_proof(doc: Dict) -> bytes:
    """Inverse of circuit_proof_to_doc (either shape); points are decompressed (ValueError on a malformed point)."""
    reciprocal = "circuit_proof" in doc
    cp = doc["circuit_proof"] if reciprocal else doc
    if len(cp["r"]) != len(cp["x"]):
        raise ValueError("r and x must have the same length")
    pt = lambda h: decompress_point(hex_to_point(h))
    sc = lambda h: bytes.fromhex(h).rjust(32, b"\0")
    out = b"".join(pt(cp[k]) for k in POINT_FIELDS) + b"".join(pt(h) for h in cp["r"]) + b"".join(pt(h) for h in cp["x"])
    if reciprocal:
        out += pt(doc["r"])
    return out + b"".join(sc(h) for h in cp["l"]) + b"".join(sc(h) for h in cp["n"])
