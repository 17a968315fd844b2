"""Deterministic synthetic inputs for the u64 range-proof path: SHAKE256 XOF, seed b"bppp-bench-v1".
Pure hashing + integer reduction -- no curve arithmetic, no oracle.  Values are uniform u64 with forced edge cases
(0, 2^64-1, 123456 = benches/range_proof.rs:13); blindings and the 52 prover scalars per proof are wide-reduced 64-byte
XOF outputs (what k256's Scalar::generate_biased does with RNG bytes)."""
from __future__ import annotations

import hashlib
import struct

import numpy as np

SEED = b"bppp-bench-v1"
LABEL = b"u64 range proof"      # benches/range_proof.rs:32
N_ORDER = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
N_RNG_DRAWS_U64 = 52


def xof(tag: bytes, idx: int, n: int, seed: bytes = SEED) -> bytes:
    return hashlib.shake_256(seed + tag + struct.pack("<Q", idx)).digest(n)


def values(n: int, first: int = 0, seed: bytes = SEED) -> np.ndarray:
    forced = {0: 0, 1: 2**64 - 1, 2: 123456}
    out = np.zeros(n, dtype=np.uint64)
    for j in range(n):
        g = first + j
        out[j] = forced[g] if g in forced else struct.unpack("<Q", xof(b"val", g, 8, seed))[0]
    return out


def _wide_scalars(tag: bytes, first: int, n: int, per: int, seed: bytes) -> np.ndarray:
    out = np.zeros((n, per * 32), dtype=np.uint8)
    for j in range(n):
        raw = xof(tag, first + j, 64 * per, seed)
        out[j] = np.frombuffer(b"".join((int.from_bytes(raw[64 * i:64 * i + 64], "big") % N_ORDER).to_bytes(32, "big")
                                        for i in range(per)), dtype=np.uint8)
    return out


def blindings(n: int, first: int = 0, seed: bytes = SEED) -> np.ndarray:
    return _wide_scalars(b"bld", first, n, 1, seed)


def prover_randomness(n: int, first: int = 0, seed: bytes = SEED) -> np.ndarray:
    return _wide_scalars(b"rng", first, n, N_RNG_DRAWS_U64, seed)


def corrupt(proofs: np.ndarray, every: int = 1024, seed: bytes = SEED):
    """Negative set: in one proof out of `every`, flip the low bit of one byte inside l0/l1/n0 (never a top byte, so the
    scalar stays canonical) -> the proof must be rejected.  Returns (proofs', expected_accept)."""
    p = proofs.copy()
    n = p.shape[0]
    expect = np.ones(n, dtype=np.uint8)
    for j in range(0, n, every):
        off = 832 + 1 + (xof(b"neg", j, 1, seed)[0] % 95)
        if off in (832, 864, 896):
            off += 1
        p[j, off] ^= 0x01
        expect[j] = 0
    return p, expect
