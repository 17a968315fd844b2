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


def corrupt_offset(j: int, seed: bytes = SEED) -> int:
    """Byte of proof j whose low bit the negative set flips: inside l0 / l1 / n0, never one of their top bytes."""
    off = 832 + 1 + (xof(b"neg", j, 1, seed)[0] % 95)
    if off in (832, 864, 896):
        off += 1
    return off


def corrupt(proofs: np.ndarray, every: int = 1024, seed: bytes = SEED):
    """Negative set: in one proof out of `every`, flip the low bit of one byte inside l0/l1/n0 (never a top byte, so the
    scalar stays canonical) -> the proof must be rejected.  Returns (proofs', expected_accept)."""
    p = proofs.copy()
    n = p.shape[0]
    expect = np.ones(n, dtype=np.uint8)
    for j in range(0, n, every):
        p[j, corrupt_offset(j, seed)] ^= 0x01
        expect[j] = 0
    return p, expect


# ---------------------------------------------------------------- bulk inputs for the full-size batches (2^17 .. 2^20 proofs)
# The per-proof XOF calls above cost ~25 us each in Python; a 2^20-proof batch needs 5.6e7 scalars (1.7 GB).  The bulk
# generators draw every chunk of BULK_CHUNK proofs from a counter-based generator (numpy's Philox: raw 64-bit output, stable
# across numpy versions) keyed by SHAKE256(seed, tag, chunk index) -- so any rank can produce exactly its own shard of a fixed
# global batch -- and make scalars canonical by clearing the top four bits (uniform below 2^252 < n) instead of a wide
# reduction: the prover takes any canonical scalars, and the bench only needs reproducible ones.  Forced edge-case values
# are kept.
BULK_CHUNK = 1 << 14


def _bulk_bytes(tag: bytes, first: int, n: int, per_proof_bytes: int, seed: bytes) -> np.ndarray:
    assert per_proof_bytes % 8 == 0
    out = np.empty((n, per_proof_bytes), dtype=np.uint8)
    j = 0
    while j < n:
        g = first + j
        chunk, off = divmod(g, BULK_CHUNK)
        take = min(n - j, BULK_CHUNK - off)
        key = int.from_bytes(hashlib.shake_256(seed + b"bulk" + tag + struct.pack("<Q", chunk)).digest(16), "little")
        raw = np.random.Philox(key=key).random_raw(BULK_CHUNK * per_proof_bytes // 8).astype("<u8", copy=False)
        out[j:j + take] = raw.view(np.uint8).reshape(BULK_CHUNK, per_proof_bytes)[off:off + take]
        j += take
    return out


def bulk_values(n: int, first: int = 0, seed: bytes = SEED) -> np.ndarray:
    v = _bulk_bytes(b"val", first, n, 8, seed).view("<u8").reshape(n).copy()
    for g, forced in ((0, 0), (1, 2**64 - 1), (2, 123456)):
        if first <= g < first + n:
            v[g - first] = forced
    return v


def _bulk_scalars(tag: bytes, first: int, n: int, per: int, seed: bytes) -> np.ndarray:
    raw = _bulk_bytes(tag, first, n, 32 * per, seed)
    raw.reshape(n, per, 32)[:, :, 0] &= 0x0F          # big-endian top byte: value < 2^252 < n, hence canonical
    return raw


def bulk_blindings(n: int, first: int = 0, seed: bytes = SEED) -> np.ndarray:
    return _bulk_scalars(b"bld", first, n, 1, seed)


def bulk_prover_randomness(n: int, first: int = 0, seed: bytes = SEED) -> np.ndarray:
    return _bulk_scalars(b"rng", first, n, N_RNG_DRAWS_U64, seed)


def bulk_reciprocal_inputs(dim_nd: int, n: int, seed: int = 20260):
    """Seeded inputs of n ReciprocalRangeProofProtocol instances with dim_np = 16 (hex digits), drawn with numpy (Philox): digits
    [n, dim_nd, 32], multiplicities m [n, 16, 32], the committed value x = sum d_i 16^i mod n [n, 32], blinding s [n, 32] and the
    20 + 2 dim_nd prover scalars rnd [n, 20 + 2 dim_nd, 32] (top nibble cleared: canonical without a wide reduction).  Instance 0 is all
    zeros, instance 1 all fifteens.  No curve arithmetic, no oracle."""
    n_rnd = 20 + 2 * dim_nd
    rng = np.random.Generator(np.random.Philox(key=seed))
    dig = rng.integers(0, 16, size=(n, dim_nd), dtype=np.uint8)
    dig[0] = 0
    if n > 1:
        dig[1] = 15
    digits = np.zeros((n, dim_nd, 32), np.uint8)
    digits[:, :, 31] = dig
    m = np.zeros((n, 16, 32), np.uint8)
    counts = np.stack([(dig == v).sum(axis=1) for v in range(16)], axis=1).astype(np.uint32)      # multiplicities, < 2^16
    m[:, :, 31] = counts & 0xFF
    m[:, :, 30] = counts >> 8
    x = np.zeros((n, 32), np.uint8)
    for b in range(n):      # the digit string read as one hexadecimal number
        v = int("".join("%x" % d for d in dig[b][::-1]), 16) % N_ORDER
        x[b] = np.frombuffer(v.to_bytes(32, "big"), np.uint8)
    raw = rng.integers(0, 256, size=(n, 1 + n_rnd, 32), dtype=np.uint8)
    raw[:, :, 0] &= 0x0F
    return dict(x=x, s=raw[:, 0, :].copy(), digits=digits, m=m, rnd=raw[:, 1:, :].copy())
