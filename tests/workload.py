"""Synthetic u64 range-proof workloads: SHAKE256-seeded generators, values, blindings and prover
randomness; proofs produced by the oracle's trapdoor prover (generators are k_i*G with known k_i, so the prover costs
~1 ms/proof on a host core while emitting byte-identical proofs to the honest prover -- tests/test_oracle_c.py).
Test / bench-setup infrastructure only."""
from __future__ import annotations

import hashlib
import os
import struct
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(_ROOT, "oracle") not in sys.path:
    sys.path.insert(0, os.path.join(_ROOT, "oracle"))

import bppp_oracle as O      # noqa: E402
import bppp_oracle_c as OC   # noqa: E402

SEED = O.SEED
LABEL = O.LABEL
N_ORDER = O.N


def _xof(tag: bytes, idx: int, n: int, seed: bytes) -> bytes:
    return hashlib.shake_256(seed + tag + struct.pack("<Q", idx)).digest(n)


def generator_dlogs(seed: bytes = SEED) -> bytes:
    return b"".join(O.sc_to_bytes(O.synth_generator_scalar(i, seed)) for i in range(49))


def generators(seed: bytes = SEED) -> bytes:
    """49 x 64 B: g, g_vec[16], h_vec[32]."""
    d = generator_dlogs(seed)
    return b"".join(OC.point_mul(None, d[32 * i:32 * i + 32]) for i in range(49))


def split_generators(gens: bytes):
    pts = [gens[64 * i:64 * i + 64] for i in range(49)]
    return pts[0], pts[1:17], pts[17:49]


def values(n: int, first: int = 0, seed: bytes = SEED) -> np.ndarray:
    return np.array([O.synth_value(first + j, seed) for j in range(n)], dtype=np.uint64)


def _wide_scalars(tag: bytes, first: int, n: int, per: int, seed: bytes) -> np.ndarray:
    out = np.zeros((n, per * 32), dtype=np.uint8)
    for j in range(n):
        raw = _xof(tag, first + j, 64 * per, seed)
        out[j] = np.frombuffer(b"".join((int.from_bytes(raw[64 * i:64 * i + 64], "big") % N_ORDER).to_bytes(32, "big")
                                        for i in range(per)), dtype=np.uint8)
    return out


def blindings(n: int, first: int = 0, seed: bytes = SEED) -> np.ndarray:
    return _wide_scalars(b"bld", first, n, 1, seed)


def prover_randomness(n: int, first: int = 0, seed: bytes = SEED) -> np.ndarray:
    return _wide_scalars(b"rng", first, n, 52, seed)


def make_batch(n: int, first: int = 0, seed: bytes = SEED, nthreads: int = 0):
    """-> (gens 49x64 bytes, commitments [n,64] u8, proofs [n,928] u8, values [n] u64)."""
    nthreads = nthreads or max(1, (os.cpu_count() or 1))
    x = values(n, first, seed)
    s = blindings(n, first, seed)
    rnd = prover_randomness(n, first, seed)
    proofs, V = OC.u64_prove_trapdoor_batch(generator_dlogs(seed), LABEL, x, s, rnd, nthreads=nthreads)
    return generators(seed), V, proofs, x


def corrupt(proofs: np.ndarray, commitments: np.ndarray, every: int = 1024, seed: bytes = SEED):
    """Negative set: in one proof out of `every`, flip one byte of a scalar (still canonical with overwhelming probability)
    -> expected reject.  Returns (proofs', expected_accept)."""
    p = proofs.copy()
    n = p.shape[0]
    expect = np.ones(n, dtype=np.uint8)
    for j in range(0, n, every):
        off = 832 + 1 + (_xof(b"neg", j, 1, seed)[0] % 95)     # inside l0,l1,n0 but never a top byte
        if off in (832, 864, 896):
            off += 1
        p[j, off] ^= 0x01
        expect[j] = 0
    return p, expect
