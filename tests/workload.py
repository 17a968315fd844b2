"""Synthetic u64 range-proof workloads for the TESTS: inputs from bp_pp_amd/synth.py, generators and proofs from the ORACLE
(trapdoor prover: generators are k_i*G with known k_i, so a proof costs ~1 ms on a host core while being byte-identical to
the honest prover's -- tests/test_oracle_c.py), i.e. independent of the product under test."""
from __future__ import annotations

import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_ROOT, os.path.join(_ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import bppp_oracle as O      # noqa: E402
import bppp_oracle_c as OC   # noqa: E402
from bp_pp_amd import synth  # noqa: E402

SEED, LABEL, N_ORDER = synth.SEED, synth.LABEL, synth.N_ORDER
values, blindings, prover_randomness = synth.values, synth.blindings, synth.prover_randomness

assert [O.synth_value(j) for j in range(6)] == [int(v) for v in synth.values(6)]     # the two generators of inputs agree


def generator_dlogs(seed: bytes = SEED) -> bytes:
    return b"".join(O.sc_to_bytes(O.synth_generator_scalar(i, seed)) for i in range(49))


def generators(seed: bytes = SEED) -> bytes:
    """49 x 64 B: g, g_vec[16], h_vec[32]."""
    d = generator_dlogs(seed)
    return b"".join(OC.point_mul(None, d[32 * i:32 * i + 32]) for i in range(49))


def split_generators(gens: bytes):
    pts = [gens[64 * i:64 * i + 64] for i in range(49)]
    return pts[0], pts[1:17], pts[17:49]


def make_batch(n: int, first: int = 0, seed: bytes = SEED, nthreads: int = 0):
    """-> (gens 49x64 bytes, commitments [n,64] u8, proofs [n,928] u8, values [n] u64), proofs by the oracle."""
    nthreads = nthreads or max(1, (os.cpu_count() or 1))
    x = values(n, first, seed)
    s = blindings(n, first, seed)
    rnd = prover_randomness(n, first, seed)
    proofs, V = OC.u64_prove_trapdoor_batch(generator_dlogs(seed), LABEL, x, s, rnd, nthreads=nthreads)
    return generators(seed), V, proofs, x


def corrupt(proofs: np.ndarray, commitments: np.ndarray = None, every: int = 1024, seed: bytes = SEED):
    return synth.corrupt(proofs, every, seed)


def edge_prover_inputs():
    """Prover inputs at the edges of their domains: x in {0, 2^64 - 1, ...}, blinding 0 / 1 / n - 1, and draws that are all zero, all
    n - 1, all one, alternating zero, or the draw index (`Scalar::generate_biased` can return any of them): unblinded and barely blinded
    proofs.  -> (x [5] u64, s [5, 32], rnd [5, 52 * 32])."""
    sc = lambda v: np.frombuffer((v % N_ORDER).to_bytes(32, "big"), np.uint8)
    cases = [(0, 0, lambda i: 0), (2**64 - 1, N_ORDER - 1, lambda i: N_ORDER - 1), (123456, 1, lambda i: 1),
             (0x0123456789ABCDEF, 5, lambda i: 0 if i % 2 else 7), (0xFFFFFFFF00000000, 0, lambda i: i)]
    x = np.array([c[0] for c in cases], dtype=np.uint64)
    s = np.stack([sc(c[1]) for c in cases]).copy()
    rnd = np.stack([np.concatenate([sc(c[2](i)) for i in range(52)]) for c in cases]).copy()
    return x, s, rnd
