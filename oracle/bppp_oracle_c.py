"""ctypes binding for the C oracle (oracle/bppp_ref.c -> oracle/libbppp_oracle.so).

TEST INFRASTRUCTURE ONLY (see the header of bppp_ref.c): used by tests/, smoke() and
bench.py's cpu_baseline leg as the checker / timed CPU baseline, never by the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbppp_oracle.so")
_lib = None

U64_PROOF_BYTES = 928
TRACE_BYTES = 704


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "bppp_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libbppp_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.bppp_oracle_u64_verify_batch.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_void_p, C.c_int]
        _lib.bppp_oracle_u64_prove_batch.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.bppp_oracle_u64_prove_trapdoor_batch.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.bppp_oracle_u64_verify.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_void_p]
        _lib.bppp_oracle_u64_prove.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_uint64, C.c_char_p, C.c_char_p, C.c_size_t,
                                               C.c_void_p, C.c_void_p]
        _lib.bppp_oracle_u64_commit_value.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_void_p]
    return _lib


def u64_commit_value(gens: bytes, x: int, s: bytes) -> bytes:
    out = C.create_string_buffer(64)
    rc = lib().bppp_oracle_u64_commit_value(gens, x, s, out)
    if rc:
        raise ValueError(f"oracle commit_value rc={rc}")
    return out.raw


def u64_verify(gens: bytes, label: bytes, V: bytes, proof: bytes, trace: bool = False):
    tb = C.create_string_buffer(TRACE_BYTES) if trace else None
    rc = lib().bppp_oracle_u64_verify(gens, label, len(label), V, proof, tb)
    return (rc, tb.raw) if trace else rc


def u64_prove(gens: bytes, label: bytes, x: int, s: bytes, rnd: bytes) -> Tuple[bytes, bytes]:
    assert len(rnd) % 32 == 0
    po = C.create_string_buffer(U64_PROOF_BYTES)
    vo = C.create_string_buffer(64)
    rc = lib().bppp_oracle_u64_prove(gens, label, len(label), x, s, rnd, len(rnd) // 32, po, vo)
    if rc:
        raise ValueError(f"oracle prove rc={rc}")
    return po.raw, vo.raw


def u64_verify_batch(gens: bytes, label: bytes, V: np.ndarray, proofs: np.ndarray, nthreads: int = 1):
    n = V.shape[0]
    V = np.ascontiguousarray(V, dtype=np.uint8)
    proofs = np.ascontiguousarray(proofs, dtype=np.uint8)
    assert V.shape == (n, 64) and proofs.shape == (n, U64_PROOF_BYTES)
    accept = np.zeros(n, dtype=np.uint8)
    status = np.zeros(n, dtype=np.int32)
    lib().bppp_oracle_u64_verify_batch(gens, label, len(label), n, V.ctypes.data, proofs.ctypes.data, accept.ctypes.data,
                                       status.ctypes.data, nthreads)
    return accept, status


def u64_prove_batch(gens: bytes, label: bytes, x: np.ndarray, s: np.ndarray, rnd: np.ndarray, nthreads: int = 1):
    n = x.shape[0]
    x = np.ascontiguousarray(x, dtype=np.uint64)
    s = np.ascontiguousarray(s, dtype=np.uint8)
    rnd = np.ascontiguousarray(rnd, dtype=np.uint8)
    assert s.shape == (n, 32) and rnd.shape == (n, 52 * 32)
    proofs = np.zeros((n, U64_PROOF_BYTES), dtype=np.uint8)
    V = np.zeros((n, 64), dtype=np.uint8)
    rc = lib().bppp_oracle_u64_prove_batch(gens, label, len(label), n, x.ctypes.data, s.ctypes.data, rnd.ctypes.data,
                                           proofs.ctypes.data, V.ctypes.data, nthreads)
    if rc:
        raise ValueError(f"oracle prove_batch rc={rc}")
    return proofs, V


def u64_prove_trapdoor_batch(gen_dlogs: bytes, label: bytes, x: np.ndarray, s: np.ndarray, rnd: np.ndarray, nthreads: int = 1):
    """Synthetic-workload prover: generators are k_i*G with KNOWN k_i (49 x 32 B), so every commitment is one
    fixed-base multiple of G.  Produces byte-identical proofs to u64_prove_batch (tests check that)."""
    n = x.shape[0]
    x = np.ascontiguousarray(x, dtype=np.uint64)
    s = np.ascontiguousarray(s, dtype=np.uint8)
    rnd = np.ascontiguousarray(rnd, dtype=np.uint8)
    assert len(gen_dlogs) == 49 * 32 and s.shape == (n, 32) and rnd.shape == (n, 52 * 32)
    proofs = np.zeros((n, U64_PROOF_BYTES), dtype=np.uint8)
    V = np.zeros((n, 64), dtype=np.uint8)
    rc = lib().bppp_oracle_u64_prove_trapdoor_batch(gen_dlogs, label, len(label), n, x.ctypes.data, s.ctypes.data,
                                                    rnd.ctypes.data, proofs.ctypes.data, V.ctypes.data, nthreads)
    if rc:
        raise ValueError(f"oracle trapdoor prove rc={rc}")
    return proofs, V


def merlin_kat(label: bytes, l1: bytes, m1: bytes, l2: bytes, n: int) -> bytes:
    out = C.create_string_buffer(n)
    lib().bppp_oracle_merlin_kat(label, C.c_size_t(len(label)), l1, m1, C.c_size_t(len(m1)), l2, out, C.c_size_t(n))
    return out.raw


def point_mul(P: Optional[bytes], k: bytes) -> bytes:
    out = C.create_string_buffer(64)
    rc = lib().bppp_oracle_point_mul(P if P is not None else bytes(64), k, out)
    if rc:
        raise ValueError(f"rc={rc}")
    return out.raw


def point_add(A: bytes, B: bytes) -> bytes:
    out = C.create_string_buffer(64)
    rc = lib().bppp_oracle_point_add(A, B, out)
    if rc:
        raise ValueError(f"rc={rc}")
    return out.raw


def scalar_inv(a: bytes) -> Tuple[bytes, bytes]:
    o1, o2 = C.create_string_buffer(32), C.create_string_buffer(32)
    rc = lib().bppp_oracle_scalar_inv(a, o1, o2)
    if rc:
        raise ZeroDivisionError(f"rc={rc}")
    return o1.raw, o2.raw
