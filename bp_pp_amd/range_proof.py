"""Host-side mirror of the reference's `range_proof::u64_proof::U64RangeProofProtocol` (u64_proof.rs:19-82) over the
C ABI of include/bppp.h, batch-first: one call verifies n independent proofs on one MI355X.

Differences from the Rust signatures, forced by the C boundary (see INTEGRATION.md for the Rust facade that hides them):
  * points are 64-byte affine big-endian x||y (identity = 64 zero bytes), scalars 32-byte big-endian;
  * `t: &mut Transcript` becomes `label: bytes` -- every reference call site creates `Transcript::new(label)`
    immediately before verify/prove (tests.rs:34,40; benches/range_proof.rs:32,47);
  * verify returns the accept bits AND a per-proof status where the reference would have panicked.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _capi

G_VEC_FULL_SZ = 16      # u64_proof.rs:12
H_VEC_CIRCUIT_SZ = 26   # u64_proof.rs:13
H_VEC_FULL_SZ = 32      # u64_proof.rs:14
U64_PROOF_BYTES = _capi.U64_PROOF_BYTES


def _as_u8(a, shape) -> np.ndarray:
    arr = np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a, dtype=np.uint8)
    return arr.reshape(shape)


def derive_generators(seed: bytes, n: int = 49, first_index: int = 0) -> bytes:
    """n x 64 bytes of nothing-up-my-sleeve generators (include/bppp.h: bppp_derive_generators; host code, no GPU needed):
    g, g_vec[16], h_vec[32] for the u64 protocol when n = 49."""
    out = C.create_string_buffer(64 * n)
    _capi.check(_capi.lib().bppp_derive_generators(seed, len(seed), first_index, n, out))
    return out.raw


def ctx_timings(ctx, reset: bool = True) -> dict:
    """Per-kernel HIP-event times accumulated by a context since the last reset (bppp_ctx_enable_timing / _get_timings)."""
    cap = 48
    names = (C.c_char_p * cap)()
    ms = (C.c_double * cap)()
    cnt = (C.c_int64 * cap)()
    k = _capi.check(_capi.lib().bppp_ctx_get_timings(ctx, cap, names, ms, cnt, 1 if reset else 0))
    return {names[i].decode(): {"total_ms": ms[i], "launches": cnt[i]} for i in range(k)}


def describe_plan(code: int, prove: bool = False) -> str:
    """Text form of a plan code (bppp_plan_describe)."""
    buf = C.create_string_buffer(256)
    _capi.check(_capi.lib().bppp_plan_describe(int(code), 1 if prove else 0, buf, len(buf)))
    return buf.value.decode()


def plan_for(n: int, prove: bool = False, n_simds: int = 1024, rlc_or_ct: bool = False, timing: bool = False) -> str:
    """The launch sequence a call of n proofs takes on a device of n_simds SIMDs (bppp_u64_plan: a pure function, no GPU needed)."""
    code = _capi.check(_capi.lib().bppp_u64_plan(1 if prove else 0, int(n), int(n_simds), (1 if rlc_or_ct else 0) | (2 if timing else 0)))
    return describe_plan(code, prove)


class U64RangeProofProtocol:
    """Public parameters g, g_vec[16], h_vec[32] (u64_proof.rs:19-28) resident on one GPU."""

    DIM_ND = 16
    DIM_NP = 16

    def __init__(self, g: bytes, g_vec: Sequence[bytes], h_vec: Sequence[bytes], device: int = 0, fb_window_bits: int = 0,
                 fb_table_budget_bytes: int = 0):
        """fb_window_bits = 0: the library sizes the fixed-base tables to the HBM that is free (include/bppp.h); fb_table_budget_bytes > 0
        bounds what they may take (bppp_wnla_ctx_create_budget)."""
        if len(g_vec) != G_VEC_FULL_SZ or len(h_vec) != H_VEC_FULL_SZ:
            raise ValueError("g_vec must hold 16 points and h_vec 32 points")
        self.g, self.g_vec, self.h_vec = bytes(g), [bytes(p) for p in g_vec], [bytes(p) for p in h_vec]
        self.device = device
        self._ctx = C.c_void_p()
        if fb_table_budget_bytes:
            _capi.check(_capi.lib().bppp_wnla_ctx_create_budget(C.byref(self._ctx), self.g, b"".join(self.g_vec), G_VEC_FULL_SZ, b"".join(self.h_vec),
                                                                H_VEC_FULL_SZ, device, fb_window_bits, int(fb_table_budget_bytes)))
        else:
            _capi.check(_capi.lib().bppp_ctx_create(C.byref(self._ctx), self.g, b"".join(self.g_vec), b"".join(self.h_vec),
                                                    device, fb_window_bits))

    @classmethod
    def _wrap(cls, ctx, g=b"", g_vec=(), h_vec=(), device=0, parent=None):
        self = cls.__new__(cls)
        self.g, self.g_vec, self.h_vec, self.device, self._ctx, self._parent = g, list(g_vec), list(h_vec), device, ctx, parent
        return self

    @classmethod
    def from_tables(cls, path: str, device: int = 0) -> "U64RangeProofProtocol":
        """A context from a table file written by save_tables (bppp_ctx_create_from_tables)."""
        ctx = C.c_void_p()
        _capi.check(_capi.lib().bppp_ctx_create_from_tables(C.byref(ctx), path.encode(), device))
        return cls._wrap(ctx, device=device)

    def save_tables(self, path: str) -> None:
        _capi.check(_capi.lib().bppp_ctx_save_tables(self._ctx, path.encode()))

    def clone_shared(self) -> "U64RangeProofProtocol":
        """Another context on the same GPU sharing this one's tables (bppp_ctx_create_shared); close it before this one."""
        ctx = C.c_void_p()
        _capi.check(_capi.lib().bppp_ctx_create_shared(C.byref(ctx), self._ctx))
        return type(self)._wrap(ctx, self.g, self.g_vec, self.h_vec, self.device, parent=self)

    @classmethod
    def borrowed(cls, ctx: int) -> "U64RangeProofProtocol":
        """A view over a context owned elsewhere (a device group's rank context: bppp_group_ctx); close() leaves it alone."""
        self = cls._wrap(C.c_void_p(ctx))
        self._is_borrowed = True
        return self

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            if not getattr(self, "_is_borrowed", False):
                _capi.lib().bppp_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- commit_value (u64_proof.rs:37-39)
    def commit_value(self, x: int, s: bytes) -> bytes:
        return bytes(self.commit_value_batch(np.array([x], dtype=np.uint64), _as_u8(s, (1, 32)))[0])

    def commit_value_batch(self, x: np.ndarray, s: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.uint64)
        n = x.shape[0]
        s = _as_u8(s, (n, 32))
        out = np.zeros((n, 64), dtype=np.uint8)
        _capi.check(_capi.lib().bppp_u64_commit_value_batch(self._ctx, n, x.ctypes.data, s.ctypes.data, out.ctypes.data))
        return out

    # ---- verify (u64_proof.rs:42-54)
    def verify(self, v: bytes, proof: bytes, label: bytes) -> bool:
        acc, _ = self.verify_batch(_as_u8(v, (1, 64)), _as_u8(proof, (1, U64_PROOF_BYTES)), label)
        return bool(acc[0])

    def verify_one(self, v: bytes, proof: bytes, transcript) -> Tuple[bool, int]:
        """The reference's own call -- ONE proof, from any number of threads at once (u64_proof.rs:42: `verify(&self, v, proof, t)`) --
        through the coalescing front end (include/bppp.h: bppp_u64_verify_one[_transcript]): the request joins whatever other
        threads have submitted and runs as part of one batched GPU call.  `transcript`: a label (bytes: Transcript::new(label)) or
        a bp_pp_amd.transcript.Transcript, which is advanced in place as the reference's `t: &mut Transcript` is.
        Returns (accept, status).  Blocking; ctypes releases the GIL while it waits."""
        v, proof = bytes(v), bytes(proof)
        if len(v) != 64 or len(proof) != U64_PROOF_BYTES:      # the C side copies exactly this many bytes from the pointers it is handed
            raise ValueError(f"commitment is 64 bytes and a u64 proof {U64_PROOF_BYTES} bytes (got {len(v)}, {len(proof)})")
        acc, st = C.c_uint8(0), C.c_int32(0)
        L = _capi.lib()
        if isinstance(transcript, (bytes, bytearray)):
            _capi.check(L.bppp_u64_verify_one(self._ctx, bytes(transcript), len(transcript), v, proof, C.byref(acc), C.byref(st)))
        else:
            _capi.check(L.bppp_u64_verify_one_transcript(self._ctx, transcript._buf, v, proof, C.byref(acc), C.byref(st)))
        return bool(acc.value), int(st.value)

    def prove_one(self, x: int, s: bytes, transcript, rnd: bytes) -> Tuple[bytes, bytes, int]:
        """`prove(&self, x, s, t, rng)` (u64_proof.rs:57) for ONE value through the coalescing front end; `rnd` = the 52 draws.
        Returns (proof 928 B, commitment 64 B, status)."""
        if len(rnd) != 52 * 32 or len(s) != 32:
            raise ValueError("s is 32 bytes, rnd 52 x 32 bytes")
        proof, com, st = C.create_string_buffer(U64_PROOF_BYTES), C.create_string_buffer(64), C.c_int32(0)
        L = _capi.lib()
        if isinstance(transcript, (bytes, bytearray)):
            _capi.check(L.bppp_u64_prove_one(self._ctx, bytes(transcript), len(transcript), x, bytes(s), bytes(rnd), proof, com, C.byref(st)))
        else:
            _capi.check(L.bppp_u64_prove_one_transcript(self._ctx, transcript._buf, x, bytes(s), bytes(rnd), proof, com, C.byref(st)))
        return proof.raw, com.raw, int(st.value)

    def coalesce_stats(self, which: str = "verify") -> dict:
        """Counters of the single-proof front end (bppp_ctx_get_coalesce_stats)."""
        out = (C.c_uint64 * 8)()
        _capi.check(_capi.lib().bppp_ctx_get_coalesce_stats(self._ctx, 1 if which == "prove" else 0, out))
        return dict(zip(("requests", "batches", "largest_batch", "sealed_full", "sealed_deadline", "run_us", "fill_wait_us"), (int(v) for v in out)))

    def verify_batch(self, commitments, proofs, label: bytes) -> Tuple[np.ndarray, np.ndarray]:
        """Host buffers in, host buffers out.  Returns (accept[n] u8, status[n] i32)."""
        commitments = _as_u8(commitments, (-1, 64))
        n = commitments.shape[0]
        proofs = _as_u8(proofs, (n, U64_PROOF_BYTES))
        accept = np.zeros(n, dtype=np.uint8)
        status = np.zeros(n, dtype=np.int32)
        _capi.check(_capi.lib().bppp_u64_verify_batch(self._ctx, label, len(label), n, commitments.ctypes.data,
                                                      proofs.ctypes.data, accept.ctypes.data, status.ctypes.data))
        return accept, status

    def verify_batch_device(self, label: bytes, n: int, d_commitments: int, d_proofs: int, d_accept: int,
                            d_status: int = 0, d_trace: int = 0, d_reject_count: int = 0) -> None:
        """Everything already resident in HBM (raw device addresses, e.g. torch tensor .data_ptr()); asynchronous on the
        context's stream."""
        _capi.check(_capi.lib().bppp_u64_verify_batch_device(self._ctx, label, len(label), n, d_commitments, d_proofs,
                                                             d_accept, d_status or None, d_trace or None,
                                                             d_reject_count or None))

    def verify_batch_transcript(self, commitments, proofs, transcripts, want_states: bool = True):
        """verify with the caller's transcripts (the reference's `t: &mut Transcript`): `transcripts` is ONE serialized state
        (203 bytes or a bp_pp_amd.transcript.Transcript) shared by the batch, or a sequence of n of them.  Returns (accept,
        status, states_out [n, 203] or None): states_out[i] is proof i's transcript as the reference's verify leaves it."""
        commitments = _as_u8(commitments, (-1, 64))
        n = commitments.shape[0]
        proofs = _as_u8(proofs, (n, U64_PROOF_BYTES))
        as_bytes = lambda t: t.state if hasattr(t, "state") else bytes(t)
        blob = as_bytes(transcripts) if hasattr(transcripts, "state") or isinstance(transcripts, (bytes, bytearray)) else \
            b"".join(as_bytes(t) for t in transcripts)
        states = _as_u8(blob, (-1, 203))
        accept, status = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.int32)
        out = np.zeros((n, 203), dtype=np.uint8) if want_states else None
        _capi.check(_capi.lib().bppp_u64_verify_batch_transcript(self._ctx, n, states.ctypes.data, states.shape[0], commitments.ctypes.data,
                                                                 proofs.ctypes.data, accept.ctypes.data, status.ctypes.data,
                                                                 out.ctypes.data if want_states else None))
        return accept, status, out

    def verify_batch_rlc(self, commitments, proofs, label: bytes, seed: bytes) -> Tuple[np.ndarray, np.ndarray]:
        """verify_batch in the optional RLC mode (host buffers); see verify_batch_rlc_device."""
        if len(seed) != 32:
            raise ValueError("seed must be 32 bytes")
        commitments = _as_u8(commitments, (-1, 64))
        n = commitments.shape[0]
        proofs = _as_u8(proofs, (n, U64_PROOF_BYTES))
        accept, status = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.int32)
        _capi.check(_capi.lib().bppp_u64_verify_batch_rlc(self._ctx, label, len(label), n, commitments.ctypes.data, proofs.ctypes.data,
                                                          accept.ctypes.data, status.ctypes.data, seed))
        return accept, status

    def verify_batch_rlc_device(self, label: bytes, n: int, d_commitments: int, d_proofs: int, d_accept: int, seed: bytes,
                                d_status: int = 0, d_reject_count: int = 0) -> None:
        """Optional batch mode (include/bppp.h: bppp_u64_verify_batch_rlc_device): the final per-proof MSM is replaced by one
        combined check per chunk of 8 proofs with secret weights derived from `seed` (32 unpredictable bytes, chosen after the
        proofs are fixed); failing chunks are re-checked exactly, so accept bits stay per proof."""
        if len(seed) != 32:
            raise ValueError("seed must be 32 bytes")
        _capi.check(_capi.lib().bppp_u64_verify_batch_rlc_device(self._ctx, label, len(label), n, d_commitments, d_proofs, d_accept,
                                                                 d_status or None, d_reject_count or None, seed))

    def verify_batch_sec1_device(self, label: bytes, n: int, d_commitments33: int, d_proofs525: int, d_accept: int, d_status: int = 0,
                                 d_trace: int = 0, d_reject_count: int = 0) -> None:
        """The wire form resident on the device (33-byte SEC1 commitments, 525-byte proofs: reciprocal.rs:37-59): decompressed on the device
        (14 square roots per proof), then the verifier; asynchronous on the context's stream (bppp_u64_verify_batch_sec1_device)."""
        _capi.check(_capi.lib().bppp_u64_verify_batch_sec1_device(self._ctx, label, len(label), n, d_commitments33, d_proofs525, d_accept, d_status,
                                                                  d_trace, d_reject_count))

    def prove_batch_sec1_device(self, label: bytes, n: int, d_x: int, d_s: int, d_rnd: int, d_proofs525: int, d_commitments33: int,
                                d_status: int = 0) -> None:
        """bppp_u64_prove_batch_sec1_device: the batch prover writing the wire form (525-byte proofs, 33-byte commitments) into device buffers."""
        _capi.check(_capi.lib().bppp_u64_prove_batch_sec1_device(self._ctx, label, len(label), n, d_x, d_s, d_rnd, d_proofs525, d_commitments33, d_status))

    def verify_batch_sec1(self, commitments33, proofs525, label: bytes) -> Tuple[np.ndarray, np.ndarray]:
        """verify over the wire content of SerializableProof: 33-byte SEC1 points, 525-byte proofs (bp_pp_amd/wire.py)."""
        commitments33 = _as_u8(commitments33, (-1, 33))
        n = commitments33.shape[0]
        proofs525 = _as_u8(proofs525, (n, 525))
        accept = np.zeros(n, dtype=np.uint8)
        status = np.zeros(n, dtype=np.int32)
        _capi.check(_capi.lib().bppp_u64_verify_batch_sec1(self._ctx, label, len(label), n, commitments33.ctypes.data,
                                                           proofs525.ctypes.data, accept.ctypes.data, status.ctypes.data))
        return accept, status

    # ---- prove (u64_proof.rs:57-82)
    def prove(self, x: int, s: bytes, label: bytes, rnd: bytes) -> bytes:
        """One proof; `rnd` = the 52 x 32 bytes the reference would draw with Scalar::generate_biased, in draw order."""
        proofs, _, st = self.prove_batch(np.array([x], dtype=np.uint64), _as_u8(s, (1, 32)), _as_u8(rnd, (1, 52 * 32)), label)
        if st[0]:
            raise ValueError(f"prove: status {int(st[0])}")
        return bytes(proofs[0])

    def prove_batch(self, x: np.ndarray, s, rnd, label: bytes):
        """Host buffers.  Returns (proofs[n,928], commitments[n,64], status[n])."""
        x = np.ascontiguousarray(x, dtype=np.uint64)
        n = x.shape[0]
        s = _as_u8(s, (n, 32))
        rnd = _as_u8(rnd, (n, 52 * 32))
        proofs = np.zeros((n, U64_PROOF_BYTES), dtype=np.uint8)
        com = np.zeros((n, 64), dtype=np.uint8)
        status = np.zeros(n, dtype=np.int32)
        _capi.check(_capi.lib().bppp_u64_prove_batch(self._ctx, label, len(label), n, x.ctypes.data, s.ctypes.data,
                                                     rnd.ctypes.data, proofs.ctypes.data, com.ctypes.data, status.ctypes.data))
        return proofs, com, status

    def prove_batch_sec1(self, x: np.ndarray, s, rnd, label: bytes):
        """prove_batch with the output in the crate's wire format: (proofs [n, 525], commitments [n, 33], status [n]) -- SEC1-compressed
        points, what verify_batch_sec1 takes."""
        x = np.ascontiguousarray(x, dtype=np.uint64)
        n = x.shape[0]
        s, rnd = _as_u8(s, (n, 32)), _as_u8(rnd, (n, 52 * 32))
        proofs, com = np.zeros((n, 525), dtype=np.uint8), np.zeros((n, 33), dtype=np.uint8)
        status = np.zeros(n, dtype=np.int32)
        _capi.check(_capi.lib().bppp_u64_prove_batch_sec1(self._ctx, label, len(label), n, x.ctypes.data, s.ctypes.data, rnd.ctypes.data,
                                                          proofs.ctypes.data, com.ctypes.data, status.ctypes.data))
        return proofs, com, status

    def prove_batch_transcript(self, x: np.ndarray, s, rnd, transcripts, want_states: bool = True):
        """prove with the caller's transcripts (u64_proof.rs:57: `t: &mut Transcript`): ONE serialized state shared by the batch
        or a sequence of n.  Returns (proofs, commitments, status, states_out [n, 203] or None)."""
        x = np.ascontiguousarray(x, dtype=np.uint64)
        n = x.shape[0]
        s, rnd = _as_u8(s, (n, 32)), _as_u8(rnd, (n, 52 * 32))
        as_bytes = lambda t: t.state if hasattr(t, "state") else bytes(t)
        blob = as_bytes(transcripts) if hasattr(transcripts, "state") or isinstance(transcripts, (bytes, bytearray)) else \
            b"".join(as_bytes(t) for t in transcripts)
        states = _as_u8(blob, (-1, 203))
        proofs, com = np.zeros((n, U64_PROOF_BYTES), dtype=np.uint8), np.zeros((n, 64), dtype=np.uint8)
        status = np.zeros(n, dtype=np.int32)
        out = np.zeros((n, 203), dtype=np.uint8) if want_states else None
        _capi.check(_capi.lib().bppp_u64_prove_batch_transcript(self._ctx, n, states.ctypes.data, states.shape[0], x.ctypes.data, s.ctypes.data,
                                                                rnd.ctypes.data, proofs.ctypes.data, com.ctypes.data, status.ctypes.data,
                                                                out.ctypes.data if want_states else None))
        return proofs, com, status, out

    def prove_batch_device(self, label: bytes, n: int, d_x: int, d_s: int, d_rnd: int, d_proofs: int, d_commitments: int,
                           d_status: int = 0) -> None:
        _capi.check(_capi.lib().bppp_u64_prove_batch_device(self._ctx, label, len(label), n, d_x, d_s, d_rnd, d_proofs,
                                                            d_commitments, d_status or None))

    # ---- plumbing
    def set_stream(self, hip_stream: Optional[int]) -> None:
        """Run the context's kernels on a caller-owned hipStream_t (its raw handle); None restores the context's own stream.
        The NULL (legacy default) stream cannot be selected -- its handle is 0, which the C ABI reads as "restore", and the
        context's own streams are non-blocking, i.e. NOT ordered against the null stream.  Callers that want ordering with
        their own work (torch ops, an RCCL all-reduce of the reject count) create a stream and run both on it:
            s = torch.cuda.Stream(); proto.set_stream(s.cuda_stream); with torch.cuda.stream(s): ..."""
        if hip_stream is not None and int(hip_stream) == 0:
            raise ValueError("the null stream (handle 0) cannot be selected; pass a non-default stream, or None for the context's own")
        _capi.check(_capi.lib().bppp_ctx_set_stream(self._ctx, hip_stream))

    def set_option(self, name: str, value: int) -> None:
        """include/bppp.h: bppp_ctx_set_option ("rlc_superchunk": 0 = bucket stage off, else 64..8192; "host_chunk": proofs per
        pipelined upload chunk of the host-buffer verify calls, 0 = upload first; "coalesce_max" / "coalesce_us" / "coalesce_lanes":
        the single-proof front end of verify_one / prove_one)."""
        _capi.check(_capi.lib().bppp_ctx_set_option(self._ctx, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        """bppp_ctx_get_option: a tunable read back, or "fb_window_bits" (the table width in use), "device", "n_generators"."""
        return int(_capi.check(_capi.lib().bppp_ctx_get_option(self._ctx, name.encode())))

    def last_plan(self, prove: bool = False) -> str:
        """Which kernels the context's last u64 verify (or prove) call ran, as text (include/bppp.h: "last_verify_plan" / "last_prove_plan"
        + bppp_plan_describe) -- e.g. "phase1=wg4 tables=beside/1 fb=l8 c0var=small round=small tail_beside=1 small=1 split=0 shared_inv=0"."""
        return describe_plan(self.get_option("last_prove_plan" if prove else "last_verify_plan"), prove)

    def synchronize(self) -> None:
        """Block until everything queued on the context's current stream (and its helper stream) has finished."""
        _capi.check(_capi.lib().bppp_ctx_synchronize(self._ctx))

    def enable_timing(self, on: bool = True) -> None:
        _capi.check(_capi.lib().bppp_ctx_enable_timing(self._ctx, 1 if on else 0))

    def timings(self, reset: bool = True) -> dict:
        return ctx_timings(self._ctx, reset)

    def device_bytes(self) -> int:
        return int(_capi.lib().bppp_ctx_device_bytes(self._ctx))
