"""Host-side mirror of the reference's `wnla::WeightNormLinearArgument` (wnla.rs:12-19, 66-121), batch-first over the C ABI.
The generators (g, g_vec, h_vec) live in the context and are shared by the batch; c, rho, mu are per instance."""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

import numpy as np

from . import _capi


def _u8(a, shape):
    arr = np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a, dtype=np.uint8)
    return arr.reshape(shape)


def _states(transcripts) -> np.ndarray:
    """ONE serialized merlin state (203 bytes or a bp_pp_amd.transcript.Transcript) shared by the batch, a sequence of them, or an
    [n, 203] array -> [n_states, 203] uint8."""
    if isinstance(transcripts, np.ndarray):
        return _u8(transcripts, (-1, 203))
    as_bytes = lambda t: t.state if hasattr(t, "state") else bytes(t)
    blob = as_bytes(transcripts) if hasattr(transcripts, "state") or isinstance(transcripts, (bytes, bytearray)) else \
        b"".join(as_bytes(t) for t in transcripts)
    return _u8(blob, (-1, 203))


class WeightNormLinearArgument:
    def __init__(self, g: bytes, g_vec: Sequence[bytes], h_vec: Sequence[bytes], device: int = 0, fb_window_bits: int = 0,
                 fb_table_budget_bytes: int = 0):
        self.ng, self.nh = len(g_vec), len(h_vec)
        self._ctx = C.c_void_p()
        _capi.check(_capi.lib().bppp_wnla_ctx_create_budget(C.byref(self._ctx), bytes(g), b"".join(g_vec), self.ng, b"".join(h_vec),
                                                            self.nh, device, fb_window_bits, int(fb_table_budget_bytes)))

    @classmethod
    def borrowed(cls, ctx: int, ng: int, nh: int) -> "WeightNormLinearArgument":
        """A view over a context somebody else owns (a device group's rank context: bppp_group_ctx); close() leaves it alone."""
        self = cls.__new__(cls)
        self.ng, self.nh, self._ctx, self._borrowed = ng, nh, C.c_void_p(ctx), True
        return self

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            if not getattr(self, "_borrowed", False):
                _capi.lib().bppp_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def commit_batch(self, c, mu, l, n) -> Tuple[np.ndarray, np.ndarray]:
        """wnla.rs:66-72 for a batch: c [B, nh, 32], mu [B, 32], l [B, nl, 32], n [B, nn, 32] -> (points [B, 64], status [B])."""
        mu = _u8(mu, (-1, 32))
        B = mu.shape[0]
        c = _u8(c, (B, self.nh, 32))
        l = _u8(l, (B, -1, 32))
        n = _u8(n, (B, -1, 32))
        out = np.zeros((B, 64), np.uint8)
        st = np.zeros(B, np.int32)
        _capi.check(_capi.lib().bppp_wnla_commit_batch(self._ctx, B, c.ctypes.data, mu.ctypes.data, l.ctypes.data, l.shape[1],
                                                       n.ctypes.data, n.shape[1], out.ctypes.data, st.ctypes.data))
        return out, st

    def prove_batch(self, label: bytes, commitments, c, rho, mu, l, n, transcripts=None):
        """wnla.rs:125-190 for a batch: commitments [B, 64], c [B, nh, 32], rho / mu [B, 32], l [B, nl, 32], n [B, nn, 32]
        -> (proof_r [B, rounds, 64], proof_x, proof_l [B, nl', 32], proof_n [B, nn', 32], status [B]).  With `transcripts` (the
        reference's `t: &mut Transcript`: one serialized state or B of them; `label` is ignored) the advanced states [B, 203] are
        appended to the result."""
        import ctypes as C
        commitments = _u8(commitments, (-1, 64))
        B = commitments.shape[0]
        c = _u8(c, (B, self.nh, 32))
        rho, mu = _u8(rho, (B, 32)), _u8(mu, (B, 32))
        l, n = _u8(l, (B, -1, 32)), _u8(n, (B, -1, 32))
        rounds, nl_f, nn_f = C.c_size_t(), C.c_size_t(), C.c_size_t()
        _capi.lib().bppp_wnla_proof_shape(l.shape[1], n.shape[1], C.byref(rounds), C.byref(nl_f), C.byref(nn_f))
        pr, px = np.zeros((B, rounds.value, 64), np.uint8), np.zeros((B, rounds.value, 64), np.uint8)
        pl, pn = np.zeros((B, nl_f.value, 32), np.uint8), np.zeros((B, nn_f.value, 32), np.uint8)
        st = np.zeros(B, np.int32)
        if transcripts is not None:
            S, out = _states(transcripts), np.zeros((B, 203), np.uint8)
            _capi.check(_capi.lib().bppp_wnla_prove_batch_transcript(self._ctx, B, S.ctypes.data, S.shape[0], commitments.ctypes.data,
                                                                     c.ctypes.data, rho.ctypes.data, mu.ctypes.data, l.ctypes.data, l.shape[1],
                                                                     n.ctypes.data, n.shape[1], pr.ctypes.data, px.ctypes.data, pl.ctypes.data,
                                                                     pn.ctypes.data, st.ctypes.data, out.ctypes.data))
            return pr, px, pl, pn, st, out
        _capi.check(_capi.lib().bppp_wnla_prove_batch(self._ctx, label, len(label), B, commitments.ctypes.data, c.ctypes.data, rho.ctypes.data,
                                                      mu.ctypes.data, l.ctypes.data, l.shape[1], n.ctypes.data, n.shape[1], pr.ctypes.data,
                                                      px.ctypes.data, pl.ctypes.data, pn.ctypes.data, st.ctypes.data))
        return pr, px, pl, pn, st

    def msm_batch(self, base_index, scalars) -> Tuple[np.ndarray, np.ndarray]:
        """sum_j scalars[i][j] * B[base_index[j]] per row over the context's generators (0 = g, 1.. = g_vec, 1 + ng.. = h_vec):
        the crate's commit functions (include/bppp.h: bppp_msm_batch).  -> (points [B, 64], status [B])."""
        idx = np.ascontiguousarray(np.asarray(base_index, dtype=np.int32))
        scalars = _u8(scalars, (-1, idx.shape[0], 32))
        B = scalars.shape[0]
        out, st = np.zeros((B, 64), np.uint8), np.zeros(B, np.int32)
        _capi.check(_capi.lib().bppp_msm_batch(self._ctx, B, idx.shape[0], idx.ctypes.data, scalars.ctypes.data, out.ctypes.data,
                                               st.ctypes.data))
        return out, st

    def verify_batch_device(self, label: bytes, n: int, d_commitments: int, d_c: int, d_rho: int, d_mu: int, rounds: int, d_proof_r: int,
                            d_proof_x: int, d_proof_l: int, nl: int, d_proof_n: int, nn: int, d_accept: int, d_status: int = 0) -> None:
        """wnla.rs:75-121 over device buffers (raw device pointers, layouts of verify_batch), asynchronous on the context's stream."""
        _capi.check(_capi.lib().bppp_wnla_verify_batch_device(self._ctx, label, len(label), n, d_commitments, d_c, d_rho, d_mu, rounds, d_proof_r,
                                                              d_proof_x, d_proof_l, nl, d_proof_n, nn, d_accept, d_status))

    def synchronize(self) -> None:
        _capi.check(_capi.lib().bppp_ctx_synchronize(self._ctx))

    def get_option(self, name: str) -> int:
        return int(_capi.check(_capi.lib().bppp_ctx_get_option(self._ctx, name.encode())))

    def enable_timing(self, on: bool = True) -> None:
        _capi.check(_capi.lib().bppp_ctx_enable_timing(self._ctx, 1 if on else 0))

    def timings(self, reset: bool = True) -> dict:
        from .range_proof import ctx_timings
        return ctx_timings(self._ctx, reset)

    def verify_batch(self, label: bytes, commitments, c, rho, mu, proof_r, proof_x, proof_l, proof_n, transcripts=None):
        """wnla.rs:75-121 for a batch; proof_r / proof_x: [B, rounds, 64] in the reference's vector order.  -> (accept, status), or
        with `transcripts` (one serialized state or B; `label` ignored) -> (accept, status, advanced states [B, 203])."""
        commitments = _u8(commitments, (-1, 64))
        B = commitments.shape[0]
        c = _u8(c, (B, self.nh, 32))
        rho, mu = _u8(rho, (B, 32)), _u8(mu, (B, 32))
        proof_r, proof_x = _u8(proof_r, (B, -1, 64)), _u8(proof_x, (B, -1, 64))
        if proof_r.shape[1] != proof_x.shape[1]:
            return np.zeros(B, np.uint8), np.zeros(B, np.int32)          # wnla.rs:76-78
        proof_l, proof_n = _u8(proof_l, (B, -1, 32)), _u8(proof_n, (B, -1, 32))
        acc = np.zeros(B, np.uint8)
        st = np.zeros(B, np.int32)
        if transcripts is not None:
            S, out = _states(transcripts), np.zeros((B, 203), np.uint8)
            _capi.check(_capi.lib().bppp_wnla_verify_batch_transcript(self._ctx, B, S.ctypes.data, S.shape[0], commitments.ctypes.data,
                                                                      c.ctypes.data, rho.ctypes.data, mu.ctypes.data, proof_r.shape[1],
                                                                      proof_r.ctypes.data, proof_x.ctypes.data, proof_l.ctypes.data,
                                                                      proof_l.shape[1], proof_n.ctypes.data, proof_n.shape[1], acc.ctypes.data,
                                                                      st.ctypes.data, out.ctypes.data))
            return acc, st, out
        _capi.check(_capi.lib().bppp_wnla_verify_batch(self._ctx, label, len(label), B, commitments.ctypes.data, c.ctypes.data,
                                                       rho.ctypes.data, mu.ctypes.data, proof_r.shape[1], proof_r.ctypes.data,
                                                       proof_x.ctypes.data, proof_l.ctypes.data, proof_l.shape[1], proof_n.ctypes.data,
                                                       proof_n.shape[1], acc.ctypes.data, st.ctypes.data))
        return acc, st


class ReciprocalRangeProofProtocol:
    """Mirror of `range_proof::reciprocal::ReciprocalRangeProofProtocol` (reciprocal.rs:64-107) for runtime dim_nd / dim_np,
    verify only.  dim_nd = dim_np = 16 is what U64RangeProofProtocol specialises."""

    def __init__(self, dim_nd: int, dim_np: int, g: bytes, g_vec, h_vec, g_vec_, h_vec_, device: int = 0, fb_window_bits: int = 0):
        if len(g_vec) != dim_nd or len(h_vec) != dim_nd + 10:
            raise ValueError("g_vec must hold dim_nd points and h_vec dim_nd + 10 (dim_nv + 9)")
        self.dim_nd, self.dim_np = dim_nd, dim_np
        self._w = WeightNormLinearArgument(g, list(g_vec) + list(g_vec_), list(h_vec) + list(h_vec_), device, fb_window_bits)

    @classmethod
    def borrowed(cls, dim_nd: int, dim_np: int, ctx: int, ng: int, nh: int) -> "ReciprocalRangeProofProtocol":
        """The protocol over a context owned elsewhere (bp_pp_amd.distributed.ReciprocalRangeProofGroup.protocol)."""
        self = cls.__new__(cls)
        self.dim_nd, self.dim_np = dim_nd, dim_np
        self._w = WeightNormLinearArgument.borrowed(ctx, ng, nh)
        return self

    def close(self):
        self._w.close()

    def prove_batch(self, label: bytes, commitments, x, s, digits, m, rnd, transcripts=None):
        """reciprocal.rs:110-146 for a batch: commitments [B, 64], x / s [B, 32], digits [B, dim_nd, 32], m [B, dim_np, 32],
        rnd [B, 20 + 2 dim_nd, 32] -> (proofs, status, (rounds, nl, nn)); with `transcripts` (`label` ignored) the advanced states
        [B, 203] are appended."""
        import ctypes as C
        commitments = _u8(commitments, (-1, 64))
        B = commitments.shape[0]
        x, s = _u8(x, (B, 32)), _u8(s, (B, 32))
        digits, m = _u8(digits, (B, self.dim_nd, 32)), _u8(m, (B, self.dim_np, 32))
        rnd = _u8(rnd, (B, 20 + 2 * self.dim_nd, 32))
        rounds, nl, nn = C.c_size_t(), C.c_size_t(), C.c_size_t()
        _capi.lib().bppp_wnla_proof_shape(self._w.nh, self._w.ng, C.byref(rounds), C.byref(nl), C.byref(nn))
        proofs = np.zeros((B, 64 * (5 + 2 * rounds.value) + 32 * (nl.value + nn.value)), np.uint8)
        st = np.zeros(B, np.int32)
        if transcripts is not None:
            S, out = _states(transcripts), np.zeros((B, 203), np.uint8)
            _capi.check(_capi.lib().bppp_reciprocal_prove_batch_transcript(self._w._ctx, B, S.ctypes.data, S.shape[0], self.dim_nd, self.dim_np,
                                                                           commitments.ctypes.data, x.ctypes.data, s.ctypes.data,
                                                                           digits.ctypes.data, m.ctypes.data, rnd.ctypes.data, proofs.ctypes.data,
                                                                           st.ctypes.data, out.ctypes.data))
            return proofs, st, (rounds.value, nl.value, nn.value), out
        _capi.check(_capi.lib().bppp_reciprocal_prove_batch(self._w._ctx, label, len(label), B, self.dim_nd, self.dim_np,
                                                            commitments.ctypes.data, x.ctypes.data, s.ctypes.data, digits.ctypes.data,
                                                            m.ctypes.data, rnd.ctypes.data, proofs.ctypes.data, st.ctypes.data))
        return proofs, st, (rounds.value, nl.value, nn.value)

    def commit_value_batch(self, x, s):
        """reciprocal.rs:88-90 for a batch: x [B, 32], s [B, 32] (big-endian scalars) -> (points, status)."""
        x, s = _u8(x, (-1, 32)), _u8(s, (-1, 32))
        return self._w.msm_batch([0, 1 + self._w.ng], np.stack([x, s], axis=1))

    def commit_poles_batch(self, r, s):
        """reciprocal.rs:93-95 for a batch: r [B, dim_nd, 32], s [B, 32] -> (points, status)."""
        r, s = _u8(r, (-1, self.dim_nd, 32)), _u8(s, (-1, 32))
        idx = [1 + self._w.ng] + [1 + self._w.ng + 9 + i for i in range(self.dim_nd)]
        return self._w.msm_batch(idx, np.concatenate([s[:, None, :], r], axis=1))

    def verify_batch(self, label: bytes, commitments, proofs, rounds: int, nl: int, nn: int, transcripts=None):
        """reciprocal.rs:98-107 for a batch -> (accept, status), or with `transcripts` (`label` ignored) (accept, status, states)."""
        commitments = _u8(commitments, (-1, 64))
        B = commitments.shape[0]
        proofs = _u8(proofs, (B, 64 * (5 + 2 * rounds) + 32 * (nl + nn)))
        acc, st = np.zeros(B, np.uint8), np.zeros(B, np.int32)
        if transcripts is not None:
            S, out = _states(transcripts), np.zeros((B, 203), np.uint8)
            _capi.check(_capi.lib().bppp_reciprocal_verify_batch_transcript(self._w._ctx, B, S.ctypes.data, S.shape[0], self.dim_nd, self.dim_np,
                                                                            commitments.ctypes.data, proofs.ctypes.data, rounds, nl, nn,
                                                                            acc.ctypes.data, st.ctypes.data, out.ctypes.data))
            return acc, st, out
        _capi.check(_capi.lib().bppp_reciprocal_verify_batch(self._w._ctx, label, len(label), B, self.dim_nd, self.dim_np,
                                                             commitments.ctypes.data, proofs.ctypes.data, rounds, nl, nn,
                                                             acc.ctypes.data, st.ctypes.data))
        return acc, st


    def verify_one(self, commitment: bytes, proof: bytes, rounds: int, nl: int, nn: int, transcript):
        """`ReciprocalRangeProofProtocol::verify(&self, commitment, proof, t)` (reciprocal.rs:98-107) for ONE instance, from any number
        of threads: the call joins the other threads' calls of the same shape in one batched GPU call (include/bppp.h:
        bppp_reciprocal_verify_one[_transcript]).  `transcript`: a label (bytes) or a bp_pp_amd.transcript.Transcript (advanced in
        place).  -> (accept, status)"""
        import ctypes as C
        commitment, proof = bytes(commitment), bytes(proof)
        want = 64 * (5 + 2 * rounds) + 32 * (nl + nn)      # what the C side reads from the proof pointer (include/bppp.h)
        if len(commitment) != 64 or len(proof) != want:
            raise ValueError(f"commitment is 64 bytes and this shape's proof {want} bytes (got {len(commitment)}, {len(proof)})")
        acc, st = C.c_uint8(0), C.c_int32(0)
        L = _capi.lib()
        if isinstance(transcript, (bytes, bytearray)):
            _capi.check(L.bppp_reciprocal_verify_one(self._w._ctx, bytes(transcript), len(transcript), self.dim_nd, self.dim_np, commitment,
                                                     proof, rounds, nl, nn, C.byref(acc), C.byref(st)))
        else:
            _capi.check(L.bppp_reciprocal_verify_one_transcript(self._w._ctx, transcript._buf, self.dim_nd, self.dim_np, commitment,
                                                                proof, rounds, nl, nn, C.byref(acc), C.byref(st)))
        return bool(acc.value), int(st.value)

    def verify_batch_device(self, label: bytes, n: int, d_commitments: int, d_proofs: int, rounds: int, nl: int, nn: int, d_accept: int,
                            d_status: int) -> None:
        """Everything resident in HBM (raw device addresses); asynchronous on the context's stream."""
        _capi.check(_capi.lib().bppp_reciprocal_verify_batch_device(self._w._ctx, label, len(label), n, self.dim_nd, self.dim_np,
                                                                    d_commitments, d_proofs, rounds, nl, nn, d_accept, d_status))

    def verify_batch_rlc(self, label: bytes, commitments, proofs, rounds: int, nl: int, nn: int, seed: bytes):
        """verify_batch in the optional random-linear-combination mode (include/bppp.h: bppp_reciprocal_verify_batch_rlc): one final MSM
        per chunk of 8 instances; accept / status stay per instance."""
        if len(seed) != 32:
            raise ValueError("seed must be 32 bytes")
        commitments = _u8(commitments, (-1, 64))
        B = commitments.shape[0]
        proofs = _u8(proofs, (B, 64 * (5 + 2 * rounds) + 32 * (nl + nn)))
        acc, st = np.zeros(B, np.uint8), np.zeros(B, np.int32)
        _capi.check(_capi.lib().bppp_reciprocal_verify_batch_rlc(self._w._ctx, label, len(label), B, self.dim_nd, self.dim_np,
                                                                 commitments.ctypes.data, proofs.ctypes.data, rounds, nl, nn,
                                                                 acc.ctypes.data, st.ctypes.data, seed))
        return acc, st

    def verify_batch_rlc_device(self, label: bytes, n: int, d_commitments: int, d_proofs: int, rounds: int, nl: int, nn: int, d_accept: int,
                                d_status: int, seed: bytes) -> None:
        if len(seed) != 32:
            raise ValueError("seed must be 32 bytes")
        _capi.check(_capi.lib().bppp_reciprocal_verify_batch_rlc_device(self._w._ctx, label, len(label), n, self.dim_nd, self.dim_np,
                                                                        d_commitments, d_proofs, rounds, nl, nn, d_accept, d_status, seed))

    def synchronize(self) -> None:
        _capi.check(_capi.lib().bppp_ctx_synchronize(self._w._ctx))

    def set_option(self, name: str, value: int) -> None:
        """bppp_ctx_set_option ("rlc_superchunk": 0 = no bucket stage in front of the RLC mode, else the superchunk size 64..8192;
        unset = chosen per call from the batch size)."""
        _capi.check(_capi.lib().bppp_ctx_set_option(self._w._ctx, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        """bppp_ctx_get_option: a tunable read back, or a fact such as "fb_window_bits" (the table width the library chose)."""
        return int(_capi.check(_capi.lib().bppp_ctx_get_option(self._w._ctx, name.encode())))

    def set_stream(self, hip_stream: int) -> None:
        """Run this context's kernels on the caller's HIP stream (0 = back to the context's own)."""
        _capi.check(_capi.lib().bppp_ctx_set_stream(self._w._ctx, hip_stream or None))

    def enable_timing(self, on: bool = True) -> None:
        _capi.check(_capi.lib().bppp_ctx_enable_timing(self._w._ctx, 1 if on else 0))

    def timings(self, reset: bool = True) -> dict:
        from .range_proof import ctx_timings
        return ctx_timings(self._w._ctx, reset)

    def device_bytes(self) -> int:
        return int(_capi.lib().bppp_ctx_device_bytes(self._w._ctx))


class ArithmeticCircuit:
    """Mirror of `circuit::ArithmeticCircuit` (circuit.rs:95-139), verify only, for a circuit shared by a batch of instances.

    W_m / W_l: row-major lists (or arrays) of 32-byte big-endian scalars, a_m / a_l likewise; `partition(typ, index)` is the
    reference's closure with typ in {"LO", "LL", "LR", "NO"}, returning an index into w_o or None -- it is sampled once here
    into the four index tables the C ABI takes."""

    def __init__(self, dim_nm: int, dim_no: int, k: int, dim_nv: int, g: bytes, g_vec, h_vec, W_m, W_l, a_m, a_l, f_l: bool, f_m: bool,
                 g_vec_, h_vec_, partition, device: int = 0, fb_window_bits: int = 0):
        import ctypes as C
        if len(g_vec) != dim_nm or len(h_vec) != dim_nv + 9:
            raise ValueError("g_vec must hold dim_nm points and h_vec dim_nv + 9")
        self.dim_nm, self.dim_no, self.k, self.dim_nv = dim_nm, dim_no, k, dim_nv
        self.dim_nl, self.dim_nw = dim_nv * k, 2 * dim_nm + dim_no
        self._w = WeightNormLinearArgument(g, list(g_vec) + list(g_vec_), list(h_vec) + list(h_vec_), device, fb_window_bits)
        Wm = _u8(W_m, (dim_nm * self.dim_nw, 32))
        Wl = _u8(W_l, (self.dim_nl * self.dim_nw, 32))
        am, al = _u8(a_m, (dim_nm, 32)), _u8(a_l, (self.dim_nl, 32))
        tab = lambda typ, size: np.array([-1 if partition(typ, j) is None else int(partition(typ, j)) for j in range(size)], np.int32)
        lo, ll, lr, no = tab("LO", dim_nv), tab("LL", dim_nv), tab("LR", dim_nv), tab("NO", dim_nm)
        dims = (C.c_size_t * 6)(dim_nm, dim_no, k, self.dim_nl, dim_nv, self.dim_nw)
        h = C.c_void_p()
        try:
            _capi.check(_capi.lib().bppp_circuit_create(self._w._ctx, C.byref(h), dims, 1 if f_l else 0, 1 if f_m else 0, Wm.ctypes.data,
                                                        Wl.ctypes.data, am.ctypes.data, al.ctypes.data, lo.ctypes.data, ll.ctypes.data,
                                                        lr.ctypes.data, no.ctypes.data))
        except Exception:
            self._w.close()
            raise
        self._circuit = h

    def close(self):
        if getattr(self, "_circuit", None):
            _capi.lib().bppp_circuit_destroy(self._circuit)
            self._circuit = None
        self._w.close()

    def prove_batch(self, label: bytes, v_commitments, v, s_v, w_l, w_r, w_o, rnd, transcripts=None):
        """circuit.rs:260-556 for a batch: v_commitments [B, k, 64], v [B, k, dim_nv, 32], s_v [B, k, 32], w_l / w_r [B, dim_nm, 32],
        w_o [B, dim_no, 32], rnd [B, 18 + dim_nv + dim_nm, 32] (the prover's random scalars in the reference's draw order)
        -> (proofs [B, proof_bytes], status [B], (rounds, nl, nn)); with `transcripts` (circuit.rs:260 `t: &mut Transcript`; `label`
        ignored) the advanced states [B, 203] are appended."""
        import ctypes as C
        v_commitments = _u8(v_commitments, (-1, self.k, 64))
        B = v_commitments.shape[0]
        v, s_v = _u8(v, (B, self.k, self.dim_nv, 32)), _u8(s_v, (B, self.k, 32))
        w_l, w_r = _u8(w_l, (B, self.dim_nm, 32)), _u8(w_r, (B, self.dim_nm, 32))
        w_o = _u8(w_o, (B, self.dim_no, 32))
        rnd = _u8(rnd, (B, 18 + self.dim_nv + self.dim_nm, 32))
        rounds, nl, nn = C.c_size_t(), C.c_size_t(), C.c_size_t()
        _capi.lib().bppp_wnla_proof_shape(self._w.nh, self._w.ng, C.byref(rounds), C.byref(nl), C.byref(nn))
        proofs = np.zeros((B, 64 * (4 + 2 * rounds.value) + 32 * (nl.value + nn.value)), np.uint8)
        st = np.zeros(B, np.int32)
        if transcripts is not None:
            S, out = _states(transcripts), np.zeros((B, 203), np.uint8)
            _capi.check(_capi.lib().bppp_circuit_prove_batch_transcript(self._w._ctx, self._circuit, B, S.ctypes.data, S.shape[0],
                                                                        v_commitments.ctypes.data, v.ctypes.data, s_v.ctypes.data, w_l.ctypes.data,
                                                                        w_r.ctypes.data, w_o.ctypes.data, rnd.ctypes.data, proofs.ctypes.data,
                                                                        st.ctypes.data, out.ctypes.data))
            return proofs, st, (rounds.value, nl.value, nn.value), out
        _capi.check(_capi.lib().bppp_circuit_prove_batch(self._w._ctx, self._circuit, label, len(label), B, v_commitments.ctypes.data,
                                                         v.ctypes.data, s_v.ctypes.data, w_l.ctypes.data, w_r.ctypes.data, w_o.ctypes.data,
                                                         rnd.ctypes.data, proofs.ctypes.data, st.ctypes.data))
        return proofs, st, (rounds.value, nl.value, nn.value)

    def commit_batch(self, v, s):
        """circuit.rs:146-151 for a batch: v [B, dim_nv, 32], s [B, 32] -> (points, status)."""
        v, s = _u8(v, (-1, self.dim_nv, 32)), _u8(s, (-1, 32))
        ng = self._w.ng
        idx = [0, 1 + ng] + [1 + ng + 9 + i for i in range(self.dim_nv - 1)]
        return self._w.msm_batch(idx, np.concatenate([v[:, :1, :], s[:, None, :], v[:, 1:, :]], axis=1))

    def verify_batch_device(self, label: bytes, n: int, d_commitments: int, d_proofs: int, rounds: int, nl: int, nn: int, d_accept: int,
                            d_status: int = 0) -> None:
        """circuit.rs:154-256 over device buffers (raw device pointers), asynchronous on the context's stream."""
        _capi.check(_capi.lib().bppp_circuit_verify_batch_device(self._w._ctx, self._circuit, label, len(label), n, d_commitments, d_proofs, rounds,
                                                                 nl, nn, d_accept, d_status))

    def synchronize(self) -> None:
        self._w.synchronize()

    def enable_timing(self, on: bool = True) -> None:
        self._w.enable_timing(on)

    def timings(self, reset: bool = True) -> dict:
        return self._w.timings(reset)

    def verify_batch(self, label: bytes, commitments, proofs, rounds: int, nl: int, nn: int, transcripts=None):
        """circuit.rs:154-256 for a batch: commitments [B, k, 64], proofs [B, 64 (4 + 2 rounds) + 32 (nl + nn)] -> (accept, status),
        or with `transcripts` (`label` ignored) (accept, status, advanced states [B, 203])."""
        commitments = _u8(commitments, (-1, self.k, 64))
        B = commitments.shape[0]
        proofs = _u8(proofs, (B, 64 * (4 + 2 * rounds) + 32 * (nl + nn)))
        acc, st = np.zeros(B, np.uint8), np.zeros(B, np.int32)
        if transcripts is not None:
            S, out = _states(transcripts), np.zeros((B, 203), np.uint8)
            _capi.check(_capi.lib().bppp_circuit_verify_batch_transcript(self._w._ctx, self._circuit, B, S.ctypes.data, S.shape[0],
                                                                         commitments.ctypes.data, proofs.ctypes.data, rounds, nl, nn,
                                                                         acc.ctypes.data, st.ctypes.data, out.ctypes.data))
            return acc, st, out
        _capi.check(_capi.lib().bppp_circuit_verify_batch(self._w._ctx, self._circuit, label, len(label), B, commitments.ctypes.data,
                                                          proofs.ctypes.data, rounds, nl, nn, acc.ctypes.data, st.ctypes.data))
        return acc, st
