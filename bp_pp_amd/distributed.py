"""Multi-GPU host logic for the batch path: one process per GPU, proofs sharded by index, no data-path collective; the
only exchange is the reject count (4 bytes), summed with one all-reduce (RCCL over xGMI on GPUs: torch.distributed's
"nccl" backend; "gloo" on CPU in the tests)."""
from __future__ import annotations

from typing import Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of proof indices [0, n_total) over `world` ranks: rank r gets [lo, hi)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    lo = n_total * rank // world
    hi = n_total * (rank + 1) // world
    return lo, hi


def _seed32(rlc_seed) -> bytes:
    """The RLC seed as the 32 bytes the C side reads (a shorter buffer would be read out of bounds and silently change the weights)."""
    b = bytes(rlc_seed)
    if len(b) != 32:
        raise ValueError("rlc_seed must be exactly 32 bytes")
    return b


def all_reduce_reject_count(count_tensor):
    """Sum the per-rank reject counts in place; returns the global number of rejected proofs (0 => batch accepted).
    `count_tensor` is a 1-element int32 tensor on this rank's device (the `d_reject_count` of bppp_u64_verify_batch_device)."""
    import torch.distributed as dist
    import os
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("BENCH_FORCE_DIST")):
        dist.all_reduce(count_tensor, op=dist.ReduceOp.SUM)
    return count_tensor


class _Group:
    """Handle of a bppp_group (include/bppp.h): a context, a stream and -- for more than one device -- an RCCL communicator per device
    inside this process.  A sharded call runs one host thread per device; the ranks vote on their return codes before any of them
    enters the accept-reduce, so a failing rank makes the call return its error instead of hanging the others (csrc/group_core.h)."""

    def _create(self, g, g_vec, h_vec, devices, fb_window_bits):
        import ctypes as C
        from . import _capi
        self._capi = _capi
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        self._grp = C.c_void_p()
        gv, hv = [bytes(p) for p in g_vec], [bytes(p) for p in h_vec]
        _capi.check(_capi.lib().bppp_wnla_group_create(C.byref(self._grp), bytes(g), b"".join(gv), len(gv), b"".join(hv), len(hv), arr,
                                                       len(self.devices), fb_window_bits))

    def _lend(self, view):
        """A rank context handed out as a protocol view: the view keeps the group alive, and close() invalidates it (its raw context
        pointer dies with the group)."""
        import weakref
        view._parent = self
        if not hasattr(self, "_views"):
            self._views = []
        self._views.append(weakref.ref(view))
        return view

    def close(self):
        if getattr(self, "_grp", None) is not None and self._grp.value:
            import ctypes as C
            for ref in getattr(self, "_views", []):
                v = ref()
                if v is not None:                       # a use after close raises instead of touching freed memory
                    if hasattr(v, "_w"):
                        v._w._ctx = C.c_void_p()
                    if hasattr(v, "_ctx"):
                        v._ctx = C.c_void_p()
            self._views = []
            self._capi.lib().bppp_group_destroy(self._grp)
            self._grp.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self._capi.lib().bppp_group_size(self._grp))

    def set_option(self, name: str, value: int) -> None:
        """bppp_group_set_option: every bppp_ctx_set_option name (applied on each device), or "inject_fault_rank" (testing aid)."""
        self._capi.check(self._capi.lib().bppp_group_set_option(self._grp, name.encode(), int(value)))

    def _ptrs(self, xs):
        import ctypes as C
        G = len(self.devices)
        if xs is None:
            xs = [0] * G
        return (C.c_void_p * G)(*[C.c_void_p(int(x) if x else None) for x in xs])


class U64RangeProofGroup(_Group):
    """One batch over several GPUs of this node INSIDE one process (include/bppp.h: bppp_group_*): the proofs split contiguously by
    `shard_range`, and one 4-byte ncclAllReduce of the reject count over RCCL (none needed for a single device).  bench.py's
    one-process-per-GPU launch (torch.distributed.run) uses `shard_range` + `all_reduce_reject_count` instead; both give the same
    split and count."""

    def __init__(self, g: bytes, g_vec, h_vec, devices, fb_window_bits: int = 0):
        if len(g_vec) != 16 or len(h_vec) != 32:
            raise ValueError("the u64 protocol has 16 + 32 generators")
        self._create(g, g_vec, h_vec, devices, fb_window_bits)

    def protocol(self, rank: int = 0):
        """Rank `rank`'s context as a U64RangeProofProtocol view (the group keeps ownership)."""
        from .range_proof import U64RangeProofProtocol
        ctx = self._capi.lib().bppp_group_ctx(self._grp, rank)
        if not ctx:
            raise ValueError("rank out of range")
        return self._lend(U64RangeProofProtocol.borrowed(ctx))

    def _host(self, commitments, proofs, cw, pw):
        import numpy as np
        commitments = np.ascontiguousarray(commitments, dtype=np.uint8).reshape(-1, cw)
        n = commitments.shape[0]
        proofs = np.ascontiguousarray(proofs, dtype=np.uint8).reshape(n, pw)
        return n, commitments, proofs, np.zeros(n, np.uint8), np.zeros(n, np.int32)

    def verify_batch(self, commitments, proofs, label: bytes, rlc_seed: bytes = None):
        """Host buffers -> (accept[n] u8, status[n] i32, global reject count); `rlc_seed` selects the optional RLC mode."""
        import ctypes as C
        n, commitments, proofs, accept, status = self._host(commitments, proofs, 64, 928)
        rej, L = C.c_int32(0), self._capi.lib()
        if rlc_seed is None:
            rc = L.bppp_u64_verify_batch_sharded(self._grp, label, len(label), n, commitments.ctypes.data, proofs.ctypes.data,
                                                 accept.ctypes.data, status.ctypes.data, C.byref(rej))
        else:
            rc = L.bppp_u64_verify_batch_rlc_sharded(self._grp, label, len(label), n, commitments.ctypes.data, proofs.ctypes.data,
                                                     accept.ctypes.data, status.ctypes.data, C.byref(rej), _seed32(rlc_seed))
        self._capi.check(rc)
        return accept, status, int(rej.value)

    def verify_batch_sec1(self, commitments33, proofs525, label: bytes):
        """SEC1-compressed host buffers (33-byte commitments, 525-byte proofs) -> (accept, status, global reject count)."""
        import ctypes as C
        n, commitments, proofs, accept, status = self._host(commitments33, proofs525, 33, 525)
        rej = C.c_int32(0)
        self._capi.check(self._capi.lib().bppp_u64_verify_batch_sec1_sharded(self._grp, label, len(label), n, commitments.ctypes.data,
                                                                             proofs.ctypes.data, accept.ctypes.data, status.ctypes.data,
                                                                             C.byref(rej)))
        return accept, status, int(rej.value)

    def verify_batch_transcripts(self, states, commitments, proofs):
        """The caller's merlin transcripts (u64_proof.rs:42): states [1 or n, 203] -> (accept, status, states_out [n, 203], rejects)."""
        import ctypes as C
        import numpy as np
        n, commitments, proofs, accept, status = self._host(commitments, proofs, 64, 928)
        states = np.ascontiguousarray(states, dtype=np.uint8).reshape(-1, 203)
        out, rej = np.zeros((n, 203), np.uint8), C.c_int32(0)
        self._capi.check(self._capi.lib().bppp_u64_verify_batch_transcript_sharded(self._grp, n, states.ctypes.data, states.shape[0],
                                                                                   commitments.ctypes.data, proofs.ctypes.data, accept.ctypes.data,
                                                                                   status.ctypes.data, out.ctypes.data, C.byref(rej)))
        return accept, status, out, int(rej.value)

    def verify_batch_device(self, label: bytes, n: int, d_commitments, d_proofs, d_accept, d_status, d_reject_count, rlc_seed: bytes = None) -> None:
        """Per-device lists of raw device addresses (rank r: its shard of `shard_range(n, r, G)` on device r); blocks until done."""
        mk, L = self._ptrs, self._capi.lib()
        if rlc_seed is None:
            rc = L.bppp_u64_verify_batch_sharded_device(self._grp, label, len(label), n, mk(d_commitments), mk(d_proofs), mk(d_accept),
                                                        mk(d_status), mk(d_reject_count))
        else:
            rc = L.bppp_u64_verify_batch_rlc_sharded_device(self._grp, label, len(label), n, mk(d_commitments), mk(d_proofs), mk(d_accept),
                                                            mk(d_status), mk(d_reject_count), _seed32(rlc_seed))
        self._capi.check(rc)

    def verify_batch_sec1_device(self, label: bytes, n: int, d_commitments33, d_proofs525, d_accept, d_status, d_reject_count) -> None:
        mk = self._ptrs
        self._capi.check(self._capi.lib().bppp_u64_verify_batch_sec1_sharded_device(self._grp, label, len(label), n, mk(d_commitments33),
                                                                                    mk(d_proofs525), mk(d_accept), mk(d_status),
                                                                                    mk(d_reject_count)))

    def verify_batch_transcripts_device(self, n: int, d_states, n_states: int, d_commitments, d_proofs, d_accept, d_status, d_reject_count,
                                        d_states_out=None) -> None:
        mk = self._ptrs
        self._capi.check(self._capi.lib().bppp_u64_verify_batch_transcript_sharded_device(self._grp, n, mk(d_states), n_states, mk(d_commitments),
                                                                                          mk(d_proofs), mk(d_accept), mk(d_status),
                                                                                          mk(d_reject_count), mk(d_states_out)))

    def prove_batch(self, x, s, rnd, label: bytes):
        """U64RangeProofProtocol::prove (u64_proof.rs:57-82) for one batch over the group's devices, host buffers: x [n] u64, s [n, 32],
        rnd [n, 52, 32] (the reference's draw order) -> (proofs [n, 928], commitments [n, 64], status [n]); byte-identical to
        U64RangeProofProtocol.prove_batch on one device.  No collective: proofs are independent."""
        import numpy as np
        x = np.ascontiguousarray(x, dtype=np.uint64).reshape(-1)
        n = x.shape[0]
        s = np.ascontiguousarray(s, dtype=np.uint8).reshape(n, 32)
        rnd = np.ascontiguousarray(rnd, dtype=np.uint8).reshape(n, 52 * 32)
        proofs, commitments, status = np.zeros((n, 928), np.uint8), np.zeros((n, 64), np.uint8), np.zeros(n, np.int32)
        self._capi.check(self._capi.lib().bppp_u64_prove_batch_sharded(self._grp, label, len(label), n, x.ctypes.data, s.ctypes.data,
                                                                       rnd.ctypes.data, proofs.ctypes.data, commitments.ctypes.data,
                                                                       status.ctypes.data))
        return proofs, commitments, status

    def prove_batch_device(self, label: bytes, n: int, d_x, d_s, d_rnd, d_proofs, d_commitments, d_status=None) -> None:
        """Per-device lists of raw device addresses of each rank's shard; blocks until every device is done."""
        mk = self._ptrs
        self._capi.check(self._capi.lib().bppp_u64_prove_batch_sharded_device(self._grp, label, len(label), n, mk(d_x), mk(d_s), mk(d_rnd),
                                                                              mk(d_proofs), mk(d_commitments), mk(d_status)))


class ReciprocalRangeProofGroup(_Group):
    """ReciprocalRangeProofProtocol::verify (reciprocal.rs:98-107) for one batch over the GPUs of a node -- BASELINE configs[4]: the
    (dim_nd 256, dim_np 16) shape, 2^18 instances on 8 GPUs.  Generators as bp_pp_amd.wnla.ReciprocalRangeProofProtocol takes them."""

    def __init__(self, dim_nd: int, dim_np: int, g: bytes, g_vec, h_vec, g_vec_, h_vec_, devices, fb_window_bits: int = 0):
        if len(g_vec) != dim_nd or len(h_vec) != dim_nd + 10:
            raise ValueError("g_vec must hold dim_nd points and h_vec dim_nd + 10")
        self.dim_nd, self.dim_np = dim_nd, dim_np
        self._ng, self._nh = len(g_vec) + len(g_vec_), len(h_vec) + len(h_vec_)
        self._create(g, list(g_vec) + list(g_vec_), list(h_vec) + list(h_vec_), devices, fb_window_bits)

    def protocol(self, rank: int = 0):
        """Rank `rank`'s context as a ReciprocalRangeProofProtocol (prove / commit / single-device verify on that GPU, over the
        group's own tables); the group keeps ownership."""
        from .wnla import ReciprocalRangeProofProtocol
        ctx = self._capi.lib().bppp_group_ctx(self._grp, rank)
        if not ctx:
            raise ValueError("rank out of range")
        return self._lend(ReciprocalRangeProofProtocol.borrowed(self.dim_nd, self.dim_np, ctx, self._ng, self._nh))

    def verify_batch(self, label: bytes, commitments, proofs, rounds: int, nl: int, nn: int, rlc_seed: bytes = None):
        """Host buffers -> (accept[n] u8, status[n] i32, global reject count)."""
        import ctypes as C
        import numpy as np
        commitments = np.ascontiguousarray(commitments, dtype=np.uint8).reshape(-1, 64)
        n = commitments.shape[0]
        proofs = np.ascontiguousarray(proofs, dtype=np.uint8).reshape(n, 64 * (5 + 2 * rounds) + 32 * (nl + nn))
        accept, status, rej, L = np.zeros(n, np.uint8), np.zeros(n, np.int32), C.c_int32(0), self._capi.lib()
        if rlc_seed is None:
            rc = L.bppp_reciprocal_verify_batch_sharded(self._grp, label, len(label), n, self.dim_nd, self.dim_np, commitments.ctypes.data,
                                                        proofs.ctypes.data, rounds, nl, nn, accept.ctypes.data, status.ctypes.data, C.byref(rej))
        else:
            rc = L.bppp_reciprocal_verify_batch_rlc_sharded(self._grp, label, len(label), n, self.dim_nd, self.dim_np, commitments.ctypes.data,
                                                            proofs.ctypes.data, rounds, nl, nn, accept.ctypes.data, status.ctypes.data,
                                                            C.byref(rej), _seed32(rlc_seed))
        self._capi.check(rc)
        return accept, status, int(rej.value)

    def verify_batch_device(self, label: bytes, n: int, d_commitments, d_proofs, rounds: int, nl: int, nn: int, d_accept, d_status,
                            d_reject_count, rlc_seed: bytes = None) -> None:
        mk, L = self._ptrs, self._capi.lib()
        if rlc_seed is None:
            rc = L.bppp_reciprocal_verify_batch_sharded_device(self._grp, label, len(label), n, self.dim_nd, self.dim_np, mk(d_commitments),
                                                               mk(d_proofs), rounds, nl, nn, mk(d_accept), mk(d_status), mk(d_reject_count))
        else:
            rc = L.bppp_reciprocal_verify_batch_rlc_sharded_device(self._grp, label, len(label), n, self.dim_nd, self.dim_np, mk(d_commitments),
                                                                   mk(d_proofs), rounds, nl, nn, mk(d_accept), mk(d_status), mk(d_reject_count),
                                                                   _seed32(rlc_seed))
        self._capi.check(rc)
