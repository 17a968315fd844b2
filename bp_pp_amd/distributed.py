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


def all_reduce_reject_count(count_tensor):
    """Sum the per-rank reject counts in place; returns the global number of rejected proofs (0 => batch accepted).
    `count_tensor` is a 1-element int32 tensor on this rank's device (the `d_reject_count` of bppp_u64_verify_batch_device)."""
    import torch.distributed as dist
    import os
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("BENCH_FORCE_DIST")):
        dist.all_reduce(count_tensor, op=dist.ReduceOp.SUM)
    return count_tensor


class U64RangeProofGroup:
    """One batch over several GPUs of this node INSIDE one process (include/bppp.h: bppp_group_*): a context and a stream per
    device, one host thread per device during a call, the proofs split contiguously by `shard_range`, and one 4-byte
    ncclAllReduce of the reject count over RCCL (none needed for a single device).  bench.py's one-process-per-GPU launch
    (torch.distributed.run) uses `shard_range` + `all_reduce_reject_count` instead; both give the same split and count."""

    def __init__(self, g: bytes, g_vec, h_vec, devices, fb_window_bits: int = 0):
        import ctypes as C
        from . import _capi
        self._capi = _capi
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        self._grp = C.c_void_p()
        _capi.check(_capi.lib().bppp_group_create(C.byref(self._grp), bytes(g), b"".join(bytes(p) for p in g_vec),
                                                  b"".join(bytes(p) for p in h_vec), arr, len(self.devices), fb_window_bits))

    def close(self):
        if getattr(self, "_grp", None) is not None and self._grp.value:
            self._capi.lib().bppp_group_destroy(self._grp)
            self._grp.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self._capi.lib().bppp_group_size(self._grp))

    def verify_batch(self, commitments, proofs, label: bytes):
        """Host buffers -> (accept[n] u8, status[n] i32, global reject count)."""
        import ctypes as C
        import numpy as np
        commitments = np.ascontiguousarray(commitments, dtype=np.uint8).reshape(-1, 64)
        n = commitments.shape[0]
        proofs = np.ascontiguousarray(proofs, dtype=np.uint8).reshape(n, 928)
        accept, status, rej = np.zeros(n, np.uint8), np.zeros(n, np.int32), C.c_int32(0)
        self._capi.check(self._capi.lib().bppp_u64_verify_batch_sharded(self._grp, label, len(label), n, commitments.ctypes.data,
                                                                        proofs.ctypes.data, accept.ctypes.data, status.ctypes.data,
                                                                        C.byref(rej)))
        return accept, status, int(rej.value)

    def verify_batch_device(self, label: bytes, n: int, d_commitments, d_proofs, d_accept, d_status, d_reject_count) -> None:
        """Per-device lists of raw device addresses (rank r: its shard of `shard_range(n, r, G)` on device r); blocks until done."""
        import ctypes as C
        G = len(self.devices)
        mk = lambda xs: (C.c_void_p * G)(*[C.c_void_p(int(x) if x else None) for x in xs])
        self._capi.check(self._capi.lib().bppp_u64_verify_batch_sharded_device(self._grp, label, len(label), n, mk(d_commitments), mk(d_proofs),
                                                                               mk(d_accept), mk(d_status), mk(d_reject_count)))
