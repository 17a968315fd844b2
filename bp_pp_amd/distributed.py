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
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(count_tensor, op=dist.ReduceOp.SUM)
    return count_tensor
