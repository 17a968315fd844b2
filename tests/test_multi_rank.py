"""N > 1 path on CPU: world_size 2 over gloo.  Each rank takes its shard of one batch (shard_range) and verifies it with the
PRODUCT's device code compiled for the host (tests/emul: the same verify_core.h functions the HIP kernels run, thread by thread --
the CPU tier has no GPU), the reject counts are all-reduced exactly as bench.py does over RCCL, and the union of the shards must
reproduce the oracle's single-process verdicts."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bp_pp_amd.distributed import all_reduce_reject_count, shard_range


def test_shard_range_partitions_exactly():
    for n in [0, 1, 7, 64, 65, 1 << 16, (1 << 20) + 3]:
        for world in [1, 2, 3, 8]:
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank, world, port, n, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bppp_oracle_c as OC
    import workload
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gens, V, P, _ = workload.make_batch(n, first=300, nthreads=1)
    P, expect = workload.corrupt(P, V, every=5)
    lo, hi = shard_range(n, rank, world)
    from emul.build import load
    L = load()
    W = 4
    tab = np.zeros(L.emul_fb_table_entries(49, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 49, W, tab.ctypes.data) == 0
    m = hi - lo
    Vs, Ps = V[lo:hi].copy(), P[lo:hi].copy()
    acc, st = np.zeros(m, np.uint8), np.zeros(m, np.int32)
    if m:
        L.emul_u64_verify_batch(tab.ctypes.data, W, workload.LABEL, len(workload.LABEL), m, Vs.ctypes.data, Ps.ctypes.data, acc.ctypes.data,
                                st.ctypes.data, None)
    oacc, _ = OC.u64_verify_batch(gens, workload.LABEL, Vs, Ps, nthreads=1)     # the checker
    assert (acc == oacc).all()
    cnt = torch.tensor([int((acc == 0).sum())], dtype=torch.int32)
    all_reduce_reject_count(cnt)
    q.put((rank, lo, hi, acc.tolist(), int(cnt.item()), expect.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_reject_count():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n, world = 11, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = res[0][5]
    merged = [None] * n
    for rank, lo, hi, acc, total, _ in res:
        merged[lo:hi] = acc
        assert total == sum(1 for e in expect if e == 0)      # every rank sees the global reject count
    assert merged == expect


def test_c_abi_shard_range_equals_the_python_split():
    """bppp_shard_range (what a Rust / C caller of the sharded entry points uses) == bp_pp_amd.distributed.shard_range, covers
    [0, n) exactly, for ragged sizes too.  No GPU needed: the function is pure host code of libbppp_hip.so."""
    import ctypes as C
    from bp_pp_amd import _capi
    from bp_pp_amd.distributed import shard_range
    L = _capi.lib()
    for n in (0, 1, 7, 8, 9, 1000, 65536, (1 << 20), (1 << 20) + 5, (1 << 62) + 3):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(world):
                lo, hi = C.c_size_t(), C.c_size_t()
                L.bppp_shard_range(n, r, world, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == shard_range(n, r, world) and lo.value == prev
                prev = hi.value
            assert prev == n
