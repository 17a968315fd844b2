"""GPU tests of the node-level sharded entry points (include/bppp.h: bppp_group_*, bppp_u64_verify_batch_sharded[_device]).
The GPU tier has ONE device, so the group has one rank: its results must equal the single-context entry point's, with and
without the RCCL accept-reduce (BPPP_FORCE_RCCL=1 routes a one-device group through a one-rank communicator, so that
dlopen(librccl), ncclCommInitAll and ncclAllReduce on the verify stream really run)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def batch():
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import workload
    n = 333
    gens, V, P, _ = workload.make_batch(n, first=77000)
    P, expect = workload.corrupt(P, V, every=10)
    P = P.copy()
    P[5, 3] ^= 0x80                       # c_l leaves the curve: status flag, counted as a reject
    expect = expect.copy(); expect[5] = 0
    return gens, V, P, expect


@pytest.mark.parametrize("force_rccl", [False, True])
def test_one_device_group_equals_single_context(batch, force_rccl, monkeypatch):
    import torch
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    from bp_pp_amd.distributed import U64RangeProofGroup
    gens, V, P, expect = batch
    g, gv, hv = workload.split_generators(gens)
    if force_rccl:
        monkeypatch.setenv("BPPP_FORCE_RCCL", "1")
    else:
        monkeypatch.delenv("BPPP_FORCE_RCCL", raising=False)
    grp = U64RangeProofGroup(g, gv, hv, [0], fb_window_bits=8)
    single = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
    try:
        assert len(grp) == 1
        acc, st, rej = grp.verify_batch(V, P, workload.LABEL)
        acc1, st1 = single.verify_batch(V, P, workload.LABEL)
        assert (acc == acc1).all() and (st == st1).all() and (acc == expect).all()
        assert rej == int((expect == 0).sum()) and st[5] == 1
        # ragged and empty batches
        for m in (0, 1, 65):
            a, s, r = grp.verify_batch(V[:m], P[:m], workload.LABEL)
            assert a.tolist() == expect[:m].tolist() and r == int((expect[:m] == 0).sum())
        # shards already resident on their device
        n = V.shape[0]
        dV, dP = torch.from_numpy(V).cuda(), torch.from_numpy(P).cuda()
        dA = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dS = torch.zeros(n, dtype=torch.int32, device="cuda")
        dR = torch.full((1,), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        grp.verify_batch_device(workload.LABEL, n, [dV.data_ptr()], [dP.data_ptr()], [dA.data_ptr()], [dS.data_ptr()], [dR.data_ptr()])
        assert (dA.cpu().numpy() == expect).all() and int(dR.item()) == rej and (dS.cpu().numpy() == st).all()
    finally:
        grp.close()
        single.close()


def test_group_rejects_bad_arguments():
    import ctypes as C
    import workload
    from bp_pp_amd import BpppError
    from bp_pp_amd.distributed import U64RangeProofGroup
    g, gv, hv = workload.split_generators(workload.generators())
    with pytest.raises(BpppError):
        U64RangeProofGroup(g, gv, hv, [0, 0], fb_window_bits=8)           # the same device twice
    with pytest.raises(BpppError):
        U64RangeProofGroup(g, gv, hv, [99], fb_window_bits=8)             # no such device


def test_table_file_round_trip_and_shared_tables(batch, tmp_path):
    """the fixed-base tables as an artefact (save -> context from the file) and one table set shared by several
    contexts on the same GPU (two host threads verifying concurrently)."""
    import threading
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    gens, V, P, expect = batch
    g, gv, hv = workload.split_generators(gens)
    a = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=10)
    path = str(tmp_path / "tables.bin")
    b = c = None
    try:
        ref_acc, ref_st = a.verify_batch(V, P, workload.LABEL)
        assert (ref_acc == expect).all()
        a.save_tables(path)
        b = U64RangeProofProtocol.from_tables(path, device=0)
        acc, st = b.verify_batch(V, P, workload.LABEL)
        assert (acc == ref_acc).all() and (st == ref_st).all()
        x = np.array([0, 5, 2**64 - 1], dtype=np.uint64)
        s = np.frombuffer(bytes(range(96)), dtype=np.uint8).reshape(3, 32) & 0x7F
        assert (a.commit_value_batch(x, s) == b.commit_value_batch(x, s)).all()
        with open(path, "r+b") as f:                       # a damaged header is refused
            f.write(b"NOTATABLE")
        with pytest.raises(Exception):
            U64RangeProofProtocol.from_tables(path, device=0)
        c = a.clone_shared()
        before = a.device_bytes()
        out = {}

        def run(name, proto):
            for _ in range(3):
                out[name] = proto.verify_batch(V, P, workload.LABEL)

        ts = [threading.Thread(target=run, args=("a", a)), threading.Thread(target=run, args=("c", c))]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert (out["a"][0] == ref_acc).all() and (out["c"][0] == ref_acc).all() and (out["c"][1] == ref_st).all()
        assert c.device_bytes() < before                    # the clone holds workspaces only, not a second table set
    finally:
        for p in (c, b, a):
            if p is not None:
                p.close()
