"""CPU tier of the device group (csrc/group_core.h -- the control flow libbppp_hip.so's bppp_group runs over HIP streams and RCCL):
the same `run_sharded` template over emulated devices (host threads running the product's device code on their shard) and an
in-process all-reduce that, like ncclAllReduce, does not return until every rank has entered it.

What is pinned here:
  * the sharded result (accept bits, statuses, every rank's all-reduced reject count) equals the unsharded one and the oracle's,
    for the u64 proofs and for the generic reciprocal verifier (reciprocal.rs:98-107, BASELINE configs[4]'s path);
  * FAIL CLOSED: a rank that fails before the collective makes the call return that rank's error -- quickly, with no rank left in
    the all-reduce -- where round 2's control flow (use_vote = 0) hangs;
  * a failing collective aborts the communicator and the call returns BPPP_ERR_RCCL instead of hanging."""
import os
import socket
import time

import numpy as np
import pytest

import recip_cases
import workload
from emul.build import load

HANG, ERR_NOMEM, ERR_RCCL = -100, -5, -6


@pytest.fixture(scope="module")
def u64_case():
    L = load()
    n = 11
    gens, V, P, _ = workload.make_batch(n, first=900, nthreads=1)
    P, expect = workload.corrupt(P, V, every=4)
    W = 4
    tab = np.zeros(L.emul_fb_table_entries(49, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 49, W, tab.ctypes.data) == 0
    return L, gens, V, P, expect, tab, W


def _group_u64(case, G, fail_rank=-1, fail_coll=-1, vote=1, timeout_ms=60000, n=None):
    L, gens, V, P, expect, tab, W = case
    n = V.shape[0] if n is None else n
    acc, st = np.zeros(n, np.uint8), np.zeros(n, np.int32)
    rej, aborted = np.full(G, -1, np.int32), np.zeros(1, np.int32)
    t0 = time.time()
    rc = L.emul_group_verify(0, G, fail_rank, fail_coll, vote, timeout_ms, tab.ctypes.data, W, 16, 32, 0, 0, workload.LABEL, len(workload.LABEL),
                             n, V.ctypes.data, P.ctypes.data, 4, 2, 1, acc.ctypes.data, st.ctypes.data, rej.ctypes.data, aborted.ctypes.data)
    return rc, acc, st, rej, int(aborted[0]), time.time() - t0


@pytest.mark.parametrize("G", [1, 2, 3])
def test_sharded_u64_equals_unsharded_and_oracle(u64_case, G):
    import bppp_oracle_c as OC
    L, gens, V, P, expect, tab, W = u64_case
    rc, acc, st, rej, aborted, _ = _group_u64(u64_case, G)
    assert rc == 0 and not aborted
    oacc, ost = OC.u64_verify_batch(gens, workload.LABEL, V, P, nthreads=2)          # the checker
    assert (acc == oacc).all() and (acc == expect).all() and (st == ost).all()
    assert rej.tolist() == [int((expect == 0).sum())] * G                            # every rank holds the GLOBAL count


def test_more_ranks_than_proofs(u64_case):
    rc, acc, st, rej, aborted, _ = _group_u64(u64_case, 3, n=2)                        # one rank's shard is empty
    assert rc == 0 and acc.tolist() == u64_case[4][:2].tolist() and len(set(rej.tolist())) == 1


def test_failing_rank_fails_closed(u64_case):
    """One rank fails before it enqueues anything (an out-of-memory workspace, say).  With the vote nobody enters the all-reduce:
    the call returns that rank's code at once.  Without it (round 2's flow) the healthy rank waits for a peer that never comes."""
    rc, _, _, _, aborted, dt = _group_u64(u64_case, 2, fail_rank=1, timeout_ms=60000)
    assert rc == ERR_NOMEM and not aborted and dt < 30          # far below the all-reduce's 60 s timeout: it was never entered
    rc, _, _, _, aborted, dt = _group_u64(u64_case, 3, fail_rank=0, timeout_ms=60000)
    assert rc == ERR_NOMEM and not aborted and dt < 30
    rc, _, _, _, _, _ = _group_u64(u64_case, 2, fail_rank=1, vote=0, timeout_ms=1500)
    assert rc == HANG                                           # what the vote prevents
    # and the group is still usable afterwards (nothing was left half-entered): a clean call right after a failed one
    rc, acc, _, rej, _, _ = _group_u64(u64_case, 2)
    assert rc == 0 and (acc == u64_case[4]).all()


def test_throwing_rank_still_votes(u64_case):
    """A rank whose prepare phase throws (std::bad_alloc in its host-side bookkeeping) must not die before the vote -- the others
    would wait for it for ever: the exception becomes that rank's return code and the call fails closed like any other failure."""
    rc, _, _, _, aborted, dt = _group_u64(u64_case, 3, fail_rank=101, timeout_ms=60000)
    assert rc == ERR_NOMEM and not aborted and dt < 30
    rc, acc, _, _, _, _ = _group_u64(u64_case, 3)
    assert rc == 0 and (acc == u64_case[4]).all()


@pytest.mark.parametrize("G,first_missing", [(2, 1), (4, 3), (4, 1), (8, 2), (4, 0)])
def test_ranks_whose_threads_cannot_start_fail_the_call_instead_of_hanging(G, first_missing):
    """group_core.h: when rank threads cannot be started, the main thread votes for ALL the missing ranks without blocking (round 4
    blocked in the first missing rank's vote, which only the same thread could have completed: two or more missing ranks hung the
    group with every lock held)."""
    import ctypes as C
    from emul.build import load
    L = load()
    ran = C.c_int(-1)
    done = []

    def call():
        done.append(L.emul_group_missing_ranks(G, first_missing, C.byref(ran)))
    import threading
    th = threading.Thread(target=call, daemon=True)
    th.start()
    th.join(20)
    assert not th.is_alive(), "run_sharded did not return"
    assert done == [-5] and ran.value == first_missing


def test_failing_collective_aborts_instead_of_hanging(u64_case):
    rc, _, _, _, aborted, dt = _group_u64(u64_case, 3, fail_coll=1, timeout_ms=60000)
    assert rc == ERR_RCCL and aborted == 1 and dt < 30


@pytest.mark.parametrize("G", [1, 2, 3])
def test_sharded_prove_equals_unsharded_and_oracle(u64_case, G):
    """bppp_u64_prove_batch_sharded's flow (u64_proof.rs:57-82 per proof, no exchange step): every proof and commitment of the
    sharded batch equals the unsharded device code's and the oracle prover's bytes; more ranks than values leaves a shard empty."""
    import bppp_oracle_c as OC
    L, gens, _, _, _, tab, W = u64_case
    for n in (7, 2):
        x = np.ascontiguousarray(workload.values(n, first=40))
        s, rnd = np.ascontiguousarray(workload.blindings(n, first=40)), np.ascontiguousarray(workload.prover_randomness(n, first=40))
        P, V, st = np.zeros((n, 928), np.uint8), np.zeros((n, 64), np.uint8), np.full(n, -1, np.int32)
        rc = L.emul_group_prove(G, -1, tab.ctypes.data, W, workload.LABEL, len(workload.LABEL), n, x.ctypes.data, s.ctypes.data, rnd.ctypes.data,
                                P.ctypes.data, V.ctypes.data, st.ctypes.data)
        assert rc == 0 and not st.any()
        op, ov = OC.u64_prove_batch(gens, workload.LABEL, x, s, rnd, nthreads=2)        # the checker
        assert (P == op).all() and (V == ov).all()


def test_sharded_prove_fails_closed(u64_case):
    L, gens, _, _, _, tab, W = u64_case
    n = 5
    x = np.ascontiguousarray(workload.values(n, first=40))
    s, rnd = np.ascontiguousarray(workload.blindings(n, first=40)), np.ascontiguousarray(workload.prover_randomness(n, first=40))
    P, V, st = np.zeros((n, 928), np.uint8), np.zeros((n, 64), np.uint8), np.zeros(n, np.int32)
    t0 = time.time()
    rc = L.emul_group_prove(2, 1, tab.ctypes.data, W, workload.LABEL, len(workload.LABEL), n, x.ctypes.data, s.ctypes.data, rnd.ctypes.data,
                            P.ctypes.data, V.ctypes.data, st.ctypes.data)
    assert rc == ERR_NOMEM and time.time() - t0 < 30


@pytest.fixture(scope="module")
def recip_case():
    L = load()
    nd, npp, B = 8, 4, 5
    case = recip_cases.make(nd, npp, B=B)
    W = 4
    gens = case["g"] + b"".join(case["gv"]) + b"".join(case["gv_"]) + b"".join(case["hv"]) + b"".join(case["hv_"])
    NB = 1 + case["NG"] + case["NH"]
    tab = np.zeros(L.emul_fb_table_entries(NB, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, NB, W, tab.ctypes.data) == 0
    P = case["proofs"].copy()
    P[1, -1] ^= 1               # wrong final scalar
    P[3, 70] ^= 1               # c_r off the curve: status flag
    return L, case, tab, W, np.ascontiguousarray(case["commitments"]), P


def _recip_rows(rc_case, lo, hi):
    """The reciprocal verifier's device code (host build) on rows [lo, hi)."""
    L, case, tab, W, com, P = rc_case
    m = hi - lo
    acc, st = np.zeros(m, np.uint8), np.zeros(m, np.int32)
    c, p = np.ascontiguousarray(com[lo:hi]), np.ascontiguousarray(P[lo:hi])
    if m:
        L.emul_recip_verify(tab.ctypes.data, W, case["NG"], case["NH"], case["nd"], case["np"], case["label"], len(case["label"]), m,
                            c.ctypes.data, p.ctypes.data, case["rounds"], case["nl"], case["nn"], acc.ctypes.data, st.ctypes.data)
    return acc, st


@pytest.mark.parametrize("G", [2, 3])
def test_sharded_reciprocal_equals_unsharded_and_oracle(recip_case, G):
    L, case, tab, W, com, P = recip_case
    n = com.shape[0]
    acc, st = np.zeros(n, np.uint8), np.zeros(n, np.int32)
    rej, aborted = np.full(G, -1, np.int32), np.zeros(1, np.int32)
    rc = L.emul_group_verify(1, G, -1, -1, 1, 60000, tab.ctypes.data, W, case["NG"], case["NH"], case["nd"], case["np"], case["label"],
                             len(case["label"]), n, com.ctypes.data, P.ctypes.data, case["rounds"], case["nl"], case["nn"], acc.ctypes.data,
                             st.ctypes.data, rej.ctypes.data, aborted.ctypes.data)
    assert rc == 0
    acc1, st1 = _recip_rows(recip_case, 0, n)
    exp = [1 if recip_cases.oracle_verify(case, bytes(com[b]), bytes(P[b])) == 1 else 0 for b in range(n)]
    assert acc.tolist() == acc1.tolist() == exp == [1, 0, 1, 0, 1] and st.tolist() == st1.tolist() == [0, 0, 0, 1, 0]
    assert rej.tolist() == [2] * G
    # a failing rank fails closed on this path too
    rc = L.emul_group_verify(1, G, G - 1, -1, 1, 60000, tab.ctypes.data, W, case["NG"], case["NH"], case["nd"], case["np"], case["label"],
                             len(case["label"]), n, com.ctypes.data, P.ctypes.data, case["rounds"], case["nl"], case["nn"], acc.ctypes.data,
                             st.ctypes.data, rej.ctypes.data, aborted.ctypes.data)
    assert rc == ERR_NOMEM


def _recip_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from bp_pp_amd.distributed import all_reduce_reject_count, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = load()
    nd, npp, B = 8, 4, 5
    case = recip_cases.make(nd, npp, B=B)
    W = 4
    gens = case["g"] + b"".join(case["gv"]) + b"".join(case["gv_"]) + b"".join(case["hv"]) + b"".join(case["hv_"])
    NB = 1 + case["NG"] + case["NH"]
    tab = np.zeros(L.emul_fb_table_entries(NB, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, NB, W, tab.ctypes.data) == 0
    P = case["proofs"].copy()
    P[1, -1] ^= 1
    P[3, 70] ^= 1
    lo, hi = shard_range(B, rank, world)
    acc, st = _recip_rows((L, case, tab, W, np.ascontiguousarray(case["commitments"]), P), lo, hi)
    cnt = torch.tensor([int((acc == 0).sum())], dtype=torch.int32)
    all_reduce_reject_count(cnt)                                   # what bench.py --workload recip256 --gpus N does over RCCL
    q.put((rank, lo, hi, acc.tolist(), st.tolist(), int(cnt.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_reciprocal_shards():
    """The one-process-per-GPU form of configs[4] (bench.py --workload recip256 --gpus N): world 2 over gloo, each rank the reciprocal
    verifier's device code (host build) on its shard_range, one all-reduce of the reject count."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_recip_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    acc, st = [None] * 5, [None] * 5
    for rank, lo, hi, a, s_, total in res:
        acc[lo:hi], st[lo:hi] = a, s_
        assert total == 2
    assert acc == [1, 0, 1, 0, 1] and st == [0, 0, 0, 1, 0]
