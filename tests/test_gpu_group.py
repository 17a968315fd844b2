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


def test_one_device_group_serves_every_sharded_form(batch, monkeypatch):
    """RLC, SEC1 and caller-transcript verifies through the group (host and device buffers) equal the single-context entry points;
    the RCCL accept-reduce really runs (one-rank communicator)."""
    import torch
    import workload
    from bp_pp_amd import U64RangeProofProtocol, wire
    from bp_pp_amd.distributed import U64RangeProofGroup
    from bp_pp_amd.transcript import Transcript
    gens, V, P, expect = batch
    g, gv, hv = workload.split_generators(gens)
    monkeypatch.setenv("BPPP_FORCE_RCCL", "1")
    grp = U64RangeProofGroup(g, gv, hv, [0], fb_window_bits=8)
    single = grp.protocol(0)                      # the same context, through the single-device entry points
    n = V.shape[0]
    n_bad = int((expect == 0).sum())
    seed = bytes(range(32))
    try:
        # RLC mode
        acc, st, rej = grp.verify_batch(V, P, workload.LABEL, rlc_seed=seed)
        a1, s1 = single.verify_batch_rlc(V, P, workload.LABEL, seed)
        assert (acc == a1).all() and (st == s1).all() and (acc == expect).all() and rej == n_bad
        dV, dP = torch.from_numpy(V).cuda(), torch.from_numpy(P).cuda()
        dA = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dS = torch.zeros(n, dtype=torch.int32, device="cuda")
        dR = torch.full((1,), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        grp.verify_batch_device(workload.LABEL, n, [dV.data_ptr()], [dP.data_ptr()], [dA.data_ptr()], [dS.data_ptr()], [dR.data_ptr()], rlc_seed=seed)
        assert (dA.cpu().numpy() == expect).all() and int(dR.item()) == n_bad
        # SEC1-compressed inputs
        V33 = np.frombuffer(b"".join(wire.compress_point(bytes(v)) for v in V), np.uint8).reshape(n, 33)
        P525 = np.zeros((n, 525), np.uint8)
        ok_rows = [i for i in range(n) if i != 5]                      # row 5 holds an off-curve point: it has no SEC1 form
        for i in ok_rows:
            P525[i] = np.frombuffer(wire.abi_to_sec1(bytes(P[i])), np.uint8)
        P525[5] = P525[4]
        acc, st, rej = grp.verify_batch_sec1(V33, P525, workload.LABEL)
        a1, s1 = single.verify_batch_sec1(V33, P525, workload.LABEL)
        exp5 = expect.copy(); exp5[5] = 0                               # proof 4's bytes under commitment 5
        assert (acc == a1).all() and (st == s1).all() and (acc == exp5).all() and rej == int((exp5 == 0).sum())
        d33, d525 = torch.from_numpy(V33).cuda(), torch.from_numpy(P525).cuda()
        dA.zero_(); dR.fill_(-7)
        torch.cuda.synchronize()
        grp.verify_batch_sec1_device(workload.LABEL, n, [d33.data_ptr()], [d525.data_ptr()], [dA.data_ptr()], [dS.data_ptr()], [dR.data_ptr()])
        assert (dA.cpu().numpy() == exp5).all() and int(dR.item()) == int((exp5 == 0).sum())
        # the caller's transcripts: Transcript::new(label) shared by the batch must reproduce the label entry point, and the
        # per-proof form must give each proof its own advanced state
        t0 = Transcript(workload.LABEL)
        acc, st, out, rej = grp.verify_batch_transcripts(np.frombuffer(t0.state, np.uint8), V, P)
        a1, s1, o1 = single.verify_batch_transcript(V, P, t0)
        assert (acc == a1).all() and (acc == expect).all() and (st == s1).all() and (out == o1).all() and rej == n_bad
        S = np.tile(np.frombuffer(t0.state, np.uint8), (n, 1))
        acc2, st2, out2, rej2 = grp.verify_batch_transcripts(S, V, P)
        assert (acc2 == acc).all() and (out2 == out).all() and rej2 == rej
        dSt_in, dO = torch.from_numpy(S).cuda(), torch.zeros((n, 203), dtype=torch.uint8, device="cuda")
        dA.zero_(); dR.fill_(-7)
        torch.cuda.synchronize()
        grp.verify_batch_transcripts_device(n, [dSt_in.data_ptr()], n, [dV.data_ptr()], [dP.data_ptr()], [dA.data_ptr()], [dS.data_ptr()],
                                            [dR.data_ptr()], [dO.data_ptr()])
        assert (dA.cpu().numpy() == expect).all() and int(dR.item()) == n_bad and (dO.cpu().numpy() == out).all()
    finally:
        single.close()
        grp.close()


@pytest.mark.parametrize("force_rccl", [False, True])
def test_injected_fault_returns_the_error_and_the_group_stays_usable(batch, force_rccl, monkeypatch):
    """A rank that fails before the collective (here: injected; in production an out-of-memory workspace) makes the sharded call
    return that rank's error -- within the test's timeout, i.e. nobody waits in an all-reduce -- and the next call works."""
    import workload
    from bp_pp_amd import BpppError, _capi
    from bp_pp_amd.distributed import U64RangeProofGroup
    gens, V, P, expect = batch
    g, gv, hv = workload.split_generators(gens)
    if force_rccl:
        monkeypatch.setenv("BPPP_FORCE_RCCL", "1")
    else:
        monkeypatch.delenv("BPPP_FORCE_RCCL", raising=False)
    grp = U64RangeProofGroup(g, gv, hv, [0], fb_window_bits=8)
    try:
        grp.set_option("inject_fault_rank", 0)
        with pytest.raises(BpppError) as e:
            grp.verify_batch(V, P, workload.LABEL)
        assert e.value.code == _capi.ERR_NOMEM and "injected" in str(e.value)
        acc, st, rej = grp.verify_batch(V, P, workload.LABEL)            # the fault was one-shot; the group is intact
        assert (acc == expect).all() and rej == int((expect == 0).sum())
        with pytest.raises(BpppError):
            grp.set_option("inject_fault_rank", 1)                       # no such rank
        grp.set_option("rlc_superchunk", 256)                            # context options reach every rank
        acc, st, rej = grp.verify_batch(V, P, workload.LABEL, rlc_seed=bytes(32))
        assert (acc == expect).all()
    finally:
        grp.close()


def test_one_device_group_proves_like_the_single_context():
    """bppp_u64_prove_batch_sharded[_device] (u64_proof.rs:57-82): byte-identical to the single-context prover and to the oracle
    prover on a sample, host and device forms, ragged sizes; an injected fault returns its error and the next call works."""
    import torch
    import bppp_oracle_c as OC
    import workload
    from bp_pp_amd import BpppError, U64RangeProofProtocol, _capi
    from bp_pp_amd.distributed import U64RangeProofGroup
    gens = workload.generators()
    g, gv, hv = workload.split_generators(gens)
    n = 777
    x = np.ascontiguousarray(workload.values(n, first=5100))
    s, rnd = np.ascontiguousarray(workload.blindings(n, first=5100)), np.ascontiguousarray(workload.prover_randomness(n, first=5100))
    grp = U64RangeProofGroup(g, gv, hv, [0], fb_window_bits=8)
    single = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
    try:
        P, V, st = grp.prove_batch(x, s, rnd, workload.LABEL)
        P1, V1, st1 = single.prove_batch(x, s, rnd, workload.LABEL)
        assert (P == P1).all() and (V == V1).all() and not st.any() and not st1.any()
        op, ov = OC.u64_prove_batch(gens, workload.LABEL, x[:24], s[:24], rnd[:24], nthreads=4)      # the checker
        assert (P[:24] == op).all() and (V[:24] == ov).all()
        acc, _ = single.verify_batch(V, P, workload.LABEL)
        assert acc.all()
        for m in (0, 1, 65):
            Pm, Vm, _ = grp.prove_batch(x[:m], s[:m], rnd[:m], workload.LABEL)
            assert (Pm == P[:m]).all() and (Vm == V[:m]).all()
        dx, ds, dr = torch.from_numpy(x.view(np.int64)).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(rnd).cuda()
        dP = torch.zeros((n, 928), dtype=torch.uint8, device="cuda")
        dV = torch.zeros((n, 64), dtype=torch.uint8, device="cuda")
        dS = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        grp.prove_batch_device(workload.LABEL, n, [dx.data_ptr()], [ds.data_ptr()], [dr.data_ptr()], [dP.data_ptr()], [dV.data_ptr()],
                               [dS.data_ptr()])
        assert (dP.cpu().numpy() == P).all() and (dV.cpu().numpy() == V).all() and not dS.cpu().numpy().any()
        dP.zero_()
        grp.prove_batch_device(workload.LABEL, n, [dx.data_ptr()], [ds.data_ptr()], [dr.data_ptr()], [dP.data_ptr()], [dV.data_ptr()])  # no status
        assert (dP.cpu().numpy() == P).all()
        with pytest.raises(BpppError):
            grp.prove_batch_device(workload.LABEL, n, [dx.data_ptr()], [0], [dr.data_ptr()], [dP.data_ptr()], [dV.data_ptr()])      # a missing shard
        grp.set_option("inject_fault_rank", 0)
        with pytest.raises(BpppError) as e:
            grp.prove_batch(x, s, rnd, workload.LABEL)
        assert e.value.code == _capi.ERR_NOMEM
        P2, _, _ = grp.prove_batch(x, s, rnd, workload.LABEL)
        assert (P2 == P).all()
    finally:
        grp.close()
        single.close()


@pytest.mark.parametrize("force_rccl", [False, True])
def test_reciprocal_group_equals_single_context(force_rccl, monkeypatch):
    """bppp_wnla_group_create + bppp_reciprocal_verify_batch[_rlc]_sharded[_device] (BASELINE configs[4]'s path) on one device: equal
    to the single-context verifier and to the oracle, exact and RLC, host and device buffers, ragged and empty batches."""
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import recip_cases
    from bp_pp_amd.distributed import ReciprocalRangeProofGroup
    if force_rccl:
        monkeypatch.setenv("BPPP_FORCE_RCCL", "1")
    else:
        monkeypatch.delenv("BPPP_FORCE_RCCL", raising=False)
    nd, npp, B = 32, 16, 21
    case = recip_cases.make(nd, npp, B)
    grp = ReciprocalRangeProofGroup(nd, npp, case["g"], case["gv"], case["hv"], case["gv_"], case["hv_"], [0], fb_window_bits=8)
    single = grp.protocol(0)
    try:
        shape = (case["rounds"], case["nl"], case["nn"])
        P, com = case["proofs"].copy(), case["commitments"].copy()
        P[0, -1] ^= 1
        P[7, 256 + 64 * case["rounds"] + 5] ^= 0x40                      # a round point leaves the curve
        com[12] = case["commitments"][13]
        exp = np.array([1 if recip_cases.oracle_verify(case, bytes(com[b]), bytes(P[b])) == 1 else 0 for b in range(B)], np.uint8)
        assert exp.sum() == B - 3
        acc, st, rej = grp.verify_batch(case["label"], com, P, *shape)
        a1, s1 = single.verify_batch(case["label"], com, P, *shape)
        assert (acc == a1).all() and (st == s1).all() and (acc == exp).all() and rej == 3 and st[7] == 1
        acc, st, rej = grp.verify_batch(case["label"], com, P, *shape, rlc_seed=bytes(range(32)))
        assert (acc == exp).all() and (st == s1).all() and rej == 3
        for m in (0, 1, 9):
            a, s, r = grp.verify_batch(case["label"], com[:m], P[:m], *shape)
            assert a.tolist() == exp[:m].tolist() and r == int((exp[:m] == 0).sum())
        dV, dP = torch.from_numpy(com).cuda(), torch.from_numpy(P).cuda()
        dA = torch.zeros(B, dtype=torch.uint8, device="cuda")
        dS = torch.zeros(B, dtype=torch.int32, device="cuda")
        dR = torch.full((1,), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for seed in (None, bytes(32)):
            dA.zero_(); dR.fill_(-7)
            torch.cuda.synchronize()
            grp.verify_batch_device(case["label"], B, [dV.data_ptr()], [dP.data_ptr()], *shape, [dA.data_ptr()], [dS.data_ptr()], [dR.data_ptr()],
                                    rlc_seed=seed)
            assert (dA.cpu().numpy() == exp).all() and int(dR.item()) == 3 and (dS.cpu().numpy() == s1).all()
        grp.set_option("inject_fault_rank", 0)
        with pytest.raises(Exception):
            grp.verify_batch(case["label"], com, P, *shape)
        acc, _, rej = grp.verify_batch(case["label"], com, P, *shape)
        assert (acc == exp).all() and rej == 3
    finally:
        single.close()
        grp.close()


def test_damaged_table_file_is_refused(batch, tmp_path):
    """The table artefact carries a checksum over generators and table body: a flipped byte anywhere, a truncated file or a generator
    moved off the curve is refused at load time instead of yielding a verifier with wrong bases."""
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    gens, V, P, expect = batch
    g, gv, hv = workload.split_generators(gens)
    a = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
    path = str(tmp_path / "t.bin")
    try:
        a.save_tables(path)
        good = open(path, "rb").read()
        b = U64RangeProofProtocol.from_tables(path, device=0)
        acc, _ = b.verify_batch(V, P, workload.LABEL)
        b.close()
        assert (acc == expect).all()
        hdr = 56                                                        # magic, 6 x u32, 3 x u64
        for off in (hdr + 3, hdr + 49 * 80 + 12345, len(good) - 1):    # a generator limb, the table body, the last byte
            bad = bytearray(good)
            bad[off] ^= 0x04
            open(path, "wb").write(bad)
            with pytest.raises(Exception):
                U64RangeProofProtocol.from_tables(path, device=0)
        open(path, "wb").write(good[:-64])
        with pytest.raises(Exception):
            U64RangeProofProtocol.from_tables(path, device=0)
    finally:
        a.close()


def test_concurrent_callers_on_shared_contexts(batch):
    """Several host threads, each with its own context over ONE set of tables (bppp_ctx_create_shared), verifying and proving small
    batches at the same time -- the way a service with a thread per request would call the library (the reference's verify is a
    plain function: callers parallelise it freely).  Every call's result must equal the single-threaded one; a context is also hit
    from two threads at once (its lock serialises them)."""
    import threading
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    gens, V, P, expect = batch
    g, gv, hv = workload.split_generators(gens)
    base = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
    clones = [base.clone_shared() for _ in range(3)]
    ctxs = [base] + clones
    n = V.shape[0]
    x, s, rnd = workload.values(64, 4200), workload.blindings(64, 4200), workload.prover_randomness(64, 4200)
    ref_p, ref_v, _ = base.prove_batch(x, s, rnd, workload.LABEL)
    errors = []

    def worker(tid):
        try:
            ctx = ctxs[tid % len(ctxs)]                      # threads 4, 5 share contexts 0, 1 with threads 0, 1
            rng = np.random.default_rng(tid)
            for it in range(25):
                lo = int(rng.integers(0, n - 1))
                m = int(rng.integers(1, min(64, n - lo) + 1))
                acc, st = ctx.verify_batch(V[lo:lo + m], P[lo:lo + m], workload.LABEL)
                if acc.tolist() != expect[lo:lo + m].tolist():
                    errors.append((tid, it, "verify"))
                if it % 5 == 0:
                    k = int(rng.integers(1, 17))
                    pp, vv, pst = ctx.prove_batch(x[:k], s[:k], rnd[:k], workload.LABEL)
                    if pst.any() or not (pp == ref_p[:k]).all() or not (vv == ref_v[:k]).all():
                        errors.append((tid, it, "prove"))
        except Exception as e:                               # noqa: BLE001 -- reported below
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    try:
        assert not any(t.is_alive() for t in threads), "a caller is stuck"
        assert not errors, errors[:5]
    finally:
        for c in clones:
            c.close()
        base.close()
