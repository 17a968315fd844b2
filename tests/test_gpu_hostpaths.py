"""GPU tier: the host side around the kernels -- allocation failures on every host-buffer path (honest BPPP_ERR_NOMEM, context still
usable), the choice of the fixed-base window width from the HBM that is free, and device buffers that are NOT 16-byte aligned (the
byte-wise load / store path of csrc/field.h: be32_to_limbs / limbs_to_be32)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ERR_NOMEM = -5


@pytest.fixture(scope="module")
def base():
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    g, gv, hv = workload.split_generators(workload.generators())
    p = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
    yield p
    p.close()


@pytest.fixture(scope="module")
def batch():
    import workload
    gens, V, P, _ = workload.make_batch(40, first=4100)
    P, expect = workload.corrupt(P, V, every=8)
    return gens, V, P, expect


def _expect_nomem_then_ok(ctx, call, check, max_k=12):
    """Fail the k-th device allocation of `call` for k = 1, 2, ...: every time the call must return BPPP_ERR_NOMEM (never BPPP_ERR_HIP,
    never a crash) and the SAME context must then serve the same call correctly.  Stops at the first k the call no longer reaches
    (all its allocations exist by then).  Returns how many allocation sites were walked."""
    from bp_pp_amd._capi import BpppError
    walked = 0
    for k in range(1, max_k + 1):
        c = ctx.clone_shared()                 # fresh workspaces: every buffer of the call is still to be allocated
        try:
            c.set_option("inject_alloc_fault", k)
            try:
                out = call(c)
            except BpppError as e:
                assert e.code == ERR_NOMEM, (k, e.code, str(e))
                walked += 1
                check(call(c))                 # the context is still usable, and right
                continue
            check(out)                         # k is beyond the call's last allocation: it simply succeeded
            c.set_option("inject_alloc_fault", 0)
            break
        finally:
            c.close()
    return walked


def test_every_u64_host_path_reports_nomem_and_recovers(base, batch, oracle_c):
    import workload
    from bp_pp_amd import wire
    from bp_pp_amd.transcript import Transcript
    gens, V, P, expect = batch
    n = V.shape[0]

    def chk_verify(out):
        assert (out[0] == expect).all() and not out[1].any()

    assert _expect_nomem_then_ok(base, lambda c: c.verify_batch(V, P, workload.LABEL), chk_verify) >= 3
    assert _expect_nomem_then_ok(base, lambda c: c.verify_batch_transcript(V, P, Transcript(workload.LABEL)), chk_verify) >= 3
    assert _expect_nomem_then_ok(base, lambda c: c.verify_batch_rlc(V, P, workload.LABEL, b"\x07" * 32), chk_verify) >= 4
    u8 = lambda blobs, w: np.frombuffer(b"".join(blobs), np.uint8).reshape(-1, w).copy()
    V33 = u8([wire.compress_point(bytes(v)) for v in V], 33)
    P525 = u8([wire.abi_to_sec1(bytes(p)) for p in P], 525)
    assert _expect_nomem_then_ok(base, lambda c: c.verify_batch_sec1(V33, P525, workload.LABEL), chk_verify) >= 4

    x, s, rnd = workload.values(n, first=50), workload.blindings(n, first=50), workload.prover_randomness(n, first=50)
    op, ov = oracle_c.u64_prove_batch(gens, workload.LABEL, x, s, rnd, nthreads=4)

    def chk_prove(out):
        assert (out[0] == op).all() and (out[1] == ov).all() and not out[2].any()

    assert _expect_nomem_then_ok(base, lambda c: c.prove_batch(x, s, rnd, workload.LABEL), chk_prove) >= 4
    assert _expect_nomem_then_ok(base, lambda c: c.prove_batch_transcript(x, s, rnd, Transcript(workload.LABEL)), chk_prove) >= 4

    def chk_prove_sec1(out):
        assert all(wire.sec1_to_abi(bytes(out[0][i])) == bytes(op[i]) for i in range(n))
        assert all(wire.decompress_point(bytes(out[1][i])) == bytes(ov[i]) for i in range(n))

    assert _expect_nomem_then_ok(base, lambda c: c.prove_batch_sec1(x, s, rnd, workload.LABEL), chk_prove_sec1) >= 5
    com = base.commit_value_batch(x, s)
    assert _expect_nomem_then_ok(base, lambda c: c.commit_value_batch(x, s), lambda out: (out == com).all() or pytest.fail("commit")) >= 2


def test_generic_host_paths_report_nomem_and_recover():
    import wnla_cases
    from bp_pp_amd._capi import BpppError
    from bp_pp_amd.wnla import WeightNormLinearArgument
    case = wnla_cases.make(4, 8, 6)
    w = WeightNormLinearArgument(case["g"], case["gv"], case["hv"], device=0, fb_window_bits=8)
    try:
        args = dict(commitments=case["commitments"], c=case["c"], rho=case["rho"], mu=case["mu"], proof_r=case["proof_r"],
                    proof_x=case["proof_x"], proof_l=case["proof_l"], proof_n=case["proof_n"])
        from bp_pp_amd import _capi
        for k in (1, 2, 3):
            _capi.check(_capi.lib().bppp_ctx_set_option(w._ctx, b"inject_alloc_fault", k))
            try:
                acc, st = w.verify_batch(case["label"], **args)
                assert acc.all()                       # fewer than k allocations left on this path
                _capi.check(_capi.lib().bppp_ctx_set_option(w._ctx, b"inject_alloc_fault", 0))
            except BpppError as e:
                assert e.code == ERR_NOMEM
            acc, st = w.verify_batch(case["label"], **args)
            assert acc.all() and not st.any()
            out, st = w.commit_batch(case["c"], case["mu"], case["l"], case["n"])
            assert (out == case["commitments"]).all()
    finally:
        w.close()


def test_default_window_width_follows_free_memory(monkeypatch):
    """fb_window_bits = 0 takes the FEWEST windows per scalar whose tables fit the free HBM (hipMemGetInfo at creation; BPPP_ASSUME_FREE_GB
    stands in for a device that is already partly taken), the windows in two widths sized to the bit (code Wb + 100 ka): 11 windows
    (523: 5 x 24 + 6 x 23 bits, 210 GB for the 49 generators) on an empty MI355X, then 12 (621, 59 GB), 13 (1119, 20 GB), 14 (618,
    8 GB), 15 (317, 3.7 GB) ... as less is free -- at most 76 % of it, leaving 50 GB or half of it.  In between, for this generator
    shape, two regions (g and g_vec at 11 windows, h_vec at 12: 112 GB).  Verdicts and commitments are the same in every layout."""
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    g, gv, hv = workload.split_generators(workload.generators())
    _, V, P, _ = workload.make_batch(20, first=10)
    P, expect = workload.corrupt(P, V, every=5)
    x = np.array([0, 7, 2**64 - 1], dtype=np.uint64)
    sb = np.frombuffer(bytes(range(96)), dtype=np.uint8).reshape(3, 32) & 0x7F
    ref = {}
    for free_gb, env, want in ((300, None, (523, 0, 0)), (300, "BPPP_NO_WIDE_TABLES", (621, 523, 17)), (200, None, (621, 523, 17)),
                               (200, "BPPP_NO_MIXED_WINDOWS", (621, 0, 0)), (100, None, (1119, 0, 0)), (58, None, (1119, 0, 0)),
                               (30, None, (618, 0, 0)), (14, None, (317, 0, 0)), (1.0, None, (1113, 0, 0)), (0.001, None, (8, 0, 0))):
        monkeypatch.setenv("BPPP_ASSUME_FREE_GB", str(free_gb))
        if env:
            monkeypatch.setenv(env, "1")
        p = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=0)
        if env:
            monkeypatch.delenv(env)
        try:
            got = (p.get_option("fb_window_bits"), p.get_option("fb_window_bits_hi"), p.get_option("fb_hi_bases"))
            assert got == want, (free_gb, env, got)
            # the parts of a large call are sized to what the tables left free (at most 70 % of it): still 2^21 proofs beside the 210 GB
            assert p.get_option("max_batch") == 1 << 21
            acc, st = p.verify_batch(V, P, workload.LABEL)
            assert (acc == expect).all() and not st.any()
            cv = p.commit_value_batch(x, sb)          # g and h_vec[0]: from both regions where there are two
            if not ref:
                ref["cv"] = cv
            assert (cv == ref["cv"]).all()
            if want[2]:
                with pytest.raises(Exception):
                    p.save_tables("/tmp/never_written.bin")          # a table in two regions is not a saveable layout
        finally:
            p.close()
    monkeypatch.delenv("BPPP_ASSUME_FREE_GB")
    # the real thing: whatever is free right now decides; a second context beside 79 GB of tables still gets a width that fits
    p = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=0)
    try:
        w1 = p.get_option("fb_window_bits")
        assert w1 in (523, 621, 1119, 618, 317)
        assert p.get_option("n_generators") == 49 and p.get_option("device") == 0
        with pytest.raises(Exception):
            p.get_option("no such option")
    finally:
        p.close()


def test_table_budget_bounds_the_automatic_layout(monkeypatch, tmp_path):
    """bppp_wnla_ctx_create_budget: fb_table_budget_bytes caps what the automatic choice may take, on top of the free-memory rule -- the
    u64 shape on a free MI355X takes 210 GB without one; 120 GB gives the two regions (112 GB), 60 GB 12 windows (59 GB), 25 GB 13,
    1 GB 18 windows; an explicit width beyond the budget is refused; the context reports what it took ("fb_table_bytes", "fb_windows",
    "fb_window_bits_widest", "fb_table_budget_bytes"); children inherit budget and part size.  A two-width layout of ONE region saves and
    loads as a table file (magic BPPPTAB4); verdicts are the same in every layout."""
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    from bp_pp_amd._capi import BpppError
    g, gv, hv = workload.split_generators(workload.generators())
    _, V, P, _ = workload.make_batch(20, first=10)
    P, expect = workload.corrupt(P, V, every=5)
    monkeypatch.setenv("BPPP_ASSUME_FREE_GB", "300")
    GB = 10**9
    for budget, want, windows in ((0, (523, 0, 0), 11), (120 * GB, (621, 523, 17), 12), (60 * GB, (621, 0, 0), 12), (25 * GB, (1119, 0, 0), 13),
                                  (1 * GB, (614, 0, 0), 18)):
        if budget == 0:
            continue                                       # (the 210 GB default is exercised by test_default_window_width_follows_free_memory)
        p = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=0, fb_table_budget_bytes=budget)
        try:
            got = (p.get_option("fb_window_bits"), p.get_option("fb_window_bits_hi"), p.get_option("fb_hi_bases"))
            assert got == want, (budget, got)
            assert 0 < p.get_option("fb_table_bytes") <= budget and p.get_option("fb_table_budget_bytes") == budget
            assert p.get_option("fb_windows") == windows
            acc, st = p.verify_batch(V, P, workload.LABEL)
            assert (acc == expect).all() and not st.any()
            c = p.clone_shared()
            try:
                assert c.get_option("fb_table_budget_bytes") == budget and c.get_option("fb_table_bytes") == 0
                assert c.get_option("max_batch") == p.get_option("max_batch")
            finally:
                c.close()
        finally:
            p.close()
    with pytest.raises(BpppError) as ei:                   # an explicit layout beyond the budget: refused, nothing built
        U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=1119, fb_table_budget_bytes=1 * GB)
    assert ei.value.code == -2
    with pytest.raises(BpppError):                         # below the smallest table there is nothing to build
        U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=0, fb_table_budget_bytes=1000)
    monkeypatch.delenv("BPPP_ASSUME_FREE_GB")
    # one region of two widths as an artefact: 24 windows (code 1810: 18 windows of 11 bits, 6 of 10: 67 MB for the 49 generators)
    a = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=1810)
    path = str(tmp_path / "two_width.bin")
    try:
        assert a.get_option("fb_windows") == 24 and a.get_option("fb_window_bits_widest") == 11
        a.save_tables(path)
        assert open(path, "rb").read(8) == b"BPPPTAB4"
        b = U64RangeProofProtocol.from_tables(path, device=0)
        try:
            assert b.get_option("fb_window_bits") == 1810
            acc, st = b.verify_batch(V, P, workload.LABEL)
            assert (acc == expect).all() and not st.any()
        finally:
            b.close()
        blob = bytearray(open(path, "rb").read())
        blob[7] = ord("3")                                 # the same bytes under the uniform-width magic: refused
        open(path, "wb").write(blob)
        with pytest.raises(Exception):
            U64RangeProofProtocol.from_tables(path, device=0)
    finally:
        a.close()


@pytest.mark.parametrize("off", [1, 4, 8])
def test_device_buffers_at_odd_offsets(base, batch, off, oracle_c):
    """Every device pointer of the verify and prove entry points shifted by 1, 4 or 8 bytes from its 16-byte-aligned allocation: the
    928-byte and 64-byte forms then cross the vector-load alignment the fast path of be32_to_limbs needs, and the outputs
    (proofs, commitments, traces) are stored through the byte-wise path too.  Same verdicts, same bytes."""
    import torch
    import workload
    gens, V, P, expect = batch
    n = V.shape[0]

    def shifted(arr):
        flat = torch.zeros(arr.size + off + 32, dtype=torch.uint8, device="cuda")
        view = flat[off:off + arr.size]
        view.copy_(torch.from_numpy(np.ascontiguousarray(arr).reshape(-1).view(np.uint8)))
        assert view.data_ptr() % 16 == off % 16
        return flat, view

    keep = []
    fV, dV = shifted(V); fP, dP = shifted(P)
    fA, dA = shifted(np.zeros(n, np.uint8)); fT, dT = shifted(np.zeros((n, 704), np.uint8))
    dS = torch.zeros(n, dtype=torch.int32, device="cuda")         # int32 arrays stay 4-byte aligned (their C type requires it)
    dR = torch.zeros(1, dtype=torch.int32, device="cuda")
    keep += [fV, fP, fA, fT]
    torch.cuda.synchronize()
    base.verify_batch_device(workload.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), dT.data_ptr(), dR.data_ptr())
    base.synchronize()
    assert (dA.cpu().numpy() == expect).all() and not dS.cpu().numpy().any() and int(dR.item()) == int((expect == 0).sum())
    # the trace of an aligned run of the same batch, byte for byte
    aT = torch.zeros((n, 704), dtype=torch.uint8, device="cuda")
    aA = torch.zeros(n, dtype=torch.uint8, device="cuda")
    aV, aP = torch.from_numpy(V).cuda(), torch.from_numpy(P).cuda()
    base.verify_batch_device(workload.LABEL, n, aV.data_ptr(), aP.data_ptr(), aA.data_ptr(), dS.data_ptr(), aT.data_ptr(), 0)
    base.synchronize()
    assert (dT.cpu().numpy().reshape(n, 704) == aT.cpu().numpy()).all()
    # the prover: inputs and outputs shifted the same way
    x, s, rnd = workload.values(n, first=90), workload.blindings(n, first=90), workload.prover_randomness(n, first=90)
    op, ov = oracle_c.u64_prove_batch(gens, workload.LABEL, x, s, rnd, nthreads=4)
    dx = torch.from_numpy(x.view(np.int64)).cuda()                # uint64 values: 8-byte aligned by their C type
    fs, ds = shifted(s); fr, dr = shifted(rnd)
    fp, dp = shifted(np.zeros((n, 928), np.uint8)); fc, dc = shifted(np.zeros((n, 64), np.uint8))
    torch.cuda.synchronize()
    base.prove_batch_device(workload.LABEL, n, dx.data_ptr(), ds.data_ptr(), dr.data_ptr(), dp.data_ptr(), dc.data_ptr(), dS.data_ptr())
    base.synchronize()
    assert not dS.cpu().numpy().any()
    assert (dp.cpu().numpy().reshape(n, 928) == op).all() and (dc.cpu().numpy().reshape(n, 64) == ov).all()
