"""Every size threshold of the u64 verifier's and prover's launch sequences (csrc/plan_core.h) at T-1, T, T+1 proofs on an MI355X,
against the oracle, with the plan that ran asserted ("last_verify_plan" / "last_prove_plan" of bppp_ctx_get_option): an off-by-one in a
threshold, or a regime that is never entered, fails here.  One batch of 2^17 + 1 proofs from the oracle's trapdoor prover serves every
size as a prefix; the CPU-tier twin (the thresholds themselves) is tests/test_plan.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THRESHOLDS = [1024, 4096, 16384, 32768, 65536, 131072]
NMAX = THRESHOLDS[-1] + 1
FIRST = 31000


@pytest.fixture(scope="module")
def batch():
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import workload
    gens, V, P, x = workload.make_batch(NMAX, first=FIRST)
    return dict(gens=gens, V=V, P=P, x=x, s=workload.blindings(NMAX, FIRST), rnd=workload.prover_randomness(NMAX, FIRST))


@pytest.fixture(scope="module")
def proto(batch):
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    g, gv, hv = workload.split_generators(batch["gens"])
    p = U64RangeProofProtocol(g, gv, hv, device=0)
    assert p.get_option("last_verify_plan") == 0 and p.get_option("last_prove_plan") == 0
    yield p
    p.close()


def _device_verify(torch, proto, label, V, P):
    n = V.shape[0]
    dV, dP = torch.from_numpy(V).cuda(), torch.from_numpy(P).cuda()
    dA = torch.zeros(n, dtype=torch.uint8, device="cuda")
    dS = torch.zeros(n, dtype=torch.int32, device="cuda")
    dT = torch.zeros((n, 704), dtype=torch.uint8, device="cuda")
    dR = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    proto.verify_batch_device(label, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), dT.data_ptr(), dR.data_ptr())
    torch.cuda.synchronize()
    return dA.cpu().numpy(), dS.cpu().numpy(), dT.cpu().numpy(), int(dR.item())


@pytest.mark.parametrize("T", THRESHOLDS)
@pytest.mark.parametrize("d", [-1, 0, 1])
def test_verify_at_every_threshold_vs_oracle(batch, proto, oracle_c, T, d):
    """Honest, tampered (every 13th) and malformed proofs (random byte flips; a coordinate = p; a scalar = n -- among them the batch's last
    proof, the one a wrong grid size would drop): accept bits, statuses, the reject count, and on a sample that holds every malformed
    proof and its neighbours the whole 704-byte trace of challenges and commitments, equal the oracle's; and the call took the plan
    the size is meant to take."""
    from bp_pp_amd.range_proof import plan_for
    n = T + d
    _verify_and_check(batch, proto, oracle_c, n, plan_for(n))


def _verify_and_check(batch, proto, oracle_c, n, expect_plan):
    import torch
    import workload
    V, P = batch["V"][:n].copy(), batch["P"][:n]
    P, expect = workload.corrupt(P, V, every=13)
    P = P.copy()
    rng = np.random.default_rng(n)
    bad = sorted(set(int(i) for i in rng.integers(0, n, 24)) | {0, n - 1})
    for i in bad[1:-1]:
        P[i, int(rng.integers(0, 928))] ^= int(rng.integers(1, 256))
    P[0, 864:896] = np.frombuffer(workload.N_ORDER.to_bytes(32, "big"), np.uint8)            # a scalar = n
    P[n - 1, 64:96] = np.frombuffer((2**256 - 2**32 - 977).to_bytes(32, "big"), np.uint8)     # a coordinate = p
    acc, st, tr, rej = _device_verify(torch, proto, workload.LABEL, V, P)
    assert proto.last_plan() == expect_plan, (n, proto.last_plan())
    sample = sorted(set(bad + [b + 1 for b in bad if b + 1 < n] + [b - 1 for b in bad if b > 0] + list(range(0, n, max(1, n // 24)))))
    flagged = 0
    for i in sample:
        rc, otr = oracle_c.u64_verify(batch["gens"], workload.LABEL, bytes(V[i]), bytes(P[i]), trace=True)
        assert int(acc[i]) == (1 if rc == 1 else 0), (n, i)
        assert (int(st[i]) != 0) == (rc < 0), (n, i, rc, int(st[i]))
        if rc >= 0:
            assert bytes(tr[i]) == otr, (n, i)
        flagged += rc < 0
    assert flagged >= 2 and st[0] != 0 and st[n - 1] != 0
    clean = np.ones(n, bool)
    clean[bad] = False
    assert (acc[clean] == expect[clean]).all() and not st[clean].any()
    assert rej == int((acc == 0).sum())


@pytest.mark.parametrize("G,n", [(2, 131073), (4, 70001), (8, 65537), (16, 70001), (16, 131073)])
def test_shared_inversions_at_the_group_sizes_of_larger_batches(batch, oracle_c, monkeypatch, G, n):
    """From 2^18 proofs the table build and the rounds take their field inversions from kernels that invert once for G proofs (plan_core.h:
    shared_inv; G = 8 from 2^18, 16 from 2^20).  BPPP_SHARED_INV forces the groups at sizes this tier can afford, ragged ones included
    (n not a multiple of G: the last lanes' groups are short).  Same checks as the sweep above."""
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    from bp_pp_amd.range_proof import plan_for
    monkeypatch.setenv("BPPP_SHARED_INV", str(G))
    g, gv, hv = workload.split_generators(batch["gens"])
    p = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=16)
    try:
        base = plan_for(n)
        assert "shared_inv=" in base
        _verify_and_check(batch, p, oracle_c, n, base[:base.index("shared_inv=")] + "shared_inv=%d" % G)
    finally:
        p.close()


@pytest.mark.parametrize("code", [618, 1119])
def test_tables_with_windows_of_two_widths(batch, oracle_c, code):
    """Fixed-base tables whose windows are sized to the bit (the library's own choice on an empty MI355X is 523 = 5 x 24 + 6 x 23 bits,
    210 GB; 618 = 6 x 19 + 8 x 18 bits and 1119 = 11 x 20 + 2 x 19 bits are the same layout at 8 and 20 GB): the verifier on a
    wavefront per sum, on 8 lanes and on one lane per sum, and the prover, against the oracle as in the sweeps above."""
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    from bp_pp_amd.range_proof import plan_for
    g, gv, hv = workload.split_generators(batch["gens"])
    p = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=code)
    try:
        assert p.get_option("fb_window_bits") == code
        for n in (1000, 20000, 131073):
            _verify_and_check(batch, p, oracle_c, n, plan_for(n))
        for n in (3, 3000, 20000):
            proofs, com, st = p.prove_batch(batch["x"][:n], batch["s"][:n], batch["rnd"][:n], workload.LABEL)
            assert not st.any() and (com == batch["V"][:n]).all() and (proofs == batch["P"][:n]).all()
    finally:
        p.close()
    for bad in (124, 3008, 507):          # a 25-bit window; more wide windows (30) than windows (29); narrow windows below 8 bits
        with pytest.raises(Exception):
            U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=bad)


def test_shared_inversions_and_the_twin_form_at_their_own_size_equal_the_per_proof_form(batch, oracle_c, monkeypatch):
    """2^18 + 1 proofs (the fixture's batch twice over: the plan's own G = 8, ragged) with tampered and malformed proofs: accept bits,
    statuses, reject count and ALL 704-byte traces of the call with shared inversions, and of the call as two half-batch chains (the
    library's own choice at this size), equal those of ONE sequence that inverts per proof (BPPP_TWIN=0 BPPP_SHARED_INV=0), byte for
    byte; a sample also goes to the oracle."""
    import torch
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    n = (1 << 18) + 1
    idx = np.arange(n) % NMAX
    V, P = batch["V"][idx].copy(), batch["P"][idx]
    P, expect = workload.corrupt(P, V, every=13)
    P = P.copy()
    rng = np.random.default_rng(18)
    bad = sorted(set(int(i) for i in rng.integers(0, n, 64)) | {0, n - 1, n - 2})
    for i in bad[1:-2]:
        P[i, int(rng.integers(0, 928))] ^= int(rng.integers(1, 256))
    P[0, 864:896] = np.frombuffer(workload.N_ORDER.to_bytes(32, "big"), np.uint8)            # a scalar = n
    P[n - 1, 64:96] = np.frombuffer((2**256 - 2**32 - 977).to_bytes(32, "big"), np.uint8)     # a coordinate = p
    P[n - 2, :832] = 0                                                                        # every proof point the identity
    g, gv, hv = workload.split_generators(batch["gens"])
    res = []
    # (1) one sequence, the plan's own shared inversions; (2) one sequence, every lane inverting for itself; (3) what the library does with
    # this size by itself: two half-batch chains (plan_core.h: twin -- 2^18 + 1 proofs are two ragged halves, 131,104 + 131,041)
    for twin, env in (("0", None), ("0", "0"), (None, None)):
        for name, val in (("BPPP_TWIN", twin), ("BPPP_SHARED_INV", env)):
            if val is None:
                monkeypatch.delenv(name, raising=False)
            else:
                monkeypatch.setenv(name, val)
        p = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=16)
        try:
            res.append(_device_verify(torch, p, workload.LABEL, V, P) + (p.last_plan(),))
        finally:
            p.close()
    monkeypatch.delenv("BPPP_TWIN", raising=False)
    (a1, s1, t1, r1, plan1), (a0, s0, t0, r0, plan0), (a2, s2, t2, r2, plan2) = res
    assert plan1.endswith("twin=1 pace=0 shared_inv=8") and plan0.endswith("twin=1 pace=0 shared_inv=0") and plan1[:-1] == plan0[:-1]
    assert plan2.endswith("twin=2 pace=0 shared_inv=0") and plan2[:plan2.index("twin=")] == plan0[:plan0.index("twin=")]
    assert (a1 == a0).all() and (s1 == s0).all() and r1 == r0 and (t1 == t0).all()
    assert (a2 == a0).all() and (s2 == s0).all() and r2 == r0 and (t2 == t0).all()
    assert r1 == int((a1 == 0).sum()) and s1[0] != 0 and s1[n - 1] != 0 and s1[n - 2] == 0 and a1[n - 2] == 0
    clean = np.ones(n, bool)
    clean[bad] = False
    assert (a1[clean] == expect[clean]).all() and not s1[clean].any()
    for i in bad + [1, n // 2]:
        rc, otr = oracle_c.u64_verify(batch["gens"], workload.LABEL, bytes(V[i]), bytes(P[i]), trace=True)
        assert int(a1[i]) == (1 if rc == 1 else 0) and (int(s1[i]) != 0) == (rc < 0), (i, rc)
        if rc >= 0:
            assert bytes(t1[i]) == otr, i


@pytest.mark.parametrize("T", THRESHOLDS)
@pytest.mark.parametrize("d", [-1, 0, 1])
def test_prove_at_every_threshold_vs_oracle(batch, proto, T, d):
    """The batch prover at T-1, T, T+1 values: every proof and commitment byte-identical to the oracle's trapdoor prover (the fixture),
    and the plan that ran is the one the size is meant to take."""
    import workload
    from bp_pp_amd.range_proof import plan_for
    n = T + d
    proofs, com, st = proto.prove_batch(batch["x"][:n], batch["s"][:n], batch["rnd"][:n], workload.LABEL)
    assert proto.last_plan(prove=True) == plan_for(n, prove=True), (n, proto.last_plan(prove=True))
    assert not st.any()
    assert (com == batch["V"][:n]).all()
    assert (proofs == batch["P"][:n]).all()


def test_every_regime_was_entered():
    """The sweep above is only worth its name if the sizes really span all the regimes: seven for the verifier, and for the prover every
    change of its plan within the sweep's range."""
    from bp_pp_amd.range_proof import plan_for
    sizes = [T + d for T in THRESHOLDS for d in (-1, 0, 1)]
    assert len({plan_for(n) for n in sizes}) == 8          # (the seven regimes; 131,072 and 131,073 differ by the pacing)
    assert len({plan_for(n, prove=True) for n in sizes}) >= 7
