"""GPU parity tests of the generic batched WeightNormLinearArgument::{commit, verify} (the crate's `wnla` API surface) through
the C ABI, against the oracle: the reference's own shape (tests.rs:139-171), odd lengths, the u64 shape, and the
"aggregated" shape of BASELINE configs[4] (|h_vec| = 512, |g_vec| = 256, 8 rounds)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ng,nh,B", [(4, 4, 5), (16, 32, 70), (3, 5, 4), (1, 2, 3), (256, 512, 3)])
def test_wnla_commit_verify_vs_oracle(ng, nh, B):
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import wnla_cases
    from bp_pp_amd.wnla import WeightNormLinearArgument
    case = wnla_cases.make(ng, nh, B)
    w = WeightNormLinearArgument(case["g"], case["gv"], case["hv"], device=0, fb_window_bits=8 if nh > 64 else 16)
    try:
        out, st = w.commit_batch(case["c"], case["mu"], case["l"], case["n"])
        assert not st.any() and (out == case["commitments"]).all()
        args = dict(commitments=case["commitments"], c=case["c"], rho=case["rho"], mu=case["mu"], proof_r=case["proof_r"],
                    proof_x=case["proof_x"], proof_l=case["proof_l"], proof_n=case["proof_n"])
        acc, st = w.verify_batch(case["label"], **args)
        assert acc.all() and not st.any()
        # tampered instances: accept bits must equal the oracle's, instance by instance
        pl = case["proof_l"].copy(); pl[0, 0, 31] ^= 1
        pn = case["proof_n"].copy(); pn[B - 1, 0, 5] ^= 0x10
        com = case["commitments"].copy()
        if B > 2:
            com[1] = case["commitments"][2]
        t = dict(args, proof_l=pl, proof_n=pn, commitments=com)
        acc, st = w.verify_batch(case["label"], **t)
        exp = [wnla_cases.oracle_verify(case, b, proof_l=pl, proof_n=pn, commitments=com) for b in range(B)]
        assert acc.tolist() == exp and not st.any()
        assert acc[0] == 0 and acc[B - 1] == 0
        # malformed: off-curve round point -> status flag, never accepted
        px = case["proof_x"].copy()
        if px.shape[1]:
            px[0, 0, 63] ^= 1
            acc, st = w.verify_batch(case["label"], **dict(args, proof_x=px))
            assert st[0] == 1 and acc[0] == 0 and acc[1:].all()
        # the resident form (bppp_wnla_verify_batch_device): the tampered batch from device buffers, same verdicts and statuses, with and
        # without a status buffer, and with per-kernel timing on
        d = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in t.items()}
        dA = torch.zeros(B, dtype=torch.uint8, device="cuda"); dS = torch.full((B,), 7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        dev = lambda st_ptr: w.verify_batch_device(case["label"], B, d["commitments"].data_ptr(), d["c"].data_ptr(), d["rho"].data_ptr(), d["mu"].data_ptr(),
                                                   case["proof_r"].shape[1], d["proof_r"].data_ptr(), d["proof_x"].data_ptr(), d["proof_l"].data_ptr(),
                                                   pl.shape[1], d["proof_n"].data_ptr(), pn.shape[1], dA.data_ptr(), st_ptr)
        dev(dS.data_ptr()); w.synchronize()
        assert dA.cpu().numpy().tolist() == exp and not dS.any().item()
        dA.zero_(); dev(0); w.synchronize()
        assert dA.cpu().numpy().tolist() == exp
        w.enable_timing(True); dA.zero_(); dev(dS.data_ptr()); w.synchronize()
        kt = w.timings(); w.enable_timing(False)
        assert dA.cpu().numpy().tolist() == exp and kt["k_wnla_msm"]["launches"] == 1 and kt["k_wnla_round"]["launches"] == case["proof_r"].shape[1]
        # proof.x.len() != proof.r.len() -> false (wnla.rs:76-78)
        acc, _ = w.verify_batch(case["label"], **dict(args, proof_r=case["proof_r"][:, :-1] if case["proof_r"].shape[1] else np.zeros((B, 1, 64), np.uint8)))
        assert not acc.any()
    finally:
        w.close()


@pytest.mark.parametrize("W", [4, 8, 10, 16, 18, 19, 20, 22])
def test_every_fixed_base_window_width_vs_oracle(W):
    """The same WNLA instances (commit, verify, prove: wnla.rs:66-190) through fixed-base tables of every window width the library
    accepts -- unsigned 4 / 8 / 16 bits, signed 10 / 18 / 19 / 20 / 22 bits (fb_core.h: fb_digit) -- against the oracle; few
    generators, so that even the 22-bit tables stay small (7 bases x 12 windows x 2^21 entries = 11 GB)."""
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import wnla_cases
    from bp_pp_amd.wnla import WeightNormLinearArgument
    B = 9
    case = wnla_cases.make(2, 4, B)
    w = WeightNormLinearArgument(case["g"], case["gv"], case["hv"], device=0, fb_window_bits=W)
    try:
        out, st = w.commit_batch(case["c"], case["mu"], case["l"], case["n"])
        assert not st.any() and (out == case["commitments"]).all()
        args = dict(commitments=case["commitments"], c=case["c"], rho=case["rho"], mu=case["mu"], proof_r=case["proof_r"],
                    proof_x=case["proof_x"], proof_l=case["proof_l"], proof_n=case["proof_n"])
        acc, st = w.verify_batch(case["label"], **args)
        assert acc.all() and not st.any()
        pl = case["proof_l"].copy(); pl[3, 0, 31] ^= 1
        acc, st = w.verify_batch(case["label"], **dict(args, proof_l=pl))
        assert acc.tolist() == [1, 1, 1, 0, 1, 1, 1, 1, 1]
        pr, px, plv, pnv, st = w.prove_batch(case["label"], case["commitments"], case["c"], case["rho"], case["mu"], case["l"], case["n"])
        assert not st.any() and (pr == case["proof_r"]).all() and (px == case["proof_x"]).all()
        assert (plv == case["proof_l"]).all() and (pnv == case["proof_n"]).all()
    finally:
        w.close()


def test_fixed_base_fast_accumulator_fallback_on_device():
    """Repeated generators + a scalar pattern that makes one lane add a table entry to itself: the incomplete (XYZZ) fixed-base
    accumulator must notice and the lane group must re-do the sum with the complete law (fb_core.h: fb_group_sum)."""
    import ctypes as C
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import bppp_oracle as O
    import bppp_oracle_c as OC
    import wnla_cases
    from bp_pp_amd.wnla import WeightNormLinearArgument
    g, gv, hv = wnla_cases.generators(2, 4)
    gv = [gv[0], gv[1], gv[0], gv[1]]                       # g_vec[2] == g_vec[0], g_vec[3] == g_vec[1]
    B = 3
    n_rows = [[7, 0, 7, 0], [0, 5 << 32, 0, 5 << 32], [3, 9, 4, 1]]    # rows 0, 1: single equal window on both copies; row 2: ordinary
    mu = [1 << 16, 1 << 16, 12345]                          # v = sum n_i^2 mu^(i+1) stays out of the colliding lane's windows
    c = np.zeros((B, 4, 32), np.uint8)
    l = np.zeros((B, 4, 32), np.uint8)
    n = np.frombuffer(b"".join(O.sc_to_bytes(v) for row in n_rows for v in row), np.uint8).reshape(B, 4, 32).copy()
    mub = np.frombuffer(b"".join(O.sc_to_bytes(v) for v in mu), np.uint8).reshape(B, 32).copy()
    L = OC.lib()
    sz = C.c_size_t
    exp = []
    for b in range(B):
        com = C.create_string_buffer(64)
        assert L.bppp_oracle_wnla_commit(g, b"".join(gv), sz(4), b"".join(hv), sz(4), c[b].tobytes(), sz(4), O.sc_to_bytes(1),
                                         mub[b].tobytes(), l[b].tobytes(), sz(4), n[b].tobytes(), sz(4), com) == 0
        exp.append(com.raw)
    w = WeightNormLinearArgument(g, gv, hv, device=0, fb_window_bits=16)
    try:
        out, st = w.commit_batch(c, mub, l, n)
        assert not st.any()
        assert [bytes(o) for o in out] == exp
    finally:
        w.close()


@pytest.mark.parametrize("ng,nh,B", [(4, 4, 5), (16, 32, 70), (3, 5, 4), (7, 9, 3), (256, 512, 2)])
def test_wnla_prove_byte_identical_and_verifies(ng, nh, B):
    """Generic WeightNormLinearArgument::prove on the GPU (bppp_wnla_prove_batch): proof bytes equal the reference-shaped
    prover's, and the GPU verifier accepts them (tests.rs:139-171 round trip)."""
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import wnla_cases
    from bp_pp_amd.wnla import WeightNormLinearArgument
    case = wnla_cases.make(ng, nh, B)
    w = WeightNormLinearArgument(case["g"], case["gv"], case["hv"], device=0, fb_window_bits=8 if nh > 64 else 16)
    try:
        pr, px, pl, pn, st = w.prove_batch(case["label"], case["commitments"], case["c"], case["rho"], case["mu"], case["l"], case["n"])
        assert not st.any()
        assert (pr == case["proof_r"]).all() and (px == case["proof_x"]).all()
        assert (pl == case["proof_l"]).all() and (pn == case["proof_n"]).all()
        acc, st = w.verify_batch(case["label"], case["commitments"], case["c"], case["rho"], case["mu"], pr, px, pl, pn)
        assert acc.all() and not st.any()
        # a non-canonical witness scalar flags its instance only
        l_bad = case["l"].copy()
        l_bad[0, 0] = 0xFF
        pr2, px2, pl2, pn2, st2 = w.prove_batch(case["label"], case["commitments"], case["c"], case["rho"], case["mu"], l_bad, case["n"])
        assert st2[0] == 1 and not st2[1:].any() and not pr2[0].any() and (pr2[1:] == pr[1:]).all() and (pl2[1:] == pl[1:]).all()
    finally:
        w.close()
