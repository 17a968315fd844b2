"""GPU parity tests of the generic batched ArithmeticCircuit::verify (circuit.rs:154-256) through the C ABI against the oracle: the
reference's own `ac_works` statement (tests.rs:45-136), k = 2 with every partition type, the f_m path, the f_l + f_m shape."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,B", [("ac_works", 70), ("mixed_k2", 9), ("fm_nv1", 5), ("fl_fm", 4)])
def test_circuit_verify_vs_oracle(name, B):
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import circuit_cases
    from bp_pp_amd.wnla import ArithmeticCircuit
    case = circuit_cases.make(name, B)
    part = lambda typ, j: (None if case["part"][typ][j] < 0 else int(case["part"][typ][j]))
    arr = lambda b: np.frombuffer(b, np.uint8).reshape(-1, 32)
    circ = ArithmeticCircuit(case["nm"], case["no"], case["k"], case["nv"], case["g"], case["gv"], case["hv"], arr(case["Wm_bytes"]),
                             arr(case["Wl_bytes"]), arr(case["am_bytes"]), arr(case["al_bytes"]), case["f_l"], case["f_m"], case["gv_"],
                             case["hv_"], part, device=0, fb_window_bits=16)
    try:
        shape = (case["rounds"], case["pl"], case["pn"])
        exp = [circuit_cases.oracle_verify(case, case["commitments"][b].tobytes(), case["proofs"][b].tobytes()) for b in range(B)]
        acc, st = circ.verify_batch(case["label"], case["commitments"], case["proofs"], *shape)
        assert acc.tolist() == exp and not st.any()
        if name != "fl_fm":
            assert all(exp)                     # shapes the prover is complete for
        # tampered instances: accept bits must equal the oracle's, instance by instance
        P = case["proofs"].copy()
        P[0, -1] ^= 1
        P[B - 1, 192:256] = P[B - 1, 0:64]
        com = case["commitments"].copy()
        com[1, 0] = case["commitments"][2, 0]
        acc, st = circ.verify_batch(case["label"], com, P, *shape)
        exp2 = [circuit_cases.oracle_verify(case, com[b].tobytes(), P[b].tobytes()) for b in range(B)]
        assert acc.tolist() == exp2 and not st.any()
        assert acc[0] == 0 and acc[1] == 0 and acc[B - 1] == 0
        # the resident form (bppp_circuit_verify_batch_device): the tampered batch from device buffers, same verdicts; kernel timing on
        dC, dP = torch.from_numpy(com).cuda(), torch.from_numpy(P).cuda()
        dA = torch.zeros(B, dtype=torch.uint8, device="cuda"); dS = torch.full((B,), 7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        circ.verify_batch_device(case["label"], B, dC.data_ptr(), dP.data_ptr(), *shape, dA.data_ptr(), dS.data_ptr()); circ.synchronize()
        assert dA.cpu().numpy().tolist() == exp2 and not dS.any().item()
        circ.enable_timing(True); dA.zero_()
        circ.verify_batch_device(case["label"], B, dC.data_ptr(), dP.data_ptr(), *shape, dA.data_ptr(), 0); circ.synchronize()
        kt = circ.timings(); circ.enable_timing(False)
        assert dA.cpu().numpy().tolist() == exp2 and kt["k_circuit_phase1"]["launches"] == 1 and kt["k_wnla_msm"]["launches"] == 1
        # malformed: off-curve point -> status flag, never accepted; the rest of the batch is unaffected
        P = case["proofs"].copy()
        P[1, 70] ^= 1
        acc, st = circ.verify_batch(case["label"], case["commitments"], P, *shape)
        assert st[1] == 1 and acc[1] == 0 and acc.tolist()[2:] == exp[2:] and acc[0] == exp[0]
    finally:
        circ.close()


@pytest.mark.parametrize("name,B", [("mixed_k2", 9), ("ac_works", 5)])
def test_circuit_verify_one_lane_kernels_vs_oracle(name, B, monkeypatch):
    """Small calls take the per-point form of C0's variable-base sum and the wavefront-per-instance fixed-base sums (bppp_generic.hip:
    per_point, generic_fb_wide); BPPP_NO_LANE_GROUPS=1 keeps the kernels large batches run -- one lane per instance, five points per
    shared-doubling pass, 8 lanes per fixed-base sum -- so that they are compared with the oracle at a size the oracle finishes too."""
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import circuit_cases
    from bp_pp_amd.wnla import ArithmeticCircuit
    monkeypatch.setenv("BPPP_NO_LANE_GROUPS", "1")
    case = circuit_cases.make(name, B)
    part = lambda typ, j: (None if case["part"][typ][j] < 0 else int(case["part"][typ][j]))
    arr = lambda b: np.frombuffer(b, np.uint8).reshape(-1, 32)
    circ = ArithmeticCircuit(case["nm"], case["no"], case["k"], case["nv"], case["g"], case["gv"], case["hv"], arr(case["Wm_bytes"]),
                             arr(case["Wl_bytes"]), arr(case["am_bytes"]), arr(case["al_bytes"]), case["f_l"], case["f_m"], case["gv_"],
                             case["hv_"], part, device=0, fb_window_bits=16)
    try:
        shape = (case["rounds"], case["pl"], case["pn"])
        P, com = case["proofs"].copy(), case["commitments"].copy()
        P[0, -1] ^= 1
        P[B - 1, 192:256] = P[B - 1, 0:64]
        com[1, 0] = case["commitments"][2, 0]
        exp = [circuit_cases.oracle_verify(case, com[b].tobytes(), P[b].tobytes()) for b in range(B)]
        acc, st = circ.verify_batch(case["label"], com, P, *shape)
        assert acc.tolist() == exp and not st.any() and sum(exp) == B - 3
    finally:
        circ.close()


def test_circuit_create_rejects_inconsistent_dimensions():
    import ctypes as C
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import circuit_cases
    from bp_pp_amd import _capi
    from bp_pp_amd.wnla import WeightNormLinearArgument
    case = circuit_cases.make("ac_works", 1)
    w = WeightNormLinearArgument(case["g"], case["gv"] + case["gv_"], case["hv"] + case["hv_"], device=0, fb_window_bits=8)
    try:
        p = case["parts"]
        h = C.c_void_p()
        call = lambda dims, lo: _capi.lib().bppp_circuit_create(w._ctx, C.byref(h), dims, 1, 0, case["Wm_bytes"], case["Wl_bytes"],
                                                                case["am_bytes"], case["al_bytes"], lo.ctypes.data, p["LL"].ctypes.data,
                                                                p["LR"].ctypes.data, p["NO"].ctypes.data)
        sz6 = C.c_size_t * 6
        assert call(sz6(1, 2, 1, 3, 2, 4), p["LO"]) == _capi.ERR_INVALID_ARG        # dim_nl != dim_nv k
        assert call(sz6(1, 2, 1, 2, 2, 5), p["LO"]) == _capi.ERR_INVALID_ARG        # dim_nw != 2 dim_nm + dim_no
        assert call(sz6(1, 2, 1, 2, 2, 4), np.array([0, 2], np.int32)) == _capi.ERR_INVALID_ARG   # partition index beyond w_o
        assert call(sz6(1, 2, 1, 2, 2, 4), p["LO"]) == 0
        _capi.lib().bppp_circuit_destroy(h)
    finally:
        w.close()


def test_commit_functions_vs_oracle():
    """ArithmeticCircuit::commit (circuit.rs:146-151) and ReciprocalRangeProofProtocol::{commit_value, commit_poles}
    (reciprocal.rs:88-95) through bppp_msm_batch against the oracle's commitments / big-integer arithmetic."""
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import bppp_oracle as O
    import circuit_cases
    import recip_cases
    from bp_pp_amd.wnla import ArithmeticCircuit, ReciprocalRangeProofProtocol
    B = 5
    case = circuit_cases.make("mixed_k2", B)
    part = lambda typ, j: (None if case["part"][typ][j] < 0 else int(case["part"][typ][j]))
    arr = lambda b: np.frombuffer(b, np.uint8).reshape(-1, 32)
    circ = ArithmeticCircuit(case["nm"], case["no"], case["k"], case["nv"], case["g"], case["gv"], case["hv"], arr(case["Wm_bytes"]),
                             arr(case["Wl_bytes"]), arr(case["am_bytes"]), arr(case["al_bytes"]), case["f_l"], case["f_m"], case["gv_"],
                             case["hv_"], part, device=0, fb_window_bits=16)
    try:
        sc = lambda v: np.frombuffer(O.sc_to_bytes(v % O.N), np.uint8)
        v = np.stack([np.stack([sc(x) for x in case["v"][j]]) for b in range(B) for j in range(case["k"])])
        s = np.stack([sc(circuit_cases._sc(b"sv", b, j)) for b in range(B) for j in range(case["k"])])
        out, st = circ.commit_batch(v, s)
        assert not st.any() and (out.reshape(B, case["k"], 64) == case["commitments"]).all()
        bad = s.copy()
        bad[1] = 0xFF                                   # >= n: flagged, identity output
        out, st = circ.commit_batch(v, bad)
        assert st.tolist() == [0, 1] + [0] * (B * case["k"] - 2) and not out[1].any()
    finally:
        circ.close()
    rc = recip_cases.make(8, 4, 3)
    proto = ReciprocalRangeProofProtocol(8, 4, rc["g"], rc["gv"], rc["hv"], rc["gv_"], rc["hv_"], device=0, fb_window_bits=16)
    try:
        xs = [7, 0, O.N - 1]
        ss = [recip_cases._sc(b"cs", i) for i in range(3)]
        out, st = proto.commit_value_batch(np.stack([sc(x) for x in xs]), np.stack([sc(x) for x in ss]))
        G, H0 = O.pt_from_xy64(rc["g"]), O.pt_from_xy64(rc["hv"][0])
        for i in range(3):
            assert out[i].tobytes() == O.pt_to_xy64(O.pt_add(O.pt_mul(G, xs[i]), O.pt_mul(H0, ss[i])))
        r = [[recip_cases._sc(b"pole", i, j) for j in range(8)] for i in range(3)]
        out, st = proto.commit_poles_batch(np.stack([np.stack([sc(x) for x in row]) for row in r]), np.stack([sc(x) for x in ss]))
        for i in range(3):
            exp = O.pt_mul(H0, ss[i])
            for j in range(8):
                exp = O.pt_add(exp, O.pt_mul(O.pt_from_xy64(rc["hv"][9 + j]), r[i][j]))
            assert out[i].tobytes() == O.pt_to_xy64(exp)
        assert not st.any()
    finally:
        proto.close()


@pytest.mark.parametrize("name,B", [("ac_works", 33), ("mixed_k2", 6), ("fm_nv1", 4), ("fl_fm", 3)])
def test_circuit_prove_byte_identical_and_verifies(name, B):
    """Generic ArithmeticCircuit::prove on the GPU (bppp_circuit_prove_batch): commitments via commit_batch, proof bytes equal to
    the reference-shaped prover's for the same witness and prover scalars, and the GPU verifier's verdict on them equals the
    oracle's (accept wherever the protocol is complete)."""
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import circuit_cases
    from bp_pp_amd.wnla import ArithmeticCircuit
    case = circuit_cases.make(name, B)
    part = lambda typ, j: (None if case["part"][typ][j] < 0 else int(case["part"][typ][j]))
    arr = lambda b: np.frombuffer(b, np.uint8).reshape(-1, 32)
    circ = ArithmeticCircuit(case["nm"], case["no"], case["k"], case["nv"], case["g"], case["gv"], case["hv"], arr(case["Wm_bytes"]),
                             arr(case["Wl_bytes"]), arr(case["am_bytes"]), arr(case["al_bytes"]), case["f_l"], case["f_m"], case["gv_"],
                             case["hv_"], part, device=0, fb_window_bits=16)
    try:
        k, nv = case["k"], case["nv"]
        com, st = circ.commit_batch(case["v_bytes"].reshape(B * k, nv, 32), case["s_v"].reshape(B * k, 32))
        assert not st.any() and (com.reshape(B, k, 64) == case["commitments"]).all()
        proofs, st, shape = circ.prove_batch(case["label"], com.reshape(B, k, 64), case["v_bytes"], case["s_v"], case["wl_bytes"],
                                             case["wr_bytes"], case["wo_bytes"], case["rnd"])
        assert not st.any() and shape == (case["rounds"], case["pl"], case["pn"])
        assert (proofs == case["proofs"]).all()
        acc, st = circ.verify_batch(case["label"], com.reshape(B, k, 64), proofs, *shape)
        assert acc.tolist() == [0 if name == "fl_fm" else 1] * B and not st.any()
        bad = case["wl_bytes"].copy()
        bad[0, 0] = 0xFF                                  # non-canonical witness scalar: that instance is flagged, its proof zeroed
        proofs2, st2, _ = circ.prove_batch(case["label"], com.reshape(B, k, 64), case["v_bytes"], case["s_v"], bad, case["wr_bytes"],
                                           case["wo_bytes"], case["rnd"])
        assert st2[0] == 1 and not st2[1:].any() and not proofs2[0].any() and (proofs2[1:] == proofs[1:]).all()
    finally:
        circ.close()
