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
        # malformed: off-curve point -> status flag, never accepted; the rest of the batch is unaffected
        P = case["proofs"].copy()
        P[1, 70] ^= 1
        acc, st = circ.verify_batch(case["label"], case["commitments"], P, *shape)
        assert st[1] == 1 and acc[1] == 0 and acc.tolist()[2:] == exp[2:] and acc[0] == exp[0]
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
