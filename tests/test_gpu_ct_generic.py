"""GPU tier of "ct_prover" for the GENERIC provers (include/bppp.h): bppp_wnla_prove_batch, bppp_circuit_prove_batch and
bppp_reciprocal_prove_batch with the option on run their sums over secret scalars in the full-scan form over a 4-bit table of the
context's own generators -- the same points, so the same proof bytes as with the option off and as the oracle provers'.  That the
form's table reads do not depend on the secrets is tested on the host build of the same device code (tests/test_ct_trace.py)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _set_ct(ctx, on):
    from bp_pp_amd import _capi
    _capi.check(_capi.lib().bppp_ctx_set_option(ctx, b"ct_prover", 1 if on else 0))
    assert _capi.lib().bppp_ctx_get_option(ctx, b"ct_prover") == (1 if on else 0)


@pytest.mark.parametrize("ng,nh,B", [(16, 32, 40), (7, 9, 5)])
def test_wnla_prover(ng, nh, B):
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import wnla_cases
    from bp_pp_amd.wnla import WeightNormLinearArgument
    case = wnla_cases.make(ng, nh, B)
    w = WeightNormLinearArgument(case["g"], case["gv"], case["hv"], device=0, fb_window_bits=16)
    try:
        _set_ct(w._ctx, True)
        pr, px, pl, pn, st = w.prove_batch(case["label"], case["commitments"], case["c"], case["rho"], case["mu"], case["l"], case["n"])
        assert not st.any()
        assert (pr == case["proof_r"]).all() and (px == case["proof_x"]).all() and (pl == case["proof_l"]).all() and (pn == case["proof_n"]).all()
        _set_ct(w._ctx, False)
        pr0, px0, pl0, pn0, st0 = w.prove_batch(case["label"], case["commitments"], case["c"], case["rho"], case["mu"], case["l"], case["n"])
        assert (pr0 == pr).all() and (px0 == px).all() and (pl0 == pl).all() and (pn0 == pn).all()
    finally:
        w.close()


@pytest.mark.parametrize("nd,npp,B", [(16, 16, 30), (12, 10, 7)])
def test_reciprocal_prover(nd, npp, B):
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import recip_cases
    from bp_pp_amd.wnla import ReciprocalRangeProofProtocol
    case = recip_cases.make(nd, npp, B)
    proto = ReciprocalRangeProofProtocol(nd, npp, case["g"], case["gv"], case["hv"], case["gv_"], case["hv_"], device=0, fb_window_bits=16)
    try:
        _set_ct(proto._w._ctx, True)
        proofs, st, shape = proto.prove_batch(case["label"], case["commitments"], case["x"], case["s"], case["digits"], case["m"], case["rnd"])
        assert not st.any() and shape == (case["rounds"], case["nl"], case["nn"])
        assert (proofs == case["proofs"]).all()
        acc, st = proto.verify_batch(case["label"], case["commitments"], proofs, *shape)
        assert acc.all() and not st.any()
        _set_ct(proto._w._ctx, False)
        proofs0, st0, _ = proto.prove_batch(case["label"], case["commitments"], case["x"], case["s"], case["digits"], case["m"], case["rnd"])
        assert (proofs0 == proofs).all()
    finally:
        proto.close()


@pytest.mark.parametrize("name,B", [("ac_works", 9), ("mixed_k2", 4)])
def test_circuit_prover(name, B):
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
        _set_ct(circ._w._ctx, True)
        proofs, st, shape = circ.prove_batch(case["label"], case["commitments"], case["v_bytes"], case["s_v"], case["wl_bytes"],
                                             case["wr_bytes"], case["wo_bytes"], case["rnd"])
        assert not st.any() and shape == (case["rounds"], case["pl"], case["pn"])
        assert (proofs == case["proofs"]).all()
        _set_ct(circ._w._ctx, False)
        proofs0, _, _ = circ.prove_batch(case["label"], case["commitments"], case["v_bytes"], case["s_v"], case["wl_bytes"],
                                         case["wr_bytes"], case["wo_bytes"], case["rnd"])
        assert (proofs0 == proofs).all()
    finally:
        circ.close()
