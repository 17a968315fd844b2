"""CPU tier for the generic WNLA device code (wnla_core.h compiled for the host): commit and verify against the oracle on the
reference's own shape (tests.rs:139-171, N = 4), on odd lengths, on mu != rho^2 and on the u64 shape."""
import numpy as np
import pytest

import wnla_cases
from emul.build import load


def _table(L, case, W=4):
    gens = case["g"] + b"".join(case["gv"]) + b"".join(case["hv"])
    nb = 1 + case["ng"] + case["nh"]
    tab = np.zeros(L.emul_fb_table_entries(nb, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, nb, W, tab.ctypes.data) == 0
    return tab, W


def _run(L, case, tab, W, commit, **over):
    B = case["commitments"].shape[0]
    d = {k: np.ascontiguousarray(over.get(k, case[k])) for k in ("commitments", "c", "rho", "mu", "proof_r", "proof_x", "proof_l", "proof_n", "l", "n")}
    out, acc, st = np.zeros((B, 64), np.uint8), np.zeros(B, np.uint8), np.zeros(B, np.int32)
    pl, pn = (d["l"], d["n"]) if commit else (d["proof_l"], d["proof_n"])
    L.emul_wnla_run(1 if commit else 0, tab.ctypes.data, W, case["ng"], case["nh"], case["label"], len(case["label"]), B,
                    d["commitments"].ctypes.data, d["c"].ctypes.data, d["rho"].ctypes.data, d["mu"].ctypes.data,
                    0 if commit else d["proof_r"].shape[1], d["proof_r"].ctypes.data, d["proof_x"].ctypes.data, pl.ctypes.data, pl.shape[1],
                    pn.ctypes.data, pn.shape[1], out.ctypes.data, acc.ctypes.data, st.ctypes.data)
    return out, acc, st


@pytest.mark.parametrize("ng,nh,musq", [(4, 4, True), (16, 32, True), (3, 5, True), (4, 8, False), (1, 2, True)])
def test_commit_and_verify_vs_oracle(ng, nh, musq):
    L = load()
    case = wnla_cases.make(ng, nh, B=3, mu_is_rho_sq=musq)
    tab, W = _table(L, case)
    out, _, st = _run(L, case, tab, W, commit=True)
    assert not st.any() and (out == case["commitments"]).all()                     # wnla.rs:66-72
    _, acc, st = _run(L, case, tab, W, commit=False)
    exp = [wnla_cases.oracle_verify(case, b) for b in range(3)]
    assert acc.tolist() == exp and not st.any()
    if musq:
        assert exp == [1, 1, 1]                                                    # honest prove => verify (tests.rs:170)
    else:
        assert exp == [0, 0, 0]                                                    # the argument is only complete for mu = rho^2
    # tampered: final l, a round point swapped, wrong commitment -- accept bits must equal the oracle's
    pl = case["proof_l"].copy(); pl[0, 0, 31] ^= 1
    px = case["proof_x"].copy()
    if px.shape[1] >= 1:
        px[1, 0], case_r0 = case["proof_r"][1, 0].copy(), None
    com = case["commitments"].copy(); com[2] = case["commitments"][0]
    _, acc, st = _run(L, case, tab, W, commit=False, proof_l=pl, proof_x=px, commitments=com)
    exp = [wnla_cases.oracle_verify(case, b, proof_l=pl, proof_x=px, commitments=com) for b in range(3)]
    assert acc.tolist() == [1 if e == 1 else 0 for e in exp] and not st.any()
    assert acc[0] == 0 and acc[2] == 0


@pytest.mark.parametrize("lg", [1, 2, 3])
@pytest.mark.parametrize("ng,nh", [(16, 32), (3, 5), (7, 9), (1, 2), (4, 4)])
def test_final_scalars_on_lane_groups(ng, nh, lg):
    """k_wnla_final_scalars_grp's arithmetic: 2^lg lanes per instance, each with its own part of the coefficient tables and the
    generators that read it -- accept bits and statuses equal the one-lane form's (and so the oracle's), also with a tampered l,
    an out-of-range scalar in each of l, n, c (status: bad encoding) and lengths that leave lanes without any generator."""
    L = load()
    case = wnla_cases.make(ng, nh, B=4)
    tab, W = _table(L, case)
    pl = case["proof_l"].copy(); pl[0, 0, 31] ^= 1
    pl[1, -1] = 0xFF                                   # >= n: not a canonical scalar
    pn = case["proof_n"].copy(); pn[2, 0] = 0xFF
    c = case["c"].copy(); c[3, -1] = 0xFF
    over = [dict(), dict(proof_l=pl), dict(proof_l=pl, proof_n=pn, c=c)]
    try:
        for o in over:
            L.emul_set_final_scalars_lg(0)
            _, acc0, st0 = _run(L, case, tab, W, commit=False, **o)
            L.emul_set_final_scalars_lg(lg)
            _, acc, st = _run(L, case, tab, W, commit=False, **o)
            assert acc.tolist() == acc0.tolist() and st.tolist() == st0.tolist()
        assert acc0.tolist() == [0, 0, 0, 0] and st0[1] != 0 and st0[2] != 0 and st0[3] != 0
    finally:
        L.emul_set_final_scalars_lg(0)


def test_length_quirks_match_reference():
    """proof.l / proof.n longer than the folded generator vectors: extra l entries multiply identities, extra n entries
    still enter |n|^2_mu (wnla.rs:67, util.rs:24-26) -- so an appended zero is harmless and an appended non-zero n is not."""
    L = load()
    case = wnla_cases.make(4, 4, B=2)
    tab, W = _table(L, case)
    zero = np.zeros((2, 1, 32), np.uint8)
    one = zero.copy(); one[:, 0, 31] = 1
    for extra_l, extra_n in [(zero, zero), (one, zero), (zero, one)]:
        pl = np.concatenate([case["proof_l"], extra_l], axis=1)
        pn = np.concatenate([case["proof_n"], extra_n], axis=1)
        _, acc, st = _run(L, case, tab, W, commit=False, proof_l=pl, proof_n=pn)
        exp = [wnla_cases.oracle_verify(case, b, proof_l=pl, proof_n=pn) for b in range(2)]
        assert acc.tolist() == exp and not st.any()
    assert exp == [0, 0]


@pytest.mark.parametrize("ng,nh,musq", [(4, 4, True), (16, 32, True), (3, 5, True), (4, 8, False), (1, 2, True), (7, 9, True), (8, 8, True)])
def test_generic_prove_is_byte_identical_to_the_oracle(ng, nh, musq):
    """wnla.rs:125-190 on the device code: the proof (r, x in the reference's vector order, final l and n) must equal the
    reference-shaped prover's bytes for the same inputs -- power-of-two and odd sizes, mu != rho^2, and the sizes that stop
    after one round or none."""
    L = load()
    B = 3
    case = wnla_cases.make(ng, nh, B=B, mu_is_rho_sq=musq)
    tab, W = _table(L, case)
    d = {k: np.ascontiguousarray(case[k]) for k in ("commitments", "c", "rho", "mu", "l", "n")}
    pr, px = np.zeros((B, 16, 64), np.uint8), np.zeros((B, 16, 64), np.uint8)
    pl, pn = np.zeros((B, 8, 32), np.uint8), np.zeros((B, 8, 32), np.uint8)
    st = np.zeros(B, np.int32)
    import ctypes as C
    r, a, b = C.c_int(), C.c_int(), C.c_int()
    # the emulation writes with the TRUE strides (rounds, nl', nn'), so size the buffers after asking for the shape
    L.emul_wnla_prove(tab.ctypes.data, W, ng, nh, case["label"], len(case["label"]), 0, None, None, None, None, None, d["l"].shape[1], None,
                      d["n"].shape[1], None, None, None, None, None, C.byref(r), C.byref(a), C.byref(b))
    assert (r.value, a.value, b.value) == (case["rounds"], case["nl"], case["nn"])
    pr, px = np.zeros((B, r.value, 64), np.uint8), np.zeros((B, r.value, 64), np.uint8)
    pl, pn = np.zeros((B, a.value, 32), np.uint8), np.zeros((B, b.value, 32), np.uint8)
    L.emul_wnla_prove(tab.ctypes.data, W, ng, nh, case["label"], len(case["label"]), B, d["commitments"].ctypes.data, d["c"].ctypes.data,
                      d["rho"].ctypes.data, d["mu"].ctypes.data, d["l"].ctypes.data, d["l"].shape[1], d["n"].ctypes.data, d["n"].shape[1],
                      pr.ctypes.data, px.ctypes.data, pl.ctypes.data, pn.ctypes.data, st.ctypes.data, C.byref(r), C.byref(a), C.byref(b))
    assert not st.any()
    assert (pr == case["proof_r"]).all() and (px == case["proof_x"]).all()
    assert (pl == case["proof_l"]).all() and (pn == case["proof_n"]).all()
