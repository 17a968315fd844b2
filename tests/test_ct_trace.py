"""The property behind "ct_prover", tested as a property: which fixed-base table entries the provers' sums over SECRET scalars read must
not depend on the secrets.  The host build of the device code (tests/emul) reports every table entry such a sum requests
(fb_core.h: FB_TRACE); for two different secrets under the same public inputs the two sequences must be IDENTICAL in the
"ct_prover" forms -- u64 prover, and the generic WNLA / circuit / reciprocal provers -- while in the default forms they differ (the
digit-addressed gathers: the side channel the mode closes).  Same proof bytes either way, equal to the oracle provers'.
Reference: k256's constant-time `ProjectivePoint * Scalar` at reciprocal.rs:118, circuit.rs:336-345,469-470, wnla.rs:152-160."""
import ctypes as C

import numpy as np
import pytest

import circuit_cases
import recip_cases
import wnla_cases
import workload
from emul.build import load


@pytest.fixture(scope="module")
def L():
    return load()


def _traced(L, ct, run):
    """run() under recording, in the ct or the default form -> the sequence of table indices its secret sums read"""
    L.emul_set_prove_ct(1 if ct else 0)
    try:
        L.emul_trace_begin()
        run()
        n = L.emul_trace_end(None, 0)
        L.emul_trace_begin()
        run()
        buf = np.zeros(n, np.uint64)
        assert L.emul_trace_end(buf.ctypes.data, n) == n          # (the same run twice reads the same entries)
    finally:
        L.emul_set_prove_ct(0)
    return buf


def _check(L, run_a, run_b):
    """two secrets, same public inputs: identical reads with ct_prover, different ones without; the ct form reads every entry of a window"""
    ta, tb = _traced(L, True, run_a), _traced(L, True, run_b)
    assert len(ta) > 0 and len(ta) % 15 == 0 and len(ta) == len(tb) and (ta == tb).all()
    w = ta.reshape(-1, 15)
    assert (w[:, 1:] - w[:, :-1] == 1).all()          # a window's 15 entries, in order, every time
    fa, fb = _traced(L, False, run_a), _traced(L, False, run_b)
    assert len(fa) > 0 and len(fa) == len(fb) and (fa != fb).any()
    assert (fa != fb).mean() > 0.5                     # digit-addressed: almost every read goes somewhere else


def test_u64_prover(L, oracle_c):
    gens = workload.generators()
    W = 4
    tab = np.zeros(L.emul_fb_table_entries(49, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 49, W, tab.ctypes.data) == 0
    x = np.ascontiguousarray(workload.values(2, first=4))
    s, rnd = np.ascontiguousarray(workload.blindings(2, first=4)), np.ascontiguousarray(workload.prover_randomness(2, first=4))
    out = {}

    def run(i):
        def f():
            proofs, V, st = np.zeros((1, 928), np.uint8), np.zeros((1, 64), np.uint8), np.zeros(1, np.int32)
            assert 0 == L.emul_u64_prove_batch(tab.ctypes.data, W, workload.LABEL, len(workload.LABEL), 1, x[i:i + 1].ctypes.data, s[i:i + 1].ctypes.data,
                                               rnd[i:i + 1].ctypes.data, proofs.ctypes.data, V.ctypes.data, st.ctypes.data)
            assert not st.any()
            out[i] = (proofs.copy(), V.copy())
        return f
    _check(L, run(0), run(1))
    op, ov = oracle_c.u64_prove_batch(gens, workload.LABEL, x, s, rnd, nthreads=2)
    for i in (0, 1):          # (the last run of each was in the default form; the ct form's bytes are checked in test_core_emul.py)
        assert (out[i][0][0] == op[i]).all() and (out[i][1][0] == ov[i]).all()


def test_generic_wnla_prover(L):
    ng, nh, B = 8, 8, 2
    case = wnla_cases.make(ng, nh, B=B, mu_is_rho_sq=True)
    W = 4
    gens = case["g"] + b"".join(case["gv"]) + b"".join(case["hv"])
    tab = np.zeros(L.emul_fb_table_entries(1 + ng + nh, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 1 + ng + nh, W, tab.ctypes.data) == 0
    d = {k: np.ascontiguousarray(case[k]) for k in ("commitments", "c", "rho", "mu", "l", "n")}
    got = {}

    def run(i):
        def f():
            r, a, b = C.c_int(), C.c_int(), C.c_int()
            pr, px = np.zeros((1, case["rounds"], 64), np.uint8), np.zeros((1, case["rounds"], 64), np.uint8)
            pl, pn = np.zeros((1, case["nl"], 32), np.uint8), np.zeros((1, case["nn"], 32), np.uint8)
            st = np.zeros(1, np.int32)
            sl = {k: np.ascontiguousarray(v[i:i + 1]) for k, v in d.items()}
            L.emul_wnla_prove(tab.ctypes.data, W, ng, nh, case["label"], len(case["label"]), 1, sl["commitments"].ctypes.data, sl["c"].ctypes.data,
                              sl["rho"].ctypes.data, sl["mu"].ctypes.data, sl["l"].ctypes.data, sl["l"].shape[1], sl["n"].ctypes.data, sl["n"].shape[1],
                              pr.ctypes.data, px.ctypes.data, pl.ctypes.data, pn.ctypes.data, st.ctypes.data, C.byref(r), C.byref(a), C.byref(b))
            assert not st.any()
            assert (pr[0] == case["proof_r"][i]).all() and (px[0] == case["proof_x"][i]).all()          # byte-identical in BOTH forms
            assert (pl[0] == case["proof_l"][i]).all() and (pn[0] == case["proof_n"][i]).all()
            got[i] = True
        return f
    _check(L, run(0), run(1))
    assert got == {0: True, 1: True}


@pytest.mark.parametrize("name", ["ac_works", "mixed_k2"])
def test_generic_circuit_prover(L, name):
    B = 2
    case = circuit_cases.make(name, B=B)
    import test_circuit_emul as TCE
    tab, W = TCE._table(L, case)
    p = case["parts"]
    c = {k_: np.ascontiguousarray(case[k_]) for k_ in ("commitments", "v_bytes", "s_v", "wl_bytes", "wr_bytes", "wo_bytes", "rnd")}

    def run(i):
        def f():
            proofs, st = np.zeros((1, case["proof_bytes"]), np.uint8), np.zeros(1, np.int32)
            sl = {k: np.ascontiguousarray(v[i:i + 1]) for k, v in c.items()}
            rc = L.emul_circuit_prove(tab.ctypes.data, W, case["NG"], case["NH"], case["dims"], int(case["f_l"]), int(case["f_m"]), case["Wm_bytes"],
                                      case["Wl_bytes"], case["am_bytes"], case["al_bytes"], p["LO"].ctypes.data, p["LL"].ctypes.data,
                                      p["LR"].ctypes.data, p["NO"].ctypes.data, case["label"], len(case["label"]), 1, sl["commitments"].ctypes.data,
                                      sl["v_bytes"].ctypes.data, sl["s_v"].ctypes.data, sl["wl_bytes"].ctypes.data, sl["wr_bytes"].ctypes.data,
                                      sl["wo_bytes"].ctypes.data, sl["rnd"].ctypes.data, proofs.ctypes.data, st.ctypes.data)
            assert rc == case["proof_bytes"] and not st.any()
            assert proofs[0].tobytes() == case["proofs"][i].tobytes()          # byte-identical in both forms
        return f
    _check(L, run(0), run(1))


def test_generic_reciprocal_prover(L):
    nd, npp, B = 8, 4, 2
    case = recip_cases.make(nd, npp, B=B)
    W = 4
    gens = case["g"] + b"".join(case["gv"]) + b"".join(case["gv_"]) + b"".join(case["hv"]) + b"".join(case["hv_"])
    NB = 1 + case["NG"] + case["NH"]
    tab = np.zeros(L.emul_fb_table_entries(NB, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, NB, W, tab.ctypes.data) == 0
    c = {k: np.ascontiguousarray(case[k]) for k in ("commitments", "x", "s", "digits", "m", "rnd")}

    def run(i):
        def f():
            proofs, st = np.zeros((1, case["proof_bytes"]), np.uint8), np.zeros(1, np.int32)
            sl = {k: np.ascontiguousarray(v[i:i + 1]) for k, v in c.items()}
            rc = L.emul_recip_prove(tab.ctypes.data, W, case["NG"], case["NH"], nd, npp, case["label"], len(case["label"]), 1, sl["commitments"].ctypes.data,
                                    sl["x"].ctypes.data, sl["s"].ctypes.data, sl["digits"].ctypes.data, sl["m"].ctypes.data, sl["rnd"].ctypes.data,
                                    proofs.ctypes.data, st.ctypes.data)
            assert rc == case["proof_bytes"] and not st.any()
            assert (proofs[0] == case["proofs"][i]).all()
        return f
    _check(L, run(0), run(1))
