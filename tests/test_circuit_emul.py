"""CPU tier for the generic ArithmeticCircuit::verify device code (circuit_core.h + wnla_core.h compiled for the host) against
the oracle: the reference's own `ac_works` statement (tests.rs:45-136), k = 2 with every partition type, the f_m path, and the
f_l + f_m shape whose intermediate values (C0, c) are compared with the big-integer oracle because neither side accepts it."""
import numpy as np
import pytest

import bppp_oracle as O
import circuit_cases
from emul.build import load


def _table(L, case, W=4):
    gens = case["g"] + b"".join(case["gv"]) + b"".join(case["gv_"]) + b"".join(case["hv"]) + b"".join(case["hv_"])
    NB = 1 + case["NG"] + case["NH"]
    tab = np.zeros(L.emul_fb_table_entries(NB, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, NB, W, tab.ctypes.data) == 0
    return tab, W


def _run(L, case, tab, W, com, proofs, want_mid=False):
    B = com.shape[0]
    acc, st = np.zeros(B, np.uint8), np.zeros(B, np.int32)
    com, proofs = np.ascontiguousarray(com), np.ascontiguousarray(proofs)
    c0 = np.zeros((B, 64), np.uint8)
    cv = np.zeros((B, case["NH"], 32), np.uint8)
    p = case["parts"]
    rc = L.emul_circuit_verify(tab.ctypes.data, W, case["NG"], case["NH"], case["dims"], int(case["f_l"]), int(case["f_m"]),
                               case["Wm_bytes"], case["Wl_bytes"], case["am_bytes"], case["al_bytes"], p["LO"].ctypes.data,
                               p["LL"].ctypes.data, p["LR"].ctypes.data, p["NO"].ctypes.data, case["label"], len(case["label"]), B,
                               com.ctypes.data, proofs.ctypes.data, case["rounds"], case["pl"], case["pn"], acc.ctypes.data,
                               st.ctypes.data, c0.ctypes.data if want_mid else None, cv.ctypes.data if want_mid else None)
    assert rc == 0
    return (acc, st, c0, cv) if want_mid else (acc, st)


@pytest.mark.parametrize("name", ["ac_works", "mixed_k2", "fm_nv1"])
def test_generic_circuit_verify_vs_oracle(name):
    L = load()
    case = circuit_cases.make(name, B=3)
    tab, W = _table(L, case)
    acc, st = _run(L, case, tab, W, case["commitments"], case["proofs"])
    assert acc.tolist() == [1, 1, 1] and not st.any()
    # tampered: last scalar, a commitment swapped between instances, c_s replaced by c_l
    P = case["proofs"].copy()
    P[0, -1] ^= 1
    P[2, 192:256] = P[2, 0:64]
    com = case["commitments"].copy()
    com[1, 0] = case["commitments"][0, 0]
    acc, st = _run(L, case, tab, W, com, P)
    exp = [circuit_cases.oracle_verify(case, com[b].tobytes(), P[b].tobytes()) for b in range(3)]
    assert acc.tolist() == exp == [0, 0, 0] and not st.any()
    P = case["proofs"].copy()
    P[1, 70] ^= 1                                  # c_r off the curve
    acc, st = _run(L, case, tab, W, case["commitments"], P)
    assert st.tolist() == [0, 1, 0] and acc.tolist() == [1, 0, 1]


def _oracle_circuit(case):
    pts = lambda lst: [O.pt_from_xy64(b) for b in lst]
    part = lambda typ, j: (None if case["part"][typ][j] < 0 else case["part"][typ][j])
    return O.ArithmeticCircuit(dim_nm=case["nm"], dim_no=case["no"], k=case["k"], dim_nl=case["nl"], dim_nv=case["nv"], dim_nw=case["nw"],
                               g=O.pt_from_xy64(case["g"]), g_vec=pts(case["gv"]), h_vec=pts(case["hv"]), W_m=case["W_m"], W_l=case["W_l"],
                               a_m=case["a_m"], a_l=case["a_l"], f_l=case["f_l"], f_m=case["f_m"], g_vec_=pts(case["gv_"]),
                               h_vec_=pts(case["hv_"]), partition=part)


@pytest.mark.parametrize("name", ["fl_fm", "mixed_k2"])
def test_circuit_intermediates_vs_bigint_oracle(name):
    """C0 (circuit.rs:230-235) and the c vector (circuit.rs:208-229) of the device code against the Python big-integer
    restatement's trace -- the check that still bites where the accept bit is 0 on both sides."""
    L = load()
    case = circuit_cases.make(name, B=2)
    tab, W = _table(L, case)
    acc, st, c0, cv = _run(L, case, tab, W, case["commitments"], case["proofs"], want_mid=True)
    circ = _oracle_circuit(case)
    R = case["rounds"]
    for b in range(2):
        pr = case["proofs"][b].tobytes()
        P = lambda i: O.pt_from_xy64(pr[64 * i:64 * i + 64])
        sc_at = lambda o: int.from_bytes(pr[o:o + 32], "big")
        off = 64 * (4 + 2 * R)
        proof = O.CircuitProof(c_l=P(0), c_r=P(1), c_o=P(2), c_s=P(3), r=[P(4 + i) for i in range(R)], x=[P(4 + R + i) for i in range(R)],
                               l=[sc_at(off + 32 * i) for i in range(case["pl"])],
                               n=[sc_at(off + 32 * case["pl"] + 32 * i) for i in range(case["pn"])])
        v = [O.pt_from_xy64(case["commitments"][b, i].tobytes()) for i in range(case["k"])]
        trace = []
        ok = circ.verify(v, O.Transcript(case["label"]), proof, trace)
        tr = dict((k_, v_) for k_, v_ in trace if isinstance(k_, str))
        assert c0[b].tobytes() == O.pt_to_xy64(tr["C0"])
        exp_c = list(tr["c"]) + [0] * (case["NH"] - len(tr["c"]))
        assert [int.from_bytes(cv[b, i].tobytes(), "big") for i in range(case["NH"])] == exp_c
        assert int(acc[b]) == int(ok) == circuit_cases.oracle_verify(case, case["commitments"][b].tobytes(), pr)
    assert not st.any()


@pytest.mark.parametrize("name", ["ac_works", "mixed_k2", "fm_nv1", "fl_fm"])
def test_generic_circuit_prove_is_byte_identical_to_the_oracle(name):
    """circuit.rs:260-556 + wnla.rs:125-190 on the device code: for the same witness and the same sequence of prover scalars the
    proof bytes must equal the reference-shaped prover's (incl. the f_l + f_m shape, whose proofs neither side's verifier accepts)."""
    L = load()
    B = 2
    case = circuit_cases.make(name, B=B)
    tab, W = _table(L, case)
    p = case["parts"]
    proofs = np.zeros((B, case["proof_bytes"]), np.uint8)
    st = np.zeros(B, np.int32)
    c = {k_: np.ascontiguousarray(case[k_]) for k_ in ("commitments", "v_bytes", "s_v", "wl_bytes", "wr_bytes", "wo_bytes", "rnd")}
    rc = L.emul_circuit_prove(tab.ctypes.data, W, case["NG"], case["NH"], case["dims"], int(case["f_l"]), int(case["f_m"]), case["Wm_bytes"],
                              case["Wl_bytes"], case["am_bytes"], case["al_bytes"], p["LO"].ctypes.data, p["LL"].ctypes.data,
                              p["LR"].ctypes.data, p["NO"].ctypes.data, case["label"], len(case["label"]), B, c["commitments"].ctypes.data,
                              c["v_bytes"].ctypes.data, c["s_v"].ctypes.data, c["wl_bytes"].ctypes.data, c["wr_bytes"].ctypes.data,
                              c["wo_bytes"].ctypes.data, c["rnd"].ctypes.data, proofs.ctypes.data, st.ctypes.data)
    assert rc == case["proof_bytes"] and not st.any()
    for b in range(B):
        got, exp = proofs[b].tobytes(), case["proofs"][b].tobytes()
        assert got[:256] == exp[:256], "c_l, c_r, c_o, c_s"
        assert got == exp
