"""CPU tier for the generic reciprocal range-proof verify device code (recip_core.h + wnla_core.h compiled for the host) against the
oracle: the u64 dimensions through the generic path, dim_nd = 32, and a non-power-of-two dim_np."""
import numpy as np
import pytest

import recip_cases
from emul.build import load


@pytest.mark.parametrize("nd,npp", [(16, 16), (32, 16), (8, 4), (12, 10)])
def test_generic_reciprocal_verify_vs_oracle(nd, npp):
    L = load()
    case = recip_cases.make(nd, npp, B=3)
    W = 4
    gens = case["g"] + b"".join(case["gv"]) + b"".join(case["gv_"]) + b"".join(case["hv"]) + b"".join(case["hv_"])
    NB = 1 + case["NG"] + case["NH"]
    tab = np.zeros(L.emul_fb_table_entries(NB, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, NB, W, tab.ctypes.data) == 0

    def run(com, proofs):
        B = com.shape[0]
        acc, st = np.zeros(B, np.uint8), np.zeros(B, np.int32)
        com, proofs = np.ascontiguousarray(com), np.ascontiguousarray(proofs)
        L.emul_recip_verify(tab.ctypes.data, W, case["NG"], case["NH"], nd, npp, case["label"], len(case["label"]), B, com.ctypes.data,
                            proofs.ctypes.data, case["rounds"], case["nl"], case["nn"], acc.ctypes.data, st.ctypes.data)
        return acc, st

    acc, st = run(case["commitments"], case["proofs"])
    assert acc.tolist() == [1, 1, 1] and not st.any()
    # tampered: final scalar, swapped commitment, c_s replaced by c_l
    P = case["proofs"].copy()
    P[0, -1] ^= 1
    P[2, 192:256] = P[2, 0:64]
    com = case["commitments"].copy()
    com[1] = case["commitments"][0]
    acc, st = run(com, P)
    exp = [recip_cases.oracle_verify(case, bytes(com[b]), bytes(P[b])) for b in range(3)]
    assert acc.tolist() == exp == [0, 0, 0] and not st.any()
    P = case["proofs"].copy()
    P[1, 70] ^= 1                                  # c_r off the curve
    acc, st = run(case["commitments"], P)
    assert st.tolist() == [0, 1, 0] and acc.tolist() == [1, 0, 1]


@pytest.mark.parametrize("nd,npp", [(16, 16), (8, 4), (12, 10), (32, 16)])
def test_generic_reciprocal_prove_is_byte_identical_to_the_oracle(nd, npp):
    """reciprocal.rs:110-146 on the device code (per-instance matrix values over the shared sparsity pattern, then the generic
    circuit and WNLA provers): proof bytes equal the reference-shaped prover's for the same witness and prover scalars.
    dim_nd = dim_np = 16 is the u64 protocol through the generic path."""
    L = load()
    B = 3
    case = recip_cases.make(nd, npp, B=B)
    W = 4
    gens = case["g"] + b"".join(case["gv"]) + b"".join(case["gv_"]) + b"".join(case["hv"]) + b"".join(case["hv_"])
    NB = 1 + case["NG"] + case["NH"]
    tab = np.zeros(L.emul_fb_table_entries(NB, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, NB, W, tab.ctypes.data) == 0
    proofs = np.zeros((B, case["proof_bytes"]), np.uint8)
    st = np.zeros(B, np.int32)
    c = {k: np.ascontiguousarray(case[k]) for k in ("commitments", "x", "s", "digits", "m", "rnd")}
    rc = L.emul_recip_prove(tab.ctypes.data, W, case["NG"], case["NH"], nd, npp, case["label"], len(case["label"]), B, c["commitments"].ctypes.data,
                            c["x"].ctypes.data, c["s"].ctypes.data, c["digits"].ctypes.data, c["m"].ctypes.data, c["rnd"].ctypes.data,
                            proofs.ctypes.data, st.ctypes.data)
    assert rc == case["proof_bytes"] and not st.any()
    assert (proofs == case["proofs"]).all()


def test_generic_reciprocal_verify_rlc_mode_equals_exact_mode():
    """The random-linear-combination mode of the generic final MSM (wnla_rlc_core.h) on the device code: 19 instances = two full
    chunks of 8 and a partial one.  A clean chunk passes its combined check; a chunk with a tampered / flagged instance and the
    partial chunk are re-checked exactly; accept bits and statuses equal exact mode's and the oracle's."""
    L = load()
    nd, npp, B = 8, 4, 19
    case = recip_cases.make(nd, npp, B=B, n_oracle=B)
    W = 4
    gens = case["g"] + b"".join(case["gv"]) + b"".join(case["gv_"]) + b"".join(case["hv"]) + b"".join(case["hv_"])
    NB = 1 + case["NG"] + case["NH"]
    tab = np.zeros(L.emul_fb_table_entries(NB, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, NB, W, tab.ctypes.data) == 0
    seed = bytes(range(100, 132))

    def run(com, proofs, rlc):
        acc, st, flags = np.zeros(B, np.uint8), np.zeros(B, np.int32), np.full(3, 9, np.uint8)
        com, proofs = np.ascontiguousarray(com), np.ascontiguousarray(proofs)
        if rlc:
            L.emul_set_rlc(seed, flags.ctypes.data)
        L.emul_recip_verify(tab.ctypes.data, W, case["NG"], case["NH"], nd, npp, case["label"], len(case["label"]), B, com.ctypes.data,
                            proofs.ctypes.data, case["rounds"], case["nl"], case["nn"], acc.ctypes.data, st.ctypes.data)
        return acc, st, flags

    acc, st, flags = run(case["commitments"], case["proofs"], True)
    assert acc.all() and not st.any() and flags.tolist() == [0, 0, 1]        # the partial chunk always goes to the exact check
    P, com = case["proofs"].copy(), case["commitments"].copy()
    P[9, -1] ^= 1                                   # n0 of instance 9 (chunk 1)
    P[12, 70] ^= 1                                  # c_r of instance 12 off the curve (chunk 1): flagged
    com[17] = case["commitments"][16]               # partial chunk
    acc0, st0, _ = run(com, P, False)
    acc1, st1, flags = run(com, P, True)
    assert (acc1 == acc0).all() and (st1 == st0).all() and flags.tolist() == [0, 1, 1]
    assert acc1.tolist() == [0 if b in (9, 12, 17) else 1 for b in range(B)] and st1[12] == 1
    for b in (8, 9, 12, 17, 18):
        rc = recip_cases.oracle_verify(case, bytes(com[b]), bytes(P[b]))
        assert int(acc1[b]) == (1 if rc == 1 else 0)


def test_generic_rlc_bucket_stage_in_front_of_the_chunks():
    """The bucket (Pippenger) stage carried over from the u64 verifier to the generic RLC mode (bucket_core.h with nb = 1 + ng + nh
    bases): superchunks of 8 instances here (the device uses 256 .. 4096).  A clean superchunk passes on its one combined check and
    its instances are accepted without any chunk-of-8 work; a superchunk holding a bad instance falls through to the chunk kernels
    and from there to the exact check.  Accept bits and statuses equal exact mode's."""
    L = load()
    nd, npp, B = 8, 4, 19
    case = recip_cases.make(nd, npp, B=B, n_oracle=B)
    W = 4
    gens = case["g"] + b"".join(case["gv"]) + b"".join(case["gv_"]) + b"".join(case["hv"]) + b"".join(case["hv_"])
    NB = 1 + case["NG"] + case["NH"]
    tab = np.zeros(L.emul_fb_table_entries(NB, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, NB, W, tab.ctypes.data) == 0
    seed = bytes(range(7, 39))

    def run(com, proofs, mode, M=8):
        acc, st, flags = np.zeros(B, np.uint8), np.zeros(B, np.int32), np.full(3, 9, np.uint8)
        passed = np.full((B + M - 1) // M, 9, np.uint8)
        com, proofs = np.ascontiguousarray(com), np.ascontiguousarray(proofs)
        if mode:
            L.emul_set_rlc(seed, flags.ctypes.data)
        if mode == 2:
            L.emul_set_rlc_superchunk(M, passed.ctypes.data)
        L.emul_recip_verify(tab.ctypes.data, W, case["NG"], case["NH"], nd, npp, case["label"], len(case["label"]), B, com.ctypes.data,
                            proofs.ctypes.data, case["rounds"], case["nl"], case["nn"], acc.ctypes.data, st.ctypes.data)
        return acc, st, flags, passed

    acc, st, flags, passed = run(case["commitments"], case["proofs"], 2)
    assert acc.all() and not st.any() and passed.tolist() == [1, 1, 1] and flags.tolist() == [0, 0, 0]   # even the ragged tail passes
    P, com = case["proofs"].copy(), case["commitments"].copy()
    P[9, -1] ^= 1                                   # superchunk 1: wrong final scalar
    P[18, 70] ^= 1                                  # superchunk 2: c_r off the curve (flagged: weight zero, rejected directly)
    acc0, st0, _, _ = run(com, P, 0)
    acc2, st2, flags, passed = run(com, P, 2)
    assert (acc2 == acc0).all() and (st2 == st0).all()
    assert passed.tolist() == [1, 0, 1] and flags.tolist() == [0, 1, 0]
    assert acc2.tolist() == [0 if b in (9, 18) else 1 for b in range(B)] and st2[18] == 1
    acc3, st3, _, passed = run(com, P, 2, M=16)     # two superchunks of 16 (the second ragged)
    assert (acc3 == acc0).all() and (st3 == st0).all() and passed.tolist() == [0, 1]


def test_fast_and_projective_round_paths_agree():
    """The generic verifiers' rounds run on affine window tables of all round points (one table-build pass, Jacobian accumulators,
    signed 5-bit windows) by default; the projective tables + complete additions they replaced are still there
    (emul_set_generic_slow_rounds / BPPP_GENERIC_SLOW_ROUNDS).  Same verdicts and statuses, tampered and malformed round points
    included."""
    L = load()
    nd, npp, B = 12, 10, 6
    case = recip_cases.make(nd, npp, B=B)
    W = 4
    gens = case["g"] + b"".join(case["gv"]) + b"".join(case["gv_"]) + b"".join(case["hv"]) + b"".join(case["hv_"])
    NB = 1 + case["NG"] + case["NH"]
    tab = np.zeros(L.emul_fb_table_entries(NB, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, NB, W, tab.ctypes.data) == 0
    P, com = case["proofs"].copy(), case["commitments"].copy()
    r = case["rounds"]
    P[1, 256 + 5] ^= 0x10                          # r[0] coordinate: off the curve
    P[2, 256 + 64 * r + 64 * (r - 1) + 40] ^= 1    # x[rounds - 1] (the first round's X): off the curve
    P[3, 256:320] = P[3, 256 + 64 * r:320 + 64 * r]  # r[0] := x[0]: a valid but wrong point
    P[4, 256:320] = 0                              # the identity as a round point
    res = []
    for slow in (0, 1):
        L.emul_set_generic_slow_rounds(slow)
        acc, st = np.zeros(B, np.uint8), np.zeros(B, np.int32)
        L.emul_recip_verify(tab.ctypes.data, W, case["NG"], case["NH"], nd, npp, case["label"], len(case["label"]), B, com.ctypes.data,
                            P.ctypes.data, r, case["nl"], case["nn"], acc.ctypes.data, st.ctypes.data)
        res.append((acc.copy(), st.copy()))
    L.emul_set_generic_slow_rounds(0)
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all()
    assert res[0][0].tolist() == [1, 0, 0, 0, 0, 1] and res[0][1].tolist() == [0, 1, 1, 0, 0, 0]
    for b in range(B):
        rc = recip_cases.oracle_verify(case, bytes(com[b]), bytes(P[b]))
        assert int(res[0][0][b]) == (1 if rc == 1 else 0) and (int(res[0][1][b]) != 0) == (rc < 0)


def test_c0_points_that_meet_in_the_window_sum():
    """The variable-base part of C0 runs on affine window tables (recip_c0_tables / recip_c0_var) with incomplete additions and a
    complete re-run when one of them met its exception: equal, opposite and identity points among c_l, c_r, c_o, c_s and a commitment
    that cancels the proof's r (V + r = identity) give the verdicts of the projective path and of the oracle."""
    L = load()
    nd, npp, B = 12, 10, 7
    case = recip_cases.make(nd, npp, B=B)
    W = 4
    gens = case["g"] + b"".join(case["gv"]) + b"".join(case["gv_"]) + b"".join(case["hv"]) + b"".join(case["hv_"])
    NB = 1 + case["NG"] + case["NH"]
    tab = np.zeros(L.emul_fb_table_entries(NB, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, NB, W, tab.ctypes.data) == 0
    P, com = case["proofs"].copy(), case["commitments"].copy()
    r = case["rounds"]
    p = 2**256 - 2**32 - 977

    def neg(xy):
        y = int.from_bytes(bytes(xy[32:]), "big")
        return np.frombuffer(bytes(xy[:32]) + ((p - y) % p).to_bytes(32, "big"), np.uint8)

    P[1, 64:128] = P[1, 0:64]                      # c_r := c_l
    P[2, 64:128] = neg(P[2, 0:64])                 # c_r := -c_l
    P[3, 192:256] = 0                              # c_s := identity
    com[4] = neg(P[4, 256 + 128 * r:320 + 128 * r])  # V := -r, so V + r is the identity
    P[5, 128:192] = P[5, 0:64]                     # c_o := c_l
    P[6, 0:256] = np.tile(P[6, 192:256], 4)        # all four the same point
    res = []
    for slow in (0, 1):
        L.emul_set_generic_slow_rounds(slow)
        acc, st = np.zeros(B, np.uint8), np.zeros(B, np.int32)
        L.emul_recip_verify(tab.ctypes.data, W, case["NG"], case["NH"], nd, npp, case["label"], len(case["label"]), B, com.ctypes.data,
                            P.ctypes.data, r, case["nl"], case["nn"], acc.ctypes.data, st.ctypes.data)
        res.append((acc.copy(), st.copy()))
    L.emul_set_generic_slow_rounds(0)
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all()
    assert res[0][0].tolist() == [1, 0, 0, 0, 0, 0, 0] and not res[0][1].any()
    for b in range(B):
        assert recip_cases.oracle_verify(case, bytes(com[b]), bytes(P[b])) == int(res[0][0][b])
