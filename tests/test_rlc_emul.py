"""CPU tier for the optional random-linear-combination batch mode (rlc_core.h compiled for the host): weights, weighted
commitments, the combined chunk check and its exact re-check.  Accept bits must equal exact mode's (the oracle's)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import bppp_oracle as O
import workload
from emul.build import load

SEED = bytes(range(7, 39))


def _weight_halves(seed: bytes, t: int):
    st = [int.from_bytes(seed[8 * i:8 * i + 8], "little") for i in range(4)] + [t, int.from_bytes(b"BPPP_RLC", "little")] + [0] * 19
    out = O.keccak_f1600(st)
    return out[0], out[1]


@pytest.fixture(scope="module")
def setup():
    L = load()
    n = 27                                          # 3 full chunks of 8 + a partial one
    gens, V, P, _ = workload.make_batch(n, first=500)
    W = 4
    tab = np.zeros(L.emul_fb_table_entries(49, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 49, W, tab.ctypes.data) == 0
    return L, gens, V, P, tab, W, n


def _run(L, tab, W, V, P, seed=SEED, want_mid=False):
    n = V.shape[0]
    acc, st = np.zeros(n, np.uint8), np.zeros(n, np.int32)
    re = C.c_int(0)
    ab = np.zeros((n, 2), np.uint64)
    lhs = np.zeros((n, 64), np.uint8)
    V, P = np.ascontiguousarray(V), np.ascontiguousarray(P)
    rc = L.emul_u64_verify_batch_rlc(tab.ctypes.data, W, workload.LABEL, len(workload.LABEL), n, V.ctypes.data, P.ctypes.data, seed,
                                     acc.ctypes.data, st.ctypes.data, C.byref(re), ab.ctypes.data if want_mid else None,
                                     lhs.ctypes.data if want_mid else None)
    assert rc == 0
    return (acc, st, re.value, ab, lhs) if want_mid else (acc, st, re.value)


def test_rlc_weights_and_weighted_commitments(setup, oracle_c):
    L, gens, V, P, tab, W, n = setup
    acc, st, re, ab, lhs = _run(L, tab, W, V[:9], P[:9], want_mid=True)
    for t in range(9):
        a, b = _weight_halves(SEED, t)
        assert (int(ab[t, 0]), int(ab[t, 1])) == (a, b)            # keyed PRF: one Keccak-f[1600] of seed | index | tag
        w = (a + b * O.LAMBDA) % O.N
        rc, tr = oracle_c.u64_verify(gens, workload.LABEL, bytes(V[t]), bytes(P[t]), trace=True)
        C4 = O.pt_from_xy64(tr[320 + 64 * 5:320 + 64 * 6])         # the oracle's final commitment
        assert lhs[t].tobytes() == O.pt_to_xy64(O.pt_mul(C4, w))


def test_rlc_mode_accept_bits_equal_exact_mode(setup, oracle_c):
    L, gens, V, P, tab, W, n = setup
    # all valid: three full chunks pass combined, the partial chunk (3 proofs) is re-checked exactly
    acc, st, re = _run(L, tab, W, V, P)
    assert acc.all() and not st.any() and re == 1
    # corruptions in chunks 0 and 2 (two in the same chunk), a malformed proof in chunk 1, the last proof of the partial chunk
    Pc = P.copy()
    Pc[3, 900] ^= 1          # final scalar n
    Pc[17, 5] ^= 0x40        # c_l: still decodes? (x change usually leaves the curve -> status) either way oracle decides
    Pc[18, 840] ^= 2         # l0
    Pc[9, 70] ^= 1           # malformed point
    Pc[26, 927] ^= 1
    Vc = V.copy()
    Vc[20] = V[21]           # wrong commitment
    acc, st, re = _run(L, tab, W, Vc, Pc)
    exp_acc, exp_st = [], []
    for t in range(n):
        rc = oracle_c.u64_verify(gens, workload.LABEL, bytes(Vc[t]), bytes(Pc[t]))
        exp_acc.append(1 if rc == 1 else 0)
        exp_st.append(1 if rc < 0 else 0)
    assert acc.tolist() == exp_acc
    assert [int(x != 0) for x in st] == exp_st
    assert re == 4           # every chunk holds a bad proof here (the partial one is always re-checked)
    # a different seed changes the weights, not the verdicts
    acc2, _, _ = _run(L, tab, W, Vc, Pc, seed=bytes(32))
    assert acc2.tolist() == exp_acc


def test_bucket_stage_superchunk_verdicts(setup):
    """bucket_core.h (single-thread form): a superchunk passes iff every unflagged proof in it is valid; a flagged (malformed)
    proof gets weight zero and does not fail its superchunk; the last superchunk may be partial."""
    L, gens, V, P, tab, W, n = setup
    M = 8

    def run(Vx, Px):
        nx = Vx.shape[0]
        ns = (nx + M - 1) // M
        passed, st = np.zeros(ns, np.uint8), np.zeros(nx, np.int32)
        Vx, Px = np.ascontiguousarray(Vx), np.ascontiguousarray(Px)
        assert L.emul_u64_bucket_stage(tab.ctypes.data, W, workload.LABEL, len(workload.LABEL), nx, Vx.ctypes.data, Px.ctypes.data, SEED,
                                       M, passed.ctypes.data, st.ctypes.data) == 0
        return passed.tolist(), st
    passed, st = run(V, P)
    assert passed == [1, 1, 1, 1] and not st.any()                 # 27 proofs: three full superchunks and one of 3
    P2 = P.copy()
    P2[9, 900] ^= 1          # wrong n0 in superchunk 1
    P2[17, 3] ^= 0x40        # c_l off the curve in superchunk 2: flagged, weight 0, the rest of the superchunk still passes
    passed, st = run(V, P2)
    assert passed == [1, 0, 1, 1] and st[17] == 1 and st[9] == 0
    V3 = V.copy()
    V3[26] = V[0]            # someone else's commitment in the partial superchunk
    passed, st = run(V3, P)
    assert passed == [1, 1, 1, 0]


def test_chunks_of_32_equal_exact_mode(oracle_c):
    """Round 5: the chunk stage on 32 proofs per chunk (rlc_core.h: RlcWs::chunk) -- one 49-base sum per 32 proofs instead of per 8.
    Two full chunks and a partial one: all valid -> both full chunks pass combined; one bad proof fails its whole chunk of 32, which is
    re-checked exactly, so the accept bits are the oracle's either way."""
    L = load()
    n = 70
    gens, V, P, _ = workload.make_batch(n, first=900)
    W = 4
    tab = np.zeros(L.emul_fb_table_entries(49, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 49, W, tab.ctypes.data) == 0
    L.emul_set_rlc_chunk(32)
    try:
        acc, st, re = _run(L, tab, W, V, P)
        assert acc.all() and not st.any() and re == 1          # only the partial chunk (6 proofs) is re-checked
        Pc = P.copy()
        Pc[40, 900] ^= 1          # chunk 1
        Pc[69, 5] ^= 0x40         # the partial chunk
        acc, st, re = _run(L, tab, W, V, Pc)
        exp = [1 if oracle_c.u64_verify(gens, workload.LABEL, bytes(V[t]), bytes(Pc[t])) == 1 else 0 for t in range(n)]
        assert acc.tolist() == exp and exp[40] == 0 and re == 2
    finally:
        L.emul_set_rlc_chunk(8)


def test_group_sizes_follow_the_previous_calls_reject_rate():
    """plan_core.h: plan_rlc -- the bucket stage's superchunk and the chunk size of an RLC call, from what the previous call rejected."""
    L = load()

    def plan(auto_m, is_auto, opt, rate):
        out = (C.c_uint * 2)()
        L.emul_plan_rlc(auto_m, is_auto, opt, rate, out)
        return out[0], out[1]
    assert plan(4096, 1, 0, -1.0) == (4096, 8)            # no history: round 4's choice
    assert plan(4096, 1, 0, 0.0) == (4096, 32)            # nothing rejected: big superchunks, and chunks of 32 behind them
    assert plan(4096, 1, 0, 1 / 65536) == (4096, 32)
    assert plan(4096, 1, 0, 1 / 8192) == (2048, 32)       # r M <= 0.3
    assert plan(4096, 1, 0, 1 / 4096) == (1024, 32)
    assert plan(4096, 1, 0, 1 / 1024) == (0, 32)          # every superchunk would fail: no bucket stage; a chunk of 32 still passes 97 %
    assert plan(4096, 1, 0, 1 / 256) == (0, 32)
    assert plan(4096, 1, 0, 1 / 255) == (0, 8)
    assert plan(4096, 1, 0, 0.5) == (0, 8)
    assert plan(512, 1, 0, 1 / 2048) == (512, 32)         # small batches: the automatic size is already small
    assert plan(512, 1, 0, 1 / 1000) == (0, 32)
    assert plan(4096, 0, 0, 1 / 1024) == (4096, 32)       # an explicit superchunk is taken as it is
    assert plan(72, 0, 0, 0.0) == (72, 8)                 # ... and a superchunk must be made of whole chunks
    assert plan(0, 0, 0, 0.0) == (0, 32)
    assert plan(4096, 1, 8, 0.0) == (4096, 8) and plan(4096, 1, 32, 0.5) == (0, 32)
