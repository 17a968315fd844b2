"""GPU tests of the optional random-linear-combination batch mode (bppp_u64_verify_batch_rlc_device): accept bits, statuses and the
reject count must equal exact mode's and the oracle's on valid, corrupted and malformed proofs, for full and partial chunks."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    from bp_pp_amd import U64RangeProofProtocol
    import workload
    n = 1000 + 5                                     # 125 full chunks + a partial one
    gens, V, P, _ = workload.make_batch(n, first=9000)
    g, gv, hv = workload.split_generators(gens)
    p = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=16)
    yield torch, p, gens, V, P, n
    p.close()


def _run(torch, proto, V, P, seed):
    import workload
    n = V.shape[0]
    dV, dP = torch.from_numpy(np.ascontiguousarray(V)).cuda(), torch.from_numpy(np.ascontiguousarray(P)).cuda()
    dA = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
    dS = torch.zeros(n, dtype=torch.int32, device="cuda")
    dR = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()   # inputs ready; the context runs on its own (non-blocking) stream, joined by proto.synchronize()
    if seed is None:
        proto.verify_batch_device(workload.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, dR.data_ptr())
    else:
        proto.verify_batch_rlc_device(workload.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), seed, dS.data_ptr(), dR.data_ptr())
    torch.cuda.synchronize()
    return dA.cpu().numpy(), dS.cpu().numpy(), int(dR.item())


def test_rlc_all_valid(env):
    torch, proto, gens, V, P, n = env
    acc, st, rej = _run(torch, proto, V, P, os.urandom(32))
    assert acc.tolist() == [1] * n and not st.any() and rej == 0


def test_rlc_matches_exact_mode_and_oracle_on_bad_proofs(env, oracle_c):
    import workload
    torch, proto, gens, V, P, n = env
    Pc, Vc = P.copy(), V.copy()
    rng = np.random.default_rng(3)
    bad = sorted(set(int(i) for i in rng.integers(0, n, 40)) | {0, 7, 8, n - 1})
    for k, i in enumerate(bad):
        kind = k % 4
        if kind == 0:
            Pc[i, 896 + int(rng.integers(0, 32))] ^= 1            # scalar n0
        elif kind == 1:
            Pc[i, 64 * int(rng.integers(4, 12)) + 31] ^= 1         # a round point's x: leaves the curve -> status flag
        elif kind == 2:
            Vc[i] = V[(i + 1) % n]                                # commitment of another proof
        else:
            Pc[i, 832 + int(rng.integers(0, 32))] ^= 0x10          # scalar l0
    e_acc, e_st, e_rej = _run(torch, proto, Vc, Pc, None)         # exact mode
    for seed in (bytes(32), os.urandom(32)):
        acc, st, rej = _run(torch, proto, Vc, Pc, seed)
        assert acc.tolist() == e_acc.tolist() and st.tolist() == e_st.tolist() and rej == e_rej
    assert e_rej == len(bad) and not e_acc[bad].any()
    # and against the oracle on the touched chunks
    for i in bad[:12]:
        rc = oracle_c.u64_verify(gens, workload.LABEL, bytes(Vc[i]), bytes(Pc[i]))
        assert int(e_acc[i]) == (1 if rc == 1 else 0)


def test_rlc_host_buffer_entry_point(env):
    import workload
    torch, proto, gens, V, P, n = env
    Pc = P.copy()
    Pc[5, 900] ^= 1
    acc, st = proto.verify_batch_rlc(V, Pc, workload.LABEL, os.urandom(32))
    exp, _ = proto.verify_batch(V, Pc, workload.LABEL)
    assert (acc == exp).all() and not st.any() and acc[5] == 0 and acc.sum() == n - 1


def test_rlc_small_batches(env):
    torch, proto, gens, V, P, n = env
    for m in (1, 7, 8, 9, 16):
        acc, st, rej = _run(torch, proto, V[:m], P[:m], os.urandom(32))
        assert acc.tolist() == [1] * m and rej == 0


@pytest.mark.parametrize("super_m", [0, 64, 256, 4096])
def test_bucket_stage_agrees_with_exact_mode(env, super_m):
    """The bucket (Pippenger) stage in front of the chunk-of-8 stage (bucket_core.h), for several superchunk sizes (0 = stage off):
    accept bits, statuses and reject count equal exact mode's; with every proof valid nothing falls through; with bad proofs only
    their superchunks fall through (k_rlc_* timings show which stage did the work)."""
    torch, proto, gens, V, P, n = env
    proto.set_option("rlc_superchunk", super_m)
    try:
        proto.enable_timing(True)
        proto.timings(reset=True)
        acc, st, rej = _run(torch, proto, V, P, os.urandom(32))
        t = proto.timings(reset=True)
        assert acc.tolist() == [1] * n and not st.any() and rej == 0
        if super_m:
            assert t["k_bkt_accumulate"]["launches"] == 1 and t["k_bkt_check"]["launches"] == 1
        else:
            assert t["k_bkt_accumulate"]["launches"] == 0
        Pc, Vc = P.copy(), V.copy()
        Pc[3, 900] ^= 1                       # wrong n0
        Pc[70, 64 * 5 + 31] ^= 1              # r[1] off the curve: status flag
        Vc[200] = V[201]                      # someone else's commitment
        Pc[n - 1, 840] ^= 4                   # l0 of the last proof (partial superchunk / partial chunk of 8)
        e_acc, e_st, e_rej = _run(torch, proto, Vc, Pc, None)
        assert e_rej == 4 and e_st[70] == 1
        for seed in (bytes(32), os.urandom(32)):
            acc, st, rej = _run(torch, proto, Vc, Pc, seed)
            assert acc.tolist() == e_acc.tolist() and st.tolist() == e_st.tolist() and rej == e_rej
    finally:
        proto.enable_timing(False)
        proto.set_option("rlc_superchunk", 4096)
    with pytest.raises(Exception):
        proto.set_option("rlc_superchunk", 100)        # not a multiple of 8 / out of range values are refused


@pytest.mark.parametrize("every", [0, 4096, 1024, 64])
def test_group_sizes_adapt_and_accept_bits_stay_exact(oracle_c, every):
    """Round 5: an RLC call sizes its groups from what the previous RLC call on the context rejected (plan_core.h: plan_rlc) -- chunks of
    32 instead of 8 while at most one proof in 256 was bad, smaller superchunks or no bucket stage when most superchunks would fail.
    At corruption rates 0, 1/4096, 1/1024 and 1/64: the first call (no history) and the following ones (with it) give exact mode's accept
    bits, statuses and reject count, which are the oracle's; and the calls really took the sizes the rate asks for."""
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    n = 20000 + 37
    gens, V, P, _ = workload.make_batch(n, first=12000)
    if every:
        P, expect = workload.corrupt(P, V, every=every)
    else:
        expect = np.ones(n, np.uint8)
    P = P.copy()
    P[n - 1, 64 * 6 + 31] ^= 1                 # and one malformed proof (a status flag), in the last, partial chunk
    g, gv, hv = workload.split_generators(gens)
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=16)
    try:
        assert proto.get_option("rlc_has_history") == 0 and proto.get_option("rlc_chunk") == 0
        e_acc, e_st, e_rej = _run(torch, proto, V, P, None)                       # exact mode
        clean = np.ones(n, bool)
        clean[n - 1] = False
        assert (e_acc[clean] == expect[clean]).all() and e_acc[n - 1] == 0 and e_st[n - 1] != 0
        for i in list(np.nonzero(e_acc == 0)[0][:6]) + [0, 1, n - 2]:
            rc = oracle_c.u64_verify(gens, workload.LABEL, bytes(V[i]), bytes(P[i]))
            assert int(e_acc[i]) == (1 if rc == 1 else 0)
        rate = e_rej / n
        used = []
        for call in range(3):
            acc, st, rej = _run(torch, proto, V, P, os.urandom(32))
            assert acc.tolist() == e_acc.tolist() and st.tolist() == e_st.tolist() and rej == e_rej
            used.append((proto.get_option("last_rlc_superchunk"), proto.get_option("last_rlc_chunk")))
            assert proto.get_option("rlc_has_history") == 1 and abs(proto.get_option("rlc_reject_ppm") - rate * 1e6) <= 1
        assert used[0] == (256, 8)                                   # no history: the automatic superchunk of this batch size, chunks of 8
        want_chunk = 32 if rate * 256 <= 1 else 8
        want_super = 0 if rate * 256 > 0.45 else 256
        assert used[1] == used[2] == (want_super, want_chunk), (used, rate)
        # forced sizes: same verdicts
        for chunk, sup in ((32, 4096), (8, 0), (32, 0)):
            proto.set_option("rlc_chunk", chunk)
            proto.set_option("rlc_superchunk", sup)
            acc, st, rej = _run(torch, proto, V, P, os.urandom(32))
            assert acc.tolist() == e_acc.tolist() and st.tolist() == e_st.tolist() and rej == e_rej
            assert (proto.get_option("last_rlc_superchunk"), proto.get_option("last_rlc_chunk")) == (sup, chunk)
        proto.set_option("rlc_history", 0)
        assert proto.get_option("rlc_has_history") == 0
        with pytest.raises(Exception):
            proto.set_option("rlc_chunk", 16)
    finally:
        proto.close()
