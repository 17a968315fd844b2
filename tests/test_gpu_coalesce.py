"""GPU tier of the single-proof front end: bppp_u64_verify_one / bppp_u64_prove_one (include/bppp.h) on an MI355X -- the reference's
calling pattern, one proof per call from many host threads at once (u64_proof.rs:42, :57), each call judged by the oracle.
The CPU-tier twin over the emulator is tests/test_coalesce_emul.py."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LABELS = [b"u64 range proof", b"another protocol", b"", b"x" * 70]


@pytest.fixture(scope="module")
def proto():
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    g, gv, hv = workload.split_generators(workload.generators())
    p = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
    yield p
    p.close()


@pytest.fixture(scope="module")
def requests_256(oracle_c):
    """256 requests as 256 callers hold them: proofs made under four different labels (64 each), every 5th proof wrong, every 9th
    malformed, and 8 of them on pre-loaded transcripts.  The expected verdicts come from the oracle, request by request."""
    import transcript_cases as TC
    import workload
    gens = workload.generators()
    dl = workload.generator_dlogs()
    reqs = []
    for li, label in enumerate(LABELS):
        x, s, rnd = workload.values(64, first=1000 * li), workload.blindings(64, first=1000 * li), workload.prover_randomness(64, first=1000 * li)
        P, V = oracle_c.u64_prove_trapdoor_batch(dl, label, x, s, rnd, nthreads=4)
        P, V = P.copy(), V.copy()
        for j in range(64):
            if j % 5 == 1:
                P[j, 870 + (j % 50)] ^= 1 + (j % 7)               # a scalar changed: well-formed, wrong
            if j % 9 == 2:
                P[j, 64 * (j % 13) + 7] ^= 0x20                   # a coordinate changed: (almost surely) off the curve
            rc = oracle_c.u64_verify(gens, label, bytes(V[j]), bytes(P[j]))
            reqs.append(dict(label=label, V=bytes(V[j]), P=bytes(P[j]), accept=1 if rc == 1 else 0, flagged=rc < 0))
    tc = TC.make(8)
    assert tc["gens"] == gens
    for j in range(8):
        reqs[32 * j + 3] = dict(state=bytes(tc["states_in"][j]), V=bytes(tc["V"][j]), P=bytes(tc["P"][j]), accept=1, flagged=False,
                                after=bytes(tc["states_after"][j]))
    assert sum(r["accept"] for r in reqs) > 150 and sum(r["flagged"] for r in reqs) > 15
    return reqs


def _call_all(proto, reqs, nthreads):
    from bp_pp_amd.transcript import Transcript
    out = [None] * len(reqs)
    errs = []
    gate = threading.Barrier(nthreads)

    def worker(t):
        try:
            gate.wait()
            for i in range(t, len(reqs), nthreads):
                r = reqs[i]
                if "state" in r:
                    tr = Transcript(state=r["state"])
                    acc, st = proto.verify_one(r["V"], r["P"], tr)
                    out[i] = (acc, st, tr.state)
                else:
                    acc, st = proto.verify_one(r["V"], r["P"], r["label"])
                    out[i] = (acc, st, None)
        except Exception as e:                # noqa: BLE001
            errs.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[:3]
    return out


def _check(reqs, out):
    for i, (r, o) in enumerate(zip(reqs, out)):
        acc, st, state = o
        assert int(acc) == r["accept"], i
        assert (st != 0) == r["flagged"], (i, st)
        if "after" in r:
            assert state == r["after"], i             # the caller's transcript, advanced exactly as merlin's


@pytest.mark.parametrize("nthreads", [256, 16])
def test_256_callers_each_get_the_oracles_verdict(proto, requests_256, nthreads):
    before = proto.coalesce_stats()
    out = _call_all(proto, requests_256, nthreads)
    _check(requests_256, out)
    after = proto.coalesce_stats()
    assert after["requests"] - before["requests"] == 256
    assert after["batches"] - before["batches"] < 256          # the calls were gathered into batched launches


def test_options_and_small_batches(proto, requests_256):
    """coalesce_max = 3 forces many small batches over 3 lanes; changing an option drains the running front end and the next call
    starts a new one.  Same verdicts."""
    proto.set_option("coalesce_max", 3)
    proto.set_option("coalesce_lanes", 3)
    proto.set_option("coalesce_us", 0)
    try:
        out = _call_all(proto, requests_256[:96], 24)
        _check(requests_256[:96], out)
        s = proto.coalesce_stats()
        assert s["largest_batch"] <= 3 and s["requests"] == 96          # a fresh front end (the counters restarted)
        with pytest.raises(Exception):
            proto.set_option("coalesce_max", 0)
        with pytest.raises(Exception):
            proto.set_option("coalesce_lanes", 9)
    finally:
        proto.set_option("coalesce_max", 1024)
        proto.set_option("coalesce_lanes", 2)
        proto.set_option("coalesce_us", 100)


def test_one_equals_batch_entry_point(proto, requests_256):
    """The single-proof answer is the batched entry point's for that row."""
    sel = [r for r in requests_256 if r.get("label") == LABELS[1]][:40]
    V = np.frombuffer(b"".join(r["V"] for r in sel), np.uint8).reshape(-1, 64)
    P = np.frombuffer(b"".join(r["P"] for r in sel), np.uint8).reshape(-1, 928)
    acc, st = proto.verify_batch(V, P, LABELS[1])
    for i, r in enumerate(sel):
        a1, s1 = proto.verify_one(r["V"], r["P"], LABELS[1])
        assert int(a1) == int(acc[i]) and s1 == int(st[i])


def test_invalid_arguments_are_refused_without_touching_the_gpu(proto):
    import ctypes as C
    from bp_pp_amd import _capi
    L = _capi.lib()
    acc, st = C.c_uint8(9), C.c_int32(9)
    bad_state = bytearray(203)
    bad_state[200] = 166
    buf = C.create_string_buffer(bytes(bad_state), 203)
    assert L.bppp_u64_verify_one_transcript(proto._ctx, buf, b"\0" * 64, b"\0" * 928, C.byref(acc), C.byref(st)) == _capi.ERR_INVALID_ARG
    assert L.bppp_u64_verify_one(proto._ctx, b"l", 1, None, b"\0" * 928, C.byref(acc), C.byref(st)) == _capi.ERR_INVALID_ARG
    assert L.bppp_u64_verify_one(None, b"l", 1, b"\0" * 64, b"\0" * 928, C.byref(acc), C.byref(st)) == _capi.ERR_INVALID_ARG
    assert acc.value == 9 and st.value == 9


def test_prove_one_from_many_threads_is_byte_identical_to_the_oracle_prover(proto, oracle_c):
    import workload
    from bp_pp_amd.transcript import Transcript
    gens = workload.generators()
    n = 48
    x, s, rnd = workload.values(n, first=7000), workload.blindings(n, first=7000), workload.prover_randomness(n, first=7000)
    out = [None] * n
    errs = []

    def worker(t):
        try:
            for i in range(t, n, 12):
                label = LABELS[i % 2]
                if i % 3 == 0:
                    tr = Transcript(label)
                    out[i] = proto.prove_one(int(x[i]), bytes(s[i]), tr, bytes(rnd[i])) + (tr.state,)
                else:
                    out[i] = proto.prove_one(int(x[i]), bytes(s[i]), label, bytes(rnd[i])) + (None,)
        except Exception as e:                # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[:3]
    for i in range(n):
        label = LABELS[i % 2]
        op, ov = oracle_c.u64_prove_batch(gens, label, x[i:i + 1], s[i:i + 1], rnd[i:i + 1], nthreads=1)
        proof, com, st, state = out[i]
        assert st == 0 and proof == bytes(op[0]) and com == bytes(ov[0]), i
        if state is not None:
            # the prover leaves the transcript where the verifier of the same proof does
            tv = Transcript(label)
            acc, _ = proto.verify_one(com, proof, tv)
            assert acc and tv.state == state
    assert proto.coalesce_stats("prove")["requests"] >= n


def test_destroy_with_callers_inside_drains():
    """bppp_ctx_destroy while single-proof callers are asleep inside the front end (a deadline of 1 s keeps them there): destroy
    seals and runs what was submitted, every caller wakes with its proper verdict long before the deadline, nobody hangs; an option
    change drains the same way and leaves the context usable."""
    import ctypes as C
    import time
    import workload
    from bp_pp_amd import U64RangeProofProtocol, _capi
    g, gv, hv = workload.split_generators(workload.generators())
    p = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=4)
    _, V, P, _ = workload.make_batch(8, first=60)
    P = P.copy()
    P[3, 900] ^= 1
    L = _capi.lib()
    ctx = p._ctx.value
    p.set_option("coalesce_us", 50)
    assert p.verify_one(V[0].tobytes(), P[0].tobytes(), workload.LABEL) == (True, 0)        # the front end exists now

    def round_of_callers(finish):
        results = [None] * 8

        def worker(t):
            acc, st = C.c_uint8(7), C.c_int32(0)
            rc = L.bppp_u64_verify_one(ctx, workload.LABEL, len(workload.LABEL), V[t].tobytes(), P[t].tobytes(), C.byref(acc), C.byref(st))
            results[t] = (rc, acc.value)

        th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        t0 = time.time()
        for t in th:
            t.start()
        time.sleep(0.25)                       # all eight are inside, asleep behind the 1 s deadline
        assert results == [None] * 8
        finish()
        for t in th:
            t.join(timeout=60)
        assert not any(t.is_alive() for t in th)
        assert time.time() - t0 < 0.9          # woken by the drain, not by the deadline
        assert results == [(0, 0 if t == 3 else 1) for t in range(8)]

    p.set_option("coalesce_us", 1_000_000)
    assert p.verify_one(V[1].tobytes(), P[1].tobytes(), workload.LABEL) == (True, 0)        # (starts the new front end; waits out one deadline)
    round_of_callers(lambda: p.set_option("coalesce_us", 999_999))     # option change: drain, context stays valid
    assert p.verify_one(V[2].tobytes(), P[2].tobytes(), workload.LABEL) == (True, 0)
    round_of_callers(p.close)                                            # destroy: drain, then the context is gone


@pytest.mark.parametrize("cmax,lanes", [(2, 1), (64, 2)])
def test_destroy_while_callers_loop_in_and_out(cmax, lanes):
    """bppp_ctx_destroy while 16 threads LOOP over bppp_u64_verify_one: at the moment of the destroy some are asleep with a claimed row,
    some wait for an open batch (back-pressure: staging for `cmax` rows per set), some are between the front end and their return.
    Round 4 freed the front-end registry after the drain and re-created it -- and a new front end, on a context that was being freed --
    for a caller that came back from a closed batch.  Now closure is sticky and the context counts the callers inside: every call
    returns 0 with the oracle's verdict or BPPP_ERR_CLOSED, nobody hangs, nothing is started on the dying context."""
    import ctypes as C
    import time
    import workload
    from bp_pp_amd import U64RangeProofProtocol, _capi
    g, gv, hv = workload.split_generators(workload.generators())
    _, V, P, _ = workload.make_batch(8, first=70)
    P = P.copy()
    P[5, 901] ^= 1
    L = _capi.lib()
    for rep in range(4):
        p = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=4)
        p.set_option("coalesce_max", cmax)
        p.set_option("coalesce_lanes", lanes)
        p.set_option("coalesce_us", 50)
        ctx = p._ctx.value
        assert p.verify_one(V[0].tobytes(), P[0].tobytes(), workload.LABEL) == (True, 0)
        gate = threading.Lock()
        state = {"stop": False}
        counts = {"ok": 0, "closed": 0}
        bad = []

        def worker(t):
            k = t
            while True:
                with gate:                      # a call is only started while the context is certainly alive
                    if state["stop"]:
                        return
                acc, st = C.c_uint8(7), C.c_int32(0)
                i = k % 8
                rc = L.bppp_u64_verify_one(ctx, workload.LABEL, len(workload.LABEL), V[i].tobytes(), P[i].tobytes(), C.byref(acc), C.byref(st))
                if rc == 0:
                    if acc.value != (0 if i == 5 else 1):
                        bad.append((t, k, acc.value))
                    counts["ok"] += 1
                elif rc == _capi.ERR_CLOSED:
                    counts["closed"] += 1
                    return                      # the context is gone: no further call
                else:
                    bad.append((t, k, "rc", rc))
                    return
                k += 1

        th = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
        for t in th:
            t.start()
        time.sleep(0.15 + 0.05 * rep)
        with gate:
            state["stop"] = True
        time.sleep(0.005)                       # whoever passed the gate is inside its call by now
        p.close()                               # destroy with callers inside
        for t in th:
            t.join(timeout=60)
        assert not any(t.is_alive() for t in th), "a caller hangs"
        assert not bad, bad[:5]
        assert counts["ok"] > 16


def test_python_single_proof_wrappers_check_buffer_lengths():
    """verify_one hands raw pointers to C, which copies 64 / 928 / proof_bytes() bytes from them: a short buffer is refused in Python."""
    from bp_pp_amd import U64RangeProofProtocol
    from bp_pp_amd.wnla import ReciprocalRangeProofProtocol
    p = object.__new__(U64RangeProofProtocol)       # (the check comes before anything touches the context)
    with pytest.raises(ValueError):
        p.verify_one(bytes(63), bytes(928), b"label")
    with pytest.raises(ValueError):
        p.verify_one(bytes(64), bytes(927), b"label")
    r = object.__new__(ReciprocalRangeProofProtocol)
    with pytest.raises(ValueError):
        r.verify_one(bytes(64), bytes(64 * 13 + 32 * 3 - 1), 4, 2, 1, b"label")
    with pytest.raises(ValueError):
        r.verify_one(bytes(10), bytes(64 * 13 + 32 * 3), 4, 2, 1, b"label")


@pytest.mark.parametrize("nd,npp,B", [(32, 16, 12), (12, 10, 7)])
def test_reciprocal_verify_one_from_many_threads(nd, npp, B):
    """bppp_reciprocal_verify_one[_transcript]: `ReciprocalRangeProofProtocol::verify` (reciprocal.rs:98-107) one instance per call from
    several threads at runtime dimensions, gathered per shape; every verdict and status the oracle's, the transcript form advancing
    the caller's state exactly as the batched transcript entry point does."""
    import recip_cases
    from bp_pp_amd.transcript import Transcript
    from bp_pp_amd.wnla import ReciprocalRangeProofProtocol
    case = recip_cases.make(nd, npp, B)
    proto = ReciprocalRangeProofProtocol(nd, npp, case["g"], case["gv"], case["hv"], case["gv_"], case["hv_"], device=0, fb_window_bits=8)
    try:
        shape = (case["rounds"], case["nl"], case["nn"])
        P, com = case["proofs"].copy(), case["commitments"].copy()
        P[0, -1] ^= 1
        P[1, 256 + 64 * case["rounds"] + 5] ^= 0x40
        com[2] = case["commitments"][3]
        exp = [recip_cases.oracle_verify(case, bytes(com[b]), bytes(P[b])) for b in range(B)]
        S = [Transcript(case["label"]) for _ in range(B)]
        _, _, ref_states = proto.verify_batch(b"", com, P, *shape, transcripts=[Transcript(case["label"]) for _ in range(B)])
        out = [None] * (2 * B)
        errs = []

        def worker(t):
            try:
                for i in range(t, 2 * B, 6):
                    b = i % B
                    if i < B:
                        out[i] = proto.verify_one(bytes(com[b]), bytes(P[b]), *shape, case["label"])
                    else:
                        out[i] = proto.verify_one(bytes(com[b]), bytes(P[b]), *shape, S[b])
            except Exception as e:                # noqa: BLE001
                errs.append(repr(e))

        th = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs[:3]
        for i in range(2 * B):
            b = i % B
            acc, st = out[i]
            assert int(acc) == (1 if exp[b] == 1 else 0) and (st != 0) == (exp[b] < 0), (i, out[i], exp[b])
        for b in range(B):
            assert S[b].state == bytes(ref_states[b])
        assert sum(1 for e in exp if e == 1) >= B - 3 and exp[0] != 1
        # a shape the context's generators cannot serve is refused before any front end exists for it
        import ctypes as C
        from bp_pp_amd import _capi
        a, s_ = C.c_uint8(9), C.c_int32(9)
        rc = _capi.lib().bppp_reciprocal_verify_one(proto._w._ctx, b"x", 1, 5000, 16, bytes(64), bytes(P[0]), shape[0], shape[1], shape[2], C.byref(a), C.byref(s_))
        assert rc == _capi.ERR_INVALID_ARG and a.value == 9
    finally:
        proto.close()
