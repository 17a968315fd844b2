"""CPU tier of the single-proof front end (csrc/coalesce_core.h -- the ring + dispatcher threads libbppp_hip.so puts behind
bppp_u64_verify_one / bppp_u64_prove_one): the same `Coalescer` template over malloc staging and the product's device code compiled
for the host, driven by real host threads that each submit ONE proof per call, the reference's calling pattern
(u64_proof.rs:42, :57; benches/range_proof.rs:47-50).

What is pinned here:
  * every caller gets exactly its own row's result -- accept bit, status, advanced transcript -- equal to the oracle's, whatever
    else shared its batch (other labels, pre-loaded transcripts, wrong and malformed proofs);
  * requests really are gathered (fewer batched calls than requests, none larger than coalesce_max), a lone caller is flushed by the
    deadline, and all staging is released;
  * a batched call that fails as a whole returns its code to ITS callers only, outputs untouched;
  * shutdown with callers still inside drains: every caller returns, with its proper result or with CLOSED -- nobody hangs."""
import ctypes as C
import time

import numpy as np
import pytest

import bppp_oracle as O
import transcript_cases as TC
import workload
from emul.build import load

ERR_NOMEM, ERR_CLOSED = -5, -7
W = 4


def _ptrs(arrays):
    return (C.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])


def _run(L, tab, kind, ins, outs, n, threads, max_batch, wait_us, lanes, fail_batch=-1, shutdown_after_ms=-1, run_delay_ms=0):
    rcs = np.full(n, -1000, np.int32)
    stats = np.zeros(5, np.uint64)
    sizes = np.zeros(4 * n + 8, np.uint64)
    live = np.zeros(1, np.int32)
    t0 = time.time()
    nb = L.emul_coalesce_run(kind, tab.ctypes.data, W, threads, n, _ptrs(ins), _ptrs(outs), rcs.ctypes.data, max_batch, wait_us, lanes,
                             fail_batch, shutdown_after_ms, run_delay_ms, stats.ctypes.data, sizes.ctypes.data, len(sizes), live.ctypes.data)
    assert nb >= 0, nb
    return rcs, [int(v) for v in stats], [int(v) for v in sizes[:nb]], int(live[0]), time.time() - t0


@pytest.fixture(scope="module")
def verify_case():
    """20 single-proof requests as 20 different callers would hold them: 16 on Transcript::new(label) (4 wrong proofs, 1 malformed),
    4 on pre-loaded transcripts of different lengths."""
    L = load()
    gens, V, P, _ = workload.make_batch(16, first=300, nthreads=2)
    P, expect = workload.corrupt(P, V, every=4)
    P = P.copy()
    P[5, 3] ^= 0x40                                            # c_l.x no longer on the curve: k256 would not deserialize it
    tc = TC.make(4)
    assert tc["gens"] == gens
    fresh = np.frombuffer(TC.ser(O.Transcript(workload.LABEL)), np.uint8)
    S = np.concatenate([np.tile(fresh, (16, 1)), tc["states_in"]]).copy()
    V, P = np.concatenate([V, tc["V"]]).copy(), np.concatenate([P, tc["P"]]).copy()
    import bppp_oracle_c as OC
    oacc, ost = OC.u64_verify_batch(gens, workload.LABEL, V[:16], P[:16], nthreads=2)       # the checker
    assert ost[5] < 0 and not oacc[5] and (oacc[[0, 4, 8, 12]] == 0).all() and (np.delete(ost, 5) == 0).all()
    ost = np.where(ost < 0, 1, 0).astype(np.int32)         # the oracle's decoding failure = BPPP_ST_BAD_ENCODING
    accept = np.concatenate([oacc, tc["accept"]])
    status = np.concatenate([ost, np.zeros(4, np.int32)])
    tab = np.zeros(L.emul_fb_table_entries(49, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 49, W, tab.ctypes.data) == 0
    return dict(L=L, tab=tab, V=V, P=P, S=S, accept=accept, status=status, after=tc["states_after"], n=20)


def _verify(case, **kw):
    n = case["n"]
    acc, st, out = np.full(n, 7, np.uint8), np.full(n, -9, np.int32), np.full((n, 203), 0xEE, np.uint8)
    rcs, stats, sizes, live, dt = _run(case["L"], case["tab"], 0, [case["V"], case["P"], case["S"]], [acc, st, out], n, **kw)
    return rcs, acc, st, out, stats, sizes, live, dt


def _check_rows(case, rows, acc, st, out):
    rows = np.asarray(rows)
    assert (acc[rows] == case["accept"][rows]).all() and (st[rows] == case["status"][rows]).all()
    for i in rows:
        if i >= 16:
            assert bytes(out[i]) == bytes(case["after"][i - 16])       # the caller's transcript, advanced as merlin's
        elif i == 5:
            assert bytes(out[i]) == bytes(case["S"][i])                # malformed: the reference's verify is never entered


@pytest.mark.parametrize("threads,max_batch,lanes", [(7, 5, 2), (20, 64, 1), (3, 2, 3)])
def test_every_caller_gets_its_own_row(verify_case, threads, max_batch, lanes):
    rcs, acc, st, out, stats, sizes, live, _ = _verify(verify_case, threads=threads, max_batch=max_batch, wait_us=3000, lanes=lanes, run_delay_ms=3)
    n = verify_case["n"]
    assert (rcs == 0).all()
    _check_rows(verify_case, range(n), acc, st, out)
    assert stats[0] == n and stats[1] == len(sizes) and sum(sizes) == n and max(sizes) <= max_batch and stats[2] == max(sizes)
    assert stats[3] + stats[4] == stats[1]
    if threads > 1:
        assert len(sizes) < n                                          # requests were gathered, not run one by one
    assert live == 0                                                   # all staging released


def test_lone_caller_is_flushed_by_the_deadline(verify_case):
    rcs, acc, st, out, stats, sizes, live, _ = _verify(verify_case, threads=1, max_batch=64, wait_us=500, lanes=2)
    assert (rcs == 0).all() and sizes == [1] * verify_case["n"] and stats[4] == verify_case["n"] and stats[3] == 0
    _check_rows(verify_case, range(verify_case["n"]), acc, st, out)


def test_full_batches_do_not_wait_for_the_deadline(verify_case):
    # a deadline of 10 s: only sealing a FULL batch can flush anything (20 requests from 20 callers, batches of 4)
    rcs, acc, st, out, stats, sizes, live, dt = _verify(verify_case, threads=20, max_batch=4, wait_us=10_000_000, lanes=2)
    assert (rcs == 0).all() and sizes == [4] * 5 and stats[3] == 5 and dt < 9
    _check_rows(verify_case, range(verify_case["n"]), acc, st, out)


def test_failed_batch_is_returned_to_its_callers_only(verify_case):
    rcs, acc, st, out, stats, sizes, live, _ = _verify(verify_case, threads=10, max_batch=5, wait_us=2000, lanes=2, fail_batch=1, run_delay_ms=3)
    failed = np.nonzero(rcs == ERR_NOMEM)[0]
    ok = np.nonzero(rcs == 0)[0]
    assert len(failed) + len(ok) == verify_case["n"] and 1 <= len(failed) <= 5 and len(failed) == sizes[1]
    assert (acc[failed] == 7).all() and (st[failed] == -9).all() and (out[failed] == 0xEE).all()     # outputs untouched
    _check_rows(verify_case, ok, acc, st, out)
    assert live == 0


def test_shutdown_with_callers_inside_drains_and_nobody_hangs(verify_case):
    # 6 callers keep submitting while the front end is shut down: what was submitted completes, the rest is refused
    rcs, acc, st, out, stats, sizes, live, dt = _verify(verify_case, threads=6, max_batch=3, wait_us=1000, lanes=1, shutdown_after_ms=25,
                                                        run_delay_ms=10)
    assert set(rcs.tolist()) <= {0, ERR_CLOSED} and dt < 60
    ok = np.nonzero(rcs == 0)[0]
    closed = np.nonzero(rcs == ERR_CLOSED)[0]
    assert len(ok) >= 1 and len(closed) >= 1
    _check_rows(verify_case, ok, acc, st, out)
    assert (acc[closed] == 7).all() and live == 0
    assert stats[0] == len(ok)


def test_prove_one_rows_equal_the_oracle_prover():
    """kind 1: x, s, the 52 draws and the transcript in; proof, commitment, status and the advanced transcript out."""
    import bppp_oracle_c as OC
    L = load()
    n = 6
    gens = workload.generators()
    x = np.ascontiguousarray(workload.values(n, first=70))
    s, rnd = np.ascontiguousarray(workload.blindings(n, first=70)), np.ascontiguousarray(workload.prover_randomness(n, first=70))
    labels = [workload.LABEL, b"other label", workload.LABEL, b"", b"other label", workload.LABEL]
    S = np.stack([np.frombuffer(TC.ser(O.Transcript(lb)), np.uint8) for lb in labels]).copy()
    tab = np.zeros(L.emul_fb_table_entries(49, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 49, W, tab.ctypes.data) == 0
    P, V, st, out = np.zeros((n, 928), np.uint8), np.zeros((n, 64), np.uint8), np.full(n, -1, np.int32), np.zeros((n, 203), np.uint8)
    rcs, stats, sizes, live, _ = _run(L, tab, 1, [x, s, rnd, S], [P, V, st, out], n, threads=3, max_batch=4, wait_us=2000, lanes=2)
    assert (rcs == 0).all() and not st.any() and sum(sizes) == n and live == 0
    for i, lb in enumerate(labels):
        op, ov = OC.u64_prove_batch(gens, lb, x[i:i + 1], s[i:i + 1], rnd[i:i + 1], nthreads=1)       # the checker
        assert (P[i] == op[0]).all() and (V[i] == ov[0]).all()
        ok, after = TC.oracle_verify(dict(proto=O.U64RangeProofProtocol(*O.synth_generators())), i, bytes(V[i]), bytes(P[i]), bytes(S[i]))
        assert ok and bytes(out[i]) == after          # prover and verifier leave the transcript in the same state (same schedule)


def test_ring_under_thread_sanitizer(tmp_path):
    """tests/emul/coalesce_stress.cpp: the same Coalescer over a trivial batched call, 8-64 caller threads, 1-4 dispatcher lanes,
    batches of 1 to 1,024, shutdown with callers inside -- built with -fsanitize=thread.  Every request must get its own row's
    answer and the race detector must stay silent."""
    import os
    import subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul", "coalesce_stress.cpp")
    exe = str(tmp_path / "coalesce_stress_tsan")
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=thread", "-std=c++17", "-pthread", "-o", exe, src], capture_output=True, text=True)
    if r.returncode != 0 and "tsan" in (r.stderr or "").lower():
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    assert r.returncode == 0, r.stderr[-3000:]
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "ASAN_OPTIONS", "UBSAN_OPTIONS")}     # (under tools/sanitize_emul.sh)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(env, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, (r.stdout[-2000:], r.stderr[-4000:])
