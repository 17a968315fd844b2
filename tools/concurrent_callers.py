"""Aggregate rate of SINGLE-proof verify calls from T host threads, each with its own context over one set of tables
(bppp_ctx_create_shared) -- a service with a thread per request.  usage: python tools/concurrent_callers.py"""
import os, sys, threading, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np
import torch, bench
from bp_pp_amd import U64RangeProofProtocol, synth
gens, g, gv, hv = bench.load_generators()
base = U64RangeProofProtocol(g, gv, hv, device=0)
dV, dP, expect, _ = bench.make_resident_batch(torch, base, synth, 0, 4096)
V, P = dV.cpu().numpy(), dP.cpu().numpy()
for T in (1, 2, 4, 8, 16, 32, 64):
    ctxs = [base] + [base.clone_shared() for _ in range(T - 1)]
    calls, bad = 100, []
    def worker(i):
        c = ctxs[i]
        for k in range(calls):
            j = (i * calls + k) % 4096
            acc, _ = c.verify_batch(V[j:j + 1], P[j:j + 1], synth.LABEL)
            if int(acc[0]) != int(expect[j]):
                bad.append((i, k))
    for c in ctxs:                                   # warm every context (workspace allocation)
        c.verify_batch(V[:1], P[:1], synth.LABEL)
    th = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"threads {T:3d}: {T * calls / dt:9.0f} single-proof verifies/s  ({dt / calls * 1e3:6.2f} ms per call per thread)  wrong {len(bad)}", flush=True)
    for c in ctxs[1:]:
        c.close()
base.close()
