"""The reference's calling pattern at GPU speed: T host threads, each making SINGLE-proof verify calls one after the other
(u64_proof.rs:42; a service with a thread per request).  Native threads (tools/cc_callers.c), so the numbers are the library's, not
the interpreter's.

Every thread calls bppp_u64_verify_one on ONE context: the coalescing front end (csrc/coalesce_core.h) gathers the callers' requests
into batched GPU calls.  (Round 3's pattern -- bppp_u64_verify_batch with n = 1, a context per thread over shared tables -- got 601
verifies/s at 64 threads: profiles/r03/r03_cc_concurrent_callers.txt.)

--prove: the same with bppp_u64_prove_one (u64_proof.rs:57): every returned proof and commitment is compared byte for byte with what
ONE batched call made of the same inputs.

usage: python tools/concurrent_callers.py [--prove] [--threads 1,8,64,256,1024] [--calls 0] [--us 100] [--max 1024] [--lanes 2] [--json out]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import hostinfo


def build_harness() -> C.CDLL:
    src, so = os.path.join(ROOT, "tools", "cc_callers.c"), os.path.join(ROOT, "tools", "libcc_callers.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", "-o", so + ".tmp", src])
        os.replace(so + ".tmp", so)
    H = C.CDLL(so)
    vp = C.c_void_p
    H.cc_run.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_char_p, C.c_size_t, vp, vp, vp, C.c_size_t, C.c_int, C.c_int, vp, C.POINTER(C.c_double),
                         C.POINTER(C.c_long), C.POINTER(C.c_long)]
    H.cc_run_prove.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_char_p, C.c_size_t, vp, vp, vp, vp, vp, C.c_size_t, C.c_int, C.c_int, vp,
                               C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    return H


def run_callers(H, fn, ctxs, label, V, P, expect, threads, calls, prove_inputs=None):
    """-> dict(rate, p50/p99/max latency in ms, wrong, failed).  prove_inputs = (x, s, rnd): fn is bppp_u64_prove_one, V / P what it must return."""
    lat = np.zeros(threads * calls, np.float64)
    el, wrong, failed = C.c_double(0), C.c_long(0), C.c_long(0)
    arr = (C.c_void_p * len(ctxs))(*ctxs)
    if prove_inputs is not None:
        x, s, rnd = prove_inputs
        rc = H.cc_run_prove(C.cast(fn, C.c_void_p), arr, len(ctxs), label, len(label), x.ctypes.data, s.ctypes.data, rnd.ctypes.data, V.ctypes.data,
                            P.ctypes.data, V.shape[0], threads, calls, lat.ctypes.data, C.byref(el), C.byref(wrong), C.byref(failed))
    else:
        rc = H.cc_run(C.cast(fn, C.c_void_p), arr, len(ctxs), label, len(label), V.ctypes.data, P.ctypes.data, expect.ctypes.data, V.shape[0], threads,
                      calls, lat.ctypes.data, C.byref(el), C.byref(wrong), C.byref(failed))
    if rc != 0:
        raise RuntimeError("could not start the caller threads")
    return {"threads": threads, "calls_per_thread": calls, ("proves_per_s" if prove_inputs is not None else "verifies_per_s"): round(threads * calls / el.value, 1),
            "latency_ms": {"mean": round(float(lat.mean()) / 1e3, 3), "p50": round(float(np.percentile(lat, 50)) / 1e3, 3),
                           "p90": round(float(np.percentile(lat, 90)) / 1e3, 3), "p99": round(float(np.percentile(lat, 99)) / 1e3, 3),
                           "max": round(float(lat.max()) / 1e3, 3)},
            "wrong": int(wrong.value), "failed": int(failed.value)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,8,64,256,1024")
    ap.add_argument("--calls", type=int, default=0, help="calls per thread (0: enough for ~1-2 s per point)")
    ap.add_argument("--us", type=int, default=-1)
    ap.add_argument("--max", type=int, default=-1)
    ap.add_argument("--lanes", type=int, default=-1)
    ap.add_argument("--pool", type=int, default=4096)
    ap.add_argument("--json", default="")
    ap.add_argument("--prove", action="store_true", help="bppp_u64_prove_one instead of bppp_u64_verify_one")
    a = ap.parse_args()
    H = build_harness()
    import torch
    import bench
    from bp_pp_amd import U64RangeProofProtocol, _capi, synth
    gens, g, gv, hv = bench.load_generators()
    base = U64RangeProofProtocol(g, gv, hv, device=0)
    dV, dP, expect, _ = bench.make_resident_batch(torch, base, synth, 0, a.pool)
    V, P = np.ascontiguousarray(dV.cpu().numpy()), np.ascontiguousarray(dP.cpu().numpy())
    expect = np.ascontiguousarray(np.asarray(expect, dtype=np.uint8))
    for name, v in (("coalesce_us", a.us), ("coalesce_max", a.max), ("coalesce_lanes", a.lanes)):
        if v >= 0:
            base.set_option(name, v)
    L = _capi.lib()
    fn, which, pin = L.bppp_u64_verify_one, "verify", None
    if a.prove:
        pin = tuple(np.ascontiguousarray(v) for v in (synth.bulk_values(a.pool), synth.bulk_blindings(a.pool), synth.bulk_prover_randomness(a.pool)))
        P, V, pst = base.prove_batch(pin[0], pin[1], pin[2], synth.LABEL)          # what every single call must reproduce
        assert not pst.any()
        P, V = np.ascontiguousarray(P), np.ascontiguousarray(V)
        fn, which = L.bppp_u64_prove_one, "prove"
    rows = []
    print(json.dumps({"host": hostinfo.summary()}), flush=True)
    for T in [int(t) for t in a.threads.split(",")]:
        calls = a.calls or max(20, min(400, 60000 // T))
        run_callers(H, fn, [base._ctx.value], synth.LABEL, V, P, expect, T, 3, pin)          # warm: front end, workspaces
        before, thr0 = base.coalesce_stats(which), hostinfo.throttle_stats()
        r = run_callers(H, fn, [base._ctx.value], synth.LABEL, V, P, expect, T, calls, pin)
        after, thr1 = base.coalesce_stats(which), hostinfo.throttle_stats()
        nb = after["batches"] - before["batches"]
        r["mean_batch"] = round((after["requests"] - before["requests"]) / max(1, nb), 1)
        r["batched_call_ms"] = round((after["run_us"] - before["run_us"]) / max(1, nb) / 1e3, 3)
        r["fill_wait_ms"] = round((after["fill_wait_us"] - before["fill_wait_us"]) / max(1, nb) / 1e3, 3)
        r["cgroup_throttled"] = {"periods": thr1["nr_throttled"] - thr0["nr_throttled"], "seconds": round(thr1["throttled_s"] - thr0["throttled_s"], 3)}
        r["mode"] = "prove_one" if a.prove else "one"
        rows.append(r)
        print(json.dumps(r), flush=True)
    base.close()
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"tool": "tools/concurrent_callers.py", "options": {"coalesce_us": a.us, "coalesce_max": a.max, "coalesce_lanes": a.lanes},
                       "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "default"), "host": hostinfo.summary(), "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
