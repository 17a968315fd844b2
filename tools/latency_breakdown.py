"""Per-kernel times of small verify / prove calls (where a call's latency goes when the chip is under-filled).
usage: python tools/latency_breakdown.py [window_bits]"""
import sys, time
sys.path[:0] = ['.']
import numpy as np
import torch, bench
from bp_pp_amd import U64RangeProofProtocol, synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 16
gens, g, gv, hv = bench.load_generators()
proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=W)
dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 0, 1 << 14)
V, P = dV.cpu().numpy(), dP.cpu().numpy()
dA = torch.zeros(1 << 14, dtype=torch.uint8, device="cuda")
dS = torch.zeros(1 << 14, dtype=torch.int32, device="cuda")
for n in (1, 64, 1024, 2048, 4096, 8192, 16384):
    for mode in ("host", "device"):
        def call():
            if mode == "host":
                return proto.verify_batch(V[:n], P[:n], synth.LABEL)[0]
            proto.verify_batch_device(synth.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr())
            proto.synchronize()
            return None
        call()
        t = time.perf_counter(); reps = 10
        for _ in range(reps): acc = call()
        t = (time.perf_counter() - t) / reps
        ok = acc is None or bool((acc == expect[:n]).all())
        if mode == "device":
            ok = bool((dA[:n].cpu().numpy() == expect[:n]).all())
        print(f"verify n {n:5d} {mode:6s} call latency {t*1e3:7.3f} ms  ok {ok}")
    proto.enable_timing(True); proto.timings(reset=True)
    for _ in range(5): proto.verify_batch_device(synth.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr())
    proto.synchronize()
    kt = proto.timings(reset=True); proto.enable_timing(False)
    print("   kernels ms/call:", {k: round(v["total_ms"] / 5, 3) for k, v in kt.items() if v["launches"]}, " sum", round(sum(v["total_ms"] for v in kt.values()) / 5, 3))
for n in (1, 64, 1024):
    x, s, rnd = synth.bulk_values(n), synth.bulk_blindings(n), synth.bulk_prover_randomness(n)
    proto.prove_batch(x, s, rnd, synth.LABEL)
    t = time.perf_counter()
    for _ in range(5): proto.prove_batch(x, s, rnd, synth.LABEL)
    print(f"prove n {n:5d} host call latency {(time.perf_counter()-t)/5*1e3:7.3f} ms")
    proto.enable_timing(True); proto.timings(reset=True)
    for _ in range(5): proto.prove_batch(x, s, rnd, synth.LABEL)
    kt = proto.timings(reset=True); proto.enable_timing(False)
    print("   kernels ms/call:", {k: round(v["total_ms"] / 5, 3) for k, v in kt.items() if v["launches"]}, " sum", round(sum(v["total_ms"] for v in kt.values()) / 5, 3))
