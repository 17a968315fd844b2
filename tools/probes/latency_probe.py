import sys, time, numpy as np
sys.path[:0] = ['.']
import torch, bench
from bp_pp_amd import U64RangeProofProtocol, synth
gens, g, gv, hv = bench.load_generators()
proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=16)
dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 0, 1 << 14)
V, P = dV.cpu().numpy(), dP.cpu().numpy()
for n in (1, 2, 8, 64, 256, 1024, 4096, 16384):
    proto.verify_batch(V[:n], P[:n], synth.LABEL)
    t = time.perf_counter(); reps = 5
    for _ in range(reps): acc, _ = proto.verify_batch(V[:n], P[:n], synth.LABEL)
    t = (time.perf_counter() - t) / reps
    print(f"n {n:6d}  host-call latency {t*1e3:8.3f} ms   {n/t:12.0f} verifies/s  ok {bool((acc == expect[:n]).all())}")
x, s, rnd = synth.bulk_values(1), synth.bulk_blindings(1), synth.bulk_prover_randomness(1)
for n in (1, 64, 1024):
    x, s, rnd = synth.bulk_values(n), synth.bulk_blindings(n), synth.bulk_prover_randomness(n)
    proto.prove_batch(x, s, rnd, synth.LABEL)
    t = time.perf_counter()
    for _ in range(5): proto.prove_batch(x, s, rnd, synth.LABEL)
    print(f"prove n {n:6d} latency {(time.perf_counter()-t)/5*1e3:8.3f} ms")
