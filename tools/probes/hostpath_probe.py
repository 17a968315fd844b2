import sys, time, numpy as np
sys.path[:0] = ['.']
import torch, bench
from bp_pp_amd import U64RangeProofProtocol, synth
gens, g, gv, hv = bench.load_generators()
proto = U64RangeProofProtocol(g, gv, hv, device=0)
n = 1 << 20
dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 0, n)
Vh, Ph = dV.cpu().numpy(), dP.cpu().numpy()
del dV, dP
for rep in range(3):
    for name, chunk in (("pipelined_2^17", 1 << 17), ("pipelined_2^18", 1 << 18), ("pipelined_2^16", 1 << 16), ("upload_first", 0)):
        proto.set_option("host_chunk", chunk)
        t = time.perf_counter()
        acc, _ = proto.verify_batch(Vh, Ph, synth.LABEL)
        t = time.perf_counter() - t
        print(f"rep {rep} {name:16s} {t*1e3:8.2f} ms  ok {bool((acc == expect).all())}", flush=True)
