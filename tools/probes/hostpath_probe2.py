"""Where the host-buffer path loses its time: bppp_u64_verify_batch from (a) pageable numpy buffers, (b) the same buffers page-locked with
hipHostRegister (torch.cuda.cudart().cudaHostRegister), (c) pinned torch tensors -- at 2^16 and 2^20 proofs, host_chunk 2^17 / 0.
usage: python tools/probes/hostpath_probe2.py"""
import os, sys, time
import numpy as np
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch, bench
from bp_pp_amd import U64RangeProofProtocol, synth
gens, g, gv, hv = bench.load_generators()
proto = U64RangeProofProtocol(g, gv, hv, device=0)
n = 1 << 20
dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 0, n)
Vh, Ph = dV.cpu().numpy(), dP.cpu().numpy()
Vp, Pp = dV.cpu().pin_memory(), dP.cpu().pin_memory()
rt = torch.cuda.cudart()
def timed(V, P, m, reps=3):
    best = None
    for _ in range(reps):
        t = time.perf_counter()
        acc, _ = proto.verify_batch(V[:m], P[:m], synth.LABEL)
        t = time.perf_counter() - t
        best = t if best is None else min(best, t)
    assert (acc == expect[:m]).all()
    return best * 1e3
for m in (1 << 16, 1 << 20):
    for chunk in ((1 << 17, 0) if m > (1 << 17) else (1 << 17,)):
        proto.set_option("host_chunk", chunk)
        timed(Vh, Ph, m, 1)
        print(f"n=2^{m.bit_length()-1} host_chunk={chunk}: pageable {timed(Vh, Ph, m):8.2f} ms", flush=True)
        print(f"n=2^{m.bit_length()-1} host_chunk={chunk}: pinned (torch pin_memory) {timed(Vp.numpy(), Pp.numpy(), m):8.2f} ms", flush=True)
# raw copy rates
for name, src in (("pageable", Ph), ("pinned", Pp.numpy())):
    d = torch.empty(Ph.nbytes, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        d.copy_(torch.from_numpy(src).view(-1), non_blocking=True); torch.cuda.synchronize()
        t = time.perf_counter() - t
    print(f"H2D of {Ph.nbytes >> 20} MB from {name}: {t*1e3:.1f} ms = {Ph.nbytes / t / 1e9:.1f} GB/s", flush=True)
t = time.perf_counter()
r1 = rt.cudaHostRegister(Vh.ctypes.data, Vh.nbytes, 0); r2 = rt.cudaHostRegister(Ph.ctypes.data, Ph.nbytes, 0)
print(f"hipHostRegister of {Vh.nbytes + Ph.nbytes >> 20} MB: {(time.perf_counter() - t) * 1e3:.1f} ms  rc {r1} {r2}", flush=True)
for m in (1 << 16, 1 << 20):
    proto.set_option("host_chunk", 1 << 17)
    print(f"n=2^{m.bit_length()-1}: registered {timed(Vh, Ph, m):8.2f} ms", flush=True)
# resident reference
dA = torch.zeros(n, dtype=torch.uint8, device="cuda"); dS = torch.zeros(n, dtype=torch.int32, device="cuda")
for m in (1 << 16, 1 << 20):
    best = None
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        proto.verify_batch_device(synth.LABEL, m, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr()); proto.synchronize()
        t = time.perf_counter() - t; best = t if best is None else min(best, t)
    print(f"n=2^{m.bit_length()-1}: resident {best*1e3:8.2f} ms", flush=True)
