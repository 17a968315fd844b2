"""Does a prove batch run faster as P sub-batches on P contexts (own stream pairs, shared tables), the latency-bound kernels of one
sub-batch under the fixed-base sums of another?  Diagnostics from the environment (BPPP_NO_SMALL_KERNELS = the 256-register builds of
every lane kernel, so that they can share a SIMD with a sum's wavefront; BPPP_NEXT_MSM_MAX, BPPP_NEXT_G4_W2, BPPP_NEXT_OVERLAP).
usage: python tools/prove_pipe_probe.py [log2 sizes ...]"""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import numpy as np
import torch, bench
from bp_pp_amd import U64RangeProofProtocol, synth
sizes = [1 << int(a) for a in sys.argv[1:]] or [1 << 14]
gens, g, gv, hv = bench.load_generators()
proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=int(os.environ.get("PROBE_W", "0")))
ctxs = [proto] + [proto.clone_shared() for _ in range(3)]
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("BPPP_")) or "default"
for n in sizes:
    dx = torch.from_numpy(synth.bulk_values(n).view(np.int64)).cuda()
    ds, dr = torch.from_numpy(synth.bulk_blindings(n)).cuda(), torch.from_numpy(synth.bulk_prover_randomness(n)).cuda()
    oP = torch.zeros((n, 928), dtype=torch.uint8, device="cuda"); oV = torch.zeros((n, 64), dtype=torch.uint8, device="cuda")
    dS = torch.zeros(n, dtype=torch.int32, device="cuda")
    ref = None
    for parts in (1, 2, 4):
        m = n // parts
        def fn():
            for i in range(parts):
                ctxs[i].prove_batch_device(synth.LABEL, m, dx.data_ptr() + 8 * i * m, ds.data_ptr() + 32 * i * m, dr.data_ptr() + 52 * 32 * i * m,
                                           oP.data_ptr() + 928 * i * m, oV.data_ptr() + 64 * i * m, dS.data_ptr() + 4 * i * m)
            for i in range(parts): ctxs[i].synchronize()
        oP.zero_(); fn()
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            for _ in range(8): fn()
            best = min(best, (time.perf_counter() - t) / 8)
        out = oP.cpu().numpy()
        if ref is None: ref = out.copy()
        print(f"[{tag}] prove n=2^{n.bit_length()-1} in {parts} part(s): {best*1e3:8.3f} ms  {n/best/1e6:6.3f} M/s  same bytes {bool((out == ref).all())}  status {int(dS.abs().sum().item())}", flush=True)
for c in ctxs[1:]: c.close()
proto.close()
