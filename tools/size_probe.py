"""Per-kernel times of the u64 verifier (and prover) at batch sizes around the chip's fill point, for the default dispatch and for the
diagnostic switches given in the environment (BPPP_NO_SMALL_KERNELS, BPPP_NO_LANE_GROUPS, ...).
usage: python tools/size_probe.py [verify|prove] [log2 sizes ...]     e.g.  BPPP_NO_SMALL_KERNELS=1 python tools/size_probe.py verify 15 16 17"""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np
import torch, bench
from bp_pp_amd import U64RangeProofProtocol, synth
what = sys.argv[1] if len(sys.argv) > 1 else "verify"
sizes = [1 << int(a) for a in sys.argv[2:]] or [1 << 15, 1 << 16, 1 << 17]
gens, g, gv, hv = bench.load_generators()
proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=int(os.environ.get("PROBE_W", "0")))
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("BPPP_")) or "default"
nmax = max(sizes)
if what == "verify":
    dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 0, nmax)
for n in sizes:
    if what == "verify":
        dA = torch.zeros(n, dtype=torch.uint8, device="cuda"); dS = torch.zeros(n, dtype=torch.int32, device="cuda")
        def fn():
            proto.verify_batch_device(synth.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, 0)
    else:
        dx = torch.from_numpy(synth.bulk_values(n).view(np.int64)).cuda()
        ds, dr = torch.from_numpy(synth.bulk_blindings(n)).cuda(), torch.from_numpy(synth.bulk_prover_randomness(n)).cuda()
        oP = torch.zeros((n, 928), dtype=torch.uint8, device="cuda"); oV = torch.zeros((n, 64), dtype=torch.uint8, device="cuda")
        dS = torch.zeros(n, dtype=torch.int32, device="cuda")
        def fn():
            proto.prove_batch_device(synth.LABEL, n, dx.data_ptr(), ds.data_ptr(), dr.data_ptr(), oP.data_ptr(), oV.data_ptr(), dS.data_ptr())
    fn(); proto.synchronize()
    reps = max(3, min(20, (1 << 21) // n))
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(reps): fn()
        proto.synchronize()
        best = min(best, (time.perf_counter() - t) / reps)
    proto.enable_timing(True); proto.timings(reset=True)
    for _ in range(reps): fn()
    proto.synchronize()
    kt = {k.replace("k_verify_", "").replace("k_prove_", ""): round(v["total_ms"] / reps, 3) for k, v in proto.timings(reset=True).items() if v["launches"]}
    proto.enable_timing(False)
    ok = bool((dA.cpu().numpy() == expect[:n]).all()) if what == "verify" else not bool(dS.any().item())
    print(f"[{tag}] {what} n=2^{n.bit_length()-1}: {best*1e3:8.3f} ms  {n/best/1e6:6.3f} M/s  ok {ok}  timed-sum {sum(kt.values()):.3f}  {kt}", flush=True)
proto.close()
