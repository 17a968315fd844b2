"""RLC batch mode: superchunk-size sweep on a resident 2^20-proof batch (VERDICT N1): exact mode vs chunks of 8 only vs the bucket
stage with M in {64, 256, 1024, 4096, 8192}, at corruption rates 0 and 1/1024.  Prints one line per point: ms per batch, M
verifies/s, per-kernel ms of the final-check stages, accept bits equal to exact mode.  usage: python tools/rlc_sweep.py [log2 n]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from bp_pp_amd import U64RangeProofProtocol, synth
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
gens, g, gv, hv = bench.load_generators()
proto = U64RangeProofProtocol(g, gv, hv, device=0)
res = []
for every, tag in ((1 << 62, "0"), (1024, "1/1024")):
    dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 0, n, corrupt_every=every)
    dA = torch.zeros(n, dtype=torch.uint8, device="cuda"); dS = torch.zeros(n, dtype=torch.int32, device="cuda"); dR = torch.zeros(1, dtype=torch.int32, device="cuda")
    def run(mode, steps=3):
        def once():
            if mode is None:
                proto.verify_batch_device(synth.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, dR.data_ptr())
            else:
                proto.verify_batch_rlc_device(synth.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), os.urandom(32), dS.data_ptr(), dR.data_ptr())
        once(); proto.synchronize()
        proto.enable_timing(True); proto.timings(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps): once()
        proto.synchronize()
        dt = (time.perf_counter() - t0) / steps
        kt = {k: round(v["total_ms"] / steps, 2) for k, v in proto.timings(reset=True).items() if v["launches"] and k.startswith(("k_rlc", "k_bkt", "k_verify_final_check", "k_verify_accept"))}
        proto.enable_timing(False)
        ok = bool((dA.cpu().numpy() == expect).all())
        return dt, kt, ok
    torch.cuda.synchronize()
    dt, kt, ok = run(None)
    res.append({"corruption": tag, "mode": "exact", "ms": dt * 1e3, "Mverifies_s": n / dt / 1e6, "kernels_ms": kt, "accept_ok": ok})
    for m in (0, 64, 256, 1024, 4096, 8192):
        proto.set_option("rlc_superchunk", m)
        dt, kt, ok = run(m)
        res.append({"corruption": tag, "mode": "rlc chunks of 8 only" if m == 0 else f"rlc bucket M={m}", "ms": dt * 1e3, "Mverifies_s": n / dt / 1e6, "kernels_ms": kt, "accept_ok": ok})
    del dV, dP
for r in res:
    print(json.dumps(r))
proto.close()
