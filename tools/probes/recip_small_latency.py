"""One small call of the reciprocal verifier on BASELINE configs[4]'s shape (dim_nd 256, dim_np 16) on device buffers: median latency for
n = 1, 64, 1024 instances, with phase 1 on one lane per instance (BPPP_RECIP_P1_GROUP=1) and as the library runs it (lane groups by size).
python tools/probes/recip_small_latency.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench_other
from bp_pp_amd.wnla import ReciprocalRangeProofProtocol


def main():
    gens5 = bench_other.recip256_generators()
    out = {}
    for name, env in (("one lane", "1"), ("by size", None)):
        if env is None:
            os.environ.pop("BPPP_RECIP_P1_GROUP", None)
        else:
            os.environ["BPPP_RECIP_P1_GROUP"] = env
        proto = ReciprocalRangeProofProtocol(256, 16, *gens5, device=0, fb_window_bits=16)
        dV, dP, expect, shape, _, _ = bench_other.recip256_resident_batch(torch, proto, 0, 1024)
        dA = torch.zeros(1024, dtype=torch.uint8, device="cuda"); dS = torch.zeros(1024, dtype=torch.int32, device="cuda")
        for n in (1, 64, 1024):
            ts = []
            for i in range(12):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                proto.verify_batch_device(bench_other.RECIP_LABEL, n, dV.data_ptr(), dP.data_ptr(), *shape, dA.data_ptr(), dS.data_ptr())
                proto.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ok = bool((dA[:n].cpu().numpy() == expect[:n]).all())
            out[(name, n)] = float(np.median(ts[2:]))
            print(f"{name:9s} n={n:5d}  median {out[(name, n)]:7.3f} ms  ok={ok}", flush=True)
        proto.enable_timing(True); proto.timings()
        for _ in range(5):
            proto.verify_batch_device(bench_other.RECIP_LABEL, 1, dV.data_ptr(), dP.data_ptr(), *shape, dA.data_ptr(), dS.data_ptr())
        proto.synchronize()
        kt = proto.timings(); proto.enable_timing(False)
        print("    n=1 kernels, ms per call:", {k.replace("k_", ""): round(v["total_ms"] / 5, 3) for k, v in kt.items() if v["launches"]}, flush=True)
        proto.close()


if __name__ == "__main__":
    main()
