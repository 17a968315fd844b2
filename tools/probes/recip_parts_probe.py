"""BASELINE configs[4]'s shape (reciprocal 256 / 16) on device buffers: one call as K parts on K streams (option "generic_parts"), K = 1 .. 4,
the parts' chains started together or out of step (option "generic_stagger" 0 .. 3), timed in turns on one context and one resident batch.
python tools/probes/recip_parts_probe.py [log2 n ...]   (default 15; VARIANTS="1:0 2:0 2:1 ..." = parts:stagger)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench, bench_other
from bp_pp_amd.wnla import ReciprocalRangeProofProtocol

REPS = int(os.environ.get("REPS", "7"))


def main():
    sizes = [1 << int(a) for a in sys.argv[1:]] or [1 << 15]
    gens5 = bench_other.recip256_generators()
    proto = ReciprocalRangeProofProtocol(256, 16, *gens5, device=0, fb_window_bits=int(os.environ.get("FB_WINDOW_BITS", "0")))
    nmax = max(sizes)
    dV, dP, expect, shape, _, _ = bench_other.recip256_resident_batch(torch, proto, 0, nmax)
    dA = torch.zeros(nmax, dtype=torch.uint8, device="cuda"); dS = torch.zeros(nmax, dtype=torch.int32, device="cuda")
    print("fb_window_bits", proto.get_option("fb_window_bits"), flush=True)
    for n in sizes:
        variants = [tuple(int(x) for x in v.split(":")) for v in os.environ.get("VARIANTS", "1:0 2:0 2:1 2:2 2:3 3:1 4:1 4:2").split()]
        times = {K: [] for K in variants}
        ok = {}
        for K in times:
            proto.set_option("generic_parts", K[0]); proto.set_option("generic_stagger", K[1])
            dA.zero_()
            for _ in range(2):
                proto.verify_batch_device(bench_other.RECIP_LABEL, n, dV.data_ptr(), dP.data_ptr(), *shape, dA.data_ptr(), dS.data_ptr())
            proto.synchronize()
            ok[K] = bool((dA[:n].cpu().numpy() == expect[:n]).all()) and not bool(dS[:n].any().item())
        for _ in range(REPS):
            for K in times:
                proto.set_option("generic_parts", K[0]); proto.set_option("generic_stagger", K[1])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(2):
                    proto.verify_batch_device(bench_other.RECIP_LABEL, n, dV.data_ptr(), dP.data_ptr(), *shape, dA.data_ptr(), dS.data_ptr())
                proto.synchronize()
                times[K].append((time.perf_counter() - t0) * 1e3 / 2)
        base = float(np.median(times[variants[0]]))
        for K, t in times.items():
            t = np.array(t)
            print(f"n=2^{n.bit_length() - 1} parts={K[0]} stagger={K[1]}  median {np.median(t):8.3f} ms  min {t.min():8.3f}  {n / np.median(t):8.1f} k/s  vs one part {np.median(t) / base - 1:+.2%}  ok={ok[K]}", flush=True)
    proto.set_option("generic_parts", 0); proto.set_option("generic_stagger", 1)
    proto.close()


if __name__ == "__main__":
    main()
