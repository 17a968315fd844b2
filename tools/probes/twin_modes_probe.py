"""Is the twin form's gain at 2^17 proofs real, and what does it depend on?  One process, few contexts; per mode the time of ONE call
(synchronised before and after) and of 8 calls back to back.
modes: one | lib:<BPPP_TWIN_STREAMS> (the library's twin plan, second pair of streams 0 normal / 1 high priority / 2 CU-mask) | ctx2 (two
child contexts, half a batch each)            python tools/probes/twin_modes_probe.py [log2 n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from bp_pp_amd import U64RangeProofProtocol, synth


def main():
    n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 17)
    gens, g, gv, hv = bench.load_generators()
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=int(os.environ.get("FB_WINDOW_BITS", "0")))
    dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 0, n)
    dA = torch.zeros(n, dtype=torch.uint8, device="cuda"); dS = torch.zeros(n, dtype=torch.int32, device="cuda")

    def child(**env):
        for k, v in env.items():
            os.environ[k] = str(v)
        c = proto.clone_shared()
        for k in env:
            os.environ.pop(k)
        return c

    def run(cs):
        m = n // len(cs)
        for i, c in enumerate(cs):
            c.verify_batch_device(synth.LABEL, m, dV[i * m:].data_ptr(), dP[i * m:].data_ptr(), dA[i * m:].data_ptr(), dS[i * m:].data_ptr(), 0, 0)

    def sync(cs):
        for c in cs:
            c.synchronize()

    modes = [("one", lambda: [child(BPPP_TWIN=0)])]
    for kind in (0, 1, 2):
        modes.append((f"lib:{kind}", lambda kind=kind: [child(BPPP_TWIN=1, BPPP_TWIN_STREAMS=kind)]))
    modes.append(("ctx2", lambda: [child(BPPP_NO_SMALL_KERNELS=1, BPPP_FB_ONE_LANE=1, BPPP_TWIN=0) for _ in range(2)]))
    modes.append(("one+pace", lambda: [child(BPPP_TWIN=0, BPPP_PACE=1)]))
    if os.environ.get("SMALL_MODES"):      # 2^16 and below: K contexts on the one-lane 256-register kernels (no lane groups), fixed-base sums on 8 lanes / 1 lane
        for K in (2, 4):
            for fb1 in (0, 1):
                modes.append((f"ctx{K}-fb{'l1' if fb1 else 'l8'}", lambda K=K, fb1=fb1: [child(BPPP_NO_SMALL_KERNELS=1, BPPP_NO_LANE_GROUPS=1, BPPP_FB_ONE_LANE=fb1, BPPP_TWIN=0)
                                                                                         for _ in range(K)]))
    for rnd in range(2):                      # the whole list twice: drift shows as a difference between the rounds
        for name, make in modes:
            cs = make()
            dA.zero_()
            run(cs); sync(cs); run(cs); sync(cs)
            ok = bool((dA.cpu().numpy() == expect).all())
            single, b2b = [], []
            for _ in range(9):
                torch.cuda.synchronize()
                t0 = time.perf_counter(); run(cs); sync(cs); single.append((time.perf_counter() - t0) * 1e3)
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(8):
                    run(cs)
                sync(cs)
                b2b.append((time.perf_counter() - t0) * 1e3 / 8)
            print(f"n=2^{n.bit_length() - 1} round {rnd} {name:9s} one call {np.median(single):7.3f} ms (min {min(single):7.3f})   8 back to back {np.median(b2b):7.3f} ms per call (min {min(b2b):7.3f})  ok={ok}  {cs[0].last_plan()[60:]}", flush=True)
            for c in cs:
                c.close()
    proto.close()


if __name__ == "__main__":
    main()
