"""One GPU's share of the 8-GPU split (2^17 proofs) and configs[1] (2^16): what do the alternatives to ONE launch sequence cost?

  a) the library's own plan (one context, one sequence);
  b) shared inversions forced on at this size (BPPP_SHARED_INV = 8 / 16: the large-batch form of the table build and the rounds);
  c) the batch as K sub-batches on K child contexts over the same tables (own streams), every sub-batch on the 256-register one-lane
     kernels (BPPP_NO_SMALL_KERNELS) so that wavefronts of DIFFERENT kernels share a SIMD, started `stagger` ms apart.

    python tools/probes/share_probe.py [log2 n ...]      (default 17 16)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from bp_pp_amd import U64RangeProofProtocol, synth

REPS = int(os.environ.get("REPS", "12"))


def child(parent, **env):
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        return parent.clone_shared()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    sizes = [1 << int(a) for a in sys.argv[1:]] or [1 << 17, 1 << 16]
    gens, g, gv, hv = bench.load_generators()
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=int(os.environ.get("FB_WINDOW_BITS", "0")))
    nmax = max(max(sizes), 1 << 20 if os.environ.get("WITH_2POW20") else 0)
    dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 0, nmax)
    dA = torch.zeros(nmax, dtype=torch.uint8, device="cuda"); dS = torch.zeros(nmax, dtype=torch.int32, device="cuda")

    def run_parts(ctxs, n, stagger_ms=0.0):
        k = len(ctxs)
        m = n // k
        for i, c in enumerate(ctxs):
            if i and stagger_ms:
                t_end = time.perf_counter() + stagger_ms * 1e-3
                while time.perf_counter() < t_end:
                    pass
            c.verify_batch_device(synth.LABEL, m, dV[i * m:].data_ptr(), dP[i * m:].data_ptr(), dA[i * m:].data_ptr(), dS[i * m:].data_ptr(), 0, 0)
        for c in ctxs:
            c.synchronize()

    def measure(tag, ctxs, n, stagger_ms=0.0):
        dA.zero_()
        run_parts(ctxs, n, stagger_ms); run_parts(ctxs, n, stagger_ms)
        ts = []
        for _ in range(REPS):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_parts(ctxs, n, stagger_ms)
            ts.append((time.perf_counter() - t0) * 1e3)
        ok = bool((dA[:n].cpu().numpy() == expect[:n]).all())
        ts = np.array(ts)
        print(f"n=2^{n.bit_length() - 1} {tag:58s} median {np.median(ts):7.3f} ms  min {ts.min():7.3f}  -> {n / np.median(ts) / 1e3:6.3f} M/s  ok={ok}  plan[0]: {ctxs[0].last_plan()}", flush=True)
        return float(np.median(ts))

    if os.environ.get("WITH_2POW20"):
        t20 = measure("one sequence (the library's plan)", [proto], 1 << 20)
        print(f"   saturated rate: {t20 / 8:.3f} ms per 2^17 proofs, {t20 / 16:.3f} per 2^16")
    for n in sizes:
        measure("one sequence (the library's plan)", [proto], n)
        for G in (8, 16):
            c = child(proto, BPPP_SHARED_INV=G)
            measure(f"one sequence, BPPP_SHARED_INV={G}", [c], n)
            c.close()
        c = child(proto, BPPP_NO_SMALL_KERNELS=1)
        measure("one sequence, 256-register kernels (BPPP_NO_SMALL_KERNELS)", [c], n)
        c.close()
        for K in (2, 4):
            for fb1 in (1, 0):
                cs = [child(proto, BPPP_NO_SMALL_KERNELS=1, BPPP_FB_ONE_LANE=fb1) for _ in range(K)]
                for stg in (0.0, 0.3, 0.8, 1.5):
                    measure(f"{K} sub-batches, 256-reg kernels, fb_one_lane={fb1}, stagger {stg} ms", cs, n, stg)
                for c in cs:
                    c.close()
        # the same split with each sub-batch on the plan the library picks for its size
        for K in (2,):
            cs = [child(proto) for _ in range(K)]
            for stg in (0.0, 0.8):
                measure(f"{K} sub-batches, each on the library's plan for its size, stagger {stg} ms", cs, n, stg)
            for c in cs:
                c.close()
    proto.close()


if __name__ == "__main__":
    main()
