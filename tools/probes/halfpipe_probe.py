"""Does a shard of 2^17 proofs (the per-GPU share of BASELINE configs[2] on 8 GPUs) run faster as TWO half batches on two
contexts / streams sharing the tables (kernels of one half under those of the other) than as one batch?   python tools/halfpipe_probe.py"""
import sys, time
sys.path[:0] = ['.']
import torch, bench
from bp_pp_amd import U64RangeProofProtocol, synth
gens, g, gv, hv = bench.load_generators()
proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=int(__import__("os").environ.get("PROBE_W", "0")))
other = proto.clone_shared()
for n in [int(a) for a in sys.argv[1:]] or (1 << 16, 1 << 17, 1 << 18):
    dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 0, n)
    dA = torch.zeros(n, dtype=torch.uint8, device="cuda"); dS = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    def one():
        proto.verify_batch_device(synth.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, 0)
        proto.synchronize()
    def split(parts):
        ctxs = [proto, other]
        m = n // parts
        for i in range(parts):
            ctxs[i % 2].verify_batch_device(synth.LABEL, m, dV.data_ptr() + i * m * 64, dP.data_ptr() + i * m * 928, dA.data_ptr() + i * m, dS.data_ptr() + i * m * 4, 0, 0)
        proto.synchronize(); other.synchronize()
    cases = [("one batch", one), ("2 halves on 2 streams", lambda: split(2)), ("4 quarters on 2 streams", lambda: split(4))]
    if n >= (1 << 19):
        cases.append(("8 eighths on 2 streams", lambda: split(8)))
    for name, fn in cases:
        fn(); dA.zero_(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            for _ in range(5): fn()
            best = min(best, (time.perf_counter() - t) / 5)
        ok = bool((dA.cpu().numpy() == expect).all())
        print(f"n {n:7d}  {name:26s} {best*1e3:8.3f} ms  {n/best/1e6:6.3f} M/s  ok {ok}")
other.close(); proto.close()
