"""Option two_stream_halves on / off at 2^18 .. 2^20 proofs, with the context on its own stream and on a torch stream (bench.py's way).
usage: python tools/twin_probe.py [log2 sizes ...]"""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch, bench
from bp_pp_amd import U64RangeProofProtocol, synth
sizes = [1 << int(a) for a in sys.argv[1:]] or [1 << 18, 1 << 20]
gens, g, gv, hv = bench.load_generators()
for use_torch_stream in (False, True):
    proto = U64RangeProofProtocol(g, gv, hv, device=0)
    st = None
    if use_torch_stream:
        st = torch.cuda.Stream()
        proto.set_stream(st.cuda_stream)
    dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 0, max(sizes))
    for n in sizes:
        dA = torch.zeros(n, dtype=torch.uint8, device="cuda"); dS = torch.zeros(n, dtype=torch.int32, device="cuda")
        for opt in (0, 1, 0, 1):
            proto.set_option("two_stream_halves", opt)
            def fn():
                proto.verify_batch_device(synth.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, 0)
            fn(); torch.cuda.synchronize(); proto.synchronize()
            best = 1e9
            for _ in range(3):
                t = time.perf_counter()
                for _ in range(4): fn()
                proto.synchronize(); torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t) / 4)
            ok = bool((dA.cpu().numpy() == expect[:n]).all())
            print(f"stream {'torch' if use_torch_stream else 'own  '}  n=2^{n.bit_length()-1}  two_stream_halves {opt}: {best*1e3:8.3f} ms  {n/best/1e6:6.3f} M/s  ok {ok}", flush=True)
    proto.close()
