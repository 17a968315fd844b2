"""When do the wavefronts of k_verify_c0_var start and end?  (-DBPPP_PHASE_TIMING build: lane 0 of every wavefront stamps s_memtime
before and after the 5-point sum, stamps 20 / 21.)   BPPP_LIB=bp_pp_amd/libbppp_hip_pt.so python tools/wave_timeline.py [log2 n ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from bp_pp_amd import U64RangeProofProtocol, synth, _capi
gens, g, gv, hv = bench.load_generators()
proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=16)
L = _capi.lib()
L.bppp_debug_read_stamps.argtypes = [C.c_void_p]
nmax = 1 << 17
dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 0, nmax)
for n in [1 << int(a) for a in sys.argv[1:]] or (1 << 15, 1 << 16, 1 << 17):
    dA = torch.zeros(n, dtype=torch.uint8, device="cuda"); dS = torch.zeros(n, dtype=torch.int32, device="cuda")
    proto.enable_timing(True)            # the C0 halves back to back, as in the per-kernel measurements
    for _ in range(2):
        proto.verify_batch_device(synth.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, 0)
    proto.synchronize()
    kt = proto.timings(reset=True)
    buf = np.zeros((1024, 32), np.uint64)
    assert L.bppp_debug_read_stamps(buf.ctypes.data) == 0
    w = min(1024, n // 64)
    d = buf[:w].astype(np.int64)
    for a, b, name in ((20, 21, "c0_var sum"), (16, 19, "tables"), (9, 13, "round (last launch)")):
        s, e = d[:, a], d[:, b]
        t0 = s.min()
        q = lambda x: [round(float(np.percentile(x, p)) / 1e5, 3) for p in (0, 10, 50, 90, 100)]          # 100 MHz ticks -> ms
        print(f"n=2^{n.bit_length()-1} first {w} wavefronts  {name:20s} start after the first (ms) p0/10/50/90/100 {q(s - t0)}  duration (ms) {q(e - s)}  "
              f"first start to last end {round(float(e.max() - t0) / 1e5, 3)} ms", flush=True)
    print("   kernel ms:", {k.replace('k_verify_', ''): round(v['total_ms'] / 2, 3) for k, v in kt.items() if v['launches']}, flush=True)
proto.close()
