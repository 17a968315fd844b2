"""Reads the phase stamps of a -DBPPP_PHASE_TIMING build (see verify_ws.h: BPPP_STAMP): shader-clock deltas between the marked
points of verify_phase1 and verify_round, averaged over the wavefronts.   BPPP_LIB=.../libbppp_hip_pt.so python tools/phase_probe.py"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bp_pp_amd import U64RangeProofProtocol, synth as workload, _capi
gens = bytes.fromhex(json.load(open(os.path.join(ROOT, "tests", "golden", "u64_golden.json")))["generators"])
g, gv, hv = gens[:64], [gens[64 * i:64 * i + 64] for i in range(1, 17)], [gens[64 * i:64 * i + 64] for i in range(17, 49)]
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 16)
proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=16)
P, V, pst = proto.prove_batch(workload.values(n), workload.blindings(n), workload.prover_randomness(n), workload.LABEL)
dV, dP = torch.from_numpy(V).cuda(), torch.from_numpy(P).cuda()
dA = torch.zeros(n, dtype=torch.uint8, device="cuda"); dS = torch.zeros(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()   # inputs ready; the context runs on its own (non-blocking) stream, joined by proto.synchronize()
for _ in range(2):
    proto.verify_batch_device(workload.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, 0)
torch.cuda.synchronize()
buf = np.zeros((1024, 32), np.uint64)
L = _capi.lib()
L.bppp_debug_read_stamps.argtypes = [C.c_void_p]
assert L.bppp_debug_read_stamps(buf.ctypes.data) == 0
names = ["p1 decode", "p1 append V + challenge e + V+r", "p1 append cl cr co v", "p1 4 challenges", "p1 append cs + tau (+stores)", "p1 inversions",
         "p1 scalars loop", "p1 cr_tau etc", None, "round: to affine", "round: loads + transcript", "round: stores + glv", "round: straus", None]
d = buf[:max(1, min(1024, n // 64))].astype(np.int64)
for i, nm in enumerate(names):
    if nm is None:
        continue
    delta = (d[:, i + 1] - d[:, i])
    print(f"{nm:36s} {delta.mean() / 100:10.1f} us  (100 MHz s_memrealtime ticks; min {delta.min()}, max {delta.max()})")
for a, b, nm in ((16, 17, "tables: forward (multiples + prefix products)"), (17, 18, "tables: inversion"), (18, 19, "tables: backward (affine, pack, store)"),
                 (20, 21, "c0_var: straus (5 points)")):
    delta = d[:, b] - d[:, a]
    print(f"{nm:48s} {delta.mean() / 100:10.1f} us")
print("n", n, "phase1 total us", (d[:, 8] - d[:, 0]).mean() / 100, " round total (9..13) us", (d[:, 13] - d[:, 9]).mean() / 100)
