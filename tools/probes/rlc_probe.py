"""Per-kernel times of the RLC batch mode (2^16 proofs resident in HBM)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bp_pp_amd import U64RangeProofProtocol, synth as workload
gens = bytes.fromhex(json.load(open(os.path.join(ROOT, "tests", "golden", "u64_golden.json")))["generators"])
g, gv, hv = gens[:64], [gens[64 * i:64 * i + 64] for i in range(1, 17)], [gens[64 * i:64 * i + 64] for i in range(17, 49)]
n = 1 << 16
proto = U64RangeProofProtocol(g, gv, hv, device=0)
P, V, pst = proto.prove_batch(workload.values(n), workload.blindings(n), workload.prover_randomness(n), workload.LABEL)
every = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P, expect = workload.corrupt(P, every=every)
dV, dP = torch.from_numpy(V).cuda(), torch.from_numpy(P).cuda()
dA = torch.zeros(n, dtype=torch.uint8, device="cuda"); dS = torch.zeros(n, dtype=torch.int32, device="cuda"); dR = torch.zeros(1, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()   # inputs ready; the context runs on its own (non-blocking) stream, joined by proto.synchronize()
seed = os.urandom(32)
for _ in range(2):
    proto.verify_batch_rlc_device(workload.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), seed, dS.data_ptr(), dR.data_ptr())
torch.cuda.synchronize()
proto.enable_timing(True); proto.timings(reset=True)
K = 4
for _ in range(K):
    proto.verify_batch_rlc_device(workload.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), seed, dS.data_ptr(), dR.data_ptr())
torch.cuda.synchronize()
t = proto.timings(reset=True)
print("corrupt every", every, {k: round(v["total_ms"] / K, 3) for k, v in t.items()}, "sum", round(sum(v["total_ms"] for v in t.values()) / K, 2),
      "accept ok", bool((dA.cpu().numpy() == expect).all()))
