"""Soak test on one GPU: fresh 2^16-proof batches (product prover), random byte corruptions, exact mode vs RLC mode (bucket stage at
several superchunk sizes, chunks of 8, exact re-check) vs expectation,
and a random sample of every batch re-verified by the CPU oracle.   python tests/soak.py [batches] [log2 of the batch size, default 16]
(batch sizes <= 2^15 run the lane-group kernels of the small-batch path)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/soak.py: test infrastructure (it uses the oracle as its checker)
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
from bp_pp_amd import U64RangeProofProtocol, synth as workload
import bppp_oracle_c as OC
gens = bytes.fromhex(json.load(open(os.path.join(ROOT, "tests", "golden", "u64_golden.json")))["generators"])
g, gv, hv = gens[:64], [gens[64 * i:64 * i + 64] for i in range(1, 17)], [gens[64 * i:64 * i + 64] for i in range(17, 49)]
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 16)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
proto = U64RangeProofProtocol(g, gv, hv, device=0)
torch.cuda.synchronize()   # inputs ready; the context runs on its own (non-blocking) stream, joined by proto.synchronize()
rng = np.random.default_rng(2026)
bad_total = 0
t0 = time.time()
for it in range(K):
    first = (it + 1) * n
    P, V, pst = proto.prove_batch(workload.bulk_values(n, first=first), workload.bulk_blindings(n, first=first), workload.bulk_prover_randomness(n, first=first), workload.LABEL)
    sm = (4096, 256, 1024, 8192, 0)[it % 5]                     # superchunk size of the RLC mode's bucket stage (0 = chunks of 8 only)
    proto.set_option("rlc_superchunk", sm)
    assert not pst.any()
    P, V = P.copy(), V.copy()
    idx = rng.choice(n, size=min(n, int(rng.integers(0, 400)) if it % 3 else int(rng.integers(0, 4))), replace=False)   # every third batch nearly clean
    P0, V0 = P.copy(), V.copy()
    for i in idx:
        kind = int(rng.integers(0, 6))
        j = int(rng.integers(0, n))
        if kind == 0:                                            # random byte anywhere (usually leaves the curve: malformed)
            P[i, int(rng.integers(0, 928))] ^= int(rng.integers(1, 256))
        elif kind == 1:                                          # a scalar byte: well-formed, wrong
            P[i, 832 + int(rng.integers(0, 96))] ^= int(rng.integers(1, 128))
        elif kind == 2:                                          # one point taken from another proof: well-formed, wrong
            k = int(rng.integers(0, 13))
            P[i, 64 * k:64 * k + 64] = P0[j, 64 * k:64 * k + 64] if j != i else P0[(i + 1) % n, 64 * k:64 * k + 64]
        elif kind == 3:                                          # another proof's commitment
            V[i] = V0[(i + 1 + j) % n] if (i + 1 + j) % n != i else V0[(i + 1) % n]
        elif kind == 4:                                          # y negated: on the curve, wrong point
            k = int(rng.integers(0, 13))
            y = int.from_bytes(P[i, 64 * k + 32:64 * k + 64].tobytes(), "big")
            P[i, 64 * k + 32:64 * k + 64] = np.frombuffer(((2**256 - 2**32 - 977 - y) % (2**256 - 2**32 - 977)).to_bytes(32, "big"), np.uint8)
        else:                                                    # two round points swapped
            P[i, 256:320], P[i, 512:576] = P0[i, 512:576].copy(), P0[i, 256:320].copy()
    dV, dP = torch.from_numpy(V).cuda(), torch.from_numpy(P).cuda()
    res = []
    for mode in ("exact", "rlc"):
        dA = torch.full((n,), 9, dtype=torch.uint8, device="cuda"); dS = torch.zeros(n, dtype=torch.int32, device="cuda"); dR = torch.zeros(1, dtype=torch.int32, device="cuda")
        if mode == "exact":
            proto.verify_batch_device(workload.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, dR.data_ptr())
        else:
            proto.verify_batch_rlc_device(workload.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), os.urandom(32), dS.data_ptr(), dR.data_ptr())
        torch.cuda.synchronize()
        res.append((dA.cpu().numpy(), dS.cpu().numpy(), int(dR.item())))
    (a0, s0, r0), (a1, s1, r1) = res
    assert (a0 == a1).all() and (s0 == s1).all() and r0 == r1, "RLC mode disagrees with exact mode"
    untouched = np.ones(n, bool); untouched[idx] = False
    assert a0[untouched].all() and not s0[untouched].any(), "an honest proof was rejected"
    assert not a0[idx].any(), "a corrupted proof was accepted"
    sample = np.concatenate([idx[:64], rng.choice(n, size=min(n, 192), replace=False)])
    oacc, ost = OC.u64_verify_batch(gens, workload.LABEL, V[sample].copy(), P[sample].copy(), nthreads=min(64, os.cpu_count() or 1))
    assert (oacc == a0[sample]).all() and ((ost != 0) == (s0[sample] != 0)).all(), "GPU and CPU oracle disagree"
    bad_total += len(idx)
    print(f"batch {it} (superchunk {sm}): {len(idx)} corrupted, {int((s0 != 0).sum())} malformed, rejects {r0}; exact == rlc == oracle sample  [{time.time() - t0:.0f} s]", flush=True)
print(f"soak ok: {K} batches, {K * n} proofs, {bad_total} corrupted")
