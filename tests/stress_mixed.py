"""Mixed-API stress on ONE context (test infrastructure; not collected by pytest): prove, verify (exact / RLC / transcript / SEC1 /
host and device buffers), commitments and the one-device group, interleaved with batch sizes that switch between the lane-group,
small and full-occupancy kernels, so that every workspace is re-used across call types.  Every result is checked against what the
first pass established (and that pass against the oracle on a sample).   python tests/stress_mixed.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import workload
import bppp_oracle_c as OC
from bp_pp_amd import U64RangeProofProtocol, synth, wire
from bp_pp_amd.transcript import Transcript

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
gens = workload.generators()
g, gv, hv = workload.split_generators(gens)
proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=16)
sizes = [1, 7, 64, 65, 1000, 5000, 16384, 16385, 40000, 70000, 140000]
nmax = max(sizes)
x, s, rnd = synth.bulk_values(nmax, first=5), synth.bulk_blindings(nmax, first=5), synth.bulk_prover_randomness(nmax, first=5)
P0, V0, st = proto.prove_batch(x, s, rnd, synth.LABEL)
assert not st.any()
Pref, Vref = OC.u64_prove_batch(gens, synth.LABEL, x[:64], s[:64], rnd[:64], nthreads=8)
assert (Pref == P0[:64]).all() and (Vref == V0[:64]).all()
Pc, expect = workload.corrupt(P0, V0, every=37)
T = Transcript(synth.LABEL)
rng = np.random.default_rng(1)
t0, it = time.time(), 0
while time.time() - t0 < budget:
    n = int(rng.choice(sizes))
    lo = int(rng.integers(0, nmax - n + 1))
    V, P, e = V0[lo:lo + n], Pc[lo:lo + n], expect[lo:lo + n]
    kind = it % 7
    if kind == 0:
        acc, stt = proto.verify_batch(V, P, synth.LABEL)
    elif kind == 1:
        acc, stt = proto.verify_batch_rlc(V, P, synth.LABEL, seed=os.urandom(32))
    elif kind == 2:
        acc, stt, _ = proto.verify_batch_transcript(V, P, T)
    elif kind == 3:
        dV, dP = torch.from_numpy(np.ascontiguousarray(V)).cuda(), torch.from_numpy(np.ascontiguousarray(P)).cuda()
        dA, dS, dR = torch.zeros(n, dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        proto.verify_batch_device(synth.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, dR.data_ptr())
        proto.synchronize()
        acc, stt = dA.cpu().numpy(), dS.cpu().numpy()
        assert int(dR.item()) == int((e == 0).sum())
    elif kind == 4:
        m = min(n, 20000)
        Pp, Vp, ps = proto.prove_batch(x[lo:lo + m], s[lo:lo + m], rnd[lo:lo + m], synth.LABEL)
        assert not ps.any() and (Pp == P0[lo:lo + m]).all() and (Vp == V0[lo:lo + m]).all()
        acc, stt = proto.verify_batch(V, P, synth.LABEL)
    elif kind == 5:
        m = min(n, 5000)
        c33 = np.frombuffer(b"".join(wire.compress_point(bytes(v)) for v in V[:m]), np.uint8).reshape(m, 33)
        p525 = np.frombuffer(b"".join(wire.abi_to_sec1(bytes(p)) for p in P0[lo:lo + m]), np.uint8).reshape(m, 525)
        a5, s5 = proto.verify_batch_sec1(c33, p525, synth.LABEL)
        assert a5.all() and not s5.any()
        acc, stt = proto.verify_batch(V, P, synth.LABEL)
    else:
        m = min(n, 4096)
        assert (proto.commit_value_batch(x[lo:lo + m], s[lo:lo + m]) == V0[lo:lo + m]).all()
        acc, stt = proto.verify_batch_rlc(V, P, synth.LABEL, seed=bytes(32))
    assert (acc == e).all() and not stt.any(), (it, kind, n, lo)
    it += 1
proto.close()
print(f"stress ok: {it} mixed calls in {time.time() - t0:.0f} s")
