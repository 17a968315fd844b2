"""Soak of the GENERIC reciprocal path on one GPU (test infrastructure; not collected by pytest): random shapes (dim_nd, dim_np), batch
sizes and transcript labels; the GPU prover's first proofs byte-identical to the oracle prover's; then one random byte of most
instances XOR-ed and the batch verified in exact and in RLC mode: the two agree everywhere and a random sample (plus every instance the
GPU accepted although it was touched) equals the C oracle's verdict.    python tests/soak_generic.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import recip_cases
from bp_pp_amd.wnla import ReciprocalRangeProofProtocol

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
SHAPES = [(8, 4), (12, 10), (16, 16), (16, 2), (32, 16), (20, 7), (64, 16), (256, 16)]
rng = np.random.default_rng(77)
t0, it, total, touched_total = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    nd, npp = SHAPES[it % len(SHAPES)]
    B = int(rng.integers(1, 700 if nd <= 32 else 200))
    label = b"soak-generic-" + os.urandom(4).hex().encode()
    n_or = 2 if nd <= 64 else 1
    case = recip_cases.make(nd, npp, B, label=label, n_oracle=n_or) if B <= 255 or nd <= 32 else recip_cases.make(nd, npp, 255, label=label, n_oracle=n_or)
    B = case["x"].shape[0]
    proto = ReciprocalRangeProofProtocol(nd, npp, case["g"], case["gv"], case["hv"], case["gv_"], case["hv_"], device=0,
                                         fb_window_bits=int(rng.choice([4, 8, 10, 16] if nd <= 64 else [8, 10])))
    try:
        com, st = proto.commit_value_batch(case["x"], case["s"])
        assert not st.any() and (com[:case["n_oracle"]] == case["commitments"]).all()
        proofs, st, shape = proto.prove_batch(label, com, case["x"], case["s"], case["digits"], case["m"], case["rnd"])
        assert not st.any() and shape == (case["rounds"], case["nl"], case["nn"])
        assert (proofs[:case["n_oracle"]] == case["proofs"]).all(), ("prover bytes", nd, npp, B)
        P, V = proofs.copy(), com.copy()
        touched = np.zeros(B, bool)
        for i in range(B):
            if rng.random() < 0.3:
                continue
            touched[i] = True
            x = int(rng.integers(1, 256))
            if rng.random() < 0.2:
                V[i, int(rng.integers(0, 64))] ^= x
            else:
                P[i, int(rng.integers(0, P.shape[1]))] ^= x
        acc, st = proto.verify_batch(label, V, P, *shape)
        acc2, st2 = proto.verify_batch_rlc(label, V, P, *shape, seed=os.urandom(32))
        assert (acc2 == acc).all() and (st2 == st).all(), ("exact vs rlc", nd, npp, B)
        assert acc[~touched].all() and not st[~touched].any()
        sample = set(int(i) for i in rng.choice(B, size=min(B, 6 if nd <= 64 else 2), replace=False)) | set(int(i) for i in np.nonzero(acc.astype(bool) & touched)[0])
        for i in sample:
            rc = recip_cases.oracle_verify(case, bytes(V[i]), bytes(P[i]))
            assert int(acc[i]) == (1 if rc == 1 else 0) and (int(st[i]) != 0) == (rc < 0), (nd, npp, B, i, rc, int(acc[i]), int(st[i]))
    finally:
        proto.close()
    it += 1; total += B; touched_total += int(touched.sum())
    if it % 8 == 0:
        print(f"iteration {it}: {total} instances so far, {touched_total} touched  [{time.time() - t0:.0f} s]", flush=True)
print(f"generic soak ok: {it} batches over {len(SHAPES)} shapes, {total} instances, {touched_total} touched; GPU prover == oracle prover (sample), exact == RLC, == oracle (sample)")
