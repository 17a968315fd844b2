"""Soak of the single-proof front end on a GPU (not collected by pytest): T host threads call verify_one / prove_one for `seconds`
seconds -- valid, wrong and malformed proofs under several labels -- while the main thread keeps changing the coalesce_* options (every
change drains the running front end under load and the next call starts a new one).  Every verdict is compared with the oracle's (computed
up front), every proof made by prove_one is verified.   usage: python tests/soak_coalesce.py [seconds=60] [threads=96]"""
import os, sys, threading, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import bppp_oracle_c as OC
import workload
from bp_pp_amd import U64RangeProofProtocol

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 96
LABELS = [workload.LABEL, b"soak label two", b""]
gens, dl = workload.generators(), workload.generator_dlogs()
g, gv, hv = workload.split_generators(gens)
pool = []
for li, label in enumerate(LABELS):
    n = 192
    x, s, rnd = workload.values(n, first=20000 + 1000 * li), workload.blindings(n, first=20000 + 1000 * li), workload.prover_randomness(n, first=20000 + 1000 * li)
    P, V = OC.u64_prove_trapdoor_batch(dl, label, x, s, rnd, nthreads=8)
    P, V = P.copy(), V.copy()
    for j in range(n):
        if j % 4 == 1: P[j, 840 + j % 80] ^= 1 + j % 5
        if j % 11 == 3: P[j, 64 * (j % 13) + 9] ^= 0x08
    acc, st = OC.u64_verify_batch(gens, label, V, P, nthreads=8)
    pool += [(label, bytes(V[j]), bytes(P[j]), int(acc[j]), int(st[j]) < 0, int(x[j]), bytes(s[j]), bytes(rnd[j])) for j in range(n)]
proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=16)
stop = threading.Event()
errors, counts = [], [0] * T

def worker(t):
    rng = random.Random(t)
    try:
        while not stop.is_set():
            label, V, P, acc, flagged, x, s, rnd = pool[rng.randrange(len(pool))]
            if rng.random() < 0.1:
                proof, com, st = proto.prove_one(x, s, label, rnd)
                if st != 0 or com != V: raise AssertionError(("prove_one", st))
                a, st2 = proto.verify_one(com, proof, label)
                if not a or st2: raise AssertionError(("prove then verify", a, st2))
            else:
                a, st = proto.verify_one(V, P, label)
                if int(a) != acc or (st != 0) != flagged: raise AssertionError(("verify_one", a, st, acc, flagged))
            counts[t] += 1
    except Exception as e:                 # noqa: BLE001
        errors.append((t, repr(e)))
        stop.set()

th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
t0 = time.time()
for t in th: t.start()
changes = 0
opts = [("coalesce_us", (0, 50, 100, 400)), ("coalesce_max", (1, 7, 64, 1024)), ("coalesce_lanes", (1, 2, 3)), ("ct_prover", (0, 1))]
rng = random.Random(99)
while time.time() - t0 < seconds and not stop.is_set():
    time.sleep(0.3)
    name, vals = opts[rng.randrange(len(opts))]
    proto.set_option(name, vals[rng.randrange(len(vals))])
    changes += 1
stop.set()
for t in th: t.join(timeout=120)
hung = [i for i, t in enumerate(th) if t.is_alive()]
total = sum(counts)
print(f"soak_coalesce: {T} threads, {time.time() - t0:.1f} s, {total} single-proof calls ({total / (time.time() - t0):.0f}/s through the interpreter), "
      f"{changes} option changes under load, errors {len(errors)}, hung threads {len(hung)}", flush=True)
if errors: print(errors[:5])
proto.close()
sys.exit(1 if errors or hung else 0)
