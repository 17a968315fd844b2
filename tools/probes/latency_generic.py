"""Latency of one call of the generic entry points (WeightNormLinearArgument N = 4: tests.rs:139-171; ReciprocalRangeProofProtocol
with the u64 shape dim_nd = dim_np = 16) -- for comparison with the u64-specialised path.  usage: python tools/latency_generic.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import wnla_cases, recip_cases, circuit_cases
from bp_pp_amd.wnla import WeightNormLinearArgument, ReciprocalRangeProofProtocol, ArithmeticCircuit

def med(f, reps=9):
    f(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e3

for ng, nh in ((4, 4), (16, 32)):
    for B in (1, 64):
        case = wnla_cases.make(ng, nh, B)
        w = WeightNormLinearArgument(case["g"], case["gv"], case["hv"], device=0, fb_window_bits=16)
        args = dict(commitments=case["commitments"], c=case["c"], rho=case["rho"], mu=case["mu"], proof_r=case["proof_r"],
                    proof_x=case["proof_x"], proof_l=case["proof_l"], proof_n=case["proof_n"])
        tv = med(lambda: w.verify_batch(case["label"], **args))
        tp = med(lambda: w.prove_batch(case["label"], case["commitments"], case["c"], case["rho"], case["mu"], case["l"], case["n"]))
        print(f"wnla ng {ng} nh {nh} rounds {case['rounds']}  B {B:3d}  verify {tv:7.3f} ms  prove {tp:7.3f} ms")
        w.enable_timing(True); w.timings()
        for _ in range(5):
            w.verify_batch(case["label"], **args)
        kt = w.timings(); w.enable_timing(False)
        print("      verify kernels, ms per call:", {k.replace("k_wnla_", ""): round(v["total_ms"] / 5, 3) for k, v in kt.items() if v["launches"]},
              "sum", round(sum(v["total_ms"] for v in kt.values()) / 5, 3))
        w.close()
for B in (1, 64):
    case = recip_cases.make(16, 16, B=B)
    r = ReciprocalRangeProofProtocol(16, 16, case["g"], case["gv"], case["hv"], case["gv_"], case["hv_"], device=0, fb_window_bits=16)
    tv = med(lambda: r.verify_batch(case["label"], case["commitments"], case["proofs"], case["rounds"], case["nl"], case["nn"]))
    print(f"reciprocal (16, 16)  B {B:3d}  verify {tv:7.3f} ms")
    r.close()

# one ArithmeticCircuit::verify (circuit.rs:154-256): the reference's own ac_works statement and the k = 2 statement of the bench line
for name in ("ac_works", "mixed_k2"):
    for B in (1, 64):
        case = circuit_cases.make(name, B)
        part = lambda typ, j: (None if case["part"][typ][j] < 0 else int(case["part"][typ][j]))
        arr = lambda b: np.frombuffer(b, np.uint8).reshape(-1, 32)
        circ = ArithmeticCircuit(case["nm"], case["no"], case["k"], case["nv"], case["g"], case["gv"], case["hv"], arr(case["Wm_bytes"]),
                                 arr(case["Wl_bytes"]), arr(case["am_bytes"]), arr(case["al_bytes"]), case["f_l"], case["f_m"], case["gv_"],
                                 case["hv_"], part, device=0, fb_window_bits=16)
        shape = (case["rounds"], case["pl"], case["pn"])
        tv = med(lambda: circ.verify_batch(case["label"], case["commitments"], case["proofs"], *shape))
        print(f"circuit {name} rounds {case['rounds']}  B {B:3d}  verify {tv:7.3f} ms")
        circ.enable_timing(True); circ.timings()
        for _ in range(5):
            circ.verify_batch(case["label"], case["commitments"], case["proofs"], *shape)
        kt = circ.timings(); circ.enable_timing(False)
        print("      verify kernels, ms per call:", {k.replace("k_", ""): round(v["total_ms"] / 5, 3) for k, v in kt.items() if v["launches"]},
              "sum", round(sum(v["total_ms"] for v in kt.values()) / 5, 3))
        circ.close()
