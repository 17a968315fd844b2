#!/usr/bin/env python3
"""Sweep of the final-check strategies of the generic reciprocal verifier at BASELINE configs[4]'s shape (dim_nd 256, dim_np 16) on
one resident batch: exact mode, RLC with chunks of 8 only, RLC with the bucket stage at several superchunk sizes and with the
automatic size -- with the bench's 1/256 corrupted instances and with every instance valid.  One JSON line per point: ms per batch,
verifies/s, per-kernel ms of the final-check stages, accept bits equal to the expectation.
usage: python tools/recip_rlc_sweep.py [log2 n = 15]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench_other as BO
from bp_pp_amd.wnla import ReciprocalRangeProofProtocol

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 15)
steps = 5
proto = ReciprocalRangeProofProtocol(BO.RECIP_ND, BO.RECIP_NP, *BO.recip256_generators(), device=0, fb_window_bits=16)
dV, dP, expect, shape, _, _ = BO.recip256_resident_batch(torch, proto, 0, n)
dA = torch.zeros(n, dtype=torch.uint8, device="cuda")
dS = torch.zeros(n, dtype=torch.int32, device="cuda")
bad = torch.from_numpy(np.nonzero(expect == 0)[0]).cuda()


def run(tag, mode, exp):
    def step():
        if mode == "exact":
            proto.verify_batch_device(BO.RECIP_LABEL, n, dV.data_ptr(), dP.data_ptr(), *shape, dA.data_ptr(), dS.data_ptr())
        else:
            proto.verify_batch_rlc_device(BO.RECIP_LABEL, n, dV.data_ptr(), dP.data_ptr(), *shape, dA.data_ptr(), dS.data_ptr(), os.urandom(32))
    step()
    proto.synchronize()
    proto.enable_timing(True)
    proto.timings(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    proto.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kt = {k: round(v["total_ms"] / steps, 2) for k, v in proto.timings(reset=True).items()
          if v["launches"] and k.startswith(("k_wnla_rlc", "k_bkt", "k_wnla_msm", "k_wnla_accept"))}
    proto.enable_timing(False)
    ok = bool((dA.cpu().numpy() == exp).all())
    print(json.dumps({"n": n, "corruption": tag, "mode": mode, "ms": round(dt * 1e3, 2), "kverifies_s": round(n / dt / 1e3, 1), "kernels_ms": kt,
                      "accept_ok": ok}), flush=True)


for tag in ("1/256 corrupted", "all valid"):
    exp = expect if tag.startswith("1/") else np.ones(n, np.uint8)
    if tag == "all valid":
        dP[bad, -1] = dP[bad, -1] ^ 1
        torch.cuda.synchronize()
    run(tag, "exact", exp)
    for m in (0, 256, 1024, 4096):
        proto.set_option("rlc_superchunk", m)
        run(tag, "rlc chunks of 8 only" if m == 0 else f"rlc bucket M={m}", exp)
proto.close()
