"""The in-process form of the multi-GPU run: ONE process, a bppp_group over N devices (a host thread, a context, a stream and -- for
N > 1 -- an RCCL communicator per device; include/bppp.h), the same fixed synthetic batch bench.py verifies, sharded contiguously,
each shard resident on its device.  Prints ONE JSON line: rccl_nranks, per-rank kernel ms, wall ms per step (all devices done),
the all-reduced reject count on every device and accept_bits_ok.
usage: python tools/group_run.py --gpus N [--total-proofs 1048576] [--steps 5] [--warmup 1] [--fb-window-bits 0]
       BENCH_ONE_DEVICE=1: every rank on device 0 is not possible for a group (devices must be distinct), so the dry run uses ONE
       device with BPPP_FORCE_RCCL=1 (a one-rank communicator: the collective path runs, the split is trivial)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--total-proofs", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--fb-window-bits", type=int, default=0)
    a = ap.parse_args()
    dry = bool(os.environ.get("BENCH_ONE_DEVICE"))
    if dry:
        os.environ["BPPP_FORCE_RCCL"] = "1"
    import numpy as np
    import torch
    import bench
    from bp_pp_amd import synth
    from bp_pp_amd.distributed import U64RangeProofGroup, shard_range
    G = 1 if dry else a.gpus
    if torch.cuda.device_count() < G:
        print(json.dumps({"error": f"{G} devices asked, {torch.cuda.device_count()} visible"}))
        sys.exit(2)
    gens, g, gv, hv = bench.load_generators()
    grp = U64RangeProofGroup(g, gv, hv, list(range(G)), fb_window_bits=a.fb_window_bits)
    total = a.total_proofs
    views, bufs, expects = [], [], []
    for r in range(G):
        lo, hi = shard_range(total, r, G)
        torch.cuda.set_device(r)
        v = grp.protocol(r)
        dV, dP, expect, _ = bench.make_resident_batch(torch, v, synth, lo, hi)
        n = hi - lo
        bufs.append((dV, dP, torch.zeros(n, dtype=torch.uint8, device=f"cuda:{r}"), torch.zeros(n, dtype=torch.int32, device=f"cuda:{r}"),
                     torch.full((1,), -1, dtype=torch.int32, device=f"cuda:{r}")))
        views.append(v)
        expects.append(expect)
    for r in range(G):
        torch.cuda.synchronize(r)
    ptr = lambda k: [b[k].data_ptr() for b in bufs]

    def step():
        grp.verify_batch_device(synth.LABEL, total, ptr(0), ptr(1), ptr(2), ptr(3), ptr(4))

    for _ in range(a.warmup):
        step()
    for v in views:
        v.enable_timing(True)
        v.timings(reset=True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    wall = (time.perf_counter() - t0) / a.steps
    per_rank = []
    for v in views:
        kt = v.timings(reset=True)
        per_rank.append(round(sum(x["total_ms"] for x in kt.values()) / a.steps, 3))
        v.enable_timing(False)
    rejects = [int(b[4].item()) for b in bufs]
    ok = all(bool((bufs[r][2].cpu().numpy() == expects[r]).all()) and not bool(bufs[r][3].any().item()) for r in range(G))
    expected_rejects = len(range(0, total, 1024))
    out = {"tool": "tools/group_run.py (one process, bppp_group)", "n_gpus": G, "rccl_nranks": G if (G > 1 or dry) else 0, "dry_run_one_device": dry,
           "total_proofs": total, "steps": a.steps, "ms_per_step_wall": round(wall * 1e3, 3), "value": round(total / wall, 1), "unit": "verifies/s",
           "per_rank_kernel_ms": per_rank, "max_rank_kernel_ms": max(per_rank), "reject_count_on_every_device": rejects,
           "reject_count_expected": expected_rejects, "accept_bits_ok": ok and all(x == expected_rejects for x in rejects),
           "note": "kernel timing on: the two halves of C0 run back to back, so wall is a few percent above bench.py's untimed step"}
    print(json.dumps(out), flush=True)
    for v in views:
        v.close()
    grp.close()
    sys.exit(0 if out["accept_bits_ok"] else 1)


if __name__ == "__main__":
    main()
