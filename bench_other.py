"""bench.py --workload prove | recip256: the two BASELINE.json configurations beside the headline verify metric, each with its own
JSON line carrying `roofline` (dominant kernel, live HIP-event time, algorithmic bytes of SURVEY.md 8d) and `cpu_baseline`
(the oracle on a bounded sample).  Inputs resident in HBM when the timed region starts.  Both shard ONE fixed batch over the ranks like
the headline metric does (`prove`: BASELINE configs[3] is 2^14 values on one GPU, N > 1 has no exchange step; `recip256`: BASELINE
configs[4], 2^18 instances on 8 GPUs).
The measure_* functions return the JSON object: bench.py's default line embeds reduced-size runs of both as secondary objects."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def _dominant(kernel_times):
    name, t = max(kernel_times.items(), key=lambda kv: kv[1]["total_ms"])
    return name, t, t["total_ms"] / max(1, t["launches"])


def two_passes(proto, step, fence, steps):
    """(seconds for `steps` steps with kernel timing OFF -- the plan a caller gets, the source of `value` --, seconds for the same steps with
    per-kernel HIP events on, the per-kernel times of that second pass).  See bench.run_verify."""
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    proto.enable_timing(True)
    proto.timings(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed_timed = time.perf_counter() - t0
    kt = {k: v for k, v in proto.timings(reset=True).items() if v["launches"]}
    proto.enable_timing(False)
    return elapsed, elapsed_timed, kt


def measure_prove(args, proto, gens, total, cpu_baseline=True, cpu_sample=4096, world=1, rank=0):
    """BASELINE configs[3] on an existing u64 context: batch-prove ONE fixed batch of `total` u64 values, this rank's contiguous shard
    of it resident in HBM (u64_proof.rs:57-82 -> circuit.rs:260-556 -> wnla.rs:125-190).  Proofs are independent: with world > 1
    there is no exchange step at all, only the barrier and the max-over-ranks clock of the bench contract.  Returns the JSON object
    (value, ms_per_step, roofline, cpu_baseline, ...)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import bench
    from bp_pp_amd import synth as workload
    from bp_pp_amd.distributed import shard_range
    lo, hi = shard_range(total, rank, world)
    n = hi - lo
    dist_on = world > 1 and dist.is_initialized()
    x_h, s_h, r_h = workload.bulk_values(n, first=lo), workload.bulk_blindings(n, first=lo), workload.bulk_prover_randomness(n, first=lo)
    dx = torch.from_numpy(x_h.view(np.int64)).cuda()
    ds, dr = torch.from_numpy(s_h).cuda(), torch.from_numpy(r_h).cuda()
    dP = torch.zeros((n, 928), dtype=torch.uint8, device="cuda")
    dV = torch.zeros((n, 64), dtype=torch.uint8, device="cuda")
    dSt = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    def step():
        proto.prove_batch_device(workload.LABEL, n, dx.data_ptr(), ds.data_ptr(), dr.data_ptr(), dP.data_ptr(), dV.data_ptr(), dSt.data_ptr())

    for _ in range(args.warmup):
        step()
    def fence():
        proto.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed, elapsed_timed, kt = two_passes(proto, step, fence, args.steps)
    plan = proto.last_plan(prove=True)
    if dist_on:
        t = torch.tensor([elapsed, elapsed_timed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_timed = float(t[0].item()), float(t[1].item())
    # what was timed is correct: no status flag, and the product verifier accepts every proof
    P, V = dP.cpu().numpy(), dV.cpu().numpy()
    acc, vst = proto.verify_batch(V, P, workload.LABEL)
    ok = bool(acc.all()) and not vst.any() and not bool(dSt.any().item())
    if dist_on:
        ok_t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        ok = bool(ok_t.item())
    dom, dom_t, avg_ms = _dominant(kt)
    launches_per_step = dom_t["launches"] / args.steps
    achieved = bench.ALGO_BYTES_PER_PROVE * n / (avg_ms * 1e-3) / 1e9
    result = {
        "metric": "u64 range proofs proved/sec (batch)", "value": total * args.steps / elapsed, "unit": "proves/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"batch prove ONE fixed batch of {total} u64 values (BASELINE configs[3]: 2^14 on one GPU)"
                               + (f", sharded contiguously over {world} GPUs ({n} per GPU), no exchange step" if world > 1 else " on one GPU")
                               + "; x, s and the 52 prover scalars per proof resident in HBM, device-side transcripts, proofs byte-identical "
                               "to the CPU prover's for the same draws",
                   "total_proofs_per_step": total, "proofs_per_gpu": n, "fb_window_bits": args.fb_window_bits or "library default",
                   "label": workload.LABEL.decode(), "parallelism": "single" if world == 1 else f"shard{world}", "plan": plan,
                   "kernel_timing_during_value": False},
        "timing_pass_ms_per_step": elapsed_timed / args.steps * 1e3,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / bench.HBM_PEAK_GBS,
                     "traffic": bench.pmc_traffic(dom, n), "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step,
                     "algorithmic_bytes_per_launch": bench.ALGO_BYTES_PER_PROVE * n,
                     "note": "2262 B/prove (SURVEY.md 8d) per launch of the dominant kernel; that kernel runs launches_per_step times per proof "
                             "batch (15 fixed-base MSMs per proof; the independent ones of a stage go out as one launch), so the per-step figure is value x 2262 B"},
        "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in kt.items()},
        "proofs_verify": ok,
    }
    if world == 1:
        # the same batch with the secret-scalar sums in the constant-address form (bppp_ctx_set_option "ct_prover")
        dP2 = torch.zeros_like(dP)
        dV2 = torch.zeros_like(dV)
        proto.set_option("ct_prover", 1)
        try:
            def step_ct():
                proto.prove_batch_device(workload.LABEL, n, dx.data_ptr(), ds.data_ptr(), dr.data_ptr(), dP2.data_ptr(), dV2.data_ptr(), dSt.data_ptr())
            step_ct()
            proto.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step_ct()
            proto.synchronize()
            t_ct = (time.perf_counter() - t0) / args.steps
        finally:
            proto.set_option("ct_prover", 0)
        result["ct_prover"] = {"value": n / t_ct, "unit": "proves/s", "ms_per_step": t_ct * 1e3, "cost_vs_default": t_ct / (elapsed / args.steps),
                               "byte_identical_to_default": bool((dP2 == dP).all().item() and (dV2 == dV).all().item()),
                               "note": "V, r_com, c_o, c_l, c_r, c_s over a 3 MB 4-bit table: every entry of every window read, masked select, complete additions"}
        del dP2, dV2
    if cpu_baseline and rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import bppp_oracle_c as OC                                   # the oracle, as the timed CPU baseline ONLY
        m = min(cpu_sample, n)                                       # ~10-20 s of host work
        host = bench.host_summary()
        hw = host["usable_cpus"]                                     # min(affinity, cgroup CPU quota)
        th = hw
        t0 = time.perf_counter()
        Pref, Vref = OC.u64_prove_batch(gens, workload.LABEL, x_h[:m], s_h[:m], r_h[:m], nthreads=th)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        OC.u64_prove_batch(gens, workload.LABEL, x_h[:8], s_h[:8], r_h[:8], nthreads=1)
        single = 8 / (time.perf_counter() - t1)
        result["cpu_baseline"] = {"value": m / dt, "unit": "proves/s", "cores": th, "kind": "port",
                                  "sample": f"first {m} values of the same batch, reference-shaped C prover (oracle/bppp_ref.c), {th} threads, "
                                            f"{dt:.2f} s wall; {host['cpu_model']}: {host['affinity_cpus']} CPUs in the affinity mask, cgroup CPU quota "
                                            f"{host['cgroup_cpu_quota']}",
                                  "host": host, "single_thread_value": single, "byte_identical_to_gpu": bool((Pref == P[:m]).all() and (Vref == V[:m]).all())}
    return result, ok


def run_prove(args):
    """bench.py --workload prove: BASELINE configs[3], its own JSON line."""
    import bench
    world, rank, local_rank = bench.setup_dist(args)
    from bp_pp_amd import U64RangeProofProtocol
    gens, g, gv, hv = bench.load_generators()
    proto = U64RangeProofProtocol(g, gv, hv, device=local_rank, fb_window_bits=args.fb_window_bits)
    result, ok = measure_prove(args, proto, gens, args.total_proofs, cpu_baseline=not args.no_cpu_baseline, world=world, rank=rank)
    if rank == 0:
        print(json.dumps(result), flush=True)
    proto.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


RECIP_ND, RECIP_NP, RECIP_LABEL, RECIP_SLICE = 256, 16, b"reciprocal bench", 1 << 14


def recip256_generators():
    """(g, g_vec[256], h_vec[266], g_vec_[0], h_vec_[246]) from the product's own derivation (SHAKE256 try-and-increment, host code)."""
    from bp_pp_amd import derive_generators
    nd, nh, NG, NH = RECIP_ND, RECIP_ND + 10, 256, 512
    raw = derive_generators(b"bppp-bench-recip256", 1 + NG + NH)
    pts = [raw[64 * i:64 * i + 64] for i in range(1 + NG + NH)]
    return (pts[0], pts[1:1 + nd], pts[1 + NG:1 + NG + nh], pts[1 + nd:1 + NG], pts[1 + NG + nh:])


def recip256_inputs(first, count):
    """Instances first .. first + count of THE fixed global batch: slice k (2^14 instances) is bp_pp_amd.synth's seeded draw number k, so
    any rank can produce exactly its own shard.  No curve arithmetic, no oracle."""
    import numpy as np
    from bp_pp_amd import synth
    parts, g = [], first
    while g < first + count:
        k, off = divmod(g, RECIP_SLICE)
        take = min(first + count - g, RECIP_SLICE - off)
        w = synth.bulk_reciprocal_inputs(RECIP_ND, RECIP_SLICE, seed=20260 + k)
        parts.append({key: v[off:off + take] for key, v in w.items()})
        g += take
    return {key: np.ascontiguousarray(np.concatenate([p_[key] for p_ in parts])) for key in parts[0]}


def recip256_resident_batch(torch, proto, lo, hi, corrupt_every=256):
    """Instances lo..hi of the global batch, proved by the product prover on this GPU slice by slice (host buffers: the prover's
    inputs are 25 KB per instance) and left resident: (dV [n, 64], dP [n, proof_bytes], expect [n], shape, seconds proving,
    host copies of the first min(n, 64) instances' inputs / commitments / proofs for the oracle sample).  One instance in
    `corrupt_every` (by GLOBAL index) gets the last bit of its final scalar flipped and must be rejected."""
    import numpy as np
    n = hi - lo
    dV = dP = shape = None
    t_prove, head = 0.0, None
    for a in range(0, n, RECIP_SLICE):
        b = min(n, a + RECIP_SLICE)
        w = recip256_inputs(lo + a, b - a)
        com, cst = proto.commit_value_batch(w["x"], w["s"])
        t0 = time.perf_counter()
        proofs, pst, shape = proto.prove_batch(RECIP_LABEL, com, w["x"], w["s"], w["digits"], w["m"], w["rnd"])
        t_prove += time.perf_counter() - t0
        assert not cst.any() and not pst.any() and shape == (8, 2, 1)
        if dV is None:
            dV = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
            dP = torch.empty((n, proofs.shape[1]), dtype=torch.uint8, device="cuda")
            m = min(b - a, 64)
            head = dict({k: v[:m].copy() for k, v in w.items()}, com=com[:m].copy(), proofs=proofs[:m].copy())
        dV[a:b] = torch.from_numpy(com).cuda()
        dP[a:b] = torch.from_numpy(proofs).cuda()
    expect = np.ones(n, np.uint8)
    bad = np.arange((-lo) % corrupt_every, n, corrupt_every, dtype=np.int64)
    if len(bad):
        tb = torch.from_numpy(bad).cuda()
        dP[tb, -1] = dP[tb, -1] ^ 1
        expect[bad] = 0
    torch.cuda.synchronize()
    return dV, dP, expect, shape, t_prove, head


def measure_recip256(args, total, W, cpu_baseline=True, rlc=True, dist_on=False, world=1, rank=0, local_rank=0):
    """BASELINE configs[4]'s shape: ReciprocalRangeProofProtocol { dim_nd: 256, dim_np: 16 } (reciprocal.rs:98-107): |g_vec| 256,
    |h_vec| 266 + 246 padding, 8 WNLA rounds, proof = 21 points + 3 scalars.  One committed value with 256 hex digits -- the closest
    thing the reference's API can express to "aggregated 16 values" (SURVEY.md 8d, config 5).  ONE fixed batch of `total` instances,
    rank r of `world` verifies shard_range(total, r, world); the single exchange is the all-reduce of the reject count."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import bench
    from bp_pp_amd.distributed import all_reduce_reject_count, shard_range
    from bp_pp_amd.wnla import ReciprocalRangeProofProtocol
    nd, npp = RECIP_ND, RECIP_NP
    lo, hi = shard_range(total, rank, world)
    n = hi - lo
    n_or = 48 if (cpu_baseline and rank == 0 and world == 1) else 0      # ~10 s of single-thread oracle work (prove + verify)
    gens5 = recip256_generators()
    t0 = time.time()
    proto = ReciprocalRangeProofProtocol(nd, npp, *gens5, device=local_rank, fb_window_bits=W)
    proto.synchronize()
    t_ctx = time.time() - t0
    t0 = time.time()
    dV, dP, expect, shape, t_prove, head = recip256_resident_batch(torch, proto, lo, hi)
    t_setup = time.time() - t0
    dA = torch.zeros(n, dtype=torch.uint8, device="cuda")
    dS = torch.zeros(n, dtype=torch.int32, device="cuda")
    dR = torch.zeros(1, dtype=torch.int32, device="cuda")
    # one explicit stream for the verify kernels, the reject count and the accept-reduce (see bench.run_verify)
    stream = torch.cuda.Stream()
    proto.set_stream(stream.cuda_stream)

    def step(seed=None, acc=dA, st=dS, rej=dR):
        with torch.cuda.stream(stream):
            if seed is None:
                proto.verify_batch_device(RECIP_LABEL, n, dV.data_ptr(), dP.data_ptr(), *shape, acc.data_ptr(), st.data_ptr())
            else:
                proto.verify_batch_rlc_device(RECIP_LABEL, n, dV.data_ptr(), dP.data_ptr(), *shape, acc.data_ptr(), st.data_ptr(), seed)
            rej.copy_((acc == 0).sum(dtype=torch.int32).reshape(1))
            all_reduce_reject_count(rej)            # the single accept-reduce (4 bytes over RCCL/xGMI); no-op at N = 1

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        if dist_on:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        step()
    elapsed, elapsed_timed, kt = two_passes(proto, step, fence, args.steps)
    elapsed, elapsed_timed = max_over_ranks(elapsed), max_over_ranks(elapsed_timed)
    acc, st = dA.cpu().numpy(), dS.cpu().numpy()
    rejects, expected_rejects = int(dR.item()), len(range(0, total, 256))
    ok_t = torch.tensor([1 if ((acc == expect).all() and not st.any()) else 0], dtype=torch.int32, device="cuda")
    if dist_on:
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
    ok = bool(ok_t.item()) and rejects == expected_rejects
    dom, dom_t, avg_ms = _dominant(kt)
    achieved = bench.ALGO_BYTES_PER_RECIP256 * n / (avg_ms * 1e-3) / 1e9
    result = {
        "metric": "reciprocal (dim_nd 256, dim_np 16) range-proof batch verifies/sec", "value": total * args.steps / elapsed, "unit": "verifies/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"batch verify ONE fixed batch of {total} ReciprocalRangeProofProtocol proofs of BASELINE configs[4]'s shape (dim_nd "
                               "256, dim_np 16: 769 generators, 8 WNLA rounds, 21 points + 3 scalars per proof), "
                               f"{'all resident on one GPU' if world == 1 else f'sharded contiguously over {world} GPUs, {n} proofs per GPU'}, through "
                               "the generic kernels, inputs resident in HBM, 1/256 proofs corrupted, one 4-byte reject-count all-reduce per step; "
                               "proofs made by the product prover (oracle-checked sample)",
                   "total_proofs_per_step": total, "proofs_per_gpu": n, "fb_window_bits": proto.get_option("fb_window_bits"),
                   "parallelism": f"shard{world}" if world > 1 else "single", "kernel_timing_during_value": False},
        "timing_pass_ms_per_step": elapsed_timed / args.steps * 1e3,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / bench.HBM_PEAK_GBS,
                     "traffic": bench.pmc_traffic(dom, n), "avg_launch_ms": avg_ms, "launches_per_step": dom_t["launches"] / args.steps,
                     "algorithmic_bytes_per_launch": bench.ALGO_BYTES_PER_RECIP256 * n},
        "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in kt.items()},
        "accept_bits_ok": ok, "reject_count_all_reduced": rejects,
        "setup_s": {"context_tables": t_ctx, "inputs_and_gpu_batch_prove": t_setup, "gpu_batch_prove_incl_pcie": t_prove},
        "prover": {"proofs_per_s_incl_pcie": n / t_prove},
        "device_bytes": proto.device_bytes(),
    }
    if rlc:
        # secondary, never `value`: the optional random-linear-combination mode of the final MSM
        seed = os.urandom(32)
        dA2 = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dS2 = torch.zeros(n, dtype=torch.int32, device="cuda")
        dR2 = torch.zeros(1, dtype=torch.int32, device="cuda")
        step(seed, dA2, dS2, dR2)
        t_rlc, _, kt2 = two_passes(proto, lambda: step(seed, dA2, dS2, dR2), fence, args.steps)
        t_rlc = max_over_ranks(t_rlc) / args.steps
        kt2 = {k: v["total_ms"] / args.steps for k, v in kt2.items()}
        result["rlc_mode"] = {"value": total / t_rlc, "unit": "verifies/s", "ms_per_step": t_rlc * 1e3, "kernels_ms_per_step": kt2,
                              "accept_bits_equal_exact_mode": bool((dA2.cpu().numpy() == acc).all() and (dS2.cpu().numpy() == st).all())
                                                              and int(dR2.item()) == rejects,
                              "note": "optional mode (bppp_reciprocal_verify_batch_rlc_device): the final 769-base MSM on secretly weighted sums of "
                                      "instances, what does not pass re-checked exactly (1/256 corrupted here); NOT the headline metric"}
    if n_or:
        # the checker: the reference-shaped C prover and verifier on the first n_or instances (same generators, same inputs)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import recip_cases
        com_h, proofs_h = head["com"], dP[:n_or].cpu().numpy()
        ocase = recip_cases.make_bulk(nd, npp, n_or, n_oracle=n_or, label=RECIP_LABEL, generators=gens5,
                                      inputs={k: np.ascontiguousarray(head[k][:n_or]) for k in ("x", "s", "digits", "m", "rnd")})
        t0 = time.perf_counter()
        agree = True
        for i in range(n_or):
            rc = recip_cases.oracle_verify(ocase, bytes(com_h[i]), bytes(proofs_h[i]))
            agree &= (rc == 1) == bool(acc[i])
        dt = time.perf_counter() - t0
        result["cpu_baseline"] = {"value": n_or / dt, "unit": "verifies/s", "cores": 1, "kind": "port",
                                  "sample": f"first {n_or} proofs of the same batch, reference-shaped C verifier (oracle/bppp_ref.c), one thread, {dt:.2f} s",
                                  "agrees_with_gpu": bool(agree),
                                  "prover_bytes_equal_oracle": bool((proofs_h[1:n_or] == ocase["proofs"][1:n_or]).all())}
    proto.close()
    return result, ok


def run_recip256(args):
    """bench.py --workload recip256 [--gpus N]: BASELINE configs[4] -- ONE fixed batch of 2^18 instances split over the N GPUs."""
    import torch.distributed as dist
    import bench
    world, rank, local_rank = bench.setup_dist(args)
    dist_on = dist.is_initialized()
    result, ok = measure_recip256(args, args.total_proofs, args.fb_window_bits, cpu_baseline=not args.no_cpu_baseline,
                                  rlc=not args.no_secondary, dist_on=dist_on, world=world, rank=rank, local_rank=local_rank)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist_on:
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)
