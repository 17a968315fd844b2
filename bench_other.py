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


def two_passes(proto, step, fence, steps, settle_s=0.25):
    """(seconds for `steps` steps with kernel timing OFF -- the plan a caller gets, the source of `value` --, seconds for the same steps with
    per-kernel HIP events on, the per-kernel times of that second pass).  See bench.run_verify.
    settle_s: these workloads' steps are a few milliseconds, and `--warmup 1` leaves the GPU short of its working clocks: the first pass then
    measures 1.5 % slower than the second (round 6, profiles/r06/r06_w9_warmup_sensitivity.txt).  So, single process only (a time-based
    loop would run a different number of barriers on every rank), further UNTIMED steps follow the caller's warm-up until settle_s
    seconds have passed; 0 = none.  (bench.py's headline workload, whose step is 140 ms, keeps exactly `--warmup` steps.)"""
    fence()
    t0 = time.perf_counter()
    while settle_s > 0 and time.perf_counter() - t0 < settle_s:
        step()
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

    elapsed, elapsed_timed, kt = two_passes(proto, step, fence, args.steps, 0.0 if dist_on else 0.25)
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
    elapsed, elapsed_timed, kt = two_passes(proto, step, fence, args.steps, 0.0 if dist_on else 0.25)
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
        t_rlc, _, kt2 = two_passes(proto, lambda: step(seed, dA2, dS2, dR2), fence, args.steps, 0.0)
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


# ---------------------------------------------------------------- the crate's generic `wnla` and `circuit` API
GENERIC_SLICE = 1 << 14


def _generic_line(metric, unit, total, args, elapsed, elapsed_timed, kt, algo_bytes, workload_text, extra_cfg, fb_bits):
    import bench
    dom, dom_t, avg_ms = _dominant(kt)
    achieved = algo_bytes * total / (avg_ms * 1e-3) / 1e9
    return {
        "metric": metric, "value": total * args.steps / elapsed, "unit": unit, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "timing_pass_ms_per_step": elapsed_timed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": dict({"workload": workload_text, "total_proofs_per_step": total, "proofs_per_gpu": total, "fb_window_bits": fb_bits, "parallelism": "single",
                        "kernel_timing_during_value": False}, **extra_cfg),
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / bench.HBM_PEAK_GBS,
                     "traffic": bench.pmc_traffic(dom, total), "avg_launch_ms": avg_ms, "launches_per_step": dom_t["launches"] / args.steps,
                     "algorithmic_bytes_per_launch": algo_bytes * total, "algorithmic_bytes_per_unit": algo_bytes,
                     "note": "VALU-issue bound like the u64 path (256-bit modular integer code); the HBM fraction is small by construction"},
        "roofline_valu": bench.valu_roofline(dom, avg_ms, total, 8 if dom in bench.FB_KERNELS else 1),
        "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in kt.items()},
    }


def measure_wnla(args, total, ng=16, nh=32, cpu_baseline=True, cpu_sample=256):
    """`WeightNormLinearArgument::verify` (wnla.rs:75-121) for ONE fixed batch of `total` instances over one generator set -- N = 16 / 32 is
    the u64 protocol's WNLA stage as a stand-alone argument (4 rounds, proof = 8 points + 3 scalars); the reference's own `wnla_works`
    (tests.rs:139-171, N = 4) is the smoke size.  Instances (c, rho, mu = rho^2, l, n) are seeded draws; commitments and proofs come from
    the product's own generic prover; everything resident in HBM when the timed region starts (bppp_wnla_verify_batch_device)."""
    import numpy as np
    import torch
    import bench
    from bp_pp_amd import derive_generators, synth
    from bp_pp_amd.wnla import WeightNormLinearArgument
    label = b"wnla test"
    raw = derive_generators(b"bppp-bench-wnla", 1 + ng + nh)
    pts = [raw[64 * i:64 * i + 64] for i in range(1 + ng + nh)]
    t0 = time.time()
    w = WeightNormLinearArgument(pts[0], pts[1:1 + ng], pts[1 + ng:], device=0, fb_window_bits=args.fb_window_bits)
    w.synchronize()
    t_ctx = time.time() - t0
    bufs = {k: [] for k in ("com", "c", "rho", "mu", "pr", "px", "pl", "pn")}
    t_prove = 0.0
    head = None
    for a in range(0, total, GENERIC_SLICE):
        m = min(GENERIC_SLICE, total - a)
        sc = synth._bulk_scalars(b"wnla", a, m, nh + 1 + nh + ng, b"bppp-bench-wnla")
        sc = sc.reshape(m, -1, 32)
        c, rho, l, n = sc[:, :nh], sc[:, nh], sc[:, nh + 1:2 * nh + 1], sc[:, 2 * nh + 1:]
        rho_i = [int.from_bytes(bytes(r), "big") for r in rho]
        mu = np.frombuffer(b"".join((r * r % synth.N_ORDER).to_bytes(32, "big") for r in rho_i), np.uint8).reshape(m, 32)
        com, cst = w.commit_batch(c, mu, l, n)
        t0 = time.perf_counter()
        pr, px, pl, pn, pst = w.prove_batch(label, com, c, rho, mu, l, n)
        t_prove += time.perf_counter() - t0
        assert not cst.any() and not pst.any()
        if head is None:
            head = dict(c=c[:cpu_sample].copy(), rho=rho[:cpu_sample].copy(), mu=mu[:cpu_sample].copy())
        for k, v in (("com", com), ("c", c), ("rho", rho), ("mu", mu), ("pr", pr), ("px", px), ("pl", pl), ("pn", pn)):
            bufs[k].append(np.ascontiguousarray(v))
    H = {k: np.concatenate(v) for k, v in bufs.items()}
    rounds, nl, nn = H["pr"].shape[1], H["pl"].shape[1], H["pn"].shape[1]
    expect = np.ones(total, np.uint8)
    bad = np.arange(0, total, 256)
    H["pn"][bad, 0, 31] ^= 1                      # one instance in 256: the last bit of n[0] flipped, must be rejected
    expect[bad] = 0
    D = {k: torch.from_numpy(v).cuda() for k, v in H.items()}
    dA = torch.zeros(total, dtype=torch.uint8, device="cuda"); dS = torch.zeros(total, dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    _capi_set_stream(w, stream)
    torch.cuda.synchronize()

    def step():
        w.verify_batch_device(label, total, D["com"].data_ptr(), D["c"].data_ptr(), D["rho"].data_ptr(), D["mu"].data_ptr(), rounds, D["pr"].data_ptr(),
                              D["px"].data_ptr(), D["pl"].data_ptr(), nl, D["pn"].data_ptr(), nn, dA.data_ptr(), dS.data_ptr())

    def fence():
        w.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    elapsed, elapsed_timed, kt = two_passes(w, step, fence, args.steps)
    acc, st = dA.cpu().numpy(), dS.cpu().numpy()
    ok = bool((acc == expect).all()) and not st.any()
    algo = 2 * rounds * 33 + 32 * (nl + nn) + 33 + 32 * nh + 64 + 1
    r = _generic_line("wnla verifies/sec (batch)", "verifies/s", total, args, elapsed, elapsed_timed, kt, algo,
                      f"batch verify ONE fixed batch of {total} WeightNormLinearArgument instances (wnla.rs:75-121) with |g_vec| = {ng}, |h_vec| = {nh} "
                      f"({rounds} rounds; proof = {2 * rounds} points + {nl + nn} scalars) through the generic kernels, inputs resident in HBM, 1/256 corrupted; "
                      "proofs made by the product prover",
                      {"ng": ng, "nh": nh, "rounds": rounds, "label": label.decode(), "algorithmic_bytes": f"{algo} = proof in SEC1 form + commitment + c ({nh} scalars) + rho, mu + accept byte"},
                      w.get_option("fb_window_bits") if hasattr(w, "get_option") else args.fb_window_bits)
    r["accept_bits_ok"] = ok
    r["setup_s"] = {"context_tables": t_ctx, "gpu_batch_prove_incl_pcie": t_prove}
    if cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
        import ctypes as C
        import bppp_oracle_c as OC                                   # the oracle, as the timed CPU baseline ONLY
        L, sz = OC.lib(), C.c_size_t
        gv, hv = b"".join(pts[1:1 + ng]), b"".join(pts[1 + ng:])
        m = min(cpu_sample, total)
        t0 = time.perf_counter()
        agree = True
        for i in range(m):
            v = L.bppp_oracle_wnla_verify(pts[0], gv, sz(ng), hv, sz(nh), bytes(H["c"][i].reshape(-1)), sz(nh), bytes(H["rho"][i]), bytes(H["mu"][i]), label,
                                          sz(len(label)), bytes(H["com"][i]), bytes(H["pr"][i].reshape(-1)), bytes(H["px"][i].reshape(-1)), sz(rounds),
                                          bytes(H["pl"][i].reshape(-1)), sz(nl), bytes(H["pn"][i].reshape(-1)), sz(nn))
            agree &= (v == 1) == bool(acc[i])
        dt = time.perf_counter() - t0
        r["cpu_baseline"] = {"value": m / dt, "unit": "verifies/s", "cores": 1, "kind": "port",
                             "sample": f"first {m} instances of the same batch, reference-shaped C verifier (oracle/bppp_ref.c: wnla.rs:75-121 as written), one thread, {dt:.2f} s",
                             "agrees_with_gpu": bool(agree)}
    w.close()
    return r, ok


def _capi_set_stream(w, stream):
    from bp_pp_amd import _capi
    _capi.check(_capi.lib().bppp_ctx_set_stream(w._ctx, stream.cuda_stream))


def measure_circuit(args, total, name="mixed_k2", cpu_baseline=True, cpu_sample=128):
    """`ArithmeticCircuit::verify` (circuit.rs:154-256) for ONE fixed batch of `total` instances of ONE circuit: `mixed_k2` (k = 2 committed
    vectors, w_o spread over all four partition types, dense W_m / W_l: the general path) or the reference's own `ac_works` statement
    (tests.rs:45-136).  The statement is data (tests/golden/statements_generic.json); generators are derived, blindings and prover draws
    seeded; commitments and proofs come from the product's own generic prover; everything resident in HBM when the timed region starts
    (bppp_circuit_verify_batch_device)."""
    import numpy as np
    import torch
    import bench
    from bp_pp_amd import derive_generators, synth
    from bp_pp_amd.wnla import ArithmeticCircuit
    with open(os.path.join(ROOT, "tests", "golden", "statements_generic.json")) as f:
        st = {c["name"]: c for c in json.load(f)["circuits"]}[name]
    nm, no, nv, k = st["dim_nm"], st["dim_no"], st["dim_nv"], st["k"]
    p2 = lambda x: 1 << max(0, (x - 1).bit_length())
    NG, NH = p2(nm), p2(nv + 9)
    raw = derive_generators(b"bppp-bench-circuit-" + name.encode(), 1 + NG + NH)
    pts = [raw[64 * i:64 * i + 64] for i in range(1 + NG + NH)]
    flat = lambda rows: np.frombuffer(b"".join(bytes.fromhex(x) for row in rows for x in row), np.uint8).reshape(-1, 32)
    vec = lambda xs: np.frombuffer(b"".join(bytes.fromhex(x) for x in xs), np.uint8).reshape(-1, 32)
    part = st["partition"]
    label = bytes.fromhex(st["label"])
    t0 = time.time()
    ac = ArithmeticCircuit(nm, no, k, nv, pts[0], pts[1:1 + nm], pts[1 + NG:1 + NG + nv + 9], flat(st["W_m"]), flat(st["W_l"]), vec(st["a_m"]), vec(st["a_l"]),
                           st["f_l"], st["f_m"], pts[1 + nm:1 + NG], pts[1 + NG + nv + 9:], lambda typ, j: (None if part[typ][j] < 0 else part[typ][j]),
                           device=0, fb_window_bits=args.fb_window_bits)
    ac.synchronize()
    t_ctx = time.time() - t0
    v_one = np.stack([vec(row) for row in st["v"]])                      # [k, nv, 32]: one witness, fresh blindings and draws per instance
    used = 18 + nv + nm
    coms, proofs = [], []
    t_prove = 0.0
    shape = None
    for a in range(0, total, GENERIC_SLICE):
        m = min(GENERIC_SLICE, total - a)
        sc = synth._bulk_scalars(b"circ", a, m, k + used, b"bppp-bench-circuit").reshape(m, -1, 32)
        s_v, rnd = sc[:, :k], sc[:, k:]
        v = np.broadcast_to(v_one, (m, k, nv, 32)).copy()
        com = np.stack([ac.commit_batch(v[:, j], s_v[:, j])[0] for j in range(k)], axis=1)      # [m, k, 64]
        rep = lambda xs: np.broadcast_to(vec(xs), (m,) + vec(xs).shape).copy()
        t0 = time.perf_counter()
        pr, pst, shape = ac.prove_batch(label, com, v, s_v, rep(st["w_l"]), rep(st["w_r"]), rep(st["w_o"]), rnd)
        t_prove += time.perf_counter() - t0
        assert not pst.any()
        coms.append(com); proofs.append(pr)
    Hc, Hp = np.concatenate(coms), np.concatenate(proofs)
    rounds, nl, nn = shape
    expect = np.ones(total, np.uint8)
    bad = np.arange(0, total, 256)
    Hp[bad, -1] ^= 1                               # one instance in 256: the last bit of the final scalar flipped, must be rejected
    expect[bad] = 0
    dC, dP = torch.from_numpy(Hc).cuda(), torch.from_numpy(Hp).cuda()
    dA = torch.zeros(total, dtype=torch.uint8, device="cuda"); dS = torch.zeros(total, dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    _capi_set_stream(ac._w, stream)
    torch.cuda.synchronize()

    def step():
        ac.verify_batch_device(label, total, dC.data_ptr(), dP.data_ptr(), rounds, nl, nn, dA.data_ptr(), dS.data_ptr())

    def fence():
        ac.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    elapsed, elapsed_timed, kt = two_passes(ac, step, fence, args.steps)
    acc, stt = dA.cpu().numpy(), dS.cpu().numpy()
    ok = bool((acc == expect).all()) and not stt.any()
    algo = (4 + 2 * rounds) * 33 + 32 * (nl + nn) + 33 * k + 1
    r = _generic_line("arithmetic-circuit verifies/sec (batch)", "verifies/s", total, args, elapsed, elapsed_timed, kt, algo,
                      f"batch verify ONE fixed batch of {total} ArithmeticCircuit instances (circuit.rs:154-256) of the statement `{name}` (dim_nm {nm}, dim_no {no}, "
                      f"dim_nv {nv}, k {k}, f_l {st['f_l']}, f_m {st['f_m']}; {1 + NG + NH} generators, {rounds} WNLA rounds) through the generic kernels, inputs "
                      "resident in HBM, 1/256 corrupted; proofs made by the product prover",
                      {"statement": name, "rounds": rounds, "label": label.decode(), "algorithmic_bytes": f"{algo} = proof in SEC1 form + {k} commitments + accept byte"},
                      ac._w.get_option("fb_window_bits") if hasattr(ac._w, "get_option") else args.fb_window_bits)
    r["accept_bits_ok"] = ok
    r["setup_s"] = {"context_tables": t_ctx, "gpu_batch_prove_incl_pcie": t_prove}
    if cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
        import ctypes as C
        import bppp_oracle_c as OC                                   # the oracle, as the timed CPU baseline ONLY
        import ref_fixture_check as RC
        L, sz = OC.lib(), C.c_size_t
        cdoc = {"g": pts[0].hex(), "g_vec": b"".join(pts[1:1 + nm]).hex(), "h_vec": b"".join(pts[1 + NG:1 + NG + nv + 9]).hex(),
                "g_vec_": b"".join(pts[1 + nm:1 + NG]).hex(), "h_vec_": b"".join(pts[1 + NG + nv + 9:]).hex()}
        head, keep = RC._circuit_call_args(st, cdoc)
        m = min(cpu_sample, total)
        t0 = time.perf_counter()
        agree = True
        for i in range(m):
            v = L.bppp_oracle_circuit_verify(*head, label, sz(len(label)), bytes(Hc[i].reshape(-1)), bytes(Hp[i]), sz(rounds), sz(nl), sz(nn))
            agree &= (v == 1) == bool(acc[i])
        dt = time.perf_counter() - t0
        r["cpu_baseline"] = {"value": m / dt, "unit": "verifies/s", "cores": 1, "kind": "port",
                             "sample": f"first {m} instances of the same batch, reference-shaped C verifier (oracle/bppp_ref.c: circuit.rs:154-256 with dense matrices), one thread, {dt:.2f} s",
                             "agrees_with_gpu": bool(agree)}
    ac.close()
    return r, ok


def run_generic(args):
    """bench.py --workload wnla | circuit: their own JSON lines (one GPU; these paths have no sharded form of their own)."""
    import torch
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the bp_pp_amd product path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    if args.workload == "wnla":
        r, ok = measure_wnla(args, args.total_proofs, cpu_baseline=not args.no_cpu_baseline)
    else:
        r, ok = measure_circuit(args, args.total_proofs, name=args.statement, cpu_baseline=not args.no_cpu_baseline)
    print(json.dumps(r), flush=True)
    if not ok:
        sys.exit(1)
