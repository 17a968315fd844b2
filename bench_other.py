"""bench.py --workload prove | recip256: the two BASELINE.json configurations beside the headline verify metric, each with its own
JSON line carrying `roofline` (dominant kernel, live HIP-event time, algorithmic bytes of SURVEY.md 8d) and `cpu_baseline`
(the oracle on a bounded sample).  One GPU; inputs resident in HBM when the timed region starts."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def _dominant(kernel_times):
    name, t = max(kernel_times.items(), key=lambda kv: kv[1]["total_ms"])
    return name, t, t["total_ms"] / max(1, t["launches"])


def run_prove(args):
    """BASELINE configs[3]: batch-prove 2^14 u64 values on one MI355X (u64_proof.rs:57-82 -> circuit.rs:260-556 -> wnla.rs:125-190)."""
    import numpy as np
    import torch
    import bench
    world, rank, local_rank = bench.setup_dist(args)
    assert world == 1, "the prove workload is a single-GPU configuration (BASELINE configs[3])"
    from bp_pp_amd import U64RangeProofProtocol, synth as workload
    gens, g, gv, hv = bench.load_generators()
    n = args.total_proofs
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=args.fb_window_bits)
    x_h, s_h, r_h = workload.bulk_values(n), workload.bulk_blindings(n), workload.bulk_prover_randomness(n)
    dx = torch.from_numpy(x_h.view(np.int64)).cuda()
    ds, dr = torch.from_numpy(s_h).cuda(), torch.from_numpy(r_h).cuda()
    dP = torch.zeros((n, 928), dtype=torch.uint8, device="cuda")
    dV = torch.zeros((n, 64), dtype=torch.uint8, device="cuda")
    dSt = torch.zeros(n, dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    proto.set_stream(stream.cuda_stream)

    def step():
        with torch.cuda.stream(stream):
            proto.prove_batch_device(workload.LABEL, n, dx.data_ptr(), ds.data_ptr(), dr.data_ptr(), dP.data_ptr(), dV.data_ptr(), dSt.data_ptr())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    proto.enable_timing(True)
    proto.timings(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kt = {k: v for k, v in proto.timings(reset=True).items() if v["launches"]}
    proto.enable_timing(False)
    # what was timed is correct: no status flag, and the product verifier accepts every proof
    P, V = dP.cpu().numpy(), dV.cpu().numpy()
    acc, vst = proto.verify_batch(V, P, workload.LABEL)
    ok = bool(acc.all()) and not vst.any() and not bool(dSt.any().item())
    dom, dom_t, avg_ms = _dominant(kt)
    launches_per_step = dom_t["launches"] / args.steps
    achieved = bench.ALGO_BYTES_PER_PROVE * n / (avg_ms * 1e-3) / 1e9
    result = {
        "metric": "u64 range proofs proved/sec (batch)", "value": n * args.steps / elapsed, "unit": "proves/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"batch prove {n} u64 values on one GPU (BASELINE configs[3]); x, s and the 52 prover scalars per proof resident "
                               "in HBM, device-side transcripts, proofs byte-identical to the CPU prover's for the same draws",
                   "proofs_per_step": n, "fb_window_bits": args.fb_window_bits or "library default", "label": workload.LABEL.decode()},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / bench.HBM_PEAK_GBS,
                     "traffic": bench.pmc_traffic(dom, n), "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step,
                     "algorithmic_bytes_per_launch": bench.ALGO_BYTES_PER_PROVE * n,
                     "note": "2262 B/prove (SURVEY.md 8d) per launch of the dominant kernel; that kernel runs launches_per_step times per proof "
                             "batch (15 fixed-base MSMs per proof), so the per-step figure is value x 2262 B"},
        "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in kt.items()},
        "proofs_verify": ok,
    }
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import bppp_oracle_c as OC                                   # the oracle, as the timed CPU baseline ONLY
        m = min(4096, n)                                             # ~10 s of host work on 64 threads
        hw = os.cpu_count() or 1
        th = min(hw, 64)
        t0 = time.perf_counter()
        Pref, Vref = OC.u64_prove_batch(gens, workload.LABEL, x_h[:m], s_h[:m], r_h[:m], nthreads=th)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        OC.u64_prove_batch(gens, workload.LABEL, x_h[:8], s_h[:8], r_h[:8], nthreads=1)
        single = 8 / (time.perf_counter() - t1)
        result["cpu_baseline"] = {"value": m / dt, "unit": "proves/s", "cores": th, "kind": "port",
                                  "sample": f"first {m} values of the same batch, reference-shaped C prover (oracle/bppp_ref.c), {th} threads, "
                                            f"{dt:.2f} s wall; box reports {hw} hardware threads",
                                  "single_thread_value": single, "byte_identical_to_gpu": bool((Pref == P[:m]).all() and (Vref == V[:m]).all())}
    print(json.dumps(result), flush=True)
    proto.close()
    if not ok:
        sys.exit(1)


def run_recip256(args):
    """BASELINE configs[4]'s shape: ReciprocalRangeProofProtocol { dim_nd: 256, dim_np: 16 } (reciprocal.rs:98-107): |g_vec| 256,
    |h_vec| 266 + 246 padding, 8 WNLA rounds, proof = 21 points + 3 scalars.  One committed value with 256 hex digits -- the closest
    thing the reference's API can express to "aggregated 16 values" (SURVEY.md 8d, config 5)."""
    import numpy as np
    import torch
    import bench
    world, rank, local_rank = bench.setup_dist(args)
    assert world == 1
    from bp_pp_amd import derive_generators, synth
    from bp_pp_amd.wnla import ReciprocalRangeProofProtocol
    nd, npp, n = 256, 16, args.total_proofs
    n_or = 0 if args.no_cpu_baseline else 48                      # ~10 s of single-thread oracle work (prove + verify)
    t0 = time.time()
    # generators from the product's own derivation (SHAKE256 try-and-increment, host code), inputs from bp_pp_amd/synth.py: the oracle
    # is only touched by the cpu_baseline leg below
    nh, NG, NH = nd + 10, 256, 512
    raw = derive_generators(b"bppp-bench-recip256", 1 + NG + NH)
    pts = [raw[64 * i:64 * i + 64] for i in range(1 + NG + NH)]
    gens5 = (pts[0], pts[1:1 + nd], pts[1 + NG:1 + NG + nh], pts[1 + nd:1 + NG], pts[1 + NG + nh:])
    case = dict(synth.bulk_reciprocal_inputs(nd, n), g=gens5[0], gv=gens5[1], hv=gens5[2], gv_=gens5[3], hv_=gens5[4], label=b"reciprocal bench")
    t_inputs = time.time() - t0
    W = args.fb_window_bits or 16
    t0 = time.time()
    proto = ReciprocalRangeProofProtocol(nd, npp, case["g"], case["gv"], case["hv"], case["gv_"], case["hv_"], device=0, fb_window_bits=W)
    proto.synchronize()
    t_ctx = time.time() - t0
    com, cst = proto.commit_value_batch(case["x"], case["s"])
    t0 = time.time()
    proofs, pst, shape = proto.prove_batch(case["label"], com, case["x"], case["s"], case["digits"], case["m"], case["rnd"])
    t_prove = time.time() - t0
    assert not cst.any() and not pst.any() and shape == (8, 2, 1)
    bad = np.arange(0, n, 256)
    proofs[bad, -1] ^= 1                                               # one instance in 256 must be rejected
    expect = np.ones(n, np.uint8); expect[bad] = 0
    dV, dP = torch.from_numpy(com).cuda(), torch.from_numpy(proofs).cuda()
    dA = torch.zeros(n, dtype=torch.uint8, device="cuda")
    dS = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    def step():
        proto.verify_batch_device(case["label"], n, dV.data_ptr(), dP.data_ptr(), *shape, dA.data_ptr(), dS.data_ptr())

    for _ in range(args.warmup):
        step()
    proto.synchronize()
    proto.enable_timing(True)
    proto.timings(reset=True)
    proto.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    proto.synchronize()
    elapsed = time.perf_counter() - t0
    kt = {k: v for k, v in proto.timings(reset=True).items() if v["launches"]}
    proto.enable_timing(False)
    acc, st = dA.cpu().numpy(), dS.cpu().numpy()
    ok = bool((acc == expect).all()) and not st.any()
    dom, dom_t, avg_ms = _dominant(kt)
    achieved = bench.ALGO_BYTES_PER_RECIP256 * n / (avg_ms * 1e-3) / 1e9
    result = {
        "metric": "reciprocal (dim_nd 256, dim_np 16) range-proof batch verifies/sec", "value": n * args.steps / elapsed, "unit": "verifies/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"batch verify {n} ReciprocalRangeProofProtocol proofs of BASELINE configs[4]'s shape (dim_nd 256, dim_np 16: 769 "
                               "generators, 8 WNLA rounds, 21 points + 3 scalars per proof) on one GPU through the generic kernels, inputs resident "
                               "in HBM, 1/256 proofs corrupted; proofs made by the product prover (oracle-checked sample)",
                   "proofs_per_step": n, "fb_window_bits": W},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / bench.HBM_PEAK_GBS,
                     "traffic": bench.pmc_traffic(dom, n), "avg_launch_ms": avg_ms, "launches_per_step": dom_t["launches"] / args.steps,
                     "algorithmic_bytes_per_launch": bench.ALGO_BYTES_PER_RECIP256 * n},
        "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in kt.items()},
        "accept_bits_ok": ok,
        "setup_s": {"inputs_host": t_inputs, "context_tables": t_ctx, "gpu_batch_prove_incl_pcie": t_prove},
        "prover": {"proofs_per_s_incl_pcie": n / t_prove},
        "device_bytes": proto.device_bytes(),
    }
    # secondary, never `value`: the optional random-linear-combination mode of the final MSM (one 769-base MSM per chunk of 8)
    seed = os.urandom(32)
    dA2 = torch.zeros(n, dtype=torch.uint8, device="cuda")
    dS2 = torch.zeros(n, dtype=torch.int32, device="cuda")

    def rlc_step():
        proto.verify_batch_rlc_device(case["label"], n, dV.data_ptr(), dP.data_ptr(), *shape, dA2.data_ptr(), dS2.data_ptr(), seed)

    rlc_step()
    proto.synchronize()
    proto.enable_timing(True)
    proto.timings(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rlc_step()
    proto.synchronize()
    t_rlc = (time.perf_counter() - t0) / args.steps
    kt2 = {k: v["total_ms"] / args.steps for k, v in proto.timings(reset=True).items() if v["launches"]}
    proto.enable_timing(False)
    result["rlc_mode"] = {"value": n / t_rlc, "unit": "verifies/s", "ms_per_step": t_rlc * 1e3, "kernels_ms_per_step": kt2,
                          "accept_bits_equal_exact_mode": bool((dA2.cpu().numpy() == acc).all() and (dS2.cpu().numpy() == st).all()),
                          "note": "optional mode (bppp_reciprocal_verify_batch_rlc_device): the final 769-base MSM once per chunk of 8 instances, "
                                  "chunks that fail re-checked exactly (1/256 corrupted here = 3 % of the chunks); NOT the headline metric"}
    if n_or:
        # the checker: the reference-shaped C prover and verifier on the first n_or instances (same generators, same inputs)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import recip_cases
        ocase = recip_cases.make_bulk(nd, npp, n_or, n_oracle=n_or, label=case["label"], generators=gens5,
                                      inputs={k: np.ascontiguousarray(case[k][:n_or]) for k in ("x", "s", "digits", "m", "rnd")})
        t0 = time.perf_counter()
        agree = True
        for i in range(n_or):
            rc = recip_cases.oracle_verify(ocase, bytes(com[i]), bytes(proofs[i]))
            agree &= (rc == 1) == bool(acc[i])
        dt = time.perf_counter() - t0
        result["cpu_baseline"] = {"value": n_or / dt, "unit": "verifies/s", "cores": 1, "kind": "port",
                                  "sample": f"first {n_or} proofs of the same batch, reference-shaped C verifier (oracle/bppp_ref.c), one thread, {dt:.2f} s",
                                  "agrees_with_gpu": bool(agree),
                                  "prover_bytes_equal_oracle": bool((proofs[1:n_or] == ocase["proofs"][1:n_or]).all())}
    print(json.dumps(result), flush=True)
    proto.close()
    if not ok:
        sys.exit(1)
