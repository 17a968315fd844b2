"""GPU parity tests at the sizes BASELINE.json quotes the metric on (run with -m gpu on an MI355X):

  * configs[2]: u64 verify at 2^17 proofs (one GPU's shard of the 8-GPU split) and at 2^20 proofs (the whole batch resident on one
    GPU, the N = 1 workload of bench.py) -- size-independent properties over the full batch (every honest proof accepted, every
    corrupted one rejected, reject count, invariance under a permutation of the batch) plus a >= 512-proof sample checked bit
    for bit against the CPU oracle; the proofs come from the product's batch prover and a sample of THEM is compared byte for
    byte with the oracle prover's output for the same inputs;
  * configs[4]: ReciprocalRangeProofProtocol (dim_nd 256, dim_np 16: |g_vec| 256, |h_vec| 512, 8 WNLA rounds) prove + verify at
    B = 2^12 instances, with an oracle-proved / oracle-verified sample.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("these tests need a GPU (they are selected with -m gpu only on the GPU box)")
    return torch


@pytest.mark.parametrize("log_n,wbits", [(17, 20), (20, 0)])
def test_u64_verify_at_baseline_sizes(torch_mod, oracle_c, log_n, wbits):
    torch = torch_mod
    import bench
    import workload                        # oracle-side helpers (trapdoor prover, generators)
    from bp_pp_amd import U64RangeProofProtocol, synth
    n = 1 << log_n
    gens, g, gv, hv = bench.load_generators()
    assert gens == workload.generators()
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=wbits)     # 0 = the library default bench.py runs with
    try:
        # commit_value (u64_proof.rs:37-39) through these tables: x walks only the windows a u64 can reach (3 of 12 / 4 of 13 signed)
        xs = np.array([0, 1, 2**64 - 1, 2**63, 2**43, 2**44 - 1, 123456], dtype=np.uint64)
        ss = np.frombuffer(b"".join(int(v).to_bytes(32, "big") for v in [0, 1, 5, 2**200, 7, 2**255, 3]), dtype=np.uint8).reshape(7, 32)
        cv = proto.commit_value_batch(xs, ss)
        for i in range(7):
            assert bytes(cv[i]) == oracle_c.u64_commit_value(gens, int(xs[i]), bytes(ss[i])), i
        lo = 3 << 20                        # a shard that does not start at proof 0 of the synthetic stream
        dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, lo, lo + n)
        assert int((expect == 0).sum()) == n // 1024
        dA = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dS = torch.zeros(n, dtype=torch.int32, device="cuda")
        dR = torch.zeros(1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        proto.verify_batch_device(synth.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, dR.data_ptr())
        proto.synchronize()
        acc, st = dA.cpu().numpy(), dS.cpu().numpy()
        # properties over the whole batch
        assert (acc == expect).all() and not st.any()
        assert int(dR.item()) == n // 1024
        # the batch order carries no state: a permuted copy gives the permuted accept bits
        perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
        dV2, dP2 = dV[perm].contiguous(), dP[perm].contiguous()
        dA2 = torch.zeros(n, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        proto.verify_batch_device(synth.LABEL, n, dV2.data_ptr(), dP2.data_ptr(), dA2.data_ptr(), dS.data_ptr(), 0, dR.data_ptr())
        proto.synchronize()
        assert bool((dA2 == dA[perm]).all().item()) and int(dR.item()) == n // 1024
        del dV2, dP2
        # oracle sample: 512 spread proofs + 64 of the corrupted ones + both ends, verified by the CPU oracle
        idx = np.unique(np.concatenate([np.arange(0, n, n // 512), np.nonzero(expect == 0)[0][:64], [1, n - 1]])).astype(np.int64)
        assert len(idx) >= 512
        ti = torch.from_numpy(idx).cuda()
        Vs, Ps = dV[ti].cpu().numpy(), dP[ti].cpu().numpy()
        oacc, ost = oracle_c.u64_verify_batch(gens, synth.LABEL, Vs, Ps, nthreads=os.cpu_count() or 1)
        assert (oacc == acc[idx]).all() and not ost.any()
        # the prover that made the batch, cross-checked: the oracle prover on the same (x, s, rnd) gives the same bytes
        p0 = n // 2 + 100                       # 96 consecutive proofs between two corrupted ones (global index = 0 mod 1024)
        pidx = np.arange(p0, p0 + 96, dtype=np.int64)
        assert expect[pidx].all()
        x, s, rnd = synth.bulk_values(96, first=lo + p0), synth.bulk_blindings(96, first=lo + p0), synth.bulk_prover_randomness(96, first=lo + p0)
        Pref, Vref = oracle_c.u64_prove_trapdoor_batch(workload.generator_dlogs(), synth.LABEL, x, s, rnd, nthreads=os.cpu_count() or 1)
        tp = torch.from_numpy(pidx).cuda()
        assert (Pref == dP[tp].cpu().numpy()).all() and (Vref == dV[tp].cpu().numpy()).all()
    finally:
        proto.close()


def test_u64_sharded_forms_at_one_gpus_share(torch_mod, monkeypatch):
    """One GPU's share of BASELINE configs[2] (2^17 proofs) through the SHARDED entry points of a one-device group with its RCCL
    communicator (the path a node-level caller takes): exact, RLC, SEC1 and caller-transcript forms on resident shards give the
    expectation over the whole shard and the all-reduced reject count; the RLC form with every proof valid passes on its bucket stage."""
    torch = torch_mod
    import bench
    from bp_pp_amd import synth, wire
    from bp_pp_amd.distributed import U64RangeProofGroup
    from bp_pp_amd.transcript import Transcript
    n = 1 << 17
    gens, g, gv, hv = bench.load_generators()
    monkeypatch.setenv("BPPP_FORCE_RCCL", "1")
    grp = U64RangeProofGroup(g, gv, hv, [0], fb_window_bits=16)
    proto = grp.protocol(0)
    try:
        dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 1 << 19, (1 << 19) + n)
        n_bad = int((expect == 0).sum())
        assert n_bad == n // 1024
        dA = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dS = torch.zeros(n, dtype=torch.int32, device="cuda")
        dR = torch.full((1,), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        L = synth.LABEL
        for seed in (None, bytes(range(32))):
            dA.zero_(); dR.fill_(-7)
            torch.cuda.synchronize()
            grp.verify_batch_device(L, n, [dV.data_ptr()], [dP.data_ptr()], [dA.data_ptr()], [dS.data_ptr()], [dR.data_ptr()], rlc_seed=seed)
            assert (dA.cpu().numpy() == expect).all() and int(dR.item()) == n_bad and not dS.any().item()
        # caller transcripts: Transcript::new(label) shared by the shard == the label form
        t0 = torch.from_numpy(np.frombuffer(Transcript(L).state, np.uint8).copy()).cuda()
        dO = torch.zeros((n, 203), dtype=torch.uint8, device="cuda")
        dA.zero_(); dR.fill_(-7)
        torch.cuda.synchronize()
        grp.verify_batch_transcripts_device(n, [t0.data_ptr()], 1, [dV.data_ptr()], [dP.data_ptr()], [dA.data_ptr()], [dS.data_ptr()], [dR.data_ptr()],
                                            [dO.data_ptr()])
        assert (dA.cpu().numpy() == expect).all() and int(dR.item()) == n_bad
        assert bool((dO[:, 200] < 166).all().item())                       # every proof got an advanced, well-formed state back
        # SEC1 form of the first 4096 proofs (the compression runs on the host in Python)
        m = 4096
        V, P = dV[:m].cpu().numpy(), dP[:m].cpu().numpy()
        V33 = np.frombuffer(b"".join(wire.compress_point(bytes(v)) for v in V), np.uint8).reshape(m, 33)
        P525 = np.frombuffer(b"".join(wire.abi_to_sec1(bytes(p)) for p in P), np.uint8).reshape(m, 525)
        d33, d525 = torch.from_numpy(V33.copy()).cuda(), torch.from_numpy(P525.copy()).cuda()
        dA.zero_(); dR.fill_(-7)
        torch.cuda.synchronize()
        grp.verify_batch_sec1_device(L, m, [d33.data_ptr()], [d525.data_ptr()], [dA.data_ptr()], [dS.data_ptr()], [dR.data_ptr()])
        assert (dA[:m].cpu().numpy() == expect[:m]).all() and int(dR.item()) == int((expect[:m] == 0).sum())
        # all valid: flip the corrupted bytes back; the RLC form accepts everything (bucket stage, superchunks of 512 at this size)
        bad = np.nonzero(expect == 0)[0]
        ti = torch.from_numpy(bad).cuda()
        to = torch.from_numpy(np.array([synth.corrupt_offset((1 << 19) + int(j)) for j in bad], dtype=np.int64)).cuda()
        dP[ti, to] = dP[ti, to] ^ 1
        dA.zero_(); dR.fill_(-7)
        torch.cuda.synchronize()
        grp.verify_batch_device(L, n, [dV.data_ptr()], [dP.data_ptr()], [dA.data_ptr()], [dS.data_ptr()], [dR.data_ptr()], rlc_seed=bytes(32))
        assert bool(dA.all().item()) and int(dR.item()) == 0
    finally:
        proto.close()
        grp.close()


def test_recip256_prove_and_verify_at_batch_scale(torch_mod):
    """BASELINE configs[4]'s shape at B = 2^12: multi-wavefront indexing, workspace growth and the fixed-base path of the
    generic kernels at batch scale (they were only exercised at B <= 4 before)."""
    import recip_cases
    from bp_pp_amd.wnla import ReciprocalRangeProofProtocol
    nd, npp, B, n_or = 256, 16, 1 << 12, 6
    case = recip_cases.make(nd, npp, B, n_oracle=n_or)
    proto = ReciprocalRangeProofProtocol(nd, npp, case["g"], case["gv"], case["hv"], case["gv_"], case["hv_"], device=0, fb_window_bits=10)
    try:
        com, st = proto.commit_value_batch(case["x"], case["s"])
        assert not st.any() and (com[:n_or] == case["commitments"]).all()
        proofs, st, shape = proto.prove_batch(case["label"], com, case["x"], case["s"], case["digits"], case["m"], case["rnd"])
        assert not st.any() and shape == (case["rounds"], case["nl"], case["nn"]) == (8, 2, 1)
        assert (proofs[:n_or] == case["proofs"]).all()                     # byte-identical to the reference-shaped prover
        acc, st = proto.verify_batch(case["label"], com, proofs, *shape)
        assert acc.all() and not st.any()
        # negatives spread over the batch (different wavefronts), each judged by the oracle too for a sample
        P, V = proofs.copy(), com.copy()
        bad = list(range(7, B, 97))
        for k, i in enumerate(bad):
            kind = k % 4
            if kind == 0:
                P[i, -1] ^= 1                                               # n0
            elif kind == 1:
                P[i, 64 * (4 + (k % 16)) + 9] ^= 0x20                       # a round point's x
            elif kind == 2:
                V[i] = com[(i + 1) % B]                                     # someone else's commitment
            else:
                P[i, 192:256] = P[i, 0:64]                                  # c_s := c_l
        acc, st = proto.verify_batch(case["label"], V, P, *shape)
        good = np.ones(B, bool); good[bad] = False
        assert acc[good].all() and not st[good].any() and not acc[bad].any()
        for i in bad[:6] + [0, 1, B - 1]:
            rc = recip_cases.oracle_verify(case, bytes(V[i]), bytes(P[i]))
            assert int(acc[i]) == (1 if rc == 1 else 0) and (int(st[i]) != 0) == (rc < 0), (i, rc)
    finally:
        proto.close()


def test_recip256_16bit_windows_at_the_benchmarked_size(torch_mod):
    """The configuration `bench.py --workload recip256` measures, tested as it is benchmarked: the (256, 16) shape with 16-bit unsigned
    windows (769 bases x 16 windows x 65,535 entries = 52 GB of tables) at 2^15 proofs -- one GPU's share of BASELINE configs[4] --
    through a one-device bppp_wnla_group (the sharded entry points) AND the single-context entry points on the same context:
    prover bytes equal the oracle's on a 48-proof sample, accept bits equal the expectation over the whole batch (1/256 corrupted by
    the bench's rule + four other kinds of damage), the oracle agrees on the sample, exact == RLC == sharded, reject count == damage."""
    torch = torch_mod
    import bench_other as BO
    import recip_cases
    from bp_pp_amd.distributed import ReciprocalRangeProofGroup
    nd, npp, n, n_or = BO.RECIP_ND, BO.RECIP_NP, 1 << 15, 48
    gens5 = BO.recip256_generators()
    grp = ReciprocalRangeProofGroup(nd, npp, *gens5, [0], fb_window_bits=16)
    proto = grp.protocol(0)
    try:
        assert proto.device_bytes() > 50e9                                  # the 52 GB tables, not a smaller stand-in
        dV, dP, expect, shape, _, head = BO.recip256_resident_batch(torch, proto, 0, n)
        assert shape == (8, 2, 1) and int((expect == 0).sum()) == n // 256
        # the checker: reference-shaped C prover on the first n_or instances, same generators and inputs
        ocase = recip_cases.make_bulk(nd, npp, n_or, n_oracle=n_or, label=BO.RECIP_LABEL, generators=gens5,
                                      inputs={k: np.ascontiguousarray(head[k][:n_or]) for k in ("x", "s", "digits", "m", "rnd")})
        assert (head["com"][:n_or] == ocase["commitments"]).all() and (head["proofs"][:n_or] == ocase["proofs"]).all()
        # more damage, inside the oracle sample and spread over the batch (different wavefronts and chunks)
        P, V = dP.cpu().numpy(), dV.cpu().numpy()
        extra = [5, 17, 33, 41] + list(range(1000, n, 4099))
        for k, i in enumerate(extra):
            kind = k % 4
            if kind == 0:
                P[i, 64 * (4 + (k % 16)) + 9] ^= 0x20                       # a round point's x
            elif kind == 1:
                V[i] = V[(i + 1) % n]                                       # someone else's commitment
            elif kind == 2:
                P[i, 192:256] = P[i, 0:64]                                  # c_s := c_l
            else:
                P[i, -40] ^= 0x01                                           # l1
            expect[i] = 0
        dV, dP = torch.from_numpy(V).cuda(), torch.from_numpy(P).cuda()
        dA = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dS = torch.zeros(n, dtype=torch.int32, device="cuda")
        dR = torch.full((1,), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        n_bad = int((expect == 0).sum())
        proto.verify_batch_device(BO.RECIP_LABEL, n, dV.data_ptr(), dP.data_ptr(), *shape, dA.data_ptr(), dS.data_ptr())
        proto.synchronize()
        acc, st = dA.cpu().numpy(), dS.cpu().numpy()
        assert (acc == expect).all() and ((st != 0) <= (expect == 0)).all()
        for i in range(n_or):                                               # the oracle on the sample, damaged rows included
            rc = recip_cases.oracle_verify(ocase, bytes(V[i]), bytes(P[i]))
            assert int(acc[i]) == (1 if rc == 1 else 0) and (int(st[i]) != 0) == (rc < 0), (i, rc)
        for seed in (None, bytes(range(32))):                               # sharded entry points, exact and RLC
            dA.zero_(); dR.fill_(-7)
            torch.cuda.synchronize()
            grp.verify_batch_device(BO.RECIP_LABEL, n, [dV.data_ptr()], [dP.data_ptr()], *shape, [dA.data_ptr()], [dS.data_ptr()], [dR.data_ptr()],
                                    rlc_seed=seed)
            assert (dA.cpu().numpy() == expect).all() and int(dR.item()) == n_bad and (dS.cpu().numpy() == st).all()
        dA.zero_()
        proto.verify_batch_rlc_device(BO.RECIP_LABEL, n, dV.data_ptr(), dP.data_ptr(), *shape, dA.data_ptr(), dS.data_ptr(), bytes(32))
        proto.synchronize()
        assert (dA.cpu().numpy() == expect).all()
        # host-buffer form of the sharded call on a ragged slice
        a, s_, r = grp.verify_batch(BO.RECIP_LABEL, V[:3001], P[:3001], *shape)
        assert (a == expect[:3001]).all() and r == int((expect[:3001] == 0).sum())
    finally:
        proto.close()
        grp.close()


def test_u64_verify_beyond_one_internal_part(torch_mod):
    """More proofs than the default `max_batch` (2^21) in ONE call: the library runs the batch as consecutive parts with a bounded
    workspace (include/bppp.h, bppp_ctx_set_option).  2^21 + 12,345 proofs resident: every honest proof accepted, every corrupted
    one rejected, the reject count accumulated across the parts, exact and RLC mode; the workspace stays that of 2^21 proofs."""
    torch = torch_mod
    import bench
    from bp_pp_amd import U64RangeProofProtocol, synth
    n = (1 << 21) + 12345
    gens, g, gv, hv = bench.load_generators()
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=16)
    try:
        dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 7, 7 + n)
        dA = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dS = torch.zeros(n, dtype=torch.int32, device="cuda")
        dR = torch.zeros(1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        bytes_before = proto.device_bytes()
        proto.verify_batch_device(synth.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, dR.data_ptr())
        proto.synchronize()
        n_bad = int((expect == 0).sum())
        assert (dA.cpu().numpy() == expect).all() and not dS.any().item() and int(dR.item()) == n_bad and n_bad >= n // 1024
        dA.zero_()
        proto.verify_batch_rlc_device(synth.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), bytes(range(32)), dS.data_ptr(), dR.data_ptr())
        proto.synchronize()
        assert (dA.cpu().numpy() == expect).all() and int(dR.item()) == n_bad
        # workspace growth is that of one part (2^21 proofs at ~30 KB), not of the whole batch
        assert proto.device_bytes() - bytes_before < 1.25 * (1 << 21) * 40000
    finally:
        proto.close()


def test_wnla_at_the_generator_count_limit(torch_mod):
    """The C ABI's own limits (the reference has none): |g_vec| = |h_vec| = 4096 generators, |l| = |n| = 4096, 11 rounds -- commit, verify
    and prove of WeightNormLinearArgument (wnla.rs:66-190) against the oracle at that size; one generator more is refused."""
    import wnla_cases
    from bp_pp_amd import BpppError
    from bp_pp_amd.wnla import WeightNormLinearArgument
    NG = NH = 4096
    B = 2
    case = wnla_cases.make(NG, NH, B)
    assert case["rounds"] == 11
    w = WeightNormLinearArgument(case["g"], case["gv"], case["hv"], device=0, fb_window_bits=8)
    try:
        out, st = w.commit_batch(case["c"], case["mu"], case["l"], case["n"])
        assert not st.any() and (out == case["commitments"]).all()
        args = dict(commitments=case["commitments"], c=case["c"], rho=case["rho"], mu=case["mu"], proof_r=case["proof_r"],
                    proof_x=case["proof_x"], proof_l=case["proof_l"], proof_n=case["proof_n"])
        acc, st = w.verify_batch(case["label"], **args)
        assert acc.all() and not st.any()
        pn = case["proof_n"].copy(); pn[1, 0, 31] ^= 1
        acc, st = w.verify_batch(case["label"], **dict(args, proof_n=pn))
        assert acc.tolist() == [1, 0]
        pr, px, pl, pnv, st = w.prove_batch(case["label"], case["commitments"], case["c"], case["rho"], case["mu"], case["l"], case["n"])
        assert not st.any() and (pr == case["proof_r"]).all() and (px == case["proof_x"]).all()
        assert (pl == case["proof_l"]).all() and (pnv == case["proof_n"]).all()
    finally:
        w.close()
    with pytest.raises(BpppError):
        WeightNormLinearArgument(case["g"], case["gv"] + [case["gv"][0]], case["hv"], device=0, fb_window_bits=8)     # 4097 generators


def test_scale_run_kit_dry_run(tmp_path):
    """tools/scale_run.sh N dry: the one-command multi-GPU readiness kit on ONE device -- bench.py --gpus 2 as two gloo ranks on
    device 0, and tools/group_run.py as a one-device group behind a one-rank RCCL communicator.  Control flow of N > 1 only (shards,
    barrier, reject-count reduce, max-over-ranks clock, per-rank report), never a measurement: it must keep running so that the day
    a multi-GPU node exists the same command produces the curve."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCALE_TOTAL_PROOFS="8192", SCALE_STEPS="2")
    env.pop("BPPP_FORCE_RCCL", None)
    r = subprocess.run(["bash", os.path.join(root, "tools", "scale_run.sh"), "2", "dry", str(tmp_path)], capture_output=True, text=True, timeout=1500, env=env)
    print(r.stdout)                          # (shown in full by pytest when the assertion below fails)
    print(r.stderr)
    assert "bench.py --gpus 2 rc=0" in r.stdout and "group_run.py --gpus 2 rc=0" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    b = [json.loads(l) for l in open(tmp_path / "bench_gpus2.json") if '"value"' in l][0]
    assert b["n_gpus"] == 2 and b["accept_bits_ok"] and b["reject_count_all_reduced"] == 8 and b["config"]["proofs_per_gpu"] == 4096
    assert len(b["ranks"]["per_rank_ms"]) == 2 and b["ranks"]["backend"] == "gloo" and b["ranks"]["one_device_dry_run"]
    assert abs(b["ranks"]["max_ms"] - b["ms_per_step"]) < 0.5 * b["ms_per_step"] + 1.0
    g = [json.loads(l) for l in open(tmp_path / "group_gpus2.json") if '"value"' in l][0]
    assert g["accept_bits_ok"] and g["rccl_nranks"] == 1 and g["reject_count_on_every_device"] == [8] and g["dry_run_one_device"]


@pytest.mark.parametrize("workload,total,rejects", [("verify", 131072, 128), ("recip256", 2048, 8)])
def test_bench_starts_its_own_ranks(workload, total, rejects):
    """`python3 bench.py --gpus 2` with NO launcher around it (how the N = 1 line is spelled, with another N): the parent -- which never
    touches the GPU -- starts the two ranks as child processes under torch.distributed.run, relays rank 0's JSON line and the exit code.
    On a one-GPU box: gloo ranks on device 0 (BENCH_ONE_DEVICE / BENCH_DIST_BACKEND), control flow only, never a measurement."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_ONE_DEVICE="1", BENCH_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BPPP_FORCE_RCCL"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--total-proofs", str(total), "--steps", "2", "--warmup", "1",
           "--no-secondary", "--no-cpu-baseline", "--no-session-rates"]
    # (explicit small tables: two ranks sizing their tables to "the free HBM" of ONE device at the same moment is not what is under test)
    cmd += ["--fb-window-bits", "16"] if workload == "verify" else ["--workload", workload, "--fb-window-bits", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and '"value"' in l]
    assert len(lines) == 1, r.stdout[-1500:]
    b = lines[0]
    assert b["n_gpus"] == 2 and b["accept_bits_ok"] and b["reject_count_all_reduced"] == rejects and b["config"]["proofs_per_gpu"] == total // 2
    assert b["config"]["kernel_timing_during_value"] is False
    if workload == "verify":
        assert len(b["ranks"]["per_rank_ms"]) == 2 and b["ranks"]["backend"] == "gloo" and b["ranks"]["one_device_dry_run"]
        assert b["timing_pass"]["ms_per_step"] > 0 and "plan" in b["config"]
