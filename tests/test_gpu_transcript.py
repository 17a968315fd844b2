"""GPU twin of tests/test_transcript_state.py: bppp_u64_verify_batch_transcript[_device] on an MI355X -- proofs made over
pre-loaded transcripts by the oracle, accept bits and the per-proof advanced merlin states equal to the oracle's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shared", [False, True])
def test_preloaded_transcripts_on_the_gpu(shared):
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import transcript_cases as TC
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    from bp_pp_amd.transcript import Transcript
    case = TC.make(5, shared=shared)
    g, gv, hv = workload.split_generators(case["gens"])
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
    try:
        V, P, S = case["V"], case["P"], case["states_in"]
        n = V.shape[0]
        acc, st, out = proto.verify_batch_transcript(V, P, S[0].tobytes() if shared else [s.tobytes() for s in S])
        assert acc.tolist() == [1] * n and not st.any() and (out == case["states_after"]).all()
        # label-only entry point on the same proofs: the bound context is missing from the transcript, so they must fail
        acc0, st0 = proto.verify_batch(V, P, workload.LABEL)
        assert not acc0.any() and not st0.any()
        # ... and the transcript entry point fed Transcript::new(label) equals the label entry point on label-made proofs
        gens2, V2, P2, _ = workload.make_batch(70, first=5)
        assert gens2 == case["gens"]
        a1, s1 = proto.verify_batch(V2, P2, workload.LABEL)
        a2, s2, o2 = proto.verify_batch_transcript(V2, P2, Transcript(workload.LABEL))
        assert a1.all() and (a1 == a2).all() and (s1 == s2).all()
        # negatives, judged by the oracle: wrong proof (state advances), malformed proof (state untouched), wrong commitment
        Pn, Vn = P.copy(), V.copy()
        Pn[0, 900] ^= 1
        Pn[1, 5] ^= 0x10
        Vn[2] = V[3]
        acc, st, out = proto.verify_batch_transcript(Vn, Pn, S[0].tobytes() if shared else [s.tobytes() for s in S])
        s_in = lambda j: bytes(S[0 if shared else j])
        assert acc.tolist() == [0, 0, 0, 1, 1] and st.tolist() == [0, 1, 0, 0, 0]
        for j in (0, 2):
            ok, after = TC.oracle_verify(case, j, bytes(Vn[j]), bytes(Pn[j]), s_in(j))
            assert not ok and bytes(out[j]) == after
        assert bytes(out[1]) == s_in(1) and (out[3:] == case["states_after"][3:]).all()
        # device-buffer variant, asynchronous on the context's stream
        dV, dP, dS = torch.from_numpy(V).cuda(), torch.from_numpy(P).cuda(), torch.from_numpy(S).cuda()
        dA = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dSt = torch.zeros(n, dtype=torch.int32, device="cuda")
        dR = torch.zeros(1, dtype=torch.int32, device="cuda")
        dO = torch.zeros((n, 203), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        from bp_pp_amd import _capi
        _capi.check(_capi.lib().bppp_u64_verify_batch_transcript_device(proto._ctx, n, dS.data_ptr(), S.shape[0], dV.data_ptr(), dP.data_ptr(),
                                                                        dA.data_ptr(), dSt.data_ptr(), dR.data_ptr(), dO.data_ptr()))
        proto.synchronize()
        assert dA.cpu().numpy().all() and int(dR.item()) == 0 and (dO.cpu().numpy() == case["states_after"]).all()
        # a state merlin cannot be in is refused
        bad = bytearray(S[0].tobytes()); bad[200] = 166
        with pytest.raises(Exception):
            proto.verify_batch_transcript(V, P, bytes(bad))
    finally:
        proto.close()


def test_prover_over_preloaded_transcripts_on_the_gpu():
    """bppp_u64_prove_batch_transcript: per-proof transcripts at two different sponge positions in one wavefront; proofs byte-identical
    to the oracle prover's, transcripts advanced as merlin's, and the GPU verifier accepts them over the same transcripts."""
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import ref_fixture_check as RC
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    doc = RC.oracle_made_document(4)
    cs = doc["cases"]
    n = len(cs)
    g, gv, hv = workload.split_generators(bytes.fromhex(doc["generators"]))
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
    try:
        u8 = lambda key, w: np.frombuffer(b"".join(bytes.fromhex(c[key]) for c in cs), dtype=np.uint8).reshape(n, w).copy()
        x = np.array([int(c["x"]) for c in cs], dtype=np.uint64)
        S = u8("state_before", 203)
        proofs, com, st, out = proto.prove_batch_transcript(x, u8("s", 32), u8("rnd", 52 * 32), [s.tobytes() for s in S])
        assert not st.any() and (proofs == u8("proof", 928)).all() and (com == u8("commitment", 64)).all()
        assert (out == u8("state_after_prove", 203)).all()
        acc, vst, vout = proto.verify_batch_transcript(com, proofs, [s.tobytes() for s in S])
        assert acc.all() and not vst.any() and (vout == u8("state_after_verify", 203)).all()
        # one shared state: Transcript::new(label) for everybody == the label entry point
        from bp_pp_amd.transcript import Transcript
        p2, c2, st2, _ = proto.prove_batch_transcript(x, u8("s", 32), u8("rnd", 52 * 32), Transcript(bytes.fromhex(doc["label"])))
        p3, c3, st3 = proto.prove_batch(x, u8("s", 32), u8("rnd", 52 * 32), bytes.fromhex(doc["label"]))
        assert (p2 == p3).all() and (c2 == c3).all() and not st2.any()
    finally:
        proto.close()


@pytest.mark.parametrize("reps", [1700, 3400])
def test_verifier_beside_modes_with_mixed_sponge_positions(reps):
    """20,400 and 40,800 proofs over PER-PROOF transcripts whose sponge positions differ inside every wavefront: the sizes at which phase 1
    runs in 256-thread workgroups (a sponge block per wavefront in LDS) beside the table kernel, and the last round as head + tail.
    Twelve oracle-made cases repeated: verdicts and advanced transcripts as the oracle's; one wrong proof judged by the oracle."""
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import transcript_cases as TC
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    case = TC.make(12, shared=False)
    g, gv, hv = workload.split_generators(case["gens"])
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
    try:
        V, P, S = np.tile(case["V"], (reps, 1)), np.tile(case["P"], (reps, 1)), np.tile(case["states_in"], (reps, 1))
        n = V.shape[0]
        assert len({bytes(s)[200] for s in S[:12]}) >= 3
        j = n - 7
        P[j, 901] ^= 4
        acc, st, out = proto.verify_batch_transcript(V, P, [s.tobytes() for s in S])
        good = np.ones(n, bool); good[j] = False
        assert acc[good].all() and not acc[j] and not st.any()
        assert (out[good] == np.tile(case["states_after"], (reps, 1))[good]).all()
        okj, after = TC.oracle_verify(case, j % 12, bytes(V[j]), bytes(P[j]), bytes(S[j]))
        assert not okj and bytes(out[j]) == after
    finally:
        proto.close()


@pytest.mark.parametrize("reps", [300, 1300, 4200])
def test_prover_lane_forms_with_mixed_sponge_positions(reps):
    """The prover's dispatch regimes beyond the small call -- sixteen lanes per value in their 256-register builds (1,200 values), four
    lanes per value with the fold leaving the next round's scalars (5,200), one lane per value with the round scalars from four
    workgroups (16,800) -- each over PER-PROOF transcripts at two different sponge positions alternating inside every wavefront, so the
    lane groups run inside for_each_position_group's divergent trips.  Four oracle-made cases repeated: proofs, commitments and
    advanced transcripts byte-identical to the oracle prover's."""
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import ref_fixture_check as RC
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    doc = RC.oracle_made_document(4)
    cs = doc["cases"]
    g, gv, hv = workload.split_generators(bytes.fromhex(doc["generators"]))
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
    try:
        u8 = lambda key, w: np.tile(np.frombuffer(b"".join(bytes.fromhex(c[key]) for c in cs), dtype=np.uint8).reshape(len(cs), w), (reps, 1))
        x = np.tile(np.array([int(c["x"]) for c in cs], dtype=np.uint64), reps)
        S = u8("state_before", 203)
        assert len({bytes(s)[200] for s in S[:4]}) >= 2                       # different byte positions among neighbours
        proofs, com, st, out = proto.prove_batch_transcript(x, u8("s", 32), u8("rnd", 52 * 32), [s.tobytes() for s in S])
        assert not st.any() and (proofs == u8("proof", 928)).all() and (com == u8("commitment", 64)).all()
        assert (out == u8("state_after_prove", 203)).all()
    finally:
        proto.close()


def test_generic_verifiers_over_preloaded_transcripts_on_the_gpu():
    """bppp_wnla_verify_batch_transcript and bppp_reciprocal_verify_batch_transcript (instances at different sponge positions in one
    wavefront), against the Python oracle's verdicts and advanced states."""
    import ctypes as C
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import generic_transcript_cases as GC
    import ref_fixture_check as RC
    import workload
    from bp_pp_amd import _capi
    from bp_pp_amd.wnla import ReciprocalRangeProofProtocol, WeightNormLinearArgument
    L = _capi.lib()
    case = GC.wnla_case()
    w = WeightNormLinearArgument(case["g"], case["gv"], case["hv"], device=0, fb_window_bits=8)
    try:
        B = case["commitments"].shape[0]
        acc, st, out = np.zeros(B, np.uint8), np.zeros(B, np.int32), np.zeros((B, 203), np.uint8)
        _capi.check(L.bppp_wnla_verify_batch_transcript(w._ctx, B, case["states_in"].ctypes.data, B, case["commitments"].ctypes.data,
                                                        case["c"].ctypes.data, case["rho"].ctypes.data, case["mu"].ctypes.data, case["rounds"],
                                                        case["proof_r"].ctypes.data, case["proof_x"].ctypes.data, case["proof_l"].ctypes.data,
                                                        case["nl"], case["proof_n"].ctypes.data, case["nn"], acc.ctypes.data, st.ctypes.data,
                                                        out.ctypes.data))
        assert acc.tolist() == [1] * B and not st.any() and (out == case["states_after"]).all()
        # the same proofs against the wrong transcripts (instance 0's state for everybody) fail for all but instance 0
        _capi.check(L.bppp_wnla_verify_batch_transcript(w._ctx, B, case["states_in"].ctypes.data, 1, case["commitments"].ctypes.data,
                                                        case["c"].ctypes.data, case["rho"].ctypes.data, case["mu"].ctypes.data, case["rounds"],
                                                        case["proof_r"].ctypes.data, case["proof_x"].ctypes.data, case["proof_l"].ctypes.data,
                                                        case["nl"], case["proof_n"].ctypes.data, case["nn"], acc.ctypes.data, st.ctypes.data, None))
        assert acc.tolist() == [1] + [0] * (B - 1)
    finally:
        w.close()
    doc = RC.oracle_made_document(4)
    cs = doc["cases"]
    n = len(cs)
    g, gv, hv = workload.split_generators(bytes.fromhex(doc["generators"]))
    r = ReciprocalRangeProofProtocol(16, 16, g, gv, hv[:26], [], hv[26:], device=0, fb_window_bits=8)
    try:
        u8 = lambda key, wd: np.frombuffer(b"".join(bytes.fromhex(c[key]) for c in cs), dtype=np.uint8).reshape(n, wd).copy()
        V, P, S = u8("commitment", 64), u8("proof", 928), u8("state_before", 203)
        acc, st, out = np.zeros(n, np.uint8), np.zeros(n, np.int32), np.zeros((n, 203), np.uint8)
        _capi.check(L.bppp_reciprocal_verify_batch_transcript(r._w._ctx, n, S.ctypes.data, n, 16, 16, V.ctypes.data, P.ctypes.data, 4, 2, 1,
                                                              acc.ctypes.data, st.ctypes.data, out.ctypes.data))
        assert acc.tolist() == [1] * n and not st.any() and (out == u8("state_after_verify", 203)).all()
    finally:
        r.close()


def test_generic_provers_over_preloaded_transcripts_on_the_gpu():
    """bppp_{wnla,reciprocal,circuit}_prove_batch_transcript + bppp_circuit_verify_batch_transcript through the Python mirror:
    instances of one wavefront at different sponge positions; proofs byte-identical to the Python oracle's, states advanced as
    merlin's; the WNLA base case hands the transcript back untouched."""
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import generic_transcript_cases as GC
    import ref_fixture_check as RC
    import workload
    from bp_pp_amd.wnla import ArithmeticCircuit, ReciprocalRangeProofProtocol, WeightNormLinearArgument
    for kw in (dict(), dict(ng=2, nh=2, B=2)):
        case = GC.wnla_case(**kw)
        B = case["commitments"].shape[0]
        w = WeightNormLinearArgument(case["g"], case["gv"], case["hv"], device=0, fb_window_bits=8)
        try:
            l, n = case["l"].reshape(B, -1, 32), case["n"].reshape(B, -1, 32)
            pr, px, pl, pn, st, out = w.prove_batch(b"", case["commitments"], case["c"], case["rho"], case["mu"], l, n, transcripts=case["states_in"])
            assert not st.any() and pr.shape[1] == case["rounds"]
            assert pr.tobytes() == case["proof_r"].tobytes() and px.tobytes() == case["proof_x"].tobytes()
            assert pl.tobytes() == case["proof_l"].tobytes() and pn.tobytes() == case["proof_n"].tobytes()
            assert (out == case["states_after_prove"]).all()
            acc, st, out = w.verify_batch(b"", case["commitments"], case["c"], case["rho"], case["mu"], pr, px, pl, pn, transcripts=case["states_in"])
            assert acc.all() and not st.any() and (out == case["states_after"]).all()
            # one shared transcript (instance 0's): instance 0's proof is reproduced, the states all start from it
            pr1, px1, pl1, pn1, st1, out1 = w.prove_batch(b"", case["commitments"], case["c"], case["rho"], case["mu"], l, n,
                                                          transcripts=case["states_in"][0].tobytes())
            assert not st1.any() and pl1[0].tobytes() == pl[0].tobytes() and (out1[0] == case["states_after_prove"][0]).all()
        finally:
            w.close()
    doc = RC.oracle_made_document(4)
    cs = doc["cases"]
    n = len(cs)
    u8 = lambda key, *sh: np.frombuffer(b"".join(bytes.fromhex(c[key]) for c in cs), dtype=np.uint8).reshape(n, *sh).copy()
    g, gv, hv = workload.split_generators(bytes.fromhex(doc["generators"]))
    r = ReciprocalRangeProofProtocol(16, 16, g, gv, hv[:26], [], hv[26:], device=0, fb_window_bits=8)
    try:
        x, digits, m = GC.recip_prover_inputs(doc)
        proofs, st, shape, out = r.prove_batch(b"", u8("commitment", 64), x, u8("s", 32), digits, m, u8("rnd", 52, 32), transcripts=u8("state_before", 203))
        assert shape == (4, 2, 1) and not st.any() and (proofs == u8("proof", 928)).all() and (out == u8("state_after_prove", 203)).all()
        acc, st, out = r.verify_batch(b"", u8("commitment", 64), proofs, *shape, transcripts=u8("state_before", 203))
        assert acc.all() and not st.any() and (out == u8("state_after_verify", 203)).all()
    finally:
        r.close()
    case = GC.circuit_case()
    part = case["part"]
    partition = lambda typ, j: part[typ][j] if j < len(part[typ]) and part[typ][j] >= 0 else None
    sc = lambda blob: np.frombuffer(blob, np.uint8).reshape(-1, 32)
    ckt = ArithmeticCircuit(case["nm"], case["no"], case["k"], case["nv"], case["g"], case["gv"], case["hv"], sc(case["Wm_bytes"]), sc(case["Wl_bytes"]),
                            sc(case["am_bytes"]), sc(case["al_bytes"]), case["f_l"], case["f_m"], case["gv_"], case["hv_"], partition, device=0,
                            fb_window_bits=8)
    try:
        proofs, st, shape, out = ckt.prove_batch(b"", case["commitments"], case["v_bytes"], case["s_v"], case["wl_bytes"], case["wr_bytes"],
                                                 case["wo_bytes"], case["rnd"], transcripts=case["states_in"])
        assert shape == (case["rounds"], case["pl"], case["pn"]) and not st.any()
        assert (proofs == case["proofs"]).all() and (out == case["states_after_prove"]).all()
        acc, st, out = ckt.verify_batch(b"", case["commitments"], proofs, *shape, transcripts=case["states_in"])
        assert acc.all() and not st.any() and (out == case["states_after_verify"]).all()
        # a proof verified on somebody else's transcript fails
        acc, st, _ = ckt.verify_batch(b"", case["commitments"], proofs, *shape, transcripts=case["states_in"][0].tobytes())
        assert acc.tolist() == [1] + [0] * (len(acc) - 1)
    finally:
        ckt.close()


@pytest.mark.parametrize("full_batch_kernels", [False, True])
def test_every_sponge_position_in_one_batch(full_batch_kernels, monkeypatch):
    """166 proofs whose transcripts sit at ALL 166 byte positions of the STROBE-128 rate (context messages of every length mod
    166): three wavefronts in which every lane is its own position group.  Proofs from the GPU prover over those transcripts, then
    verified over them: all accepted, the advanced states equal the prover's, and a sample (every 17th position) reproduced by the
    Python oracle -- proof bytes, verdict and both advanced states."""
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    if full_batch_kernels:
        # the two-wave builds that full batches run -- sponge state in LDS (merlin.h: strobe_lds), one lane per proof -- forced onto
        # this 166-proof batch, so that the position groups are exercised on THEM too (the context reads the switches when it is made)
        monkeypatch.setenv("BPPP_NO_SMALL_KERNELS", "1")
        monkeypatch.setenv("BPPP_NO_LANE_GROUPS", "1")
    import bppp_oracle as O
    import workload
    from transcript_cases import ser
    from bp_pp_amd import U64RangeProofProtocol, synth
    from bp_pp_amd.transcript import Transcript
    gens = workload.generators()
    g, gv, hv = workload.split_generators(gens)
    ts, seen = [], set()
    for j in range(400):
        t = Transcript(synth.LABEL)
        t.append_message(b"ctx", bytes([j & 0xFF]) * j)
        pos = t.state[200]
        if pos not in seen:
            seen.add(pos)
            ts.append((pos, j, t))
        if len(seen) == 166:
            break
    assert len(seen) == 166
    ts.sort()
    n = len(ts)
    S = np.frombuffer(b"".join(t.state for _, _, t in ts), np.uint8).reshape(n, 203).copy()
    x, s, rnd = synth.bulk_values(n, first=900), synth.bulk_blindings(n, first=900), synth.bulk_prover_randomness(n, first=900)
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
    try:
        proofs, com, st, after_prove = proto.prove_batch_transcript(x, s, rnd, S)
        assert not st.any()
        acc, vst, after_verify = proto.verify_batch_transcript(com, proofs, S)
        assert acc.all() and not vst.any() and (after_verify == after_prove).all()       # prover and verifier leave the same transcript
        acc2, _, _ = proto.verify_batch_transcript(com, proofs, np.roll(S, 1, axis=0))  # everybody on a neighbour's transcript
        assert not acc2.any()
    finally:
        proto.close()
    pts = [O.pt_from_xy64(gens[64 * i:64 * i + 64]) for i in range(49)]
    oproto = O.U64RangeProofProtocol(pts[0], pts[1:17], pts[17:49])
    for i in range(0, n, 17):
        _, j, _ = ts[i]
        t = O.Transcript(synth.LABEL)
        t.append_message(b"ctx", bytes([j & 0xFF]) * j)
        assert ser(t) == bytes(S[i])
        tp = t.clone()
        rs = [int.from_bytes(bytes(rnd[i, 32 * k:32 * k + 32]), "big") for k in range(52)]
        pr = oproto.prove(int(x[i]), int.from_bytes(bytes(s[i]), "big"), tp, O.ScalarRng(rs))
        assert O.u64_proof_to_bytes(pr) == bytes(proofs[i]) and ser(tp) == bytes(after_prove[i])
        tv = t.clone()
        assert oproto.verify(O.pt_from_xy64(bytes(com[i])), pr, tv) and ser(tv) == bytes(after_verify[i])


@pytest.mark.parametrize("full_batch_kernels", [False, True])
def test_invalid_state_does_not_disturb_its_wavefront(full_batch_kernels, monkeypatch):
    """Per-proof isolation through the DEVICE entry point (the host variants refuse such a state up front): one pre-loaded state merlin
    cannot be in (pos_begin = 200) sits in lane 0 -- the wavefront's first active lane, the leader of its position group -- among valid
    per-proof states at the same byte position.  That lane is flagged and rejected; its transcript restarts from the context's base, at
    ANOTHER position, and must not drag its neighbours there: they are accepted and come back with the prover's advanced states."""
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    if full_batch_kernels:                                  # the LDS-sponge builds of full batches, forced onto this small one
        monkeypatch.setenv("BPPP_NO_SMALL_KERNELS", "1")
        monkeypatch.setenv("BPPP_NO_LANE_GROUPS", "1")
    import workload
    from bp_pp_amd import U64RangeProofProtocol, _capi, synth
    from bp_pp_amd.transcript import Transcript
    g, gv, hv = workload.split_generators(workload.generators())
    n = 70                                                  # two wavefronts; the bad lanes are 0 (leader) and 64 + 3
    ts = []
    for j in range(n):
        t = Transcript(synth.LABEL)
        t.append_message(b"ctx", bytes([j]) * 37)           # same length => same sponge position, different contents
        ts.append(t)
    S = np.frombuffer(b"".join(t.state for t in ts), np.uint8).reshape(n, 203).copy()
    assert len(set(S[:, 200].tolist())) == 1 and S[0, 200] != Transcript(b"").state[200]
    x, s, rnd = synth.bulk_values(n, first=40), synth.bulk_blindings(n, first=40), synth.bulk_prover_randomness(n, first=40)
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
    try:
        proofs, com, st, after = proto.prove_batch_transcript(x, s, rnd, S)
        assert not st.any()
        bad = S.copy()
        bad[0, 201] = 200
        bad[67, 201] = 200
        dV, dP, dS = torch.from_numpy(com).cuda(), torch.from_numpy(proofs).cuda(), torch.from_numpy(bad).cuda()
        dA = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dSt = torch.zeros(n, dtype=torch.int32, device="cuda")
        dR = torch.zeros(1, dtype=torch.int32, device="cuda")
        dO = torch.zeros((n, 203), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        _capi.check(_capi.lib().bppp_u64_verify_batch_transcript_device(proto._ctx, n, dS.data_ptr(), n, dV.data_ptr(), dP.data_ptr(),
                                                                        dA.data_ptr(), dSt.data_ptr(), dR.data_ptr(), dO.data_ptr()))
        proto.synchronize()
        acc, vst, out = dA.cpu().numpy(), dSt.cpu().numpy(), dO.cpu().numpy()
        good = np.ones(n, bool); good[[0, 67]] = False
        assert acc[good].all() and not vst[good].any() and (out[good] == after[good]).all()
        assert not acc[~good].any() and (vst[~good] == 1).all() and int(dR.item()) == 2
    finally:
        proto.close()
