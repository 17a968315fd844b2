"""The `t: &mut Transcript` boundary: serialized merlin states through the C ABI's host helpers and through the
device code of the verifier (compiled for the host, tests/emul).  CPU tier; the -m gpu twin is in tests/test_gpu_transcript.py."""
import os

import numpy as np
import pytest

import bppp_oracle as O
import transcript_cases as TC
from bp_pp_amd.transcript import Transcript
from emul.build import load


def test_host_transcript_ops_equal_merlin():
    """bppp_transcript_{new, append_message, challenge_bytes}: merlin's published known answer, and state-for-state equality with
    the oracle's Strobe128 through operations that cross the sponge rate."""
    t = Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    o, t = O.Transcript(b"u64 range proof"), Transcript(b"u64 range proof")
    assert TC.ser(o) == t.state
    rng = np.random.default_rng(9)
    for k in range(40):
        label = bytes(rng.integers(0, 256, int(rng.integers(0, 50)), dtype=np.uint8))
        if k % 3 == 2:
            n = int(rng.integers(0, 400))
            assert o.challenge_bytes(label, n) == t.challenge_bytes(label, n)
        elif k % 3 == 1:
            x = int(rng.integers(0, 2**63))
            o.append_u64(label, x); t.append_u64(label, x)
        else:
            m = bytes(rng.integers(0, 256, int(rng.integers(0, 700)), dtype=np.uint8))
            o.append_message(label, m); t.append_message(label, m)
        assert TC.ser(o) == t.state, k
    assert Transcript(state=t.state).state == t.state and t.clone().state == t.state


@pytest.mark.parametrize("shared", [False, True])
def test_preloaded_transcripts_through_the_device_code(shared):
    L = load()
    case = TC.make(4, shared=shared)
    W = 4
    tab = np.zeros(L.emul_fb_table_entries(49, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(case["gens"], 49, W, tab.ctypes.data) == 0
    V, P, S = case["V"].copy(), case["P"].copy(), case["states_in"]
    n = V.shape[0]
    acc, st, out = np.zeros(n, np.uint8), np.zeros(n, np.int32), np.zeros((n, 203), np.uint8)
    L.emul_u64_verify_batch_transcript(tab.ctypes.data, W, n, S.ctypes.data, S.shape[0], V.ctypes.data, P.ctypes.data, acc.ctypes.data,
                                       st.ctypes.data, out.ctypes.data)
    assert acc.tolist() == [1] * n and not st.any()
    assert (out == case["states_after"]).all()                      # the caller's transcript, advanced exactly as merlin's
    # the transcript content matters: the same proofs against Transcript::new(label) alone must fail
    plain = np.frombuffer(Transcript(b"u64 range proof").state, dtype=np.uint8).copy()
    L.emul_u64_verify_batch_transcript(tab.ctypes.data, W, n, plain.ctypes.data, 1, V.ctypes.data, P.ctypes.data, acc.ctypes.data,
                                       st.ctypes.data, out.ctypes.data)
    assert not acc.any() and not st.any()
    # negatives: a wrong proof (state still advances, as the reference's verify runs to the end), a malformed one (state untouched)
    P2 = P.copy()
    P2[0, 900] ^= 1
    P2[1, 5] ^= 0x10
    L.emul_u64_verify_batch_transcript(tab.ctypes.data, W, n, S.ctypes.data, S.shape[0], V.ctypes.data, P2.ctypes.data, acc.ctypes.data,
                                       st.ctypes.data, out.ctypes.data)
    s_in = lambda j: bytes(S[0 if shared else j])
    ok0, after0 = TC.oracle_verify(case, 0, bytes(V[0]), bytes(P2[0]), s_in(0))
    assert acc.tolist() == [0, 0, 1, 1] and not ok0 and st.tolist() == [0, 1, 0, 0]
    assert bytes(out[0]) == after0 and bytes(out[1]) == s_in(1) and (out[2:] == case["states_after"][2:]).all()


def test_derived_generators_equal_the_oracle_and_are_valid_points():
    """bppp_derive_generators (host code of the library: SHAKE256 try-and-increment, even y) against the oracle's restatement."""
    from bp_pp_amd import derive_generators
    for seed in (b"", b"bppp-bench-v1", b"x" * 200):
        gens = derive_generators(seed, 49)
        assert len(gens) == 49 * 64 and len({gens[64 * i:64 * i + 64] for i in range(49)}) == 49
        for i in (0, 1, 16, 17, 48):
            pt = O.derive_generator(seed, i)
            assert gens[64 * i:64 * i + 64] == O.pt_to_xy64(pt) and O.on_curve(pt) and pt[1] % 2 == 0
    assert derive_generators(b"s", 3, first_index=5) == derive_generators(b"s", 8)[5 * 64:]


def test_prover_over_preloaded_transcripts_through_the_device_code():
    """`prove(x, s, t: &mut Transcript, rng)` (u64_proof.rs:57): proofs made on transcripts that already hold context are
    byte-identical to the oracle prover's and the transcripts come back advanced exactly as merlin's."""
    import ref_fixture_check as RC
    L = load()
    doc = RC.oracle_made_document(4)          # cases 1 and 3 carry context, so the four states sit at two different positions
    cs = doc["cases"]
    W = 4
    tab = np.zeros(L.emul_fb_table_entries(49, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(bytes.fromhex(doc["generators"]), 49, W, tab.ctypes.data) == 0
    n = len(cs)
    u8 = lambda key, w: np.frombuffer(b"".join(bytes.fromhex(c[key]) for c in cs), dtype=np.uint8).reshape(n, w).copy()
    x = np.array([int(c["x"]) for c in cs], dtype=np.uint64)
    s, rnd, S = u8("s", 32), u8("rnd", 52 * 32), u8("state_before", 203)
    proofs, V, st, out = np.zeros((n, 928), np.uint8), np.zeros((n, 64), np.uint8), np.zeros(n, np.int32), np.zeros((n, 203), np.uint8)
    assert L.emul_u64_prove_batch_transcript(tab.ctypes.data, W, n, S.ctypes.data, n, x.ctypes.data, s.ctypes.data, rnd.ctypes.data,
                                             proofs.ctypes.data, V.ctypes.data, st.ctypes.data, out.ctypes.data) == 0
    assert not st.any() and (proofs == u8("proof", 928)).all() and (V == u8("commitment", 64)).all()
    assert (out == u8("state_after_prove", 203)).all()


def test_generic_verifiers_over_preloaded_transcripts_through_the_device_code():
    """`t: &mut Transcript` of WeightNormLinearArgument::verify (wnla.rs:75) and ReciprocalRangeProofProtocol::verify
    (reciprocal.rs:98) in the generic device code: accept bits and advanced states equal the Python oracle's."""
    import generic_transcript_cases as GC
    import ref_fixture_check as RC
    L = load()
    W = 4
    # ---- wnla
    case = GC.wnla_case()
    gens = case["g"] + b"".join(case["gv"]) + b"".join(case["hv"])
    NB = 1 + case["ng"] + case["nh"]
    tab = np.zeros(L.emul_fb_table_entries(NB, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, NB, W, tab.ctypes.data) == 0
    B = case["commitments"].shape[0]
    acc, st, out = np.zeros(B, np.uint8), np.zeros(B, np.int32), np.zeros((B, 203), np.uint8)
    L.emul_set_transcripts(case["states_in"].ctypes.data, B, out.ctypes.data)
    L.emul_wnla_run(0, tab.ctypes.data, W, case["ng"], case["nh"], b"", 0, B, case["commitments"].ctypes.data, case["c"].ctypes.data,
                    case["rho"].ctypes.data, case["mu"].ctypes.data, case["rounds"], case["proof_r"].ctypes.data, case["proof_x"].ctypes.data,
                    case["proof_l"].ctypes.data, case["nl"], case["proof_n"].ctypes.data, case["nn"], None, acc.ctypes.data, st.ctypes.data)
    assert acc.tolist() == [1] * B and not st.any() and (out == case["states_after"]).all()
    # the base case (|l| + |n| < 6, wnla.rs:80-82) performs no transcript operation: the state comes back untouched, its
    # cur_flags byte (2 after append_message) included
    case = GC.wnla_case(ng=2, nh=2, B=2)
    assert case["rounds"] == 0 and (case["states_after"] == case["states_in"]).all() and case["states_in"][0, 202] == 2
    gens = case["g"] + b"".join(case["gv"]) + b"".join(case["hv"])
    tab = np.zeros(L.emul_fb_table_entries(5, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 5, W, tab.ctypes.data) == 0
    acc, st, out = np.zeros(2, np.uint8), np.zeros(2, np.int32), np.zeros((2, 203), np.uint8)
    L.emul_set_transcripts(case["states_in"].ctypes.data, 2, out.ctypes.data)
    L.emul_wnla_run(0, tab.ctypes.data, W, 2, 2, b"", 0, 2, case["commitments"].ctypes.data, case["c"].ctypes.data, case["rho"].ctypes.data,
                    case["mu"].ctypes.data, 0, case["proof_r"].ctypes.data, case["proof_x"].ctypes.data, case["proof_l"].ctypes.data, case["nl"],
                    case["proof_n"].ctypes.data, case["nn"], None, acc.ctypes.data, st.ctypes.data)
    assert acc.tolist() == [1, 1] and not st.any() and (out == case["states_in"]).all()
    # ---- reciprocal at the u64 dimensions, on the u64 proofs of the oracle-made document (two of them carry context)
    doc = RC.oracle_made_document(4)
    cs = doc["cases"]
    gens = bytes.fromhex(doc["generators"])
    tab = np.zeros(L.emul_fb_table_entries(49, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 49, W, tab.ctypes.data) == 0
    n = len(cs)
    u8 = lambda key, w: np.frombuffer(b"".join(bytes.fromhex(c[key]) for c in cs), dtype=np.uint8).reshape(n, w).copy()
    V, P, S = u8("commitment", 64), u8("proof", 928), u8("state_before", 203)
    acc, st, out = np.zeros(n, np.uint8), np.zeros(n, np.int32), np.zeros((n, 203), np.uint8)
    L.emul_set_transcripts(S.ctypes.data, n, out.ctypes.data)
    L.emul_recip_verify(tab.ctypes.data, W, 16, 32, 16, 16, b"", 0, n, V.ctypes.data, P.ctypes.data, 4, 2, 1, acc.ctypes.data, st.ctypes.data)
    assert acc.tolist() == [1] * n and not st.any() and (out == u8("state_after_verify", 203)).all()


def test_committed_round2_golden_vectors(oracle_c):
    """tests/golden/r02_golden.json (made by make_golden_r02.py): the host transcript operations replay the scripted states byte
    for byte, the oracle reproduces the document (proofs over pre-loaded transcripts, states before / after), and the generator
    derivation reproduces the committed points."""
    import json
    import ref_fixture_check as RC
    from bp_pp_amd import derive_generators
    with open(os.path.join(os.path.dirname(__file__), "golden", "r02_golden.json")) as f:
        doc = json.load(f)
    ts = doc["transcript_script"]
    t = Transcript(bytes.fromhex(ts["label"]))
    assert t.state.hex() == ts["states"][0]
    for op, st in zip(ts["ops"], ts["states"][1:]):
        if op["op"] == "append_message":
            t.append_message(bytes.fromhex(op["label"]), bytes.fromhex(op["message"]))
        elif op["op"] == "append_u64":
            t.append_u64(bytes.fromhex(op["label"]), int(op["value"]))
        else:
            assert t.challenge_bytes(bytes.fromhex(op["label"]), op["n"]).hex() == op["output"]
        assert t.state.hex() == st
    assert RC.check_document(doc, oracle_c) == len(doc["cases"])
    dg = doc["derived_generators"]
    assert derive_generators(bytes.fromhex(dg["seed"]), 5, dg["first_index"]).hex() == dg["points"]


def _table(L, gens: bytes, nb: int, W: int = 4):
    tab = np.zeros(L.emul_fb_table_entries(nb, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, nb, W, tab.ctypes.data) == 0
    return tab


def test_generic_provers_over_preloaded_transcripts_through_the_device_code():
    """`t: &mut Transcript` of WeightNormLinearArgument::prove (wnla.rs:125), ReciprocalRangeProofProtocol::prove (reciprocal.rs:109)
    and ArithmeticCircuit::prove / verify (circuit.rs:260,154) in the generic device code: proofs byte-identical to the Python
    oracle's on transcripts that already hold per-instance context (different sponge positions), advanced states equal."""
    import ctypes as C
    import generic_transcript_cases as GC
    import ref_fixture_check as RC
    L = load()
    W = 4
    # ---- wnla prover (3 rounds) and its base case (no rounds: the transcript is not touched)
    for kw in (dict(), dict(ng=2, nh=2, B=2)):
        case = GC.wnla_case(**kw)
        B, ng, nh = case["commitments"].shape[0], case["ng"], case["nh"]
        tab = _table(L, case["g"] + b"".join(case["gv"]) + b"".join(case["hv"]), 1 + ng + nh)
        pr, px = np.zeros((B, max(case["rounds"], 1), 64), np.uint8), np.zeros((B, max(case["rounds"], 1), 64), np.uint8)
        pl, pn = np.zeros((B, case["nl"], 32), np.uint8), np.zeros((B, case["nn"], 32), np.uint8)
        st, out = np.zeros(B, np.int32), np.zeros((B, 203), np.uint8)
        r, a, b = C.c_int(), C.c_int(), C.c_int()
        l, n = case["l"].reshape(B, -1, 32), case["n"].reshape(B, -1, 32)
        L.emul_set_transcripts(case["states_in"].ctypes.data, B, out.ctypes.data)
        L.emul_wnla_prove(tab.ctypes.data, W, ng, nh, b"", 0, B, case["commitments"].ctypes.data, case["c"].ctypes.data, case["rho"].ctypes.data,
                          case["mu"].ctypes.data, l.ctypes.data, l.shape[1], n.ctypes.data, n.shape[1], pr.ctypes.data, px.ctypes.data,
                          pl.ctypes.data, pn.ctypes.data, st.ctypes.data, C.byref(r), C.byref(a), C.byref(b))
        assert (r.value, a.value, b.value) == (case["rounds"], case["nl"], case["nn"]) and not st.any()
        if case["rounds"]:
            assert pr.tobytes() == case["proof_r"].tobytes() and px.tobytes() == case["proof_x"].tobytes()
        assert pl.tobytes() == case["proof_l"].tobytes() and pn.tobytes() == case["proof_n"].tobytes()
        assert (out == case["states_after_prove"]).all()
    # ---- reciprocal prover at the u64 dimensions on the oracle-made document (two of its cases carry context)
    doc = RC.oracle_made_document(4)
    cs = doc["cases"]
    n = len(cs)
    u8 = lambda key, *sh: np.frombuffer(b"".join(bytes.fromhex(c[key]) for c in cs), dtype=np.uint8).reshape(n, *sh).copy()
    tab = _table(L, bytes.fromhex(doc["generators"]), 49)
    x, digits, m = GC.recip_prover_inputs(doc)
    V, s, rnd, S = u8("commitment", 64), u8("s", 32), u8("rnd", 52, 32), u8("state_before", 203)
    proofs, st, out = np.zeros((n, 928), np.uint8), np.zeros(n, np.int32), np.zeros((n, 203), np.uint8)
    L.emul_set_transcripts(S.ctypes.data, n, out.ctypes.data)
    rc = L.emul_recip_prove(tab.ctypes.data, W, 16, 32, 16, 16, b"", 0, n, V.ctypes.data, x.ctypes.data, s.ctypes.data, digits.ctypes.data,
                            m.ctypes.data, rnd.ctypes.data, proofs.ctypes.data, st.ctypes.data)
    assert rc == 928 and not st.any() and (proofs == u8("proof", 928)).all() and (out == u8("state_after_prove", 203)).all()
    # ---- generic circuit: prove, then verify the proofs, both over the pre-loaded transcripts
    case = GC.circuit_case()
    B, p = case["commitments"].shape[0], case["parts"]
    gens = case["g"] + b"".join(case["gv"] + case["gv_"] + case["hv"] + case["hv_"])
    tab = _table(L, gens, 1 + case["NG"] + case["NH"])
    dims = (C.c_size_t * 6)(case["nm"], case["no"], case["k"], case["nl"], case["nv"], case["nw"])
    proofs, st, out = np.zeros_like(case["proofs"]), np.zeros(B, np.int32), np.zeros((B, 203), np.uint8)
    L.emul_set_transcripts(case["states_in"].ctypes.data, B, out.ctypes.data)
    rc = L.emul_circuit_prove(tab.ctypes.data, W, case["NG"], case["NH"], dims, int(case["f_l"]), int(case["f_m"]), case["Wm_bytes"],
                              case["Wl_bytes"], case["am_bytes"], case["al_bytes"], p["LO"].ctypes.data, p["LL"].ctypes.data, p["LR"].ctypes.data,
                              p["NO"].ctypes.data, b"", 0, B, case["commitments"].ctypes.data, case["v_bytes"].ctypes.data, case["s_v"].ctypes.data,
                              case["wl_bytes"].ctypes.data, case["wr_bytes"].ctypes.data, case["wo_bytes"].ctypes.data, case["rnd"].ctypes.data,
                              proofs.ctypes.data, st.ctypes.data)
    assert rc == proofs.shape[1] and not st.any() and (proofs == case["proofs"]).all() and (out == case["states_after_prove"]).all()
    acc, st, out = np.zeros(B, np.uint8), np.zeros(B, np.int32), np.zeros((B, 203), np.uint8)
    L.emul_set_transcripts(case["states_in"].ctypes.data, B, out.ctypes.data)
    rc = L.emul_circuit_verify(tab.ctypes.data, W, case["NG"], case["NH"], dims, int(case["f_l"]), int(case["f_m"]), case["Wm_bytes"],
                               case["Wl_bytes"], case["am_bytes"], case["al_bytes"], p["LO"].ctypes.data, p["LL"].ctypes.data, p["LR"].ctypes.data,
                               p["NO"].ctypes.data, b"", 0, B, case["commitments"].ctypes.data, proofs.ctypes.data, case["rounds"], case["pl"],
                               case["pn"], acc.ctypes.data, st.ctypes.data, None, None)
    assert rc == 0 and acc.tolist() == [1] * B and not st.any() and (out == case["states_after_verify"]).all()
