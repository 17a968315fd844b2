"""The C restatement (oracle/bppp_ref.c) against the Python oracle's committed golden vectors, byte for byte, plus the
trapdoor (known-discrete-log) batch prover against the honest reference-shaped prover."""
import json
import os

import numpy as np

import bppp_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _gold():
    with open(os.path.join(GOLD, "u64_golden.json")) as f:
        return json.load(f)


def test_c_merlin_and_points(oracle_c):
    assert oracle_c.merlin_kat(b"test protocol", b"some label", b"some data", b"challenge", 32).hex() == \
        "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    two_g = oracle_c.point_mul(None, (2).to_bytes(32, "big"))
    assert two_g == O.pt_to_xy64(O.pt_mul(O.G, 2))
    assert oracle_c.point_mul(None, (O.N - 1).to_bytes(32, "big")) == O.pt_to_xy64(O.pt_neg(O.G))
    assert oracle_c.point_add(two_g, O.pt_to_xy64(O.pt_neg(O.pt_mul(O.G, 2)))) == bytes(64)
    assert oracle_c.point_add(two_g, two_g) == O.pt_to_xy64(O.pt_mul(O.G, 4))
    for a in [1, 2, O.N - 1, 0x1234567890ABCDEF << 100]:
        f, v = oracle_c.scalar_inv(a.to_bytes(32, "big"))
        assert f == v == pow(a, -1, O.N).to_bytes(32, "big")


def test_c_prove_verify_match_golden(oracle_c):
    gold = _gold()
    gens, label = bytes.fromhex(gold["generators"]), bytes.fromhex(gold["label"])
    for c in gold["cases"]:
        proof, V = oracle_c.u64_prove(gens, label, c["x"], bytes.fromhex(c["s"]), bytes.fromhex(c["rnd"]))
        assert proof.hex() == c["proof"] and V.hex() == c["commitment"]
        assert oracle_c.u64_commit_value(gens, c["x"], bytes.fromhex(c["s"])).hex() == c["commitment"]
        rc, tr = oracle_c.u64_verify(gens, label, V, proof, trace=True)
        assert rc == 1
        exp = bytes.fromhex(c["trace_challenges_and_points"])
        assert tr[:320 + 64 * 5] == exp                        # challenges, V+r, C0..C3 (C4 is not in the python trace)
    for c in gold["negative_cases"]:
        rc = oracle_c.u64_verify(gens, label, bytes.fromhex(c["commitment"]), bytes.fromhex(c["proof"]))
        assert rc == (0 if c["status"] == 0 else -1), c["what"]


def test_trapdoor_prover_is_byte_identical(oracle_c):
    gold = _gold()
    gens, label, dlogs = bytes.fromhex(gold["generators"]), bytes.fromhex(gold["label"]), bytes.fromhex(gold["generator_dlogs"])
    cases = gold["cases"]
    x = np.array([c["x"] for c in cases], dtype=np.uint64)
    s = np.frombuffer(b"".join(bytes.fromhex(c["s"]) for c in cases), dtype=np.uint8).reshape(len(cases), 32)
    rnd = np.frombuffer(b"".join(bytes.fromhex(c["rnd"]) for c in cases), dtype=np.uint8).reshape(len(cases), 52 * 32)
    p1, v1 = oracle_c.u64_prove_batch(gens, label, x, s, rnd, nthreads=2)
    p2, v2 = oracle_c.u64_prove_trapdoor_batch(dlogs, label, x, s, rnd, nthreads=2)
    assert (p1 == p2).all() and (v1 == v2).all()
    for i, c in enumerate(cases):
        assert bytes(p2[i]).hex() == c["proof"]
    acc, st = oracle_c.u64_verify_batch(gens, label, v2, p2, nthreads=2)
    assert acc.all() and not st.any()


def test_c_wnla_golden(oracle_c):
    import ctypes as C
    with open(os.path.join(GOLD, "wnla_golden.json")) as f:
        w = json.load(f)
    L = oracle_c.lib()
    b = lambda k: bytes.fromhex(w[k])
    l = b"".join(int(v).to_bytes(32, "big") for v in w["l"])
    n = b"".join(int(v).to_bytes(32, "big") for v in w["n"])
    out = C.create_string_buffer(64)
    sz = C.c_size_t
    assert L.bppp_oracle_wnla_commit(b("g"), b("g_vec"), sz(4), b("h_vec"), sz(4), b("c"), sz(4), b("rho"), b("mu"), l, sz(4), n, sz(4), out) == 0
    assert out.raw.hex() == w["commitment"]
    r_out, x_out = C.create_string_buffer(64 * 8), C.create_string_buffer(64 * 8)
    l_out, n_out = C.create_string_buffer(32 * 8), C.create_string_buffer(32 * 8)
    nr, nl, nn = sz(0), sz(0), sz(0)
    rc = L.bppp_oracle_wnla_prove(b("g"), b("g_vec"), sz(4), b("h_vec"), sz(4), b("c"), sz(4), b("rho"), b("mu"), b("label"),
                                  sz(len(b("label"))), b("commitment"), l, sz(4), n, sz(4), r_out, x_out, C.byref(nr), l_out,
                                  C.byref(nl), n_out, C.byref(nn))
    assert rc == 0
    assert r_out.raw[:64 * nr.value].hex() == w["proof_r"] and x_out.raw[:64 * nr.value].hex() == w["proof_x"]
    assert l_out.raw[:32 * nl.value].hex() == w["proof_l"] and n_out.raw[:32 * nn.value].hex() == w["proof_n"]
    rc = L.bppp_oracle_wnla_verify(b("g"), b("g_vec"), sz(4), b("h_vec"), sz(4), b("c"), sz(4), b("rho"), b("mu"), b("label"),
                                   sz(len(b("label"))), b("commitment"), b("proof_r"), b("proof_x"), sz(nr.value), b("proof_l"),
                                   sz(nl.value), b("proof_n"), sz(nn.value))
    assert rc == 1


def test_c_generic_circuit_matches_bigint_oracle(oracle_c):
    """The C restatement's generic ArithmeticCircuit::{prove, verify} (circuit.rs:154-556) against the Python big-integer one: the
    reference's `ac_works` statement (tests.rs:45-136) must verify; so must k = 2 with all partition types and the f_m path at
    dim_nv = 1; f_l + f_m is rejected by BOTH restatements (the shape the reference's coefficient helpers do not complete)."""
    import circuit_cases
    from test_circuit_emul import _oracle_circuit
    import bppp_oracle as O
    for name, want in (("ac_works", 1), ("mixed_k2", 1), ("fm_nv1", 1), ("fl_fm", 0)):
        case = circuit_cases.make(name, B=1)
        pr = case["proofs"][0].tobytes()
        assert circuit_cases.oracle_verify(case, case["commitments"][0].tobytes(), pr) == want
        R = case["rounds"]
        P = lambda i: O.pt_from_xy64(pr[64 * i:64 * i + 64])
        off = 64 * (4 + 2 * R)
        sc_at = lambda o: int.from_bytes(pr[o:o + 32], "big")
        proof = O.CircuitProof(c_l=P(0), c_r=P(1), c_o=P(2), c_s=P(3), r=[P(4 + i) for i in range(R)], x=[P(4 + R + i) for i in range(R)],
                               l=[sc_at(off + 32 * i) for i in range(case["pl"])],
                               n=[sc_at(off + 32 * case["pl"] + 32 * i) for i in range(case["pn"])])
        v = [O.pt_from_xy64(case["commitments"][0, i].tobytes()) for i in range(case["k"])]
        assert int(_oracle_circuit(case).verify(v, O.Transcript(case["label"]), proof)) == want
