"""Pins the Python big-int oracle: public known answers for the third-party layers (secp256k1 = k256's curve,
Merlin/STROBE/Keccak = merlin 3.0.0), the reference's own completeness tests (src/tests.rs: u64_proof_works, ac_works,
wnla_works -- honest prove => verify true), negative cases the reference lacks, and the committed golden fixtures."""
import hashlib
import json
import os

import pytest

import bppp_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_secp256k1_known_answers():
    assert O.on_curve(O.G)
    assert O.pt_mul(O.G, 2) == (0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5,
                                0x1AE168FEA63DC339A3C58419466CEAEEF7F632653266D0E1236431A950CFE52A)
    assert O.pt_mul(O.G, 3) == (0xF9308A019258C31049344F85F89D5229B531C845836F99B08601F113BCE036F9,
                                0x388F7B0F632DE8140FE337E62A37F3566500A99934C2231B6CB9FD7584B8E672)
    assert O.pt_mul_affine_only(O.G, O.N) is None
    assert O.pt_mul(O.G, O.N - 1) == O.pt_neg(O.G)
    assert O.pt_mul(O.G, O.LAMBDA) == (O.BETA * O.GX % O.P, O.GY)          # GLV endomorphism
    assert pow(O.BETA, 3, O.P) == 1 and pow(O.LAMBDA, 3, O.N) == 1
    # SEC1 compressed encoding of G, identity -> 33 zero bytes
    assert O.pt_to_bytes(O.G).hex() == "0279be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798"
    assert O.pt_to_bytes(None) == bytes(33)
    assert O.pt_from_bytes(O.pt_to_bytes(O.pt_mul(O.G, 7))) == O.pt_mul(O.G, 7)


def test_jacobian_accelerator_matches_affine_law():
    for k in [1, 2, 3, 0xDEADBEEF, O.N - 2, 2**255 + 12345]:
        assert O.pt_mul(O.G, k) == O.pt_mul_affine_only(O.G, k)
    p = O.pt_mul(O.G, 99)
    assert O.pt_add(p, O.pt_neg(p)) is None
    assert O.pt_add(p, None) == p and O.pt_add(None, p) == p
    assert O.pt_add(p, p) == O.pt_mul(O.G, 198)


def _sha3_256_via_keccak(msg: bytes) -> bytes:
    st, rate = bytearray(200), 136
    m = bytearray(msg) + b"\x06"
    while len(m) % rate:
        m.append(0)
    m[-1] |= 0x80
    for off in range(0, len(m), rate):
        for i in range(rate):
            st[i] ^= m[off + i]
        O.keccak_f1600_bytes(st)
    return bytes(st[:32])


def test_keccak_f1600_against_hashlib():
    for msg in [b"", b"abc", b"x" * 135, b"y" * 136, b"z" * 500]:
        assert _sha3_256_via_keccak(msg) == hashlib.sha3_256(msg).digest()


def test_merlin_known_answer():
    # upstream merlin known-answer (also used by independent Merlin ports)
    t = O.Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_util_semantics():
    # util.rs:7-22 reduce = even/odd split; :28-44 weight exponent starts at 1; :24-26 zero extension
    assert O.reduce([1, 2, 3, 4, 5]) == ([1, 3, 5], [2, 4])
    assert O.weight_vector_mul([1, 1], [1, 1], 3) == 3 + 9
    assert O.vector_add([1, 2, 3], [10]) == [11, 2, 3]
    assert O.vector_sub([1], [0, 5]) == [1, O.N - 5]
    assert O.e_vec(2, 4) == [1, 2, 4, 8]
    assert O.minus(1) == O.N - 1
    assert O.vector_tensor_mul([1, 2], [3, 4]) == [3, 6, 4, 8]
    d = O.diag_inv(2, 2)
    assert d[0][0] == pow(2, -1, O.N) and d[1][1] == pow(4, -1, O.N) and d[0][1] == 0
    assert O.u64_to_hex(0x123) == [3, 2, 1] + [0] * 13
    assert O.u64_to_hex_mapped(0x1123) == [12, 2, 1, 1] + [0] * 12


def test_u64_proof_works_golden():
    """src/tests.rs:13-42 (x = 123456) on the seeded generators + every committed golden case."""
    with open(os.path.join(GOLD, "u64_golden.json")) as f:
        gold = json.load(f)
    gens = bytes.fromhex(gold["generators"])
    pts = [O.pt_from_xy64(gens[64 * i:64 * i + 64]) for i in range(49)]
    assert pts == [O.pt_mul(O.G, int.from_bytes(bytes.fromhex(gold["generator_dlogs"])[32 * i:32 * i + 32], "big")) for i in range(49)]
    pub = O.U64RangeProofProtocol(pts[0], pts[1:17], pts[17:49])
    label = bytes.fromhex(gold["label"])
    case = gold["cases"][2]
    assert case["x"] == 123456
    s = int(case["s"], 16)
    rnd = [int.from_bytes(bytes.fromhex(case["rnd"])[32 * i:32 * i + 32], "big") for i in range(52)]
    rng = O.ScalarRng(rnd)
    proof = pub.prove(case["x"], s, O.Transcript(label), rng)
    assert rng.drawn == O.N_RNG_DRAWS_U64                       # 52 generate_biased draws
    assert O.u64_proof_to_bytes(proof).hex() == case["proof"]
    assert O.pt_to_xy64(pub.commit_value(case["x"], s)).hex() == case["commitment"]
    for c in gold["cases"]:
        assert pub.verify(O.pt_from_xy64(bytes.fromhex(c["commitment"])), O.u64_proof_from_bytes(bytes.fromhex(c["proof"])),
                          O.Transcript(label))
    for c in gold["negative_cases"]:
        if c["status"] == 0:
            assert not pub.verify(O.pt_from_xy64(bytes.fromhex(c["commitment"])),
                                  O.u64_proof_from_bytes(bytes.fromhex(c["proof"])), O.Transcript(label)), c["what"]
        else:
            with pytest.raises(ValueError):
                O.u64_proof_from_bytes(bytes.fromhex(c["proof"]))


def test_ac_works():
    """src/tests.rs:45-136: x + y = r, x * y = z with dim_nm=1, dim_no=2, dim_nv=2, k=1."""
    x, y, r, z = 3, 5, 8, 15
    W_m = [[0, 0, 1, 0]]
    a_m = [0]
    W_l = [[0, 1, 0, 0], [0, O.N - 1, 1, 0]]
    a_l = [O.minus(r), O.minus(z)]
    g = O.pt_mul(O.G, 11)
    g_vec = [O.pt_mul(O.G, 12)]
    h_vec = [O.pt_mul(O.G, 100 + i) for i in range(16)]
    circuit = O.ArithmeticCircuit(
        dim_nm=1, dim_no=2, k=1, dim_nl=2, dim_nv=2, dim_nw=4, g=g, g_vec=g_vec[:1], h_vec=h_vec[:11], W_m=W_m, W_l=W_l,
        a_m=a_m, a_l=a_l, f_l=True, f_m=False, g_vec_=g_vec[1:], h_vec_=h_vec[11:],
        partition=lambda typ, index: index if typ == O.LL else None)
    wit = O.CircuitWitness(v=[[x, y]], s_v=[777], w_l=[x], w_r=[y], w_o=[z, r])
    v = [circuit.commit(wit.v[0], wit.s_v[0])]
    rnd = [int.from_bytes(hashlib.sha256(b"ac" + bytes([i])).digest(), "big") % O.N for i in range(64)]
    proof = circuit.prove(v, wit, O.Transcript(b"circuit test"), O.ScalarRng(rnd))
    assert circuit.verify(v, O.Transcript(b"circuit test"), proof)
    proof.l[0] = (proof.l[0] + 1) % O.N
    assert not circuit.verify(v, O.Transcript(b"circuit test"), proof)


def test_wnla_works_golden():
    """src/tests.rs:139-171: N = 4, l = [1,2,3,4], n = [8,7,6,5]."""
    with open(os.path.join(GOLD, "wnla_golden.json")) as f:
        w = json.load(f)
    pts = lambda h: [O.pt_from_xy64(bytes.fromhex(h)[64 * i:64 * i + 64]) for i in range(len(h) // 128)]
    scs = lambda h: [int.from_bytes(bytes.fromhex(h)[32 * i:32 * i + 32], "big") for i in range(len(h) // 64)]
    wn = O.WeightNormLinearArgument(g=pts(w["g"])[0], g_vec=pts(w["g_vec"]), h_vec=pts(w["h_vec"]), c=scs(w["c"]),
                                    rho=scs(w["rho"])[0], mu=scs(w["mu"])[0])
    com = wn.commit(w["l"], w["n"])
    assert O.pt_to_xy64(com).hex() == w["commitment"]
    proof = wn.prove(com, O.Transcript(bytes.fromhex(w["label"])), list(w["l"]), list(w["n"]))
    assert b"".join(map(O.pt_to_xy64, proof.r)).hex() == w["proof_r"]
    assert b"".join(map(O.pt_to_xy64, proof.x)).hex() == w["proof_x"]
    assert wn.verify(com, O.Transcript(bytes.fromhex(w["label"])), proof)
    bad = O.WnlaProof(r=proof.r[:-1], x=proof.x, l=proof.l, n=proof.n)
    assert not wn.verify(com, O.Transcript(bytes.fromhex(w["label"])), bad)      # wnla.rs:76-78 length check
