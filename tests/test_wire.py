"""Wire format, CPU tier: bp_pp_amd/wire.py against the oracle's SEC1 encoding, the JSON mirror of
`SerializableProof`, and the device decompression code (compiled for the host) against both."""
import json
import os

import numpy as np
import pytest

import bppp_oracle as O
from bp_pp_amd import wire
from emul.build import load

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "u64_golden.json")) as f:
        return json.load(f)


def test_point_compression_matches_k256_semantics():
    for k in [1, 2, 3, 0xDEADBEEF, O.N - 1]:
        p = O.pt_mul(O.G, k)
        assert wire.compress_point(O.pt_to_xy64(p)) == O.pt_to_bytes(p)
        assert wire.decompress_point(O.pt_to_bytes(p)) == O.pt_to_xy64(p)
    assert wire.compress_point(bytes(64)) == bytes(33) and wire.decompress_point(bytes(33)) == bytes(64)
    with pytest.raises(ValueError):
        wire.decompress_point(b"\x04" + bytes(32))
    with pytest.raises(ValueError):
        wire.decompress_point(b"\x02" + (5).to_bytes(32, "big"))      # x = 5: x^3 + 7 = 132 is a non-residue mod p
    with pytest.raises(ValueError):
        wire.decompress_point(b"\x02" + O.P.to_bytes(32, "big"))


def test_proof_forms_round_trip(gold):
    for c in gold["cases"]:
        abi = bytes.fromhex(c["proof"])
        sec1 = wire.abi_to_sec1(abi)
        assert len(sec1) == 525 and wire.sec1_to_abi(sec1) == abi
        text = wire.sec1_to_json(sec1)
        doc = json.loads(text)
        assert set(doc) == {"circuit_proof", "r"} and set(doc["circuit_proof"]) == {"c_l", "c_r", "c_o", "c_s", "r", "x", "l", "n"}
        assert wire.json_to_sec1(text) == sec1 and wire.json_to_sec1(text.lower()) == sec1
        proof = O.u64_proof_from_bytes(abi)                              # field order = the reference's struct order
        assert bytes.fromhex(doc["circuit_proof"]["c_s"]) == O.pt_to_bytes(proof.circuit_proof.c_s)
        assert bytes.fromhex(doc["circuit_proof"]["x"][3]) == O.pt_to_bytes(proof.circuit_proof.x[3])
        assert bytes.fromhex(doc["r"]) == O.pt_to_bytes(proof.r)
        assert bytes.fromhex(doc["circuit_proof"]["n"][0]) == O.sc_to_bytes(proof.circuit_proof.n[0])


def test_device_decompression_code(gold):
    L = load()
    cases = gold["cases"]
    PRIME = 2**256 - 2**32 - 977
    c33 = [wire.compress_point(bytes.fromhex(c["commitment"])) for c in cases]
    p525 = [wire.abi_to_sec1(bytes.fromhex(c["proof"])) for c in cases]
    exp_c = [bytes.fromhex(c["commitment"]) for c in cases]
    exp_p = [bytes.fromhex(c["proof"]) for c in cases]
    # encodings k256's from_bytes rejects; the last three have x = 0 mod p and must NOT come out as the identity (0, 0)
    rejects = [b"\x04" + p525[0][1:33],                        # bad tag on c_l
               b"\x02" + (5).to_bytes(32, "big"),              # x with no square root
               b"\x02" + bytes(32),                            # x = 0: 7 is a non-residue, so 02||0 is off the curve
               b"\x05" + bytes(32),                            # bad tag over x = 0
               b"\x02" + PRIME.to_bytes(32, "big"),            # x = p: out of range
               b"\x03" + PRIME.to_bytes(32, "big")]
    first_bad = len(c33)
    for j, enc in enumerate(rejects):
        bad = bytearray(p525[0]); bad[33 * (j % 13):33 * (j % 13) + 33] = enc
        c33.append(c33[0]); p525.append(bytes(bad)); exp_c.append(exp_c[0]); exp_p.append(None)
    idp = bytearray(p525[0]); idp[132:165] = bytes(33)                    # r[0] = identity
    c33.append(bytes(33)); p525.append(bytes(idp)); exp_c.append(bytes(64)); exp_p.append(wire.sec1_to_abi(bytes(idp)))
    flip = bytearray(p525[1]); flip[0] ^= 1                               # the other square root of c_l
    c33.append(c33[1]); p525.append(bytes(flip)); exp_c.append(exp_c[1]); exp_p.append(wire.sec1_to_abi(bytes(flip)))
    c33.append(rejects[2]); p525.append(p525[0]); exp_c.append(None); exp_p.append(exp_p[0])   # an undecodable COMMITMENT
    n = len(c33)
    C33 = np.frombuffer(b"".join(c33), dtype=np.uint8).copy()
    P525 = np.frombuffer(b"".join(p525), dtype=np.uint8).copy()
    C64, P928 = np.zeros(n * 64, np.uint8), np.zeros(n * 928, np.uint8)
    L.emul_sec1_expand(n, C33.ctypes.data, P525.ctypes.data, C64.ctypes.data, P928.ctypes.data)
    for i in range(n):
        if exp_c[i] is not None:
            assert bytes(C64[64 * i:64 * i + 64]) == exp_c[i]
        got = bytes(P928[928 * i:928 * i + 928])
        if exp_p[i] is not None:
            assert got == exp_p[i]
    # undecodable points come out as (1, 0): off the curve and never the identity, so the verifier flags them
    sentinel = (1).to_bytes(32, "big") + bytes(32)
    assert not O.on_curve((1, 0))
    for j in range(len(rejects)):
        i = first_bad + j
        assert bytes(P928[928 * i + 64 * (j % 13):928 * i + 64 * (j % 13) + 64]) == sentinel, j
    assert bytes(C64[64 * (n - 1):64 * n]) == sentinel


def test_device_compression_equals_the_wire_module(gold):
    """The prover's output in the wire format (bppp_u64_prove_batch_sec1: sec1_compress_lane on the device code): equal to
    bp_pp_amd.wire's host-side compression for every golden proof, for a proof with identity points, and back again through the
    expansion the SEC1 verifier uses."""
    from emul.build import load
    L = load()
    abi = [bytes.fromhex(c["proof"]) for c in gold["cases"]]
    com = [bytes.fromhex(c["commitment"]) for c in gold["cases"]]
    idp = bytearray(abi[0]); idp[64 * 5:64 * 6] = bytes(64); idp[64 * 12:64 * 13] = bytes(64)
    abi.append(bytes(idp)); com.append(bytes(64))
    n = len(abi)
    C64 = np.frombuffer(b"".join(com), np.uint8).copy()
    P928 = np.frombuffer(b"".join(abi), np.uint8).copy()
    C33, P525 = np.zeros(n * 33, np.uint8), np.zeros(n * 525, np.uint8)
    L.emul_sec1_compress(n, C64.ctypes.data, P928.ctypes.data, C33.ctypes.data, P525.ctypes.data)
    for i in range(n):
        assert bytes(P525[525 * i:525 * i + 525]) == wire.abi_to_sec1(abi[i]), i
        assert bytes(C33[33 * i:33 * i + 33]) == wire.compress_point(com[i]), i
    B64, B928 = np.zeros(n * 64, np.uint8), np.zeros(n * 928, np.uint8)
    L.emul_sec1_expand(n, C33.ctypes.data, P525.ctypes.data, B64.ctypes.data, B928.ctypes.data)
    assert (B64 == C64).all() and (B928 == P928).all()


def test_identity_in_json_is_00(gold):
    """serde writes the identity AffinePoint as the one SEC1 byte 0x00 ("00"), not as 33 zero bytes."""
    abi = bytearray(bytes.fromhex(gold["cases"][0]["proof"]))
    abi[64 * 5:64 * 6] = bytes(64)                                        # r[1] = identity
    sec1 = wire.abi_to_sec1(bytes(abi))
    doc = json.loads(wire.sec1_to_json(sec1))
    assert doc["circuit_proof"]["r"][1] == "00" and len(doc["circuit_proof"]["r"][0]) == 66
    assert wire.json_to_sec1(json.dumps(doc)) == sec1
    d2 = wire.circuit_proof_to_doc(bytes(abi), 4, 2, 1, reciprocal=True)
    assert d2["circuit_proof"]["r"][1] == "00" and wire.doc_to_circuit_proof(d2) == bytes(abi)
    with pytest.raises(ValueError):
        wire.hex_to_point("0000")


def test_generic_proof_documents_round_trip():
    """circuit / reciprocal / wnla proofs of any shape <-> the serde-shaped documents, on oracle-made proofs."""
    import circuit_cases
    import recip_cases
    import wnla_cases
    from bp_pp_amd import wire
    c = circuit_cases.make("mixed_k2", 1)
    doc = wire.circuit_proof_to_doc(c["proofs"][0].tobytes(), c["rounds"], c["pl"], c["pn"])
    assert set(doc) == {"c_l", "c_r", "c_o", "c_s", "r", "x", "l", "n"} and len(doc["r"]) == len(doc["x"]) == c["rounds"]
    assert all(len(h) == 66 for h in doc["r"] + doc["x"] + [doc["c_l"]]) and all(len(h) == 64 for h in doc["l"] + doc["n"])
    assert wire.doc_to_circuit_proof(json.loads(json.dumps(doc))) == c["proofs"][0].tobytes()
    r = recip_cases.make(8, 4, 1)
    doc = wire.circuit_proof_to_doc(r["proofs"][0].tobytes(), r["rounds"], r["nl"], r["nn"], reciprocal=True)
    assert set(doc) == {"circuit_proof", "r"} and wire.doc_to_circuit_proof(doc) == r["proofs"][0].tobytes()
    # the u64 shape agrees with the dedicated 928-byte converters
    u = recip_cases.make(16, 16, 1)
    assert json.loads(wire.sec1_to_json(wire.abi_to_sec1(u["proofs"][0].tobytes()))) == \
        wire.circuit_proof_to_doc(u["proofs"][0].tobytes(), u["rounds"], u["nl"], u["nn"], reciprocal=True)
    w = wnla_cases.make(4, 4, 1)
    d = wire.wnla_proof_to_doc(w["proof_r"][0].tobytes(), w["proof_x"][0].tobytes(), w["proof_l"][0].tobytes(), w["proof_n"][0].tobytes())
    assert len(d["r"]) == w["rounds"] and len(d["l"]) == w["nl"] and len(d["n"]) == w["nn"]
    with pytest.raises(ValueError):
        wire.circuit_proof_to_doc(c["proofs"][0].tobytes()[:-1], c["rounds"], c["pl"], c["pn"])


def test_compress_expand_round_trip_on_random_points():
    """sec1_compress_lane then sec1_expand_lane (the device code of both wire directions, compiled for the host) is the identity on
    well-formed 64-byte points -- random multiples of G, both parities of y, the identity -- and agrees with the wire module."""
    import random
    L = load()
    rnd = random.Random(77)
    n = 24
    pts = [O.pt_mul(O.G, rnd.getrandbits(256) % O.N) for _ in range(14 * n - 3)] + [None, None, None]
    rnd.shuffle(pts)
    enc = [O.pt_to_xy64(p) for p in pts]
    C64 = np.frombuffer(b"".join(enc[:n]), np.uint8).copy()
    scal = [rnd.getrandbits(256).to_bytes(32, "big") for _ in range(3 * n)]
    P928 = np.frombuffer(b"".join(b"".join(enc[n + 13 * i:n + 13 * i + 13]) + b"".join(scal[3 * i:3 * i + 3]) for i in range(n)), np.uint8).copy()
    C33, P525 = np.zeros(n * 33, np.uint8), np.zeros(n * 525, np.uint8)
    L.emul_sec1_compress(n, C64.ctypes.data, P928.ctypes.data, C33.ctypes.data, P525.ctypes.data)
    for i in range(n):
        assert bytes(C33[33 * i:33 * i + 33]) == wire.compress_point(enc[i])
        assert bytes(P525[525 * i:525 * i + 525]) == wire.abi_to_sec1(bytes(P928[928 * i:928 * i + 928]))
    B64, B928 = np.zeros(n * 64, np.uint8), np.zeros(n * 928, np.uint8)
    L.emul_sec1_expand(n, C33.ctypes.data, P525.ctypes.data, B64.ctypes.data, B928.ctypes.data)
    assert (B64 == C64).all() and (B928 == P928).all()
    assert {int(C33[33 * i]) for i in range(n)} | {int(P525[525 * i + 33 * j]) for i in range(n) for j in range(13)} >= {0, 2, 3}
