"""Consumer of tests/golden/ref_*.json -- fixtures written by facade/src/bin/gen_fixtures.rs from the REAL reference crate (bp-pp
0.1.1 on k256 0.13.3 / merlin 3.0.0).  `check_document` is everything the CPU tier demands of such a file; the GPU tier adds the
HIP path (tests/test_gpu_ref_fixtures.py).  `oracle_made_document` writes the same format from this repository's own oracle: it
exists to keep the consumer itself tested while no reference-made file is available, and pins nothing."""
import json
import os

import numpy as np

import bppp_oracle as O
from bp_pp_amd import wire
from bp_pp_amd.transcript import Transcript

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def reference_fixture_paths():
    return sorted(os.path.join(GOLD, f) for f in os.listdir(GOLD) if f.startswith("ref_") and f.endswith(".json"))


def _protocol(doc):
    gens = bytes.fromhex(doc["generators"])
    pts = [O.pt_from_xy64(gens[64 * i:64 * i + 64]) for i in range(49)]
    return gens, O.U64RangeProofProtocol(pts[0], pts[1:17], pts[17:49])


def _transcript_from(state: bytes) -> "O.Transcript":
    t = O.Transcript(b"x")
    t.strobe.state = bytearray(state[:200])
    t.strobe.pos, t.strobe.pos_begin, t.strobe.cur_flags = state[200], state[201], state[202]
    return t


def _ser(t) -> bytes:
    return bytes(t.strobe.state) + bytes([t.strobe.pos, t.strobe.pos_begin, t.strobe.cur_flags])


def check_document(doc, oracle_c=None):
    label = bytes.fromhex(doc["label"])
    gens, proto = _protocol(doc)
    assert len(gens) == 49 * 64 and all(O.on_curve(p) for p in [proto.g] + proto.g_vec + proto.h_vec)
    if "identity_to_bytes" in doc:      # GroupEncoding::to_bytes of the identity: what transcript.rs:7 would hash
        assert bytes.fromhex(doc["identity_to_bytes"]) == O.pt_to_bytes(None) == bytes(33)
    if "identity_json" in doc:          # serde of the identity AffinePoint
        assert doc["identity_json"] == wire.point_to_hex(bytes(33))
    if "merlin_kat" in doc:
        t = Transcript(b"test protocol")
        t.append_message(b"some label", b"some data")
        assert t.challenge_bytes(b"challenge", 32).hex() == doc["merlin_kat"]["challenge"]
    for c in doc["cases"]:
        x, s = int(c["x"]), O.sc_from_bytes(bytes.fromhex(c["s"]))
        ctx = bytes.fromhex(c.get("context", ""))
        # merlin: the state the caller held before prove / verify, rebuilt with this repository's host transcript
        t = Transcript(label)
        if ctx:
            t.append_message(b"ctx", ctx)
        assert t.state == bytes.fromhex(c["state_before"])
        # k256 Scalar::generate_biased: 64 RNG bytes, big-endian, reduced mod n -- per draw, 52 draws per proof
        raw, rnd = bytes.fromhex(c["rng_bytes"]), bytes.fromhex(c["rnd"])
        assert c["rng_calls"] == [64] * 52 and len(raw) == 52 * 64 and len(rnd) == 52 * 32
        scalars = [O.wide_reduce(raw[64 * i:64 * i + 64]) for i in range(52)]
        assert b"".join(O.sc_to_bytes(v) for v in scalars) == rnd
        if "s_rng_bytes" in c:
            assert O.wide_reduce(bytes.fromhex(c["s_rng_bytes"])) == s
        # commit_value, then the prover replayed on the recorded draws: byte-identical proof, identical transcript afterwards
        V = proto.commit_value(x, s)
        assert O.pt_to_xy64(V) == bytes.fromhex(c["commitment"])
        tp = _transcript_from(bytes.fromhex(c["state_before"]))
        proof = proto.prove(x, s, tp, O.ScalarRng(scalars))
        assert O.u64_proof_to_bytes(proof) == bytes.fromhex(c["proof"])
        assert _ser(tp) == bytes.fromhex(c["state_after_prove"])
        # verify: accept bit and the advanced transcript
        tv = _transcript_from(bytes.fromhex(c["state_before"]))
        assert proto.verify(V, O.u64_proof_from_bytes(bytes.fromhex(c["proof"])), tv) == c["accept"]
        assert _ser(tv) == bytes.fromhex(c["state_after_verify"])
        if oracle_c is not None and not ctx:
            assert oracle_c.u64_verify(gens, label, bytes.fromhex(c["commitment"]), bytes.fromhex(c["proof"])) == (1 if c["accept"] else 0)
            pb, vb = oracle_c.u64_prove(gens, label, x, bytes.fromhex(c["s"]), rnd)
            assert pb == bytes.fromhex(c["proof"]) and vb == bytes.fromhex(c["commitment"])
        # serde: the JSON the reference prints (tests.rs:37-38) against this repository's wire module, case-sensitively
        if "proof_json" in c:
            assert json.loads(wire.sec1_to_json(wire.abi_to_sec1(bytes.fromhex(c["proof"])))) == c["proof_json"]
            assert wire.json_to_sec1(json.dumps(c["proof_json"])) == wire.abi_to_sec1(bytes.fromhex(c["proof"]))
        if "commitment_json" in c:
            assert wire.point_to_hex(wire.compress_point(bytes.fromhex(c["commitment"]))) == c["commitment_json"]
    for c in doc.get("negative_cases", []):
        tv = _transcript_from(bytes.fromhex(c["state_before"]))
        ok = proto.verify(O.pt_from_xy64(bytes.fromhex(c["commitment"])), O.u64_proof_from_bytes(bytes.fromhex(c["proof"])), tv)
        assert ok == c["accept"] and _ser(tv) == bytes.fromhex(c["state_after_verify"])
    return len(doc["cases"])


def oracle_made_document(n_cases: int = 3):
    """The ref_*.json format written by THIS repository's oracle (test of the consumer; not a pin)."""
    import hashlib
    label = b"u64 range proof"
    g, gv, hv = O.synth_generators()
    proto = O.U64RangeProofProtocol(g, gv, hv)
    gens = b"".join(O.pt_to_xy64(p) for p in [g] + list(gv) + list(hv))
    cases, negs = [], []
    for j in range(n_cases):
        x = [123456, 0, 2**64 - 1, 77][j % 4]
        s_raw = hashlib.shake_256(b"s" + bytes([j])).digest(64)
        s = O.wide_reduce(s_raw)
        raw = hashlib.shake_256(b"rng" + bytes([j])).digest(52 * 64)
        scalars = [O.wide_reduce(raw[64 * i:64 * i + 64]) for i in range(52)]
        ctx = b"tx-%d" % j if j % 2 else b""
        t0 = O.Transcript(label)
        if ctx:
            t0.append_message(b"ctx", ctx)
        V = proto.commit_value(x, s)
        tp, tv = t0.clone(), t0.clone()
        proof = proto.prove(x, s, tp, O.ScalarRng(scalars))
        assert proto.verify(V, proof, tv)
        pb = O.u64_proof_to_bytes(proof)
        cases.append({"x": str(x), "s": O.sc_to_bytes(s).hex(), "s_rng_bytes": s_raw.hex(), "context": ctx.hex(), "rng_bytes": raw.hex(),
                      "rng_calls": [64] * 52, "rnd": b"".join(O.sc_to_bytes(v) for v in scalars).hex(), "commitment": O.pt_to_xy64(V).hex(),
                      "proof": pb.hex(), "proof_json": json.loads(wire.sec1_to_json(wire.abi_to_sec1(pb))),
                      "commitment_json": wire.point_to_hex(wire.compress_point(O.pt_to_xy64(V))), "state_before": _ser(t0).hex(),
                      "state_after_prove": _ser(tp).hex(), "state_after_verify": _ser(tv).hex(), "accept": True})
        bad = bytearray(pb); bad[927] ^= 1
        tb = t0.clone()
        ok = proto.verify(V, O.u64_proof_from_bytes(bytes(bad)), tb)
        negs.append({"commitment": O.pt_to_xy64(V).hex(), "proof": bytes(bad).hex(), "state_before": _ser(t0).hex(),
                     "state_after_verify": _ser(tb).hex(), "accept": ok})
    return {"source": "THIS repository's oracle (consumer self-test; pins nothing)", "label": label.hex(), "generators": gens.hex(),
            "cases": cases, "negative_cases": negs, "identity_to_bytes": bytes(33).hex(), "identity_json": "00",
            "merlin_kat": {"challenge": "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"}}
