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


# ---- the `generic` documents (gen_fixtures.rs generic: the crate's circuit.rs and wnla.rs on the statements of tests/golden/statements_generic.json)
def load_statements():
    with open(os.path.join(GOLD, "statements_generic.json")) as f:
        st = json.load(f)
    return {c["name"]: c for c in st["circuits"]}, {w["name"]: w for w in st["wnla"]}


def _circuit_call_args(st, cdoc):
    """ctypes arguments of bppp_oracle_circuit_{prove,verify} for statement `st` over the generators recorded in `cdoc`."""
    import ctypes as C
    sz = C.c_size_t
    nm, no, nv, k = st["dim_nm"], st["dim_no"], st["dim_nv"], st["k"]
    dims = (sz * 6)(nm, no, k, nv * k, nv, 2 * nm + no)
    hx = lambda rows: b"".join(bytes.fromhex(x) for row in rows for x in row)
    part = {t: np.array(st["partition"][t], np.int32) for t in ("LO", "LL", "LR", "NO")}
    gv_, hv_ = bytes.fromhex(cdoc["g_vec_"]), bytes.fromhex(cdoc["h_vec_"])
    head = (bytes.fromhex(cdoc["g"]), bytes.fromhex(cdoc["g_vec"]), bytes.fromhex(cdoc["h_vec"]), gv_, sz(len(gv_) // 64), hv_, sz(len(hv_) // 64), dims,
            int(st["f_l"]), int(st["f_m"]), hx(st["W_m"]), hx(st["W_l"]), b"".join(bytes.fromhex(x) for x in st["a_m"]),
            b"".join(bytes.fromhex(x) for x in st["a_l"]), *(part[t].ctypes.data_as(C.c_void_p) for t in ("LO", "LL", "LR", "NO")))
    return head, part          # (part is returned to keep the arrays alive)


def check_generic_document(doc, oracle_c):
    """Everything the CPU tier demands of a reference-made generic document: per circuit instance the oracle's prover, fed the recorded
    generators, blindings and draws, must emit the recorded proof byte for byte, and the oracle's verifier must return the verdict the
    REFERENCE's verifier returned (`accept`, true or false); the same for the WNLA instances.  Returns the number of instances checked."""
    import ctypes as C
    L, sz = oracle_c.lib(), C.c_size_t
    circuits, wnlas = load_statements()
    checked = 0
    for cdoc in doc.get("circuits", []):
        st = circuits[cdoc["name"]]
        label = bytes.fromhex(cdoc["label"])
        head, keep = _circuit_call_args(st, cdoc)
        hs = lambda xs: b"".join(bytes.fromhex(x) for x in xs)
        for inst in cdoc["instances"]:
            rnd = bytes.fromhex(inst["rnd"])
            assert inst["rng_calls"] == [64] * (len(rnd) // 32)
            raw = bytes.fromhex(inst["rng_bytes"])
            assert b"".join(O.sc_to_bytes(O.wide_reduce(raw[64 * i:64 * i + 64])) for i in range(len(rnd) // 32)) == rnd
            com = C.create_string_buffer(64 * st["k"])
            pbuf = C.create_string_buffer(64 * (4 + 2 * 16) + 32 * 16)
            rounds, pl, pn = sz(0), sz(0), sz(0)
            rc = L.bppp_oracle_circuit_prove(*head, label, sz(len(label)), b"".join(hs(row) for row in st["v"]), bytes.fromhex(inst["s_v"]), hs(st["w_l"]),
                                             hs(st["w_r"]), hs(st["w_o"]), rnd, sz(len(rnd) // 32), com, pbuf, C.byref(rounds), C.byref(pl), C.byref(pn))
            assert rc == 0, (cdoc["name"], rc)
            assert (rounds.value, pl.value, pn.value) == (inst["rounds"], inst["nl"], inst["nn"]), cdoc["name"]
            nbytes = 64 * (4 + 2 * rounds.value) + 32 * (pl.value + pn.value)
            assert com.raw == bytes.fromhex(inst["commitments"]), cdoc["name"]
            assert pbuf.raw[:nbytes] == bytes.fromhex(inst["proof"]), cdoc["name"]
            v = L.bppp_oracle_circuit_verify(*head, label, sz(len(label)), bytes.fromhex(inst["commitments"]), bytes.fromhex(inst["proof"]),
                                             sz(inst["rounds"]), sz(inst["nl"]), sz(inst["nn"]))
            assert (v == 1) == bool(inst["accept"]), (cdoc["name"], v, inst["accept"])
            checked += 1
        del keep
    for w in doc.get("wnla", []):
        st = wnlas[w["name"]]
        label = bytes.fromhex(w["label"])
        ng, nh = w["ng"], w["nh"]
        b = lambda k: bytes.fromhex(w[k])
        assert b("l") == b"".join(bytes.fromhex(x) for x in st["l"]) and b("n") == b"".join(bytes.fromhex(x) for x in st["n"])
        com = C.create_string_buffer(64)
        assert L.bppp_oracle_wnla_commit(b("g"), b("g_vec"), sz(ng), b("h_vec"), sz(nh), b("c"), sz(nh), b("rho"), b("mu"), b("l"), sz(nh), b("n"), sz(ng), com) == 0
        assert com.raw == b("commitment"), w["name"]
        r_out, x_out = C.create_string_buffer(64 * 16), C.create_string_buffer(64 * 16)
        l_out, n_out = C.create_string_buffer(32 * 8), C.create_string_buffer(32 * 8)
        nr, nl, nn = sz(0), sz(0), sz(0)
        assert L.bppp_oracle_wnla_prove(b("g"), b("g_vec"), sz(ng), b("h_vec"), sz(nh), b("c"), sz(nh), b("rho"), b("mu"), label, sz(len(label)), com.raw,
                                        b("l"), sz(nh), b("n"), sz(ng), r_out, x_out, C.byref(nr), l_out, C.byref(nl), n_out, C.byref(nn)) == 0
        assert r_out.raw[:64 * nr.value] == b("proof_r") and x_out.raw[:64 * nr.value] == b("proof_x"), w["name"]
        assert l_out.raw[:32 * nl.value] == b("proof_l") and n_out.raw[:32 * nn.value] == b("proof_n"), w["name"]
        v = L.bppp_oracle_wnla_verify(b("g"), b("g_vec"), sz(ng), b("h_vec"), sz(nh), b("c"), sz(nh), b("rho"), b("mu"), label, sz(len(label)), b("commitment"),
                                      b("proof_r"), b("proof_x"), sz(len(b("proof_r")) // 64), b("proof_l"), sz(len(b("proof_l")) // 32), b("proof_n"),
                                      sz(len(b("proof_n")) // 32))
        assert (v == 1) == bool(w["accept"]), (w["name"], v)
        checked += 1
    return checked


def oracle_made_generic_document(oracle_c):
    """The generic document format written by THIS repository's oracle on the same statements (test of the consumer; pins nothing).
    `accept` is the oracle's own verdict -- false for the f_l-and-f_m shape, which is exactly the claim a reference-made file settles."""
    import ctypes as C
    import hashlib
    L, sz = oracle_c.lib(), C.c_size_t
    circuits, wnlas = load_statements()
    sc = lambda tag, *idx: O.wide_reduce(hashlib.shake_256(b"oracle-made-generic" + tag + b"".join(int(i).to_bytes(4, "little") for i in idx)).digest(64))
    pt = lambda tag, i: oracle_c.point_mul(None, O.sc_to_bytes(sc(b"gen" + tag, i)))
    p2 = lambda n: 1 << max(0, (n - 1).bit_length())
    out_c, out_w = [], []
    for name, st in circuits.items():
        nm, nv, k = st["dim_nm"], st["dim_nv"], st["k"]
        ng, nh = p2(nm), p2(nv + 9)
        gall, hall = [pt(name.encode() + b"g", i) for i in range(ng)], [pt(name.encode() + b"h", i) for i in range(nh)]
        cdoc = {"name": name, "label": st["label"], "g": pt(name.encode(), 0).hex(), "g_vec": b"".join(gall[:nm]).hex(), "h_vec": b"".join(hall[:nv + 9]).hex(),
                "g_vec_": b"".join(gall[nm:]).hex(), "h_vec_": b"".join(hall[nv + 9:]).hex(), "instances": []}
        head, keep = _circuit_call_args(st, cdoc)
        label = bytes.fromhex(st["label"])
        hs = lambda xs: b"".join(bytes.fromhex(x) for x in xs)
        used = 18 + nv + nm
        for j in range(st.get("instances", 1)):
            s_v = b"".join(O.sc_to_bytes(sc(b"sv" + name.encode(), j, i)) for i in range(k))
            raw = hashlib.shake_256(b"rng" + name.encode() + bytes([j])).digest(64 * used)
            rnd = b"".join(O.sc_to_bytes(O.wide_reduce(raw[64 * i:64 * i + 64])) for i in range(used))
            com = C.create_string_buffer(64 * k)
            pbuf = C.create_string_buffer(64 * (4 + 2 * 16) + 32 * 16)
            rounds, pl, pn = sz(0), sz(0), sz(0)
            assert L.bppp_oracle_circuit_prove(*head, label, sz(len(label)), b"".join(hs(row) for row in st["v"]), s_v, hs(st["w_l"]), hs(st["w_r"]),
                                               hs(st["w_o"]), rnd, sz(used), com, pbuf, C.byref(rounds), C.byref(pl), C.byref(pn)) == 0
            nbytes = 64 * (4 + 2 * rounds.value) + 32 * (pl.value + pn.value)
            v = L.bppp_oracle_circuit_verify(*head, label, sz(len(label)), com.raw, pbuf.raw[:nbytes], sz(rounds.value), sz(pl.value), sz(pn.value))
            cdoc["instances"].append({"s_v": s_v.hex(), "rng_bytes": raw.hex(), "rng_calls": [64] * used, "rnd": rnd.hex(), "commitments": com.raw.hex(),
                                      "proof": pbuf.raw[:nbytes].hex(), "rounds": rounds.value, "nl": pl.value, "nn": pn.value, "accept": v == 1})
        out_c.append(cdoc)
        del keep
    for name, st in wnlas.items():
        ng, nh = st["ng"], st["nh"]
        label = bytes.fromhex(st["label"])
        g = pt(name.encode(), 0)
        gv, hv = b"".join(pt(name.encode() + b"g", i) for i in range(ng)), b"".join(pt(name.encode() + b"h", i) for i in range(nh))
        c = b"".join(O.sc_to_bytes(sc(b"c" + name.encode(), i)) for i in range(nh))
        rho = sc(b"rho" + name.encode(), 0)
        rho_b, mu_b = O.sc_to_bytes(rho), O.sc_to_bytes(rho * rho % O.N)
        lb, nb = b"".join(bytes.fromhex(x) for x in st["l"]), b"".join(bytes.fromhex(x) for x in st["n"])
        com = C.create_string_buffer(64)
        assert L.bppp_oracle_wnla_commit(g, gv, sz(ng), hv, sz(nh), c, sz(nh), rho_b, mu_b, lb, sz(nh), nb, sz(ng), com) == 0
        r_out, x_out = C.create_string_buffer(64 * 16), C.create_string_buffer(64 * 16)
        l_out, n_out = C.create_string_buffer(32 * 8), C.create_string_buffer(32 * 8)
        nr, nl, nn = sz(0), sz(0), sz(0)
        assert L.bppp_oracle_wnla_prove(g, gv, sz(ng), hv, sz(nh), c, sz(nh), rho_b, mu_b, label, sz(len(label)), com.raw, lb, sz(nh), nb, sz(ng), r_out, x_out,
                                        C.byref(nr), l_out, C.byref(nl), n_out, C.byref(nn)) == 0
        pr, px, pl_, pn_ = r_out.raw[:64 * nr.value], x_out.raw[:64 * nr.value], l_out.raw[:32 * nl.value], n_out.raw[:32 * nn.value]
        v = L.bppp_oracle_wnla_verify(g, gv, sz(ng), hv, sz(nh), c, sz(nh), rho_b, mu_b, label, sz(len(label)), com.raw, pr, px, sz(nr.value), pl_, sz(nl.value),
                                      pn_, sz(nn.value))
        out_w.append({"name": name, "label": st["label"], "ng": ng, "nh": nh, "g": g.hex(), "g_vec": gv.hex(), "h_vec": hv.hex(), "c": c.hex(), "rho": rho_b.hex(),
                      "mu": mu_b.hex(), "l": lb.hex(), "n": nb.hex(), "commitment": com.raw.hex(), "proof_r": pr.hex(), "proof_x": px.hex(), "proof_l": pl_.hex(),
                      "proof_n": pn_.hex(), "accept": v == 1})
    return {"source": "THIS repository's oracle (consumer self-test; pins nothing)", "circuits": out_c, "wnla": out_w}
