"""GPU twin of tests/test_ref_fixtures.py: the HIP path (through the C ABI) against reference-made fixtures when present, and
against an oracle-made document of the same format always (so the hook itself is known to work)."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check_on_gpu(doc):
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    gens, label = bytes.fromhex(doc["generators"]), bytes.fromhex(doc["label"])
    g, gv, hv = workload.split_generators(gens)
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
    try:
        cs = doc["cases"] + doc.get("negative_cases", [])
        u8 = lambda key, w: np.frombuffer(b"".join(bytes.fromhex(c[key]) for c in cs), dtype=np.uint8).reshape(len(cs), w).copy()
        V, P, S = u8("commitment", 64), u8("proof", 928), u8("state_before", 203)
        acc, st, out = proto.verify_batch_transcript(V, P, [s.tobytes() for s in S])
        assert acc.tolist() == [1 if c["accept"] else 0 for c in cs] and not st.any()
        assert (out == u8("state_after_verify", 203)).all()
        plain = [c for c in doc["cases"] if not c.get("context")]
        if plain:       # the label entry points: verify, and the batch prover replayed on the recorded draws
            Vp = np.frombuffer(b"".join(bytes.fromhex(c["commitment"]) for c in plain), dtype=np.uint8).reshape(-1, 64)
            Pp = np.frombuffer(b"".join(bytes.fromhex(c["proof"]) for c in plain), dtype=np.uint8).reshape(-1, 928)
            a, s = proto.verify_batch(Vp, Pp, label)
            assert a.all() and not s.any()
            x = np.array([int(c["x"]) for c in plain], dtype=np.uint64)
            sb = np.frombuffer(b"".join(bytes.fromhex(c["s"]) for c in plain), dtype=np.uint8).reshape(-1, 32)
            rnd = np.frombuffer(b"".join(bytes.fromhex(c["rnd"]) for c in plain), dtype=np.uint8).reshape(-1, 52 * 32)
            proofs, coms, pst = proto.prove_batch(x, sb, rnd, label)
            assert not pst.any() and (proofs == Pp).all() and (coms == Vp).all()
    finally:
        proto.close()


def _check_generic_on_gpu(doc):
    """gen_fixtures.rs generic: the product's ArithmeticCircuit / WeightNormLinearArgument (HIP, through the C ABI) on the recorded
    generators and instances -- verify must return the recorded verdict (the reference's own, true or false), prove on the recorded
    blindings and draws the recorded proof bytes."""
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    import ref_fixture_check as RC
    from bp_pp_amd.wnla import ArithmeticCircuit, WeightNormLinearArgument
    circuits, wnlas = RC.load_statements()
    pts = lambda h: [bytes.fromhex(h)[64 * i:64 * i + 64] for i in range(len(h) // 128)]
    for cdoc in doc.get("circuits", []):
        st = circuits[cdoc["name"]]
        label = bytes.fromhex(cdoc["label"])
        nm, no, nv, k = st["dim_nm"], st["dim_no"], st["dim_nv"], st["k"]
        flat = lambda rows: np.frombuffer(b"".join(bytes.fromhex(x) for row in rows for x in row), np.uint8).reshape(-1, 32)
        vec = lambda xs: np.frombuffer(b"".join(bytes.fromhex(x) for x in xs), np.uint8).reshape(-1, 32)
        part = st["partition"]
        ac = ArithmeticCircuit(nm, no, k, nv, bytes.fromhex(cdoc["g"]), pts(cdoc["g_vec"]), pts(cdoc["h_vec"]), flat(st["W_m"]), flat(st["W_l"]), vec(st["a_m"]),
                               vec(st["a_l"]), st["f_l"], st["f_m"], pts(cdoc["g_vec_"]), pts(cdoc["h_vec_"]),
                               lambda typ, j: (None if part[typ][j] < 0 else part[typ][j]), device=0, fb_window_bits=8)
        try:
            ins = cdoc["instances"]
            B = len(ins)
            u8 = lambda key, *shape: np.frombuffer(b"".join(bytes.fromhex(i[key]) for i in ins), np.uint8).reshape(B, *shape).copy()
            rounds, nl, nn = ins[0]["rounds"], ins[0]["nl"], ins[0]["nn"]
            coms, proofs = u8("commitments", k, 64), u8("proof", -1)
            acc, stt = ac.verify_batch(label, coms, proofs, rounds, nl, nn)
            assert acc.tolist() == [1 if i["accept"] else 0 for i in ins] and not stt.any(), cdoc["name"]
            rep = lambda xs: np.broadcast_to(vec(xs), (B,) + vec(xs).shape).copy()
            v = np.broadcast_to(np.stack([vec(row) for row in st["v"]]), (B, k, nv, 32)).copy()
            out, pst, shape = ac.prove_batch(label, coms, v, u8("s_v", k, 32), rep(st["w_l"]), rep(st["w_r"]), rep(st["w_o"]), u8("rnd", -1, 32))
            assert shape == (rounds, nl, nn) and not pst.any() and (out == proofs).all(), cdoc["name"]
        finally:
            ac.close()
    for w in doc.get("wnla", []):
        b = lambda key, *shape: np.frombuffer(bytes.fromhex(w[key]), np.uint8).reshape(1, *shape).copy()
        label = bytes.fromhex(w["label"])
        arg = WeightNormLinearArgument(bytes.fromhex(w["g"]), pts(w["g_vec"]), pts(w["h_vec"]), device=0, fb_window_bits=8)
        try:
            args = dict(commitments=b("commitment", 64), c=b("c", -1, 32), rho=b("rho", 32), mu=b("mu", 32))
            acc, stt = arg.verify_batch(label, proof_r=b("proof_r", -1, 64), proof_x=b("proof_x", -1, 64), proof_l=b("proof_l", -1, 32),
                                        proof_n=b("proof_n", -1, 32), **args)
            assert int(acc[0]) == (1 if w["accept"] else 0) and not stt.any(), w["name"]
            pr, px, pl, pn, pst = arg.prove_batch(label, l=b("l", -1, 32), n=b("n", -1, 32), **args)
            assert not pst.any() and (pr == b("proof_r", -1, 64)).all() and (px == b("proof_x", -1, 64)).all(), w["name"]
            assert (pl == b("proof_l", -1, 32)).all() and (pn == b("proof_n", -1, 32)).all(), w["name"]
            com, cst = arg.commit_batch(args["c"], args["mu"], b("l", -1, 32), b("n", -1, 32))
            assert not cst.any() and (com == args["commitments"]).all(), w["name"]
        finally:
            arg.close()


def test_gpu_on_an_oracle_made_generic_document(oracle_c):
    import ref_fixture_check as RC
    _check_generic_on_gpu(RC.oracle_made_generic_document(oracle_c))


def test_gpu_on_an_oracle_made_document():
    import ref_fixture_check as RC
    _check_on_gpu(RC.oracle_made_document(4))


def test_gpu_reproduces_the_reference_made_fixtures():
    import ref_fixture_check as RC
    paths = RC.reference_fixture_paths()
    if not paths:
        pytest.skip("parity UNPINNED: no tests/golden/ref_*.json (run facade/src/bin/gen_fixtures.rs where a Rust toolchain exists)")
    for p in paths:
        with open(p) as f:
            doc = json.load(f)
        if "cases" in doc:
            _check_on_gpu(doc)
        if "circuits" in doc or "wnla" in doc:
            _check_generic_on_gpu(doc)
