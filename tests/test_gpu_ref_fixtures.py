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
            _check_on_gpu(json.load(f))
