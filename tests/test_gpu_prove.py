"""GPU parity tests of the batch PROVER: proofs produced by the HIP path through the C ABI must be
byte-identical to the oracle prover's for the same (x, s, 52 prover scalars), and must verify (HIP verifier and oracle)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "u64_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def proto(gold):
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    from bp_pp_amd import U64RangeProofProtocol
    import workload
    g, gv, hv = workload.split_generators(bytes.fromhex(gold["generators"]))
    p = U64RangeProofProtocol(g, gv, hv, device=0)
    yield p
    p.close()


def test_golden_proofs_byte_exact(proto, gold):
    label = bytes.fromhex(gold["label"])
    cases = gold["cases"]
    x = np.array([c["x"] for c in cases], dtype=np.uint64)
    s = np.frombuffer(b"".join(bytes.fromhex(c["s"]) for c in cases), dtype=np.uint8).reshape(-1, 32)
    rnd = np.frombuffer(b"".join(bytes.fromhex(c["rnd"]) for c in cases), dtype=np.uint8).reshape(-1, 52 * 32)
    proofs, com, st = proto.prove_batch(x, s, rnd, label)
    assert not st.any()
    for i, c in enumerate(cases):
        assert bytes(com[i]).hex() == c["commitment"]
        assert bytes(proofs[i]).hex() == c["proof"]
    c = cases[2]
    assert proto.prove(c["x"], bytes.fromhex(c["s"]), label, bytes.fromhex(c["rnd"])).hex() == c["proof"]   # tests.rs:13-42 shape
    assert proto.verify(bytes.fromhex(c["commitment"]), bytes.fromhex(c["proof"]), label)


# the prover's dispatch regimes (bppp_u64.hip): 1, 65, 1024 the small-call path (a wavefront per sum, sixteen lanes per value in the stage
# and fold kernels, next commitments as sums); 1500 the same lane forms in their 256-register builds; 5000 four lanes per value in the
# stage kernels, round scalars from four workgroups, next commitments still sums; 9000 next commitments by the variable-base path on
# four lanes, in line; 2^14 the same on the helper stream under the next round's sums; 20000 one lane per value, helper stream
@pytest.mark.parametrize("n", [1, 65, 1024, 1500, 5000, 9000, 1 << 14, 20000])
def test_batch_prove_vs_oracle(proto, oracle_c, n):
    """n = 2^14 is BASELINE config 4.  Every proof equals the oracle's trapdoor prover byte for byte (itself equal to the
    honest reference-shaped prover, tests/test_oracle_c.py); a sample is also proved by the honest prover directly."""
    import workload
    first = 5000
    x, s, rnd = workload.values(n, first), workload.blindings(n, first), workload.prover_randomness(n, first)
    proofs, com, st = proto.prove_batch(x, s, rnd, workload.LABEL)
    assert not st.any()
    op, ov = oracle_c.u64_prove_trapdoor_batch(workload.generator_dlogs(), workload.LABEL, x, s, rnd, nthreads=os.cpu_count() or 1)
    assert (ov == com).all()
    assert (op == proofs).all()
    k = min(n, 8)
    hp, hv = oracle_c.u64_prove_batch(workload.generators(), workload.LABEL, x[:k], s[:k], rnd[:k], nthreads=k)
    assert (hp == proofs[:k]).all() and (hv == com[:k]).all()
    acc, vst = proto.verify_batch(com, proofs, workload.LABEL)          # prove -> verify round trip on the GPU
    assert acc.all() and not vst.any()
    m = min(n, 256)
    oacc, ost = oracle_c.u64_verify_batch(workload.generators(), workload.LABEL, com[:m].copy(), proofs[:m].copy(), nthreads=os.cpu_count() or 1)
    assert oacc.all() and not ost.any()


@pytest.mark.parametrize("n", [1, 300, 5000])
def test_prover_output_in_the_wire_format(proto, n):
    """bppp_u64_prove_batch_sec1: the same proofs as bppp_u64_prove_batch, SEC1-compressed on the device (what serde gives for
    SerializableProof, circuit.rs:36-76) -- equal to the wire module's host-side compression, and accepted by the SEC1 verifier."""
    import workload
    from bp_pp_amd import wire
    first = 9100
    x, s, rnd = workload.values(n, first), workload.blindings(n, first), workload.prover_randomness(n, first)
    p525, c33, st = proto.prove_batch_sec1(x, s, rnd, workload.LABEL)
    proofs, com, st0 = proto.prove_batch(x, s, rnd, workload.LABEL)
    assert not st.any() and not st0.any()
    for i in range(0, n, max(1, n // 50)):
        assert bytes(p525[i]) == wire.abi_to_sec1(bytes(proofs[i])) and bytes(c33[i]) == wire.compress_point(bytes(com[i])), i
    acc, vst = proto.verify_batch_sec1(c33, p525, workload.LABEL)
    assert acc.all() and not vst.any()


def test_non_canonical_prover_input_is_flagged(proto):
    import workload
    x, s, rnd = workload.values(2, 9), workload.blindings(2, 9).copy(), workload.prover_randomness(2, 9).copy()
    s[0] = np.frombuffer(workload.N_ORDER.to_bytes(32, "big"), dtype=np.uint8)        # s = n: not a canonical Scalar
    rnd[1, 32 * 20:32 * 21] = 0xFF                                                      # one prover draw >= n
    _, _, st = proto.prove_batch(x, s, rnd, workload.LABEL)
    assert (st & 1).all()


def test_prover_at_the_edges_of_its_inputs(proto, gold, oracle_c):
    """GPU twin of tests/test_core_emul.py::test_prover_at_the_edges_of_its_inputs (x = 0 / 2^64 - 1, blinding 0 / n - 1, draws all
    zero / n - 1 / one): byte-identical to the oracle prover, accepted by the GPU verifier (exact and RLC) and by the oracle."""
    import workload
    gens, label = bytes.fromhex(gold["generators"]), bytes.fromhex(gold["label"])
    x, s, rnd = workload.edge_prover_inputs()
    proofs, com, st = proto.prove_batch(x, s, rnd, label)
    op, ov = oracle_c.u64_prove_batch(gens, label, x, s, rnd, nthreads=2)
    assert not st.any() and (ov == com).all() and (op == proofs).all()
    acc, vst = proto.verify_batch(com, proofs, label)
    acc2, vst2 = proto.verify_batch_rlc(com, proofs, label, seed=bytes(32))
    oacc, ost = oracle_c.u64_verify_batch(gens, label, com, proofs, nthreads=2)
    assert acc.all() and acc2.all() and oacc.all() and not vst.any() and not vst2.any() and not ost.any()


@pytest.mark.parametrize("n", [1, 100, 3000, 1 << 14])
def test_ct_prover_mode_emits_the_same_bytes(proto, oracle_c, gold, n):
    """ "ct_prover" (include/bppp.h): the sums over the witness and its blindings -- V, r_com, c_o, c_l, c_r, c_s, and commit_value --
    in the form with no secret-dependent address (every entry of every 4-bit window read, masked select, complete additions).  Same
    points, so the same proofs and commitments, byte for byte, as the default mode and as the oracle prover; edge inputs included."""
    import workload
    first = 9000
    x, s, rnd = workload.values(n, first), workload.blindings(n, first), workload.prover_randomness(n, first)
    if n >= 100:
        ex, es, ernd = workload.edge_prover_inputs()
        x, s, rnd = np.concatenate([ex, x[5:]]), np.concatenate([es, s[5:]]), np.concatenate([ernd, rnd[5:]])
    x, s, rnd = np.ascontiguousarray(x), np.ascontiguousarray(s), np.ascontiguousarray(rnd)
    p0, c0, st0 = proto.prove_batch(x, s, rnd, workload.LABEL)
    v0 = proto.commit_value_batch(x, s)
    proto.set_option("ct_prover", 1)
    try:
        assert proto.get_option("ct_prover") == 1
        p1, c1, st1 = proto.prove_batch(x, s, rnd, workload.LABEL)
        v1 = proto.commit_value_batch(x, s)
        if n == 100:                                   # the single-proof front end picks the mode up (its lanes share the 4-bit table)
            one = proto.prove_one(int(x[7]), bytes(s[7]), workload.LABEL, bytes(rnd[7]))
            assert one[0] == bytes(p0[7]) and one[1] == bytes(c0[7]) and one[2] == 0
    finally:
        proto.set_option("ct_prover", 0)
    assert not st0.any() and not st1.any()
    assert (p1 == p0).all() and (c1 == c0).all() and (v1 == v0).all() and (v0 == c0).all()
    k = min(n, 64)
    hp, hv = oracle_c.u64_prove_batch(workload.generators(), workload.LABEL, x[:k], s[:k], rnd[:k], nthreads=8)
    assert (hp == p1[:k]).all() and (hv == c1[:k]).all()
    with pytest.raises(Exception):
        proto.set_option("ct_prover", 2)
