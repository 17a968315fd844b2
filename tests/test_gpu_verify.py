"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI of include/bppp.h, against the
oracle on the same seeded inputs.  Bit-exact bar: accept bits, per-proof status and every transcript challenge /
hashed commitment (the trace) must equal the oracle's bytes."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("these tests need a GPU (they are selected with -m gpu only on the GPU box)")
    return torch


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "u64_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module", params=[8, 16])
def proto(request, torch_mod, gold):
    from bp_pp_amd import U64RangeProofProtocol
    import workload
    g, gv, hv = workload.split_generators(bytes.fromhex(gold["generators"]))
    p = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=request.param)
    yield p
    p.close()


def _device_verify(torch, proto, label, V, P, want_trace=True):
    n = V.shape[0]
    dV = torch.from_numpy(V).cuda()
    dP = torch.from_numpy(P).cuda()
    dA = torch.zeros(n, dtype=torch.uint8, device="cuda")
    dS = torch.zeros(n, dtype=torch.int32, device="cuda")
    dT = torch.zeros((n, 704), dtype=torch.uint8, device="cuda") if want_trace else None
    dR = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    torch.cuda.synchronize()   # inputs ready; the context runs on its own (non-blocking) stream, joined by proto.synchronize()
    proto.verify_batch_device(label, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(),
                              dT.data_ptr() if want_trace else 0, dR.data_ptr())
    torch.cuda.synchronize()
    return dA.cpu().numpy(), dS.cpu().numpy(), (dT.cpu().numpy() if want_trace else None), int(dR.item())


def test_golden_vectors_bit_exact(torch_mod, proto, gold, oracle_c):
    gens, label = bytes.fromhex(gold["generators"]), bytes.fromhex(gold["label"])
    items = [(c["commitment"], c["proof"], 1, 0) for c in gold["cases"]]
    items += [(c["commitment"], c["proof"], 0, c["status"]) for c in gold["negative_cases"]]
    n = len(items)
    V = np.frombuffer(b"".join(bytes.fromhex(i[0]) for i in items), dtype=np.uint8).reshape(n, 64).copy()
    P = np.frombuffer(b"".join(bytes.fromhex(i[1]) for i in items), dtype=np.uint8).reshape(n, 928).copy()
    acc, st, tr, rej = _device_verify(torch_mod, proto, label, V, P)
    assert acc.tolist() == [i[2] for i in items]
    assert st.tolist() == [i[3] for i in items]
    assert rej == sum(1 for i in items if not i[2])
    for k, c in enumerate(gold["cases"]):
        exp = bytes.fromhex(c["trace_challenges_and_points"])
        assert bytes(tr[k][:len(exp)]) == exp
    for k in range(n):
        if items[k][3] == 0:
            rc, otr = oracle_c.u64_verify(gens, label, bytes(V[k]), bytes(P[k]), trace=True)
            assert rc == items[k][2] and bytes(tr[k]) == otr
    # host-pointer entry point (copies in and out) agrees
    acc2, st2 = proto.verify_batch(V, P, label)
    assert (acc2 == acc).all() and (st2 == st).all()
    assert proto.verify(bytes(V[2]), bytes(P[2]), label) is True


def test_commit_value_matches_reference_formula(torch_mod, proto, gold, oracle_c):
    gens = bytes.fromhex(gold["generators"])
    for c in gold["cases"]:
        assert proto.commit_value(c["x"], bytes.fromhex(c["s"])).hex() == c["commitment"]
    x = np.array([0, 1, 2**64 - 1, 123456, 16**15], dtype=np.uint64)
    s = np.frombuffer(b"".join(int(v).to_bytes(32, "big") for v in [0, 1, 5, 2**200, 7]), dtype=np.uint8).reshape(5, 32)
    out = proto.commit_value_batch(x, s)
    for i in range(5):
        assert bytes(out[i]) == oracle_c.u64_commit_value(gens, int(x[i]), bytes(s[i]))


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4096, 6000])   # <= 4096: small-call path (a lane per table / half stream); 6000: lane groups
def test_ragged_batches_vs_oracle(torch_mod, proto, oracle_c, n):
    import workload
    gens, V, P, _ = workload.make_batch(n, first=1000)
    P, expect = workload.corrupt(P, V, every=7)
    acc, st, tr, rej = _device_verify(torch_mod, proto, workload.LABEL, V, P)
    oacc, ost = oracle_c.u64_verify_batch(gens, workload.LABEL, V, P, nthreads=os.cpu_count() or 1)
    assert (acc == oacc).all() and (acc == expect).all()
    assert (st == 0).all() and (ost == 0).all()
    assert rej == int((expect == 0).sum())
    for k in range(0, n, max(1, n // 16)):       # sampled intermediates, byte for byte
        rc, otr = oracle_c.u64_verify(gens, workload.LABEL, bytes(V[k]), bytes(P[k]), trace=True)
        assert bytes(tr[k]) == otr


def test_empty_batch_is_a_no_op(torch_mod, proto):
    acc, st = proto.verify_batch(np.zeros((0, 64), np.uint8), np.zeros((0, 928), np.uint8), b"u64 range proof")
    assert acc.shape == (0,) and st.shape == (0,)


def test_full_size_batch_properties(torch_mod, gold):
    """BASELINE config 2: 2^16 independent proofs on one GPU.  Size-independent properties: every honest proof is accepted,
    exactly the corrupted ones are rejected, the result is idempotent and permutation-equivariant, and a sampled subset
    is bit-exact against the oracle."""
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    import bppp_oracle_c as OC
    torch = torch_mod
    n = 1 << 16
    gens, V, P, _ = workload.make_batch(n)
    P, expect = workload.corrupt(P, V, every=1024)
    g, gv, hv = workload.split_generators(gens)
    proto = U64RangeProofProtocol(g, gv, hv, device=0)
    try:
        acc, st, _, rej = _device_verify(torch, proto, workload.LABEL, V, P, want_trace=False)
        assert (acc == expect).all() and (st == 0).all() and rej == n // 1024
        acc2, _, _, _ = _device_verify(torch, proto, workload.LABEL, V, P, want_trace=False)
        assert (acc2 == acc).all()
        perm = np.random.default_rng(1).permutation(n)
        acc3, _, _, _ = _device_verify(torch, proto, workload.LABEL, V[perm].copy(), P[perm].copy(), want_trace=False)
        assert (acc3 == acc[perm]).all()
        idx = np.concatenate([np.arange(0, n, 1024)[:16], np.arange(5, n, 257)[:240]])
        oacc, _ = OC.u64_verify_batch(gens, workload.LABEL, V[idx].copy(), P[idx].copy(), nthreads=os.cpu_count() or 1)
        assert (oacc == acc[idx]).all()
    finally:
        proto.close()


@pytest.mark.parametrize("n", [20000, 40000])
def test_sizes_that_run_kernels_beside_each_other_vs_oracle(torch_mod, proto, oracle_c, n):
    """2^14 < n <= 2^16: the one-lane table kernel runs BESIDE phase 1 from its own decode of the proof bytes (verify_tables_own);
    2^15 < n <= 2^16: the last round goes out as head and tail, the tail beside the final fixed-base sum (bppp_u64.hip: tables_beside,
    tail_beside).  Honest, tampered and MALFORMED proofs (byte flips, a coordinate = p, a scalar = n): accept bits, statuses and -- on a
    sample that includes every malformed proof's neighbours -- the whole trace of challenges and intermediate commitments equal the
    oracle's."""
    import workload
    gens, V, P, _ = workload.make_batch(n, first=200)
    P, expect = workload.corrupt(P, V, every=11)
    V, P = V.copy(), P.copy()
    rng = np.random.default_rng(n)
    bad = sorted(set(int(i) for i in rng.integers(0, n, 96)))
    for i in bad:
        P[i, int(rng.integers(0, 928))] ^= int(rng.integers(1, 256))
    pbytes = (2**256 - 2**32 - 977).to_bytes(32, "big")
    nbytes = int("FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141", 16).to_bytes(32, "big")
    P[n - 1, 64:96] = np.frombuffer(pbytes, np.uint8)
    P[n // 2, 864:896] = np.frombuffer(nbytes, np.uint8)
    bad += [n - 1, n // 2]
    acc, st, tr, rej = _device_verify(torch_mod, proto, workload.LABEL, V, P)
    sample = sorted(set(bad + [b + 1 for b in bad if b + 1 < n] + list(range(0, n, n // 64))))
    flagged = 0
    for i in sample:
        rc, otr = oracle_c.u64_verify(gens, workload.LABEL, bytes(V[i]), bytes(P[i]), trace=True)
        assert int(acc[i]) == (1 if rc == 1 else 0), i
        assert (int(st[i]) != 0) == (rc < 0), (i, rc, int(st[i]))
        if rc >= 0:
            assert bytes(tr[i]) == otr, i
        flagged += rc < 0
    assert flagged >= 2 and st[n - 1] != 0 and st[n // 2] != 0
    clean = np.ones(n, bool)
    clean[bad] = False
    assert (acc[clean] == expect[clean]).all() and not st[clean].any()
    assert rej == int((acc == 0).sum())


def test_sec1_wire_inputs(torch_mod, proto, gold, oracle_c):
    """the same verify fed with the reference's wire content (33-byte SEC1 points, 525-byte proofs)."""
    import workload
    from bp_pp_amd import wire
    n = 300
    gens, V, P, _ = workload.make_batch(n, first=2000)
    P, expect = workload.corrupt(P, V, every=11)
    C33 = np.frombuffer(b"".join(wire.compress_point(bytes(V[i])) for i in range(n)), dtype=np.uint8).reshape(n, 33).copy()
    P525 = np.frombuffer(b"".join(wire.abi_to_sec1(bytes(P[i])) for i in range(n)), dtype=np.uint8).reshape(n, 525).copy()
    P525[5, 0] = 4                      # bad SEC1 tag
    P525[6, 33:66] = np.frombuffer(b"\x02" + (5).to_bytes(32, "big"), dtype=np.uint8)     # x without a square root
    P525[7, 0] ^= 1                     # the other root of c_l: decodes, must be rejected by the protocol
    # encodings with x = 0 mod p that k256's from_bytes refuses: they must be flagged, never read as the identity
    pbytes = (2**256 - 2**32 - 977).to_bytes(32, "big")
    P525[8, 66:99] = np.frombuffer(b"\x02" + bytes(32), dtype=np.uint8)       # 02 || 0: 7 is a non-residue
    P525[9, 99:132] = np.frombuffer(b"\x05" + bytes(32), dtype=np.uint8)      # bad tag over x = 0
    P525[10, 132:165] = np.frombuffer(b"\x02" + pbytes, dtype=np.uint8)       # x = p
    C33[12] = np.frombuffer(b"\x03" + pbytes, dtype=np.uint8)                 # the commitment itself
    acc, st = proto.verify_batch_sec1(C33, P525, workload.LABEL)
    acc_ref, st_ref = proto.verify_batch(V, P, workload.LABEL)
    touched = [5, 6, 7, 8, 9, 10, 12]
    keep = np.ones(n, bool); keep[touched] = False
    assert (acc[keep] == acc_ref[keep]).all() and (acc[keep] == expect[keep]).all() and not st[keep].any()
    for i in (5, 6, 8, 9, 10, 12):
        assert st[i] == 1 and acc[i] == 0, i
    assert st[7] == 0 and acc[7] == 0


def test_random_byte_corruptions_vs_oracle(torch_mod, proto, oracle_c):
    """Fuzz: one random byte of every proof (or its commitment) is XOR-ed with a random value.  Accept bit and the
    malformed-input status (non-canonical coordinate / scalar, point off the curve: the inputs k256 deserialisation refuses)
    must equal the oracle's verdict for each proof; the oracle's negative return codes are its decoding failures."""
    import workload
    n = 768
    gens, V, P, _ = workload.make_batch(n, first=31000)
    rng = np.random.default_rng(11)
    V, P = V.copy(), P.copy()
    for i in range(n):
        if i % 16 == 0:
            continue                                         # keep some untouched
        x = int(rng.integers(1, 256))
        if i % 7 == 0:
            V[i, int(rng.integers(0, 64))] ^= x
        else:
            P[i, int(rng.integers(0, 928))] ^= x
    # force the rare encodings too: coordinate = p (non-canonical), scalar = n (non-canonical), y negated (valid other point)
    pbytes = (2**256 - 2**32 - 977).to_bytes(32, "big")
    nbytes = int("FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141", 16).to_bytes(32, "big")
    P[16, 0:32] = np.frombuffer(pbytes, np.uint8)             # (these three proofs were left untouched above)
    P[32, 832:864] = np.frombuffer(nbytes, np.uint8)
    y = int.from_bytes(P[48, 96:128].tobytes(), "big")
    P[48, 96:128] = np.frombuffer(((2**256 - 2**32 - 977) - y).to_bytes(32, "big"), np.uint8)
    acc, st, _, rej = _device_verify(torch_mod, proto, workload.LABEL, V, P, want_trace=False)
    n_flag = 0
    for i in range(n):
        rc = oracle_c.u64_verify(gens, workload.LABEL, bytes(V[i]), bytes(P[i]))
        assert int(acc[i]) == (1 if rc == 1 else 0), i
        assert (int(st[i]) != 0) == (rc < 0), (i, rc, int(st[i]))
        n_flag += rc < 0
    assert rej == int((acc == 0).sum()) and n_flag > 50 and acc[64::16].all()
    assert st[16] != 0 and st[32] != 0 and st[48] == 0 and acc[48] == 0


# (13024, 1024): three parts -- 1,024, then 7 x 1,024, then the rest (round 5: the parts of a pipelined host-buffer call grow)
@pytest.mark.parametrize("n,chunk", [(5000, 1024), (1537, 1024), (3072, 1024), (4097, 2048), (13024, 1024)])
def test_host_buffer_path_pipelined_in_chunks(torch_mod, proto, oracle_c, n, chunk):
    """bppp_u64_verify_batch over host buffers uploads a large batch chunk by chunk while the previous chunk is verified
    (include/bppp.h, "host_chunk").  With a small chunk size so that a test batch spans several chunks -- tail merged into the
    last chunk, exact multiples, one proof over -- the results are those of the unchunked call and of the oracle, in exact and
    in RLC mode, including flagged (bad-encoding) proofs that sit on chunk boundaries."""
    import workload
    gens, V, P, _ = workload.make_batch(n, first=77)
    P, expect = workload.corrupt(P, V, every=13)
    st_expect = np.zeros(n, np.int32)
    for j in (chunk - 1, chunk, n - 1):
        P[j, 0:32] = 0xFF                                   # x coordinate >= p: k256 would refuse to deserialize it
        expect[j], st_expect[j] = 0, 1
    try:
        proto.set_option("host_chunk", 0)
        acc0, st0 = proto.verify_batch(V, P, workload.LABEL)
        proto.set_option("host_chunk", chunk)
        acc1, st1 = proto.verify_batch(V, P, workload.LABEL)
        acc2, st2 = proto.verify_batch_rlc(V, P, workload.LABEL, seed=bytes(range(32)))
    finally:
        proto.set_option("host_chunk", 1 << 17)
    assert (acc0 == expect).all() and (st0 == st_expect).all()
    assert (acc1 == acc0).all() and (st1 == st0).all() and (acc2 == acc0).all() and (st2 == st0).all()
    idx = np.unique(np.concatenate([np.arange(0, n, 41), [chunk - 1, chunk, chunk + 1, n - 1]]))
    oacc, ost = oracle_c.u64_verify_batch(gens, workload.LABEL, V[idx].copy(), P[idx].copy(), nthreads=os.cpu_count() or 1)
    assert (oacc == acc1[idx]).all() and ((ost != 0) == (st1[idx] != 0)).all()   # the oracle's own code for a bad encoding is -1
    with pytest.raises(Exception):
        proto.set_option("host_chunk", 1000)                # not a multiple of 64


def test_internal_parts_bound_the_workspace(torch_mod, proto, oracle_c):
    """A device-buffer batch larger than "max_batch" runs as consecutive parts (include/bppp.h): accept bits, status, the trace, the
    reject count and the RLC mode are those of the one-part call; the per-proof workspace stops growing with n."""
    import workload
    torch = torch_mod
    n = 3000
    gens, V, P, _ = workload.make_batch(n, first=4000)
    P, expect = workload.corrupt(P, V, every=11)
    P[1024, 64:96] = 0xFF                                    # a flagged proof right at a part boundary
    expect[1024] = 0
    ref = _device_verify(torch, proto, workload.LABEL, V, P)
    try:
        proto.set_option("max_batch", 1024)
        got = _device_verify(torch, proto, workload.LABEL, V, P)
        dV, dP = torch.from_numpy(V).cuda(), torch.from_numpy(P).cuda()
        dA = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dS = torch.zeros(n, dtype=torch.int32, device="cuda")
        dR = torch.zeros(1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        proto.verify_batch_rlc_device(workload.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), bytes(range(32)), dS.data_ptr(), dR.data_ptr())
        proto.synchronize()
    finally:
        proto.set_option("max_batch", 1 << 21)
    for a, b in zip(ref[:3], got[:3]):
        assert (a == b).all()
    assert ref[3] == got[3] == int((expect == 0).sum()) == int(dR.item())
    assert (got[0] == expect).all() and got[1][1024] == 1
    assert (dA.cpu().numpy() == expect).all() and (dS.cpu().numpy() == got[1]).all()
    with pytest.raises(Exception):
        proto.set_option("max_batch", 100)


def test_calls_from_several_threads_on_one_context(torch_mod, proto):
    """SURVEY 8b "Threading": the reference's types are Send + Sync.  Calls on ONE context from several host threads are serialized
    by the context's lock; each gets its own batch's results."""
    import threading
    import workload
    gens, V, P, _ = workload.make_batch(700, first=9000)
    batches = []
    for k in range(6):
        lo, hi = 100 * k, 100 * k + 150 + 10 * k
        Pk, ek = workload.corrupt(P[lo:hi].copy(), V[lo:hi], every=5 + k)
        batches.append((V[lo:hi].copy(), Pk, ek))
    results, errors = [None] * len(batches), []

    def work(k):
        try:
            for _ in range(4):
                Vk, Pk, ek = batches[k]
                acc, st = (proto.verify_batch if k % 2 else lambda v, p, l: proto.verify_batch_rlc(v, p, l, seed=bytes([k]) * 32))(Vk, Pk, workload.LABEL)
                assert (acc == ek).all() and not st.any()
            results[k] = True
        except Exception as e:          # noqa: BLE001
            errors.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(len(batches))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors and all(results)


def test_a_part_that_cannot_be_allocated_is_halved_not_fatal(torch_mod, proto, oracle_c):
    """max_batch is a guess from the memory that was free when the context was created.  When the workspace of a part cannot be allocated
    after all (another context or process took the memory: here the 1st, 2nd, 3rd allocation of the call is made to fail), the call
    releases what it holds, halves the part size -- which stays halved: "max_batch" -- and runs the part again; verdicts, statuses and the
    reject count are those of an undisturbed call.  Only a part of at most 4,096 proofs that still cannot be allocated fails the call,
    with BPPP_ERR_NOMEM and a context that keeps working (include/bppp.h)."""
    import workload
    from bp_pp_amd import _capi
    torch = torch_mod
    n = 9000
    gens, V, P, _ = workload.make_batch(n, first=77000)
    P, expect = workload.corrupt(P, V, every=500)
    dV, dP = torch.from_numpy(V).cuda(), torch.from_numpy(P).cuda()
    dA = torch.zeros(n, dtype=torch.uint8, device="cuda"); dS = torch.zeros(n, dtype=torch.int32, device="cuda"); dR = torch.zeros(1, dtype=torch.int32, device="cuda")
    for k, want_parts in ((1, 4544), (2, 4544), (3, 4544)):
        c = proto.clone_shared()                # fresh workspaces: every buffer of the call is still to be allocated
        try:
            assert c.get_option("max_batch") == proto.get_option("max_batch")          # children inherit the parent's part size
            c.set_option("inject_alloc_fault", k)
            dA.zero_()
            c.verify_batch_device(workload.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, dR.data_ptr())
            c.synchronize()
            assert (dA.cpu().numpy() == expect).all() and not dS.any().item() and int(dR.item()) == int((expect == 0).sum())
            assert c.get_option("max_batch") == want_parts, (k, c.get_option("max_batch"))         # 9000 -> ceil(4500 / 64) * 64
            c.set_option("inject_alloc_fault", 0)
            dA.zero_()
            c.verify_batch_device(workload.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, dR.data_ptr())
            c.synchronize()
            assert (dA.cpu().numpy() == expect).all() and int(dR.item()) == int((expect == 0).sum())
        finally:
            c.close()
    # at or below the smallest part there is nothing left to halve: the failure is reported, and the context keeps working
    c = proto.clone_shared()
    try:
        c.set_option("inject_alloc_fault", 1)
        with pytest.raises(_capi.BpppError) as ei:
            c.verify_batch_device(workload.LABEL, 4096, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, 0)
        assert ei.value.code == -5 and "memory" in str(ei.value).lower()
        dA.zero_()
        c.verify_batch_device(workload.LABEL, 4096, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, 0)
        c.synchronize()
        assert (dA[:4096].cpu().numpy() == expect[:4096]).all()
    finally:
        c.close()


def test_identity_points_in_proofs_vs_oracle(torch_mod, proto, oracle_c):
    """The identity is a point k256 deserializes (serde "00"; 64 zero bytes at this ABI): a proof or commitment carrying it is
    well-formed, gets hashed as 33 zero bytes (GroupEncoding) and runs through every window table and sum as the neutral element.
    Each of the 14 points set to the identity in turn, and all 14 at once: accept bit, status and the whole trace (challenges and
    hashed commitments) equal the oracle's; a duplicated point (X_1 := R_1, c_l := c_r) exercises the P + P cases of the tables."""
    import workload
    gens, V0, P0, _ = workload.make_batch(2, first=555)
    rows_v, rows_p = [], []
    for j in range(14):
        V, P = V0[0].copy(), P0[0].copy()
        if j == 13:
            V[:] = 0
        else:
            P[64 * j:64 * j + 64] = 0
        rows_v.append(V); rows_p.append(P)
    V, P = V0[1].copy(), P0[1].copy()
    V[:] = 0; P[:832] = 0
    rows_v.append(V); rows_p.append(P)
    for a, b in ((8, 4), (0, 1), (12, 3)):                   # X_1 := R_1, c_l := c_r, reciprocal r := c_s
        V, P = V0[1].copy(), P0[1].copy()
        P[64 * a:64 * a + 64] = P[64 * b:64 * b + 64]
        rows_v.append(V); rows_p.append(P)
        V, P = V0[1].copy(), P0[1].copy()                     # ... and the negated copy: P + (-P) inside the sums
        P[64 * a:64 * a + 32] = P[64 * b:64 * b + 32]
        y = int.from_bytes(P[64 * b + 32:64 * b + 64].tobytes(), "big")
        P[64 * a + 32:64 * a + 64] = np.frombuffer(((2**256 - 2**32 - 977) - y).to_bytes(32, "big"), np.uint8)
        rows_v.append(V); rows_p.append(P)
    rows_v.append(V0[0].copy()); rows_p.append(P0[0].copy())  # untouched control
    V, P = np.stack(rows_v), np.stack(rows_p)
    acc, st, tr, rej = _device_verify(torch_mod, proto, workload.LABEL, V, P)
    assert acc[-1] == 1 and not acc[:-1].any() and not st.any() and rej == len(acc) - 1
    for i in range(len(acc)):
        rc, otr = oracle_c.u64_verify(gens, workload.LABEL, bytes(V[i]), bytes(P[i]), trace=True)
        assert int(acc[i]) == (1 if rc == 1 else 0) and rc >= 0, (i, rc)
        assert bytes(tr[i]) == otr, i
    # the same rows through the RLC mode and the host-buffer path
    acc2, st2 = proto.verify_batch_rlc(V, P, workload.LABEL, seed=bytes(range(32)))
    acc3, st3 = proto.verify_batch(V, P, workload.LABEL)
    assert (acc2 == acc).all() and (acc3 == acc).all() and not st2.any() and not st3.any()


@pytest.mark.parametrize("n", [3, 300, 6000, 20000])
def test_lane_groups_and_one_lane_per_proof_agree(torch_mod, gold, oracle_c, n):
    """Calls of up to 4,096 proofs take the small-call path (a lane per window table and per HALF GLV stream, a wavefront per
    fixed-base sum); beyond that the variable-base sums run on groups of four (n <= 16,384) or two (n <= 32,768) lanes per proof.
    A context created with BPPP_NO_SPLIT falls back to the lane groups at every small size, one created with BPPP_NO_LANE_GROUPS
    keeps one lane per proof.  All three: same accept bits, statuses, reject count and -- byte for byte -- the same trace (every
    challenge and every hashed commitment), also for proofs with repeated / negated / identity points, whose sums hit the
    exceptional-addition fallback inside a group."""
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    gens, V, P, _ = workload.make_batch(min(n, 400), first=70000)
    if n > 400:                                             # large case: tile the 400 proofs (the group size is what matters)
        reps = (n + 399) // 400
        V, P = np.tile(V, (reps, 1))[:n].copy(), np.tile(P, (reps, 1))[:n].copy()
    P, expect = workload.corrupt(P, V, every=9)
    if n >= 300:
        P[5, 64 * 8:64 * 9] = P[5, 64 * 4:64 * 5]           # X_1 := R_1
        P[6, 64 * 4:64 * 5] = 0                             # R_1 := identity
        y = int.from_bytes(P[7, 64 * 4 + 32:64 * 5].tobytes(), "big")
        P[7, 64 * 8:64 * 8 + 32] = P[7, 64 * 4:64 * 4 + 32]
        P[7, 64 * 8 + 32:64 * 9] = np.frombuffer(((2**256 - 2**32 - 977) - y).to_bytes(32, "big"), np.uint8)   # X_1 := -R_1
    g, gv, hv = workload.split_generators(gens)
    res = []
    for switch in (None, "BPPP_NO_SPLIT", "BPPP_NO_LANE_GROUPS"):
        if switch:
            os.environ[switch] = "1"
        try:
            proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
        finally:
            if switch:
                os.environ.pop(switch, None)
        try:
            res.append(_device_verify(torch_mod, proto, workload.LABEL, V, P))
        finally:
            proto.close()
    (a0, s0, t0, r0) = res[0]
    for (a1, s1, t1, r1) in res[1:]:
        assert (a0 == a1).all() and (s0 == s1).all() and r0 == r1 and (t0 == t1).all()
    for i in ([0, 1, 2] if n < 300 else [0, 5, 6, 7, 9, 18, n - 1]):
        rc, otr = oracle_c.u64_verify(gens, workload.LABEL, bytes(V[i]), bytes(P[i]), trace=True)
        assert int(a0[i]) == (1 if rc == 1 else 0) and (rc < 0 or bytes(t0[i]) == otr)


@pytest.mark.gpu
def test_fixed_base_one_lane_and_eight_lanes_agree(torch_mod, gold, oracle_c):
    """From 2^17 proofs up the two fixed-base sums of the u64 verifier (C0's fixed half, the final check) run on one lane per proof
    instead of eight (k_verify_*_l1); BPPP_FB_ONE_LANE=1 / =0 forces either form at any size.  Same accept bits, statuses, reject
    count and byte-identical traces, including proofs whose points make the sums degenerate."""
    import workload
    from bp_pp_amd import U64RangeProofProtocol
    gens, V, P, _ = workload.make_batch(300, first=81000)
    P, expect = workload.corrupt(P, V, every=7)
    P[5, 64 * 8:64 * 9] = P[5, 64 * 4:64 * 5]
    P[6, 64 * 4:64 * 5] = 0
    g, gv, hv = workload.split_generators(gens)
    res = []
    for mode in ("0", "1"):
        os.environ["BPPP_FB_ONE_LANE"] = mode
        try:
            proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
        finally:
            os.environ.pop("BPPP_FB_ONE_LANE", None)
        try:
            res.append(_device_verify(torch_mod, proto, workload.LABEL, V, P))
        finally:
            proto.close()
    (a0, s0, t0, r0), (a1, s1, t1, r1) = res
    assert (a0 == a1).all() and (s0 == s1).all() and r0 == r1 and (t0 == t1).all()
    assert (a0[[i for i in range(300) if i not in (5, 6)]] == expect[[i for i in range(300) if i not in (5, 6)]]).all()
    for i in (0, 5, 6, 7, 14, 299):
        rc, otr = oracle_c.u64_verify(gens, workload.LABEL, bytes(V[i]), bytes(P[i]), trace=True)
        assert int(a1[i]) == (1 if rc == 1 else 0) and (rc < 0 or bytes(t1[i]) == otr)
