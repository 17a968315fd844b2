"""u64 range proofs over PRE-LOADED transcripts, built with the Python oracle: the caller binds context into its
`merlin::Transcript` before prove / verify (the reference's `t: &mut Transcript`, u64_proof.rs:42,57) and keeps using the
transcript afterwards, so both the accept bit and the advanced state are part of the contract."""
import numpy as np

import bppp_oracle as O
from bp_pp_amd import synth


def ser(t: "O.Transcript") -> bytes:
    """merlin's STROBE-128 state as the 203 bytes the C ABI exchanges."""
    return bytes(t.strobe.state) + bytes([t.strobe.pos, t.strobe.pos_begin, t.strobe.cur_flags])


def make(n: int = 4, shared: bool = False):
    """n proofs, proof j made on Transcript::new(label) + append_message(b"ctx", <j-dependent bytes>) (+ one append_u64).
    Returns dict(gens, V, P, states_in [n or 1], states_after [n] (oracle, after verify), accept [n])."""
    g, gv, hv = O.synth_generators()
    proto = O.U64RangeProofProtocol(g, gv, hv)
    gens = b"".join(O.pt_to_xy64(p) for p in [g] + list(gv) + list(hv))
    Vs, Ps, tin, tout, acc = [], [], [], [], []
    for j in range(n):
        t = O.Transcript(synth.LABEL)
        t.append_message(b"ctx", b"order-book/7" if shared else (b"tx-" + bytes([j]) * (1 + 60 * j)))   # crosses the rate for j >= 3
        t.append_u64(b"height", 1000 if shared else 1000 + j)
        x, s = O.synth_value(40 + j), O.synth_blinding(40 + j)
        V = proto.commit_value(x, s)
        proof = proto.prove(x, s, t.clone(), O.ScalarRng(O.synth_rng_scalars(40 + j)))
        tv = t.clone()
        ok = proto.verify(V, proof, tv)
        assert ok
        Vs.append(O.pt_to_xy64(V)); Ps.append(O.u64_proof_to_bytes(proof)); tin.append(ser(t)); tout.append(ser(tv)); acc.append(1)
    u8 = lambda blobs, w: np.frombuffer(b"".join(blobs), dtype=np.uint8).reshape(len(blobs), w).copy()
    return dict(gens=gens, V=u8(Vs, 64), P=u8(Ps, 928), states_in=u8(tin[:1] if shared else tin, 203), states_after=u8(tout, 203),
                accept=np.array(acc, np.uint8), proto=proto)


def oracle_verify(case, j: int, V: bytes, P: bytes, state_in: bytes):
    """Oracle verdict and advanced state for arbitrary (V, proof, state) -- for the negative cases."""
    t = O.Transcript(b"x")
    t.strobe.state = bytearray(state_in[:200])
    t.strobe.pos, t.strobe.pos_begin, t.strobe.cur_flags = state_in[200], state_in[201], state_in[202]
    ok = case["proto"].verify(O.pt_from_xy64(V), O.u64_proof_from_bytes(P), t)
    return bool(ok), ser(t)
