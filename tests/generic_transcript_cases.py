"""Generic-layer instances over PRE-LOADED transcripts, made with the Python oracle (the C oracle's provers take a label only):
  * WeightNormLinearArgument (wnla.rs:75,125): small generator sets, transcripts holding per-instance context;
  * ReciprocalRangeProofProtocol at the u64 dimensions (16, 16) -- a u64 proof IS a reciprocal proof over g, g_vec, h_vec[..26],
    h_vec_ = h_vec[26..] -- reusing ref_fixture_check.oracle_made_document."""
import hashlib

import numpy as np

import bppp_oracle as O
import wnla_cases
from transcript_cases import ser


def wnla_case(ng: int = 4, nh: int = 8, B: int = 3, label: bytes = b"wnla transcript test"):
    g, gv, hv = wnla_cases.generators(ng, nh)
    G, GV, HV = O.pt_from_xy64(g), [O.pt_from_xy64(p) for p in gv], [O.pt_from_xy64(p) for p in hv]
    sc = lambda tag, *i: O.wide_reduce(hashlib.shake_256(b"gtc" + tag + bytes(i)).digest(64))
    out = dict(g=g, gv=gv, hv=hv, ng=ng, nh=nh)
    cs, rhos, mus, coms, prs, pxs, pls, pns, tin, tout = ([] for _ in range(10))
    for b in range(B):
        c = [sc(b"c", b, i) for i in range(nh)]
        rho = sc(b"rho", b)
        mu = rho * rho % O.N
        l = [sc(b"l", b, i) for i in range(nh)]
        n = [sc(b"n", b, i) for i in range(ng)]
        arg = O.WeightNormLinearArgument(G, list(GV), list(HV), list(c), rho, mu)
        com = arg.commit(l, n)
        t = O.Transcript(label)
        t.append_message(b"ctx", b"instance-" + bytes([65 + b]) * (1 + 50 * b))      # different lengths: different sponge positions
        proof = arg.prove(com, t.clone(), list(l), list(n))
        tv = t.clone()
        assert arg.verify(com, tv, proof)
        cs.append(b"".join(O.sc_to_bytes(v) for v in c)); rhos.append(O.sc_to_bytes(rho)); mus.append(O.sc_to_bytes(mu))
        coms.append(O.pt_to_xy64(com))
        prs.append(b"".join(O.pt_to_xy64(p) for p in proof.r)); pxs.append(b"".join(O.pt_to_xy64(p) for p in proof.x))
        pls.append(b"".join(O.sc_to_bytes(v) for v in proof.l)); pns.append(b"".join(O.sc_to_bytes(v) for v in proof.n))
        tin.append(ser(t)); tout.append(ser(tv))
    u8 = lambda lst: np.frombuffer(b"".join(lst), dtype=np.uint8).reshape(B, -1).copy()
    out.update(c=u8(cs), rho=u8(rhos), mu=u8(mus), commitments=u8(coms), proof_r=u8(prs), proof_x=u8(pxs), proof_l=u8(pls), proof_n=u8(pns),
               rounds=len(prs[0]) // 64, nl=len(pls[0]) // 32, nn=len(pns[0]) // 32, states_in=u8(tin), states_after=u8(tout))
    return out
