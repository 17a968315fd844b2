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
    cs, rhos, mus, coms, prs, pxs, pls, pns, tin, tout, tprove, lin, nin = ([] for _ in range(13))
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
        tp = t.clone()
        proof = arg.prove(com, tp, list(l), list(n))
        tprove.append(ser(tp)); lin.append(b"".join(O.sc_to_bytes(v) for v in l)); nin.append(b"".join(O.sc_to_bytes(v) for v in n))
        tv = t.clone()
        assert arg.verify(com, tv, proof)
        cs.append(b"".join(O.sc_to_bytes(v) for v in c)); rhos.append(O.sc_to_bytes(rho)); mus.append(O.sc_to_bytes(mu))
        coms.append(O.pt_to_xy64(com))
        prs.append(b"".join(O.pt_to_xy64(p) for p in proof.r)); pxs.append(b"".join(O.pt_to_xy64(p) for p in proof.x))
        pls.append(b"".join(O.sc_to_bytes(v) for v in proof.l)); pns.append(b"".join(O.sc_to_bytes(v) for v in proof.n))
        tin.append(ser(t)); tout.append(ser(tv))
    u8 = lambda lst: np.frombuffer(b"".join(lst), dtype=np.uint8).reshape(B, -1).copy()
    out.update(c=u8(cs), rho=u8(rhos), mu=u8(mus), commitments=u8(coms), proof_r=u8(prs), proof_x=u8(pxs), proof_l=u8(pls), proof_n=u8(pns),
               rounds=len(prs[0]) // 64, nl=len(pls[0]) // 32, nn=len(pns[0]) // 32, states_in=u8(tin), states_after=u8(tout),
               states_after_prove=u8(tprove), l=u8(lin), n=u8(nin))
    return out


def recip_prover_inputs(doc):
    """The u64 cases of ref_fixture_check.oracle_made_document as ReciprocalRangeProofProtocol::prove inputs at (16, 16): x and s as
    scalars, hex digits, digit multiplicities, the 52 draws (u64_proof.rs:57-82)."""
    cs = doc["cases"]
    n = len(cs)
    sc32 = lambda v: O.sc_to_bytes(v % O.N)
    x = np.frombuffer(b"".join(sc32(int(c["x"])) for c in cs), np.uint8).reshape(n, 32).copy()
    digits = np.frombuffer(b"".join(b"".join(sc32(d) for d in O.u64_to_hex(int(c["x"]))) for c in cs), np.uint8).reshape(n, 16, 32).copy()
    m = np.frombuffer(b"".join(b"".join(sc32(d) for d in O.u64_to_hex_mapped(int(c["x"]))) for c in cs), np.uint8).reshape(n, 16, 32).copy()
    return x, digits, m


def circuit_case(name: str = "mixed_k2", B: int = 3, label: bytes = b"circuit transcript test"):
    """A generic ArithmeticCircuit statement of circuit_cases.py proved and verified by the PYTHON oracle over transcripts that
    already hold per-instance context (circuit.rs:154,260 `t: &mut Transcript`); layouts as the C ABI takes them."""
    import circuit_cases as CC
    st = CC.STATEMENTS[name]()
    nm, no, nv, k = st["nm"], st["no"], st["nv"], st["k"]
    nl, nw, nh = nv * k, 2 * nm + no, nv + 9
    NG, NH = CC._pow2_at_least(nm), CC._pow2_at_least(nh)
    gp = lambda tag, i: O.pt_mul(O.G, CC._sc(b"gen" + tag, i))
    g, gv, hv = gp(b"g", 0), [gp(b"gv", i) for i in range(nm)], [gp(b"hv", i) for i in range(nh)]
    gv_, hv_ = [gp(b"gv_", i) for i in range(NG - nm)], [gp(b"hv_", i) for i in range(NH - nh)]
    part = st["part"]

    def partition(typ, j):
        tab = part[typ]
        return tab[j] if j < len(tab) and tab[j] >= 0 else None

    ckt = O.ArithmeticCircuit(dim_nm=nm, dim_no=no, k=k, dim_nl=nl, dim_nv=nv, dim_nw=nw, g=g, g_vec=gv, h_vec=hv, W_m=st["W_m"], W_l=st["W_l"],
                              a_m=st["a_m"], a_l=st["a_l"], f_l=st["f_l"], f_m=st["f_m"], g_vec_=gv_, h_vec_=hv_, partition=partition)
    used = 18 + nv + nm
    coms, proofs, svs, rnds, tin, tprove, tverify = ([] for _ in range(7))
    shape = None
    for b in range(B):
        s_v = [CC._sc(b"tsv", b, j) for j in range(k)]
        rnd = [CC._sc(b"trnd", b, i) for i in range(used)]
        vp = [ckt.commit(st["v"][j], s_v[j]) for j in range(k)]
        t = O.Transcript(label)
        t.append_message(b"ctx", b"circuit-" + bytes([97 + b]) * (3 + 70 * b))
        tp = t.clone()
        rng = O.ScalarRng(rnd)
        pr = ckt.prove(vp, O.CircuitWitness(v=[list(r) for r in st["v"]], s_v=list(s_v), w_l=list(st["w_l"]), w_r=list(st["w_r"]), w_o=list(st["w_o"])),
                       tp, rng)
        assert rng.drawn == used
        tv = t.clone()
        assert ckt.verify(vp, tv, pr)
        sh = (len(pr.r), len(pr.l), len(pr.n))
        shape = shape or sh
        assert sh == shape and len(pr.x) == len(pr.r)
        coms.append(b"".join(O.pt_to_xy64(p) for p in vp))
        proofs.append(b"".join(O.pt_to_xy64(p) for p in [pr.c_l, pr.c_r, pr.c_o, pr.c_s] + list(pr.r) + list(pr.x)) +
                      b"".join(O.sc_to_bytes(v) for v in list(pr.l) + list(pr.n)))
        svs.append(CC._b(s_v)); rnds.append(CC._b(rnd)); tin.append(ser(t)); tprove.append(ser(tp)); tverify.append(ser(tv))
    xy = lambda pts: [O.pt_to_xy64(p) for p in pts]
    u8 = lambda blobs, *sh: np.frombuffer(b"".join(blobs), dtype=np.uint8).reshape(B, *sh).copy()
    parts = {t: np.array(part[t], np.int32) for t in CC.TYPES}
    return dict(st, g=O.pt_to_xy64(g), gv=xy(gv), hv=xy(hv), gv_=xy(gv_), hv_=xy(hv_), NG=NG, NH=NH, nl=nl, nw=nw, parts=parts,
                Wm_bytes=CC._b(x for row in st["W_m"] for x in row), Wl_bytes=CC._b(x for row in st["W_l"] for x in row),
                am_bytes=CC._b(st["a_m"]), al_bytes=CC._b(st["a_l"]), rounds=shape[0], pl=shape[1], pn=shape[2],
                commitments=u8(coms, k, 64), proofs=u8(proofs, -1), s_v=u8(svs, k, 32), rnd=u8(rnds, used, 32),
                v_bytes=u8([CC._b(x for row in st["v"] for x in row)] * B, k, nv, 32), wl_bytes=u8([CC._b(st["w_l"])] * B, nm, 32),
                wr_bytes=u8([CC._b(st["w_r"])] * B, nm, 32), wo_bytes=u8([CC._b(st["w_o"])] * B, no, 32),
                states_in=u8(tin, 203), states_after_prove=u8(tprove, 203), states_after_verify=u8(tverify, 203))
