"""Seeded generic reciprocal range-proof instances (reciprocal.rs, any dim_nd / dim_np) built with the oracle.
dim_nd = 16, dim_np = 16 is the u64 protocol; dim_nd = 256, dim_np = 16 is the "aggregated 16-value" shape of BASELINE configs[4]
(|g_vec| = 256, |h_vec| = 266 + 246 padding = 512, 8 WNLA rounds)."""
import ctypes as C
import hashlib

import numpy as np

import bppp_oracle as O
import bppp_oracle_c as OC


def _sc(tag: bytes, *idx) -> int:
    return O.wide_reduce(hashlib.shake_256(b"bppp-recip-cases" + tag + b"".join(int(i).to_bytes(4, "little") for i in idx)).digest(64))


def _pow2_at_least(n):
    p = 1
    while p < n:
        p *= 2
    return p


def make(dim_nd: int, dim_np: int, B: int, label: bytes = b"reciprocal test", n_oracle: int = None):
    """B seeded instances (witnesses + prover randomness).  The oracle proves the first `n_oracle` of them (default: all B);
    case["commitments"] / case["proofs"] then hold only those rows -- large batches get their proofs from the product prover and
    use the oracle rows as the cross-check sample."""
    n_oracle = B if n_oracle is None else min(B, n_oracle)
    L = OC.lib()
    sz = C.c_size_t
    nh = dim_nd + 10
    NH, NG = _pow2_at_least(nh), _pow2_at_least(dim_nd)
    pt = lambda tag, i: OC.point_mul(None, O.sc_to_bytes(_sc(tag, i)))
    g = pt(b"g", 0)
    gv = [pt(b"gv", i) for i in range(dim_nd)]
    hv = [pt(b"hv", i) for i in range(nh)]
    gv_ = [pt(b"gv_", i) for i in range(NG - dim_nd)]
    hv_ = [pt(b"hv_", i) for i in range(NH - nh)]
    case = dict(g=g, gv=gv, hv=hv, gv_=gv_, hv_=hv_, nd=dim_nd, np=dim_np, label=label, NG=NG, NH=NH)
    coms, proofs, shape = [], [], None
    n_rnd = 20 + 2 * dim_nd
    wx, ws, wd, wm, wr = [], [], [], [], []
    for b in range(B):
        dg = hashlib.shake_256(b"dig" + (bytes([b]) if b < 256 else b.to_bytes(4, "little"))).digest(2 * dim_nd) if b >= 256 else None
        digits = [int.from_bytes(dg[2 * i:2 * i + 2] if dg else
                                 hashlib.shake_256(b"dig" + bytes([b]) + i.to_bytes(4, "little")).digest(2), "little") % dim_np
                  for i in range(dim_nd)]
        if b == 0:
            digits = [0] * dim_nd
        if b == 1:
            digits = [dim_np - 1] * dim_nd
        x = sum(d * pow(dim_np, i, O.N) for i, d in enumerate(digits)) % O.N
        m = [digits.count(v) for v in range(dim_np)]
        s = _sc(b"s", b)
        if b < 256:
            rnd = b"".join(O.sc_to_bytes(_sc(b"rnd", b, i)) for i in range(n_rnd))
        else:       # large batches: one XOF call per instance, top four bits cleared (canonical without a wide reduction)
            raw = bytearray(hashlib.shake_256(b"bppp-recip-cases-rnd" + b.to_bytes(4, "little")).digest(32 * n_rnd))
            raw[0::32] = bytes(v & 0x0F for v in raw[0::32])
            rnd = bytes(raw)
        wx.append(O.sc_to_bytes(x)); ws.append(O.sc_to_bytes(s)); wd.append(b"".join(O.sc_to_bytes(d) for d in digits))
        wm.append(b"".join(O.sc_to_bytes(v) for v in m)); wr.append(rnd)
        if b >= n_oracle:
            continue
        com = C.create_string_buffer(64)
        pbuf = C.create_string_buffer(64 * (5 + 2 * 16) + 32 * 16)
        rounds, nl, nn = sz(0), sz(0), sz(0)
        rc = L.bppp_oracle_reciprocal_prove(g, b"".join(gv), sz(dim_nd), sz(dim_np), b"".join(hv), b"".join(gv_), sz(len(gv_)), b"".join(hv_),
                                            sz(len(hv_)), label, sz(len(label)), O.sc_to_bytes(x), O.sc_to_bytes(s),
                                            b"".join(O.sc_to_bytes(d) for d in digits), b"".join(O.sc_to_bytes(v) for v in m), rnd, sz(n_rnd),
                                            com, pbuf, C.byref(rounds), C.byref(nl), C.byref(nn))
        assert rc == 0, rc
        sh = (rounds.value, nl.value, nn.value)
        shape = shape or sh
        assert sh == shape
        nbytes = 64 * (5 + 2 * sh[0]) + 32 * (sh[1] + sh[2])
        coms.append(com.raw)
        proofs.append(pbuf.raw[:nbytes])
    u8 = lambda blobs, *sh: np.frombuffer(b"".join(blobs), dtype=np.uint8).reshape(B, *sh).copy()
    if not proofs:      # n_oracle == 0: shape from the sizes alone
        rounds, nl, nn = 0, NH, NG
        while nl + nn >= 6:
            nl, nn, rounds = (nl + 1) // 2, (nn + 1) // 2, rounds + 1
        shape = (rounds, nl, nn)
        proofs, coms = [bytes(64 * (5 + 2 * rounds) + 32 * (nl + nn))], [bytes(64)]
    case.update(x=u8(wx, 32), s=u8(ws, 32), digits=u8(wd, dim_nd, 32), m=u8(wm, dim_np, 32), rnd=u8(wr, n_rnd, 32))
    case.update(rounds=shape[0], nl=shape[1], nn=shape[2], proof_bytes=len(proofs[0]),
                commitments=np.frombuffer(b"".join(coms), dtype=np.uint8).reshape(len(coms), 64).copy(),
                proofs=np.frombuffer(b"".join(proofs), dtype=np.uint8).reshape(len(proofs), -1).copy(), n_oracle=n_oracle)
    return case


def oracle_verify(case, commitment: bytes, proof: bytes) -> int:
    L = OC.lib()
    sz = C.c_size_t
    return L.bppp_oracle_reciprocal_verify(case["g"], b"".join(case["gv"]), sz(case["nd"]), sz(case["np"]), b"".join(case["hv"]),
                                           b"".join(case["gv_"]), sz(len(case["gv_"])), b"".join(case["hv_"]), sz(len(case["hv_"])),
                                           case["label"], sz(len(case["label"])), commitment, proof, sz(case["rounds"]), sz(case["nl"]),
                                           sz(case["nn"]))


def make_bulk(dim_nd: int, dim_np: int, B: int, n_oracle: int = 0, label: bytes = b"reciprocal bench", seed: int = 20260, generators=None, inputs=None):
    """Large batches: the same generators as make(), witnesses drawn with numpy (Philox, seeded) instead of one hash call per
    digit -- dim_np must be 16 (hex digits, so x is one big-integer conversion per instance).  The oracle proves the first
    `n_oracle` instances from exactly these inputs (case["proofs"] / case["commitments"] hold those rows)."""
    assert dim_np == 16
    L = OC.lib()
    sz = C.c_size_t
    nh = dim_nd + 10
    NH, NG = _pow2_at_least(nh), _pow2_at_least(dim_nd)
    if generators is not None:                       # (g, gv, hv, gv_, hv_) given by the caller (bench_other.py derives them with the product)
        g, gv, hv, gv_, hv_ = generators
    else:
        pt = lambda tag, i: OC.point_mul(None, O.sc_to_bytes(_sc(tag, i)))
        g = pt(b"g", 0)
        gv = [pt(b"gv", i) for i in range(dim_nd)]
        hv = [pt(b"hv", i) for i in range(nh)]
        gv_ = [pt(b"gv_", i) for i in range(NG - dim_nd)]
        hv_ = [pt(b"hv_", i) for i in range(NH - nh)]
    n_rnd = 20 + 2 * dim_nd
    from bp_pp_amd import synth
    w = inputs if inputs is not None else synth.bulk_reciprocal_inputs(dim_nd, B, seed)   # inputs: rows the caller already drew
    x, s, digits, m, rnd = w["x"], w["s"], w["digits"], w["m"], w["rnd"]
    case = dict(g=g, gv=gv, hv=hv, gv_=gv_, hv_=hv_, nd=dim_nd, np=dim_np, label=label, NG=NG, NH=NH, x=x, s=s, digits=digits, m=m, rnd=rnd)
    rounds, nl, nn = 0, NH, NG
    while nl + nn >= 6:
        nl, nn, rounds = (nl + 1) // 2, (nn + 1) // 2, rounds + 1
    nbytes = 64 * (5 + 2 * rounds) + 32 * (nl + nn)
    coms, proofs = [], []
    for b in range(min(B, n_oracle)):
        com = C.create_string_buffer(64)
        pbuf = C.create_string_buffer(nbytes + 64)
        r_, l_, n_ = sz(0), sz(0), sz(0)
        rc = L.bppp_oracle_reciprocal_prove(g, b"".join(gv), sz(dim_nd), sz(dim_np), b"".join(hv), b"".join(gv_), sz(len(gv_)), b"".join(hv_),
                                            sz(len(hv_)), label, sz(len(label)), x[b].tobytes(), s[b].tobytes(), digits[b].tobytes(), m[b].tobytes(),
                                            rnd[b].tobytes(), sz(n_rnd), com, pbuf, C.byref(r_), C.byref(l_), C.byref(n_))
        assert rc == 0 and (r_.value, l_.value, n_.value) == (rounds, nl, nn), rc
        coms.append(com.raw); proofs.append(pbuf.raw[:nbytes])
    case.update(rounds=rounds, nl=nl, nn=nn, proof_bytes=nbytes, n_oracle=len(coms),
                commitments=np.frombuffer(b"".join(coms) or bytes(64), dtype=np.uint8).reshape(-1, 64).copy(),
                proofs=np.frombuffer(b"".join(proofs) or bytes(nbytes), dtype=np.uint8).reshape(-1, nbytes).copy())
    return case
