"""Seeded generic WNLA instances (wnla.rs shapes) built with the oracle: shared generators, per-instance c / rho / l / n, proofs by
the reference-shaped C prover.  Shapes include the reference's own test (tests.rs:139-171: N = 4) and odd lengths."""
import ctypes as C
import hashlib

import numpy as np

import bppp_oracle as O
import bppp_oracle_c as OC


def _sc(tag: bytes, *idx) -> int:
    return O.wide_reduce(hashlib.shake_256(b"bppp-wnla-cases" + tag + b"".join(int(i).to_bytes(4, "little") for i in idx)).digest(64))


def generators(ng: int, nh: int):
    g = OC.point_mul(None, O.sc_to_bytes(_sc(b"g", 0)))
    gv = [OC.point_mul(None, O.sc_to_bytes(_sc(b"gv", i))) for i in range(ng)]
    hv = [OC.point_mul(None, O.sc_to_bytes(_sc(b"hv", i))) for i in range(nh)]
    return g, gv, hv


def make(ng: int, nh: int, B: int, label: bytes = b"wnla test", mu_is_rho_sq: bool = True):
    """-> dict with generators, per-instance inputs and oracle proofs/commitments (all as numpy byte arrays)."""
    L = OC.lib()
    g, gv, hv = generators(ng, nh)
    sz = C.c_size_t
    out = {"g": g, "gv": gv, "hv": hv, "label": label, "ng": ng, "nh": nh}
    cs, rhos, mus, ls, ns, coms, prs, pxs, pls, pns = ([] for _ in range(10))
    shape = None
    for b in range(B):
        c = [_sc(b"c", b, i) for i in range(nh)]
        rho = _sc(b"rho", b)
        mu = rho * rho % O.N if mu_is_rho_sq else _sc(b"mu", b)
        l = [(i + 1 + b) % O.N for i in range(nh)] if b == 0 else [_sc(b"l", b, i) for i in range(nh)]
        n = [(8 - i + b) % O.N for i in range(ng)] if b == 0 else [_sc(b"n", b, i) for i in range(ng)]
        cb, lb, nb = (b"".join(O.sc_to_bytes(v) for v in vec) for vec in (c, l, n))
        com = C.create_string_buffer(64)
        assert L.bppp_oracle_wnla_commit(g, b"".join(gv), sz(ng), b"".join(hv), sz(nh), cb, sz(nh), O.sc_to_bytes(rho), O.sc_to_bytes(mu),
                                         lb, sz(nh), nb, sz(ng), com) == 0
        r_out, x_out = C.create_string_buffer(64 * 16), C.create_string_buffer(64 * 16)
        l_out, n_out = C.create_string_buffer(32 * 8), C.create_string_buffer(32 * 8)
        nr, nl, nn = sz(0), sz(0), sz(0)
        assert L.bppp_oracle_wnla_prove(g, b"".join(gv), sz(ng), b"".join(hv), sz(nh), cb, sz(nh), O.sc_to_bytes(rho), O.sc_to_bytes(mu),
                                        label, sz(len(label)), com.raw, lb, sz(nh), nb, sz(ng), r_out, x_out, C.byref(nr), l_out,
                                        C.byref(nl), n_out, C.byref(nn)) == 0
        if shape is None:
            shape = (nr.value, nl.value, nn.value)
        assert shape == (nr.value, nl.value, nn.value)
        cs.append(cb); rhos.append(O.sc_to_bytes(rho)); mus.append(O.sc_to_bytes(mu)); ls.append(lb); ns.append(nb); coms.append(com.raw)
        prs.append(r_out.raw[:64 * nr.value]); pxs.append(x_out.raw[:64 * nr.value])
        pls.append(l_out.raw[:32 * nl.value]); pns.append(n_out.raw[:32 * nn.value])
    arr = lambda lst, w: np.frombuffer(b"".join(lst), dtype=np.uint8).reshape(B, -1, w).copy() if lst[0] else np.zeros((B, 0, w), np.uint8)
    out.update(rounds=shape[0], nl=shape[1], nn=shape[2], c=arr(cs, 32), rho=arr(rhos, 32).reshape(B, 32), mu=arr(mus, 32).reshape(B, 32),
               l=arr(ls, 32), n=arr(ns, 32), commitments=arr(coms, 64).reshape(B, 64), proof_r=arr(prs, 64), proof_x=arr(pxs, 64),
               proof_l=arr(pls, 32), proof_n=arr(pns, 32))
    return out


def oracle_verify(case, b, commitments=None, proof_r=None, proof_x=None, proof_l=None, proof_n=None) -> int:
    L = OC.lib()
    sz = C.c_size_t
    g = lambda k, d: bytes((d if d is not None else case[k])[b].reshape(-1))
    pr, px, pl, pn = g("proof_r", proof_r), g("proof_x", proof_x), g("proof_l", proof_l), g("proof_n", proof_n)
    com = bytes((commitments if commitments is not None else case["commitments"])[b])
    return L.bppp_oracle_wnla_verify(case["g"], b"".join(case["gv"]), sz(case["ng"]), b"".join(case["hv"]), sz(case["nh"]),
                                     bytes(case["c"][b].reshape(-1)), sz(case["nh"]), bytes(case["rho"][b]), bytes(case["mu"][b]),
                                     case["label"], sz(len(case["label"])), com, pr, px, sz(len(pr) // 64), pl, sz(len(pl) // 32), pn,
                                     sz(len(pn) // 32))
