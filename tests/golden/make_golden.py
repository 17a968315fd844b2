"""Generates tests/golden/*.json from the Python big-int oracle (oracle/bppp_oracle.py).

The reference (distributed-lab/bp-pp) holds no golden vectors and cannot be built in this environment (no cargo/rustc),
so these fixtures pin the oracle to ITSELF across languages (Python big-int <-> C limbs <-> HIP kernels) and pin the
third-party layers to their public known answers (secp256k1, Merlin).  Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import bppp_oracle as O  # noqa: E402


def hx(b):
    return b.hex()


def trace_bytes(tr):
    d, ys, cs = {}, [], []
    for k, v in tr:
        if k == "wnla_y":
            ys.append(v)
        elif k == "wnla_com":
            cs.append(v)
        else:
            d[k] = v
    chal = [d["e"], d["rho"], d["lambda"], d["beta"], d["delta"], d["tau"]] + ys
    return b"".join(O.sc_to_bytes(c) for c in chal) + O.pt_to_xy64(d["V+r"]) + b"".join(O.pt_to_xy64(c) for c in cs)


def main():
    g, gv, hv = O.synth_generators()
    pub = O.U64RangeProofProtocol(g, gv, hv)
    out = {
        "about": "u64 range proofs from oracle/bppp_oracle.py; seed b'bppp-bench-v1'; label b'u64 range proof'",
        "label": hx(O.LABEL),
        "generators": hx(b"".join(O.pt_to_xy64(p) for p in [g] + gv + hv)),
        "generator_dlogs": hx(b"".join(O.sc_to_bytes(O.synth_generator_scalar(i)) for i in range(49))),
        "cases": [],
    }
    for j in [0, 1, 2, 3, 11]:
        x, s, rnd = O.synth_value(j), O.synth_blinding(j), O.synth_rng_scalars(j)
        com = pub.commit_value(x, s)
        proof = pub.prove(x, s, O.Transcript(O.LABEL), O.ScalarRng(rnd))
        tr = []
        ok = pub.verify(com, proof, O.Transcript(O.LABEL), tr)
        assert ok
        tb = trace_bytes(tr)
        out["cases"].append({
            "index": j, "x": x, "s": hx(O.sc_to_bytes(s)), "rnd": hx(b"".join(O.sc_to_bytes(r) for r in rnd)),
            "commitment": hx(O.pt_to_xy64(com)), "proof": hx(O.u64_proof_to_bytes(proof)),
            "trace_challenges_and_points": hx(tb), "accept": True,
        })
    # negative cases derived from case index 2 (x = 123456, the value of benches/range_proof.rs:13)
    base = out["cases"][2]
    pb = bytearray(bytes.fromhex(base["proof"]))
    neg = []
    t = bytearray(pb); t[832 + 31] ^= 1
    neg.append({"what": "l0 low bit flipped", "commitment": base["commitment"], "proof": hx(bytes(t)), "accept": False, "status": 0})
    t = bytearray(pb); t[896 + 5] ^= 0x80
    neg.append({"what": "n0 bit flipped", "commitment": base["commitment"], "proof": hx(bytes(t)), "accept": False, "status": 0})
    t = bytearray(pb); t[0:64], t[64:128] = pb[64:128], pb[0:64]
    neg.append({"what": "c_l and c_r swapped", "commitment": base["commitment"], "proof": hx(bytes(t)), "accept": False, "status": 0})
    t = bytearray(pb); t[4 * 64:5 * 64] = bytes(64)
    neg.append({"what": "r[0] replaced by the identity", "commitment": base["commitment"], "proof": hx(bytes(t)), "accept": False, "status": 0})
    wrong = pub.commit_value(123457, O.synth_blinding(2))
    neg.append({"what": "commitment to x+1", "commitment": hx(O.pt_to_xy64(wrong)), "proof": base["proof"], "accept": False, "status": 0})
    t = bytearray(pb); t[10] ^= 1
    neg.append({"what": "c_l off the curve", "commitment": base["commitment"], "proof": hx(bytes(t)), "accept": False, "status": 1})
    t = bytearray(pb); t[832:864] = O.N.to_bytes(32, "big")
    neg.append({"what": "l0 = n (non-canonical scalar)", "commitment": base["commitment"], "proof": hx(bytes(t)), "accept": False, "status": 1})
    # sanity: the oracle rejects every decodable negative case
    for c in neg:
        if c["status"] == 0:
            assert not pub.verify(O.pt_from_xy64(bytes.fromhex(c["commitment"])), O.u64_proof_from_bytes(bytes.fromhex(c["proof"])),
                                  O.Transcript(O.LABEL)), c["what"]
    out["negative_cases"] = neg
    with open(os.path.join(HERE, "u64_golden.json"), "w") as f:
        json.dump(out, f, indent=1)

    # --- wnla_works shape (tests.rs:139-171): N = 4, l = [1,2,3,4], n = [8,7,6,5]
    import hashlib
    def sc(tag, i):
        return O.wide_reduce(hashlib.shake_256(b"bppp-golden-wnla" + tag + bytes([i])).digest(64))
    wg = O.pt_mul(O.G, sc(b"g", 0))
    wgv = [O.pt_mul(O.G, sc(b"gv", i)) for i in range(4)]
    whv = [O.pt_mul(O.G, sc(b"hv", i)) for i in range(4)]
    c = [sc(b"c", i) for i in range(4)]
    rho = sc(b"rho", 0)
    w = O.WeightNormLinearArgument(g=wg, g_vec=wgv, h_vec=whv, c=c, rho=rho, mu=rho * rho % O.N)
    l, n = [1, 2, 3, 4], [8, 7, 6, 5]
    com = w.commit(l, n)
    pr = w.prove(com, O.Transcript(b"wnla test"), list(l), list(n))
    assert w.verify(com, O.Transcript(b"wnla test"), pr)
    wn = {
        "label": hx(b"wnla test"), "g": hx(O.pt_to_xy64(wg)), "g_vec": hx(b"".join(map(O.pt_to_xy64, wgv))),
        "h_vec": hx(b"".join(map(O.pt_to_xy64, whv))), "c": hx(b"".join(map(O.sc_to_bytes, c))),
        "rho": hx(O.sc_to_bytes(rho)), "mu": hx(O.sc_to_bytes(rho * rho % O.N)), "l": l, "n": n,
        "commitment": hx(O.pt_to_xy64(com)), "proof_r": hx(b"".join(map(O.pt_to_xy64, pr.r))),
        "proof_x": hx(b"".join(map(O.pt_to_xy64, pr.x))), "proof_l": hx(b"".join(map(O.sc_to_bytes, pr.l))),
        "proof_n": hx(b"".join(map(O.sc_to_bytes, pr.n))),
    }
    with open(os.path.join(HERE, "wnla_golden.json"), "w") as f:
        json.dump(wn, f, indent=1)


if __name__ == "__main__":
    main()
