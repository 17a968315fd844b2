#!/usr/bin/env python3
"""Writes tests/golden/statements_generic.json: the STATEMENTS (no proofs, no generators) that facade/src/bin/gen_fixtures.rs feeds to the real
reference crate in its `generic` mode -- the reference's own `ac_works` circuit (tests.rs:45-136), the three further circuit shapes of
tests/circuit_cases.py (k > 1 with all four partition types; f_m with one-element vectors; f_l and f_m together, for which this
repository's oracle says the reference's own prover output does NOT verify, circuit.rs:559-614 -- a statement about the reference that
only the reference can confirm), and WNLA instances (tests.rs:139-171: N = 4 with l = [1, 2, 3, 4], n = [8, 7, 6, 5]; 16 / 32 and a
ragged 5 / 7).  Scalars are 32-byte big-endian hex; partition tables hold an index into w_o or -1 for None.

    python tests/golden/make_statements_generic.py        (run from the repository root; needs the oracle only for the scalar helpers)"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))

import circuit_cases as CC          # noqa: E402
import bppp_oracle as O             # noqa: E402

H = lambda v: O.sc_to_bytes(v % O.N).hex()


def _sc(tag: bytes, *idx) -> int:
    return O.wide_reduce(hashlib.shake_256(b"bppp-ref-statements" + tag + b"".join(int(i).to_bytes(4, "little") for i in idx)).digest(64))


def main():
    circuits = []
    for name in ("ac_works", "mixed_k2", "fm_nv1", "fl_fm"):
        st = CC.STATEMENTS[name]()
        circuits.append({
            "name": name, "label": b"circuit test".hex(), "dim_nm": st["nm"], "dim_no": st["no"], "dim_nv": st["nv"], "k": st["k"],
            "f_l": bool(st["f_l"]), "f_m": bool(st["f_m"]),
            "W_m": [[H(x) for x in row] for row in st["W_m"]], "W_l": [[H(x) for x in row] for row in st["W_l"]],
            "a_m": [H(x) for x in st["a_m"]], "a_l": [H(x) for x in st["a_l"]],
            "partition": {t: [int(i) for i in st["part"][t]] for t in CC.TYPES},
            "w_l": [H(x) for x in st["w_l"]], "w_r": [H(x) for x in st["w_r"]], "w_o": [H(x) for x in st["w_o"]],
            "v": [[H(x) for x in row] for row in st["v"]],
            "instances": 2,          # proofs per statement (fresh generators are drawn once per statement, fresh s_v / prover draws per instance)
        })
    wnla = [{"name": "wnla_works", "label": b"wnla test".hex(), "ng": 4, "nh": 4, "l": [H(v) for v in (1, 2, 3, 4)], "n": [H(v) for v in (8, 7, 6, 5)]}]
    for ng, nh in ((16, 32), (5, 7)):
        wnla.append({"name": f"wnla_{ng}_{nh}", "label": b"wnla test".hex(), "ng": ng, "nh": nh,
                     "l": [H(_sc(b"l", ng, i)) for i in range(nh)], "n": [H(_sc(b"n", ng, i)) for i in range(ng)]})
    doc = {"about": "statements for facade/src/bin/gen_fixtures.rs generic (made by tests/golden/make_statements_generic.py)", "circuits": circuits, "wnla": wnla}
    with open(os.path.join(HERE, "statements_generic.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", os.path.join(HERE, "statements_generic.json"), len(circuits), "circuits,", len(wnla), "wnla shapes")


if __name__ == "__main__":
    main()
