"""Generates tests/golden/r02_golden.json from the Python big-int oracle: fixed vectors for what round 2 added at the boundary --
serialized merlin transcripts (203 bytes) after a scripted sequence of operations, u64 proofs over pre-loaded transcripts with the
states before / after prove and verify, and derived generators.  Like u64_golden.json these pin the implementations to the oracle
and to each other across languages, not to the Rust crate (see facade/).  Run:  python tests/golden/make_golden_r02.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import bppp_oracle as O  # noqa: E402
import ref_fixture_check as RC  # noqa: E402
from transcript_cases import ser  # noqa: E402


def main():
    t = O.Transcript(b"u64 range proof")
    script, states = [], [ser(t).hex()]
    ops = [("append", b"ctx", b"order-book/7"), ("u64", b"height", 1000), ("challenge", b"c0", 32), ("append", b"long" * 11, bytes(range(256)) * 2),
           ("challenge", b"c1", 200), ("append", b"", b""), ("u64", b"n", 2**64 - 1), ("challenge", b"c2", 1)]
    for op in ops:
        if op[0] == "append":
            t.append_message(op[1], op[2]); script.append({"op": "append_message", "label": op[1].hex(), "message": op[2].hex()})
        elif op[0] == "u64":
            t.append_u64(op[1], op[2]); script.append({"op": "append_u64", "label": op[1].hex(), "value": str(op[2])})
        else:
            out = t.challenge_bytes(op[1], op[2]); script.append({"op": "challenge_bytes", "label": op[1].hex(), "n": op[2], "output": out.hex()})
        states.append(ser(t).hex())
    doc = RC.oracle_made_document(4)
    doc["about"] = "round-2 boundary vectors from oracle/bppp_oracle.py (tests/golden/make_golden_r02.py)"
    doc["transcript_script"] = {"label": b"u64 range proof".hex(), "ops": script, "states": states}
    doc["derived_generators"] = {"seed": b"bppp-bench-v1".hex(), "first_index": 0,
                                 "points": b"".join(O.pt_to_xy64(O.derive_generator(b"bppp-bench-v1", i)) for i in range(5)).hex()}
    with open(os.path.join(HERE, "r02_golden.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote r02_golden.json:", len(doc["cases"]), "cases,", len(script), "transcript ops")


if __name__ == "__main__":
    main()
