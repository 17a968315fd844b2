import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import ref_fixture_check as RC, workload
from bp_pp_amd import U64RangeProofProtocol
doc = RC.oracle_made_document(4)
cs = doc["cases"]
gens = bytes.fromhex(doc["generators"])
g, gv, hv = workload.split_generators(gens)
proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=8)
u8 = lambda sel, key, w: np.frombuffer(b"".join(bytes.fromhex(cs[i][key]) for i in sel), dtype=np.uint8).reshape(len(sel), w).copy()
for sel in ([0, 2], [1, 3], [1], [0, 1], [1, 0], [0, 1, 2, 3], [1, 1, 1, 1]):
    V, P, S = u8(sel, "commitment", 64), u8(sel, "proof", 928), u8(sel, "state_before", 203)
    acc, st, out = proto.verify_batch_transcript(V, P, [s.tobytes() for s in S])
    exp = u8(sel, "state_after_verify", 203)
    print(sel, "acc", acc.tolist(), "st", st.tolist(), "state_ok", [(out[i] == exp[i]).all() for i in range(len(sel))], "pos", S[:, 200].tolist())
# shared state, ctx case
V, P, S = u8([1], "commitment", 64), u8([1], "proof", 928), u8([1], "state_before", 203)
print("shared", proto.verify_batch_transcript(V, P, S[0].tobytes())[0].tolist())
