#!/usr/bin/env python3
"""Per-kernel static ISA summary from a hipcc -save-temps .s file: instruction class counts for the whole kernel and for each
backward-branch loop (label .. branch back), including scratch (spill) traffic inside loops.
usage: isa_loops.py file.s [kernel-name-substring]"""
import re, sys
from collections import Counter
path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else ""
HALF = ("v_mad_u64_u32", "v_mad_i64_i32", "v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshl_add_u64")
def klass(op):
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu_half" if op.startswith(HALF) else "valu"
    return "other"
cur = None; kernels = {}
for line in open(path):
    m = re.match(r"^(_Z\w+|k_\w+):", line)
    if m and ("k_" in m.group(1)): cur = m.group(1); kernels[cur] = []; continue
    if cur is None: continue
    if line.startswith("\t.end_amdhsa_kernel") or re.match(r"^\s*\.size\s", line): pass
    s = line.strip()
    if re.match(r"^\.LBB\d+_\d+:", s): kernels[cur].append(("label", s[:-1].split(":")[0])); continue
    if not s or s.startswith((".", ";", "//")): continue
    op = s.split()[0]
    tgt = None
    if op.startswith("s_cbranch") or op == "s_branch":
        tgt = s.split()[1]
    kernels[cur].append(("inst", op, tgt))
for k, items in kernels.items():
    if want not in k: continue
    tot = Counter(klass(i[1]) for i in items if i[0] == "inst")
    n = sum(tot.values())
    print(f"== {k}: {n} instructions", dict(tot), f"half_rate_frac_of_valu={tot['valu_half']/max(1,tot['valu']+tot['valu_half']):.3f}")
    pos = {it[1]: idx for idx, it in enumerate(items) if it[0] == "label"}
    for idx, it in enumerate(items):
        if it[0] == "inst" and it[2] and it[2] in pos and pos[it[2]] < idx:
            body = [x for x in items[pos[it[2]]:idx + 1] if x[0] == "inst"]
            c = Counter(klass(x[1]) for x in body)
            if len(body) >= 200:
                print(f"   loop {it[2]:>12s}: {len(body):6d} insts  valu {c['valu']:6d} half {c['valu_half']:6d} vmem {c['vmem']:4d} scratch {c['scratch']:4d} salu {c['salu']:5d}")
