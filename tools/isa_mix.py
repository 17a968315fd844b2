#!/usr/bin/env python3
"""Static instruction mix of every kernel in the shipped code object: disassembles bp_pp_amd/libbppp_hip.so (llvm-objdump on the
embedded gfx950 code object) and counts, per kernel, VALU instructions and how many of them are HALF-RATE on CDNA4 (64-bit
multiply-add v_mad_u64_u32, 32-bit full multiplies, 64-bit shifts / adds -- 4 cycles per wave64 instruction against 2).  bench.py
uses profiles/isa_mix.json for the per-kernel VALU issue ceiling (roofline_valu) instead of a hard-coded mix.  The count is static
(every instruction once); the kernels are straight-line field arithmetic inlined many times, so the static mix tracks the
dynamic one closely -- tools/isa_loops.py gives the same split for the hot loops alone.
usage: python tools/isa_mix.py [path/to/lib.so] > profiles/isa_mix.json"""
import json, os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "bp_pp_amd", "libbppp_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
HALF = ("v_mad_u64_u32", "v_mad_i64_i32", "v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32",
        "v_lshl_add_u64", "v_add_u64", "v_sub_u64")
with tempfile.TemporaryDirectory() as td:
    # the fat binary holds one bundle per translation unit; clang-offload-bundler cannot list them all at once, so carve the ELF
    # images out of the .hip_fatbin section by their magic
    raw = os.path.join(td, "fatbin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, raw])
    blob = open(raw, "rb").read()
    out = {}
    pos = 0
    idx = 0
    while True:
        i = blob.find(b"\x7fELF", pos)
        if i < 0:
            break
        # ELF64 header: e_shoff at 0x28, e_shentsize 0x3A, e_shnum 0x3C -> image size
        e_shoff = int.from_bytes(blob[i + 0x28:i + 0x30], "little")
        e_shentsize = int.from_bytes(blob[i + 0x3A:i + 0x3C], "little")
        e_shnum = int.from_bytes(blob[i + 0x3C:i + 0x3E], "little")
        size = e_shoff + e_shentsize * e_shnum
        pos = i + 4
        if blob[i + 0x12:i + 0x14] != (224).to_bytes(2, "little"):   # EM_AMDGPU
            continue
        co = os.path.join(td, f"co{idx}.elf"); idx += 1
        open(co, "wb").write(blob[i:i + size])
        pos = i + size
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
        cur = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
            if m:
                name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
                cur = out.setdefault(name, Counter()) if name.startswith("k_") else None
                continue
            if cur is None:
                continue
            t = line.strip().split()
            if not t:
                continue
            op = t[0]
            if op.startswith("v_"):
                cur["valu"] += 1
                if op.startswith(HALF):
                    cur["valu_half_rate"] += 1
                if op.startswith("v_mad_u64_u32"):
                    cur["v_mad_u64_u32"] += 1
            elif op.startswith("s_"):
                cur["salu"] += 1
            elif op.startswith(("global_", "flat_", "buffer_")):
                cur["vmem"] += 1
            elif op.startswith("scratch_"):
                cur["scratch"] += 1
            elif op.startswith("ds_"):
                cur["lds"] += 1
res = {}
for k, c in sorted(out.items()):
    if not c["valu"]:
        continue
    res[k] = dict(c)
    res[k]["half_rate_frac"] = c["valu_half_rate"] / c["valu"]
    res[k]["mad_u64_frac"] = c["v_mad_u64_u32"] / c["valu"]
print(json.dumps(res, indent=1))
