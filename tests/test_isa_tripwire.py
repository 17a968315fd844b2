"""A static tripwire for the built library's gfx950 code (CPU tier: llvm-objdump cross-disassembles without a GPU).

Round 6 lost 75 % of every fixed-base kernel's speed to a header change with every test green: a second call site made the inliner
leave the software pipeline's closures as calls and their captures went to SCRATCH memory (profiles/r06/r06_q_fb_blocks_and_lambda_inlining.txt).
The disassembly shows that at once -- 150-190 `scratch_*` instructions in kernels that have 14-36 -- so this test compares, kernel by
kernel, the scratch-instruction count of the library as built with the committed static mix (profiles/isa_mix.json, the file bench.py's
roofline_valu reads; regenerate it with `python tools/isa_mix.py > profiles/isa_mix.json` whenever kernels change on purpose)."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "bp_pp_amd", "libbppp_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
HOT = ("k_verify_round", "k_verify_c0_var", "k_verify_tables", "k_verify_final_check_l1", "k_verify_final_check", "k_verify_c0_fixed_l1",
       "k_verify_c0_fixed", "k_wnla_msm", "k_wnla_msm_l1", "k_recip_c0_fixed", "k_prove_msm_l4x")


@pytest.mark.skipif(not (os.path.exists(SO) and os.path.exists(OBJDUMP) and shutil.which("c++filt")), reason="needs the built library and the ROCm llvm tools")
def test_no_kernel_gained_scratch_traffic():
    with open(os.path.join(ROOT, "profiles", "isa_mix.json")) as f:
        base = json.load(f)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py"), SO], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    cur = json.loads(out.stdout)
    for k in HOT:
        assert k in cur, f"{k} is not in the library any more: update HOT and profiles/isa_mix.json"
    worse = []
    for k, c in cur.items():
        if k not in base:
            continue                      # a new kernel: nothing to compare with until profiles/isa_mix.json is regenerated
        was, now = base[k].get("scratch", 0), c.get("scratch", 0)
        if now > was * 3 // 2 + 24:
            worse.append((k, was, now))
    assert not worse, ("kernels whose scratch-memory instruction count grew (name, committed, built) -- spilled pipeline state? "
                       f"regenerate profiles/isa_mix.json only if this is intended: {worse}")
    # the hot loops themselves stay (nearly) scratch-free whatever the baseline file says
    for k in ("k_verify_final_check_l1", "k_verify_c0_fixed_l1", "k_verify_c0_var", "k_wnla_msm", "k_recip_c0_fixed"):
        assert cur[k].get("scratch", 0) <= 64, (k, cur[k].get("scratch", 0))
