#!/usr/bin/env python3
"""Compile bppp_hip.hip with -Rpass-analysis=kernel-resource-usage and print one line per kernel
(VGPRs, AGPRs, scratch bytes per lane, occupancy, spills).  Extra hipcc flags can be passed after `--`."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNIT = os.environ.get("UNIT", "unity/bppp_unity.hip")   # e.g. UNIT=k_verify_var.hip for one kernel group
extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
out = "/tmp/resusage.so"
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-fvisibility=hidden",
       "-Rpass-analysis=kernel-resource-usage", "-o", out, os.path.join(ROOT, "bp_pp_amd/csrc", UNIT)] + extra
p = subprocess.run(cmd, capture_output=True, text=True)
if p.returncode:
    sys.stderr.write(p.stderr[-4000:]); sys.exit(1)
cur = {}
rows = []
for line in p.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
filt = [a for a in sys.argv[1:] if a != "--" and a not in extra]
print(f"{'kernel':48s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'occ':>4s} {'sgprSp':>6s} {'vgprSp':>6s}")
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    if filt and not any(f in n for f in filt): continue
    print(f"{n:48s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('SGPRs Spill','?'):>6s} {r.get('VGPRs Spill','?'):>6s}")
