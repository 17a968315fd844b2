#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ_* passes into per-kernel averages (profiles/pmc_valu.json): VALU instructions per wave, the share of
half-rate (INT64) instructions, and wait / active fractions of the wave cycles.  usage: sq_summarize.py <dir holding the passes>
The "_meta" entry carries the SHA-256 of the library's device code the counters were collected on."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bp_pp_amd import _build   # noqa: E402

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0]
            if not k.startswith("k_"):
                continue
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
res = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in acc.items()}
for k, d in res.items():
    if d.get("SQ_WAVES"):
        d["valu_insts_per_wave"] = d.get("SQ_INSTS_VALU", 0) / d["SQ_WAVES"]
        if "SQ_INSTS_VALU_INT64" in d:
            d["int64_frac_of_valu"] = d["SQ_INSTS_VALU_INT64"] / max(1.0, d.get("SQ_INSTS_VALU", 0))
    if d.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY"):
            if c in d:
                d[c + "_frac_of_wave_cycles"] = d[c] / d["SQ_WAVE_CYCLES"]
for k, d in res.items():
    if k.startswith("k_verify") or k.startswith("k_prove"):
        print(k, {c: round(v, 4) for c, v in d.items() if c.endswith("frac_of_wave_cycles") or c in ("valu_insts_per_wave", "int64_frac_of_valu")})
res["_meta"] = {"code_object_sha256": _build.device_code_sha256()}
json.dump(res, open(os.path.join(out, "pmc_valu.json"), "w"), indent=1)
