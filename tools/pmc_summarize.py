#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch (profiles/pmc_traffic.json).
Units: FETCH_SIZE / WRITE_SIZE count kilobytes (MI355X_MICROARCH.md, HBM section).  The calibration stream (tools/membench,
1 GiB read + 1 GiB written per kernel) gives the correction factor for each access width on this chip."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bp_pp_amd import _build   # noqa: E402  (no GPU call: only reads the library file)


def load(prefix, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(out_dir, f"{prefix}_{counter}", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0]
                acc[name][0] += float(row["Counter_Value"])
                acc[name][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


# usage: pmc_summarize.py <dir> [proofs per launch of the bench passes] [kernel-name prefixes to keep, comma separated] [bench prefix]
N_PER_LAUNCH = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
KEEP = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 and sys.argv[3] else ("k_",)
BENCH_PREFIX = sys.argv[4] if len(sys.argv) > 4 else "bench"
# code_object_sha256: the kernels these counters were collected on (bench.py nulls `traffic` when the loaded library differs)
res = {"units": "bytes per launch; raw = counter * 1024", "calibration": {}, "kernels": {}, "proofs_per_launch": N_PER_LAUNCH,
       "code_object_sha256": _build.device_code_sha256()}
GIB = float(1 << 30)
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    cal = load("cal", counter)
    for k, (avg, cnt) in cal.items():
        if k.startswith("k_copy") or k.startswith("k_gather64"):
            res["calibration"].setdefault(k, {})[counter] = {"raw_bytes": avg * 1024, "true_bytes": GIB, "factor_true_over_raw": GIB / (avg * 1024) if avg else None}
    for k, (avg, cnt) in load(BENCH_PREFIX, counter).items():
        if k.startswith(KEEP):
            res["kernels"].setdefault(k, {})[counter + "_raw_bytes"] = avg * 1024
            res["kernels"][k]["launches_seen"] = cnt
# corrected traffic.  Calibration on this chip (tools/membench): coalesced streams (4 or 16 B/lane) -> FETCH_SIZE reads exactly
# 1/2 of the bytes; random per-lane record gathers (the table look-ups) -> FETCH_SIZE is exact; WRITE_SIZE is exact.
# Kernels whose reads are dominated by per-lane table gathers use the gather factor, the rest the stream factor; both
# bounds are kept so the choice is visible.
def factor(kernel, counter):
    return ((res["calibration"].get(kernel) or {}).get(counter) or {}).get("factor_true_over_raw")
f_stream = factor("k_copy_dword", "FETCH_SIZE") or 2.0
f_gather = factor("k_gather64", "FETCH_SIZE") or 1.0
wf = factor("k_copy_dword", "WRITE_SIZE") or 1.0
GATHER_DOMINATED = ("k_verify_c0_fixed", "k_verify_final_check", "k_verify_c0_fixed_l1", "k_verify_final_check_l1", "k_verify_c0_var", "k_verify_round", "k_prove_msm_x", "k_prove_msm_l4x", "k_prove_msm_l1x", "k_prove_msm_l64x", "k_prove_round_fold",
                    "k_wnla_msm", "k_recip_c0_fixed", "k_rlc_chunk")
for k, v in res["kernels"].items():
    fr, wr = v.get("FETCH_SIZE_raw_bytes", 0.0), v.get("WRITE_SIZE_raw_bytes", 0.0)
    ff = f_gather if k in GATHER_DOMINATED else f_stream
    v["hbm_bytes_per_launch"] = fr * ff + wr * wf
    v["hbm_bytes_per_launch_bounds"] = [fr * min(f_gather, f_stream) + wr * wf, fr * max(f_gather, f_stream) + wr * wf]
    v["correction"] = {"fetch_factor": ff, "write_factor": wf, "class": "gather" if k in GATHER_DOMINATED else "stream"}
    v["proofs_per_launch"] = N_PER_LAUNCH
json.dump(res, open(os.path.join(out_dir, f"pmc_traffic_{BENCH_PREFIX}.json" if BENCH_PREFIX != "bench" else "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
