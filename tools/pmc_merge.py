#!/usr/bin/env python3
"""Fold the per-workload traffic summaries of one session (pmc_traffic_prove.json, pmc_traffic_recip.json) into its
pmc_traffic.json.  usage: pmc_merge.py <dir>"""
import json
import os
import sys

d = sys.argv[1]
base = json.load(open(os.path.join(d, "pmc_traffic.json")))
for extra in ("pmc_traffic_prove.json", "pmc_traffic_recip.json"):
    p = os.path.join(d, extra)
    if os.path.exists(p):
        base["kernels"].update(json.load(open(p))["kernels"])
base["note"] = ("proofs_per_launch is per kernel: 2^20 for the verify bench, 2^14 for --workload prove, 2^15 for --workload recip256 "
                "--total-proofs 32768")
json.dump(base, open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
print("merged", sorted(base["kernels"]))
