#!/bin/bash
# round-3 session E: SQ-counter passes of the headline workload on the current build (pmc_valu), fixed-base window width A/B (20 vs
# (BPPP_PROVE_UNCAPPED was this session's A/B switch; removed afterwards: profiles/r03_e_prove_*)
# 22 bits), prover lane kernels capped / uncapped at 2^16..2^18 values, TCC hit/miss of k_wnla_msm.
# usage: tools/r03_session_e.sh <tag>
set -u
TAG=${1:-r03_e}
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
  timeout 600 $B > $OUT/bench_w20_$rep.json 2> $OUT/bench_w20_$rep.err; echo "w20 $rep rc=$?" >> $OUT/log.txt
  timeout 600 $B --fb-window-bits 22 > $OUT/bench_w22_$rep.json 2> $OUT/bench_w22_$rep.err; echo "w22 $rep rc=$?" >> $OUT/log.txt
done
for LOGN in 16 17 18; do
  N=$((1 << LOGN))
  timeout 600 python bench.py --workload prove --total-proofs $N --steps 5 --no-cpu-baseline > $OUT/prove_${LOGN}_capped.json 2> $OUT/prove_${LOGN}_capped.err; echo "prove $LOGN capped rc=$?" >> $OUT/log.txt
  BPPP_PROVE_UNCAPPED=1 timeout 600 python bench.py --workload prove --total-proofs $N --steps 5 --no-cpu-baseline > $OUT/prove_${LOGN}_uncapped.json 2> $OUT/prove_${LOGN}_uncapped.err; echo "prove $LOGN uncapped rc=$?" >> $OUT/log.txt
done
cd /tmp
P="python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq/p1 -- $P > $OUT/sq_p1.json 2> $OUT/sq_p1.err; echo "sq1 rc=$?" >> $OUT/log.txt
timeout 900 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d $OUT/sq/p2 -- $P > $OUT/sq_p2.json 2> $OUT/sq_p2.err; echo "sq2 rc=$?" >> $OUT/log.txt
R="python3 $REPO/bench.py --workload recip256 --total-proofs 32768 --steps 3 --no-cpu-baseline --no-secondary"
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/tcc -- $R > $OUT/tcc.json 2> $OUT/tcc.err; echo "tcc rc=$?" >> $OUT/log.txt
python3 - "$OUT/sq" <<'PY' > $OUT/sq_summary.txt 2>&1
import csv, glob, os, sys, json
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0]
            if not k.startswith("k_"): continue
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
res = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in acc.items()}
for k, d in res.items():
    if d.get("SQ_WAVES"):
        d["valu_insts_per_wave"] = d.get("SQ_INSTS_VALU", 0) / d["SQ_WAVES"]
        if "SQ_INSTS_VALU_INT64" in d: d["int64_frac_of_valu"] = d["SQ_INSTS_VALU_INT64"] / max(1.0, d.get("SQ_INSTS_VALU", 0))
    if d.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY"):
            if c in d: d[c + "_frac_of_wave_cycles"] = d[c] / d["SQ_WAVE_CYCLES"]
json.dump(res, open(os.path.join(out, "pmc_valu.json"), "w"), indent=1)
for k, d in res.items():
    print(k, {c: round(v, 4) for c, v in d.items() if c.endswith("frac_of_wave_cycles") or c in ("valu_insts_per_wave", "int64_frac_of_valu", "SQ_WAVES")})
PY
python3 - "$OUT/tcc" <<'PY' > $OUT/tcc_summary.txt 2>&1
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0]
            if not k.startswith("k_"): continue
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    v = {c: x[0] / x[1] for c, x in d.items()}
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    print(k, {c: round(x) for c, x in v.items()}, "hit rate", round(h / (h + m), 4) if h + m else None)
PY
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*kernel_trace*" -size +4M -delete
cd "$REPO"
cat $OUT/log.txt
python tools/show_bench.py $OUT/bench_w20_1.json $OUT/bench_w22_1.json $OUT/bench_w20_2.json $OUT/bench_w22_2.json
for f in $OUT/prove_*.json; do echo $f; python - $f <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if '"value"' in l:
        d = json.loads(l); print("  ", round(d["value"]), d["unit"], round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["kernels_ms_per_step"].items()}, d["proofs_verify"])
PY
done
cat $OUT/sq_summary.txt | cut -c1-400; cat $OUT/tcc_summary.txt | cut -c1-300; tail -3 $OUT/tcc.err
