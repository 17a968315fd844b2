#!/bin/bash
# round-2 profile session on the GPU box: tests, bench, rocprofv3 kernel stats, PMC traffic + SQ passes for the 2^20 workload,
# and the prove / recip256 workloads with their kernel stats.  usage: tools/r02_profile_session.sh <tag> [skip-tests]
set -u
TAG=${1:-r02_a}
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
N=1048576
if [ "${2:-}" != "skip-tests" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" > $OUT/log.txt
fi
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/log.txt
[ -x tools/membench ] || hipcc --offload-arch=gfx950 -O3 -o tools/membench tools/membench.hip >> $OUT/log.txt 2>&1
cd /tmp
B="python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- $B > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?" >> $OUT/log.txt
find $OUT/prof -name "*kernel_trace*" -size +4M -delete
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/cal_$C -- $REPO/tools/membench > $OUT/cal_$C.log 2>&1
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/bench_$C -- $B > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
  echo "$C rc=$?" >> $OUT/log.txt
done
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/prove_$C -- python3 $REPO/bench.py --workload prove --no-cpu-baseline > /dev/null 2> $OUT/pmc_prove_$C.err
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/recip_$C -- python3 $REPO/bench.py --workload recip256 --no-cpu-baseline > /dev/null 2> $OUT/pmc_recip_$C.err
done
# the verify bench's kernels at 2^20 proofs per launch; the prover's at 2^14; the generic reciprocal verifier's at 2^15
python3 $REPO/tools/pmc_summarize.py $OUT/pmc $N k_verify,k_rlc,k_bkt,k_fb,k_decode > $OUT/pmc_summary.txt 2>&1
python3 $REPO/tools/pmc_summarize.py $OUT/pmc 16384 k_prove prove >> $OUT/pmc_summary.txt 2>&1
python3 $REPO/tools/pmc_summarize.py $OUT/pmc 32768 k_recip,k_wnla,k_msm recip >> $OUT/pmc_summary.txt 2>&1
python3 - $OUT/pmc <<'PY'
import json, os, sys
d = sys.argv[1]
base = json.load(open(os.path.join(d, "pmc_traffic.json")))
for extra in ("pmc_traffic_prove.json", "pmc_traffic_recip.json"):
    p = os.path.join(d, extra)
    if os.path.exists(p):
        base["kernels"].update(json.load(open(p))["kernels"])
base["note"] = "proofs_per_launch is per kernel: 2^20 for the verify bench, 2^14 for --workload prove, 2^15 for --workload recip256"
json.dump(base, open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
PY
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq/p1 -- $B > $OUT/sq_p1.json 2> $OUT/sq_p1.err; echo "sq1 rc=$?" >> $OUT/log.txt
timeout 900 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d $OUT/sq/p2 -- $B > $OUT/sq_p2.json 2> $OUT/sq_p2.err; echo "sq2 rc=$?" >> $OUT/log.txt
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_INSTS_[A-Z0-9_]*" | sort -u > $OUT/sq_counters.txt
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
print(json.dumps(res, indent=1))
PY
cd "$REPO"
timeout 600 python bench.py --workload prove > $OUT/prove.json 2> $OUT/prove.err; echo "prove rc=$?" >> $OUT/log.txt
timeout 900 python bench.py --workload recip256 > $OUT/recip256.json 2> $OUT/recip256.err; echo "recip rc=$?" >> $OUT/log.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_prove -- python3 $REPO/bench.py --workload prove --no-cpu-baseline > /dev/null 2> $OUT/prof_prove.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_recip -- python3 $REPO/bench.py --workload recip256 --no-cpu-baseline > /dev/null 2> $OUT/prof_recip.err
find $OUT -name "*kernel_trace*" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
cd "$REPO"
timeout 900 python tools/rlc_sweep.py 20 2> $OUT/rlc_sweep.err | grep '^{' > $OUT/rlc_sweep.jsonl; echo "sweep rc=$?" >> $OUT/log.txt
tail -8 $OUT/pytest_gpu.txt 2>/dev/null
cat $OUT/log.txt
python tools/show_bench.py $OUT/bench.json; tail -3 $OUT/bench.err
find $OUT -name "*kernel_stats.csv" | head; du -sh $OUT
head -c 1500 $OUT/pmc_summary.txt; tail -5 $OUT/sq_p2.err
