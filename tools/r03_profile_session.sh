#!/bin/bash
# round-3 profile session on the GPU box: full GPU test tier, the default bench line, rocprofv3 kernel stats of the same command, PMC
# traffic (FETCH / WRITE in separate passes, calibrated) and SQ counter passes for the 2^20 workload, the prove / recip256 workloads
# with their kernel stats, one GPU's share of configs[2] (2^17 proofs), soaks.   usage: tools/r03_profile_session.sh <tag> [skip-tests]
set -u
TAG=${1:-r03_p}
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8; echo "host cores: $(nproc)"; grep -m1 "model name" /proc/cpuinfo; free -g | head -2) > $OUT/box.txt 2>&1
if [ "${2:-}" != "skip-tests" ]; then
  timeout 2700 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" > $OUT/log.txt
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/log.txt
fi
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/log.txt
timeout 600 python bench.py --total-proofs 131072 --no-cpu-baseline --no-secondary > $OUT/bench_shard17.json 2> $OUT/bench_shard17.err; echo "shard17 rc=$?" >> $OUT/log.txt
timeout 600 python bench.py --workload prove > $OUT/prove.json 2> $OUT/prove.err; echo "prove rc=$?" >> $OUT/log.txt
timeout 1200 python bench.py --workload recip256 > $OUT/recip256.json 2> $OUT/recip256.err; echo "recip rc=$?" >> $OUT/log.txt
[ -x tools/membench ] || hipcc --offload-arch=gfx950 -O3 -w -o tools/membench tools/membench.hip >> $OUT/log.txt 2>&1
cd /tmp
B="python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $REPO/bench.py --no-cpu-baseline --no-secondary > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?" >> $OUT/log.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_prove -- python3 $REPO/bench.py --workload prove --no-cpu-baseline > /dev/null 2> $OUT/prof_prove.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_recip -- python3 $REPO/bench.py --workload recip256 --total-proofs 32768 --no-cpu-baseline > /dev/null 2> $OUT/prof_recip.err
find $OUT -name "*kernel_trace*" -size +4M -delete
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/cal_$C -- $REPO/tools/membench > $OUT/cal_$C.log 2>&1
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/bench_$C -- $B > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
  echo "$C rc=$?" >> $OUT/log.txt
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/prove_$C -- python3 $REPO/bench.py --workload prove --no-cpu-baseline > /dev/null 2> $OUT/pmc_prove_$C.err
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/recip_$C -- python3 $REPO/bench.py --workload recip256 --total-proofs 32768 --no-cpu-baseline > /dev/null 2> $OUT/pmc_recip_$C.err
done
python3 $REPO/tools/pmc_summarize.py $OUT/pmc 1048576 k_verify,k_rlc,k_bkt,k_fb,k_decode > $OUT/pmc_summary.txt 2>&1
python3 $REPO/tools/pmc_summarize.py $OUT/pmc 16384 k_prove prove >> $OUT/pmc_summary.txt 2>&1
python3 $REPO/tools/pmc_summarize.py $OUT/pmc 32768 k_recip,k_wnla,k_msm,k_bkt recip >> $OUT/pmc_summary.txt 2>&1
python3 - $OUT/pmc <<'PY'
import json, os, sys
d = sys.argv[1]
base = json.load(open(os.path.join(d, "pmc_traffic.json")))
for extra in ("pmc_traffic_prove.json", "pmc_traffic_recip.json"):
    p = os.path.join(d, extra)
    if os.path.exists(p):
        base["kernels"].update(json.load(open(p))["kernels"])
base["note"] = "proofs_per_launch is per kernel: 2^20 for the verify bench, 2^14 for --workload prove, 2^15 for --workload recip256 --total-proofs 32768"
json.dump(base, open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
PY
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq/p1 -- $B > $OUT/sq_p1.json 2> $OUT/sq_p1.err; echo "sq1 rc=$?" >> $OUT/log.txt
timeout 900 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d $OUT/sq/p2 -- $B > $OUT/sq_p2.json 2> $OUT/sq_p2.err; echo "sq2 rc=$?" >> $OUT/log.txt
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
    if k.startswith("k_verify"):
        print(k, {c: round(v, 4) for c, v in d.items() if c.endswith("frac_of_wave_cycles") or c in ("valu_insts_per_wave", "int64_frac_of_valu")})
PY
find $OUT -name "*counter_collection.csv" -size +8M -delete
cd "$REPO"
timeout 900 python tests/soak.py 40 16 > $OUT/soak_2pow16.txt 2>&1; echo "soak16 rc=$?" >> $OUT/log.txt
timeout 900 python tests/soak.py 6 20 > $OUT/soak_2pow20.txt 2>&1; echo "soak20 rc=$?" >> $OUT/log.txt
timeout 900 python tests/soak.py 200 1 > $OUT/soak_2pow1.txt 2>&1; echo "soak1 rc=$?" >> $OUT/log.txt
timeout 900 python tests/soak.py 200 5 > $OUT/soak_2pow5.txt 2>&1; echo "soak5 rc=$?" >> $OUT/log.txt
timeout 900 python tests/soak.py 100 10 > $OUT/soak_2pow10.txt 2>&1; echo "soak10 rc=$?" >> $OUT/log.txt
timeout 900 python tests/soak.py 60 12 > $OUT/soak_2pow12.txt 2>&1; echo "soak12 rc=$?" >> $OUT/log.txt
timeout 600 python tools/latency_breakdown.py 22 > $OUT/latency_w22.txt 2>&1; echo "latency rc=$?" >> $OUT/log.txt
timeout 900 python tests/stress_mixed.py > $OUT/stress_mixed.txt 2>&1; echo "stress rc=$?" >> $OUT/log.txt
timeout 900 python tests/soak_generic.py > $OUT/soak_generic.txt 2>&1; echo "soak_generic rc=$?" >> $OUT/log.txt
tail -12 $OUT/pytest_gpu.txt 2>/dev/null; tail -2 $OUT/smoke.txt 2>/dev/null
cat $OUT/log.txt
python tools/show_bench.py $OUT/bench.json $OUT/bench_shard17.json; tail -3 $OUT/bench.err
python tools/show_bench.py $OUT/recip256.json | head -5
head -c 600 $OUT/prove.json; echo
cat $OUT/sq_summary.txt | cut -c1-330
tail -3 $OUT/soak_2pow16.txt $OUT/soak_2pow20.txt $OUT/stress_mixed.txt $OUT/soak_generic.txt
find $OUT -name "*kernel_stats.csv" | head; du -sh $OUT
