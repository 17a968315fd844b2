#!/bin/bash
# round-3 session AH: rocprofv3 kernel stats of small calls (n = 64)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=$REPO/gpurun_out/${1:-r03_ah}; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $REPO/tools/prove_small_loop.py 64 > $OUT/run.txt 2>&1; echo "rc=$?"
F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $F $OUT/small_call_kernel_stats.csv
python3 - $OUT/small_call_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if r["Name"].startswith("k_"):
        print(f'{r["Name"][:44]:44s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.3f}')
PY
find $OUT/prof -name "*kernel_trace*" -delete
