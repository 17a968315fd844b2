#!/bin/bash
# PMC passes (separate runs, --kernel-trace only): HBM bytes per launch of the verify kernels + a calibration stream.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$REPO/gpurun_out/pmc"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/cal_$C -- $REPO/tools/membench > $OUT/cal_$C.log 2>&1
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/bench_$C -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_$C.json 2> $OUT/bench_$C.err
  echo "$C rc=$?"
done
find $OUT -name "*counter_collection.csv" | head
python3 $REPO/tools/pmc_summarize.py $OUT
