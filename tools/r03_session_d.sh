#!/bin/bash
# round-3 session D: full GPU tier on the current build, recip256 RLC sweeps (bucket stage in front of the generic RLC mode) at one
# GPU's share of configs[4] (2^15) and at the whole batch (2^18), the default bench line, and kernel-trace + FETCH/WRITE PMC passes of
# the headline workload (k_verify_tables' traffic with block running products).
# usage: tools/r03_session_d.sh <tag>
set -u
TAG=${1:-r03_d}
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
timeout 2700 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" > $OUT/log.txt
timeout 900 python tools/recip_rlc_sweep.py 15 2> $OUT/recip_sweep15.err | grep '^{' > $OUT/recip_rlc_sweep_2pow15.jsonl; echo "sweep15 rc=$?" >> $OUT/log.txt
timeout 1200 python tools/recip_rlc_sweep.py 18 2> $OUT/recip_sweep18.err | grep '^{' > $OUT/recip_rlc_sweep_2pow18.jsonl; echo "sweep18 rc=$?" >> $OUT/log.txt
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/log.txt
[ -x tools/membench ] || hipcc --offload-arch=gfx950 -O3 -w -o tools/membench tools/membench.hip >> $OUT/log.txt 2>&1
cd /tmp
B="python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- $B > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?" >> $OUT/log.txt
find $OUT/prof -name "*kernel_trace*" -size +4M -delete
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/cal_$C -- $REPO/tools/membench > $OUT/cal_$C.log 2>&1
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/bench_$C -- $B > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
  echo "$C rc=$?" >> $OUT/log.txt
done
python3 $REPO/tools/pmc_summarize.py $OUT/pmc 1048576 k_verify,k_rlc,k_bkt,k_fb,k_decode > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*counter_collection.csv" -size +8M -delete
cd "$REPO"
tail -12 $OUT/pytest_gpu.txt; cat $OUT/log.txt
cat $OUT/recip_rlc_sweep_2pow15.jsonl $OUT/recip_rlc_sweep_2pow18.jsonl; tail -3 $OUT/recip_sweep15.err
python tools/show_bench.py $OUT/bench.json; tail -3 $OUT/bench.err
head -c 2500 $OUT/pmc_summary.txt
