#!/bin/bash
# round-3 session B (the fused final check it A/B-tested was 16 % slower -- final_check 50.7 against 41.9 + 1.9 ms -- and was removed;
# BPPP_NO_FUSED_FINAL no longer exists): microbenchmarks of the session (ratebench: issue rates + shader clock; fe52bench: FP64-FMA limb products against
# v_mad_u64_u32), then A/B of the fused final check on the headline workload, then the GPU tests that run at the sizes selecting it.
# usage: tools/r03_session_b.sh <tag>
set -u
TAG=${1:-r03_b}
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
timeout 120 ./tools/ratebench > $OUT/ratebench.json 2> $OUT/ratebench.err; echo "ratebench rc=$?" > $OUT/log.txt
timeout 300 ./tools/fe52bench > $OUT/fe52bench.txt 2>&1; echo "fe52bench rc=$?" >> $OUT/log.txt
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
  timeout 600 $B > $OUT/bench_fused_$rep.json 2> $OUT/bench_fused_$rep.err; echo "fused $rep rc=$?" >> $OUT/log.txt
  BPPP_NO_FUSED_FINAL=1 timeout 600 $B > $OUT/bench_twostage_$rep.json 2> $OUT/bench_twostage_$rep.err; echo "twostage $rep rc=$?" >> $OUT/log.txt
done
timeout 1500 python -m pytest tests/test_gpu_scale.py tests/test_gpu_verify.py tests/test_gpu_rlc.py -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
cat $OUT/log.txt $OUT/ratebench.json; cat $OUT/fe52bench.txt
python tools/show_bench.py $OUT/bench_fused_1.json $OUT/bench_twostage_1.json $OUT/bench_fused_2.json $OUT/bench_twostage_2.json
tail -5 $OUT/pytest_gpu.txt
