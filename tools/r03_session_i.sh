#!/bin/bash
# round-3 session I: fixed-base sums with a three-deep gather pipeline (-DBPPP_FB_PIPE3) against the shipped two-deep one
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/r03_i; mkdir -p $OUT
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
  BPPP_LIB=$REPO/bp_pp_amd/libbppp_hip_pipe3.so timeout 600 $B > $OUT/bench_pipe3_$rep.json 2> $OUT/bench_pipe3_$rep.err; echo "pipe3 $rep rc=$?" >> $OUT/log.txt
  timeout 600 $B > $OUT/bench_pipe2_$rep.json 2> $OUT/bench_pipe2_$rep.err; echo "pipe2 $rep rc=$?" >> $OUT/log.txt
done
BPPP_LIB=$REPO/bp_pp_amd/libbppp_hip_pipe3.so timeout 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_prove.py tests/test_gpu_recip.py -m gpu -x -q > $OUT/pytest_pipe3.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
cat $OUT/log.txt
python tools/show_bench.py $OUT/bench_pipe3_1.json $OUT/bench_pipe2_1.json $OUT/bench_pipe3_2.json $OUT/bench_pipe2_2.json | grep -v "roofline\|setup"
tail -n 3 $OUT/pytest_pipe3.txt
