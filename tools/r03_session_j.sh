#!/bin/bash
# round-3 session J: the u64 table builder as five kernels (one per pass; BPPP_TABLES_SPLIT=1) -- passes D, E at 2 waves per SIMD
# (default build) or at 3 (variant t333) -- against the single kernel
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/r03_j; mkdir -p $OUT
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
  timeout 600 $B > $OUT/bench_one_$rep.json 2> $OUT/bench_one_$rep.err; echo "one $rep rc=$?" >> $OUT/log.txt
  BPPP_TABLES_SPLIT=1 timeout 600 $B > $OUT/bench_split22_$rep.json 2> $OUT/bench_split22_$rep.err; echo "split22 $rep rc=$?" >> $OUT/log.txt
  BPPP_TABLES_SPLIT=1 BPPP_LIB=$REPO/bp_pp_amd/libbppp_hip_t333.so timeout 600 $B > $OUT/bench_split33_$rep.json 2> $OUT/bench_split33_$rep.err; echo "split33 $rep rc=$?" >> $OUT/log.txt
done
BPPP_TABLES_SPLIT=1 timeout 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_scale.py -m gpu -x -q > $OUT/pytest_split.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
cat $OUT/log.txt
python tools/show_bench.py $OUT/bench_one_1.json $OUT/bench_split22_1.json $OUT/bench_split33_1.json $OUT/bench_one_2.json $OUT/bench_split22_2.json $OUT/bench_split33_2.json | grep -v "roofline\|setup"
tail -n 3 $OUT/pytest_split.txt
