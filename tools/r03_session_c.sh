#!/bin/bash
# round-3 session C: A/B of k_verify_tables with block running products (one per four denominators at levels 3 and 4) at 3 and at 2
# wavefronts per SIMD, on the headline workload; then the GPU tests that cover the table builder (u64 + generic rounds).
# usage: tools/r03_session_c.sh <tag>
set -u
TAG=${1:-r03_c}
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
  timeout 600 $B > $OUT/bench_w3_$rep.json 2> $OUT/bench_w3_$rep.err; echo "w3 $rep rc=$?" >> $OUT/log.txt
  BPPP_LIB=$REPO/bp_pp_amd/libbppp_hip_tw2.so timeout 600 $B > $OUT/bench_w2_$rep.json 2> $OUT/bench_w2_$rep.err; echo "w2 $rep rc=$?" >> $OUT/log.txt
done
timeout 1500 python -m pytest tests/test_gpu_scale.py tests/test_gpu_verify.py tests/test_gpu_recip.py tests/test_gpu_wnla.py tests/test_gpu_prove.py -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
cat $OUT/log.txt
python tools/show_bench.py $OUT/bench_w3_1.json $OUT/bench_w2_1.json $OUT/bench_w3_2.json $OUT/bench_w2_2.json
tail -5 $OUT/pytest_gpu.txt
