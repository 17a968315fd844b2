set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out/r03_h; OUT=gpurun_out/r03_h
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
  timeout 600 $B > $OUT/bench_lds_$rep.json 2> $OUT/bench_lds_$rep.err; echo "lds $rep rc=$?" >> $OUT/log.txt
  BPPP_LIB=$REPO/bp_pp_amd/libbppp_hip_base.so timeout 600 $B > $OUT/bench_base_$rep.json 2> $OUT/bench_base_$rep.err; echo "base $rep rc=$?" >> $OUT/log.txt
done
timeout 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_transcript.py -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
cat $OUT/log.txt
python tools/show_bench.py $OUT/bench_lds_1.json $OUT/bench_base_1.json $OUT/bench_lds_2.json $OUT/bench_base_2.json | grep -v "roofline\|setup"
tail -4 $OUT/pytest_gpu.txt
