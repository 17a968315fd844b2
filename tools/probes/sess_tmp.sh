cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_v; mkdir -p $O
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/log.txt
timeout 1500 python -m pytest tests/test_gpu_verify.py tests/test_gpu_plan_boundaries.py tests/test_gpu_scale.py tests/test_gpu_rlc.py -x -q -m gpu > $O/pytest_u64.txt 2>&1; echo "pytest rc=$?" >> $O/log.txt
cat $O/log.txt
