cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_x; mkdir -p $O
ONLY_ENV="one kernel:BPPP_TABLES_STAGED=0;by stage:BPPP_TABLES_STAGED=1" REPS=11 timeout 1200 python tools/probes/twin_pace_probe.py 65536 49152 32768 24576 40000 > $O/staged.txt 2> $O/staged.err; echo "probe rc=$?" >> $O/log.txt
timeout 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_plan_boundaries.py -x -q -m gpu > $O/pytest_u64.txt 2>&1; echo "pytest rc=$?" >> $O/log.txt
cat $O/log.txt; cat $O/staged.txt; tail -3 $O/staged.err; tail -5 $O/pytest_u64.txt
