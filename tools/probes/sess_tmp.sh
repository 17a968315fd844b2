cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_k; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/log.txt
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/log.txt
cat $O/log.txt
