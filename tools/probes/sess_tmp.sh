cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_y2; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/log.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/log.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/log.txt
cat $O/log.txt; tail -3 $O/pytest_gpu.txt; tail -2 $O/smoke.txt; python tools/show_bench.py $O/bench.json 2>/dev/null | head -40
