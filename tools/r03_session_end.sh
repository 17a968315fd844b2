#!/bin/bash
# what the driver runs at round end, on the final build: GPU tier, smoke, the default bench line
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/${1:-r03_end}; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/log.txt
( time timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench_time.txt; echo "bench rc=$?" >> $OUT/log.txt
timeout 900 python tests/soak.py 30 13 > $OUT/soak_2pow13.txt 2>&1; echo "soak13 rc=$?" >> $OUT/log.txt
timeout 900 python tests/soak.py 20 14 > $OUT/soak_2pow14.txt 2>&1; echo "soak14 rc=$?" >> $OUT/log.txt
timeout 600 python tests/stress_mixed.py > $OUT/stress_mixed.txt 2>&1; echo "stress rc=$?" >> $OUT/log.txt
tail -n 1 $OUT/soak_2pow13.txt $OUT/soak_2pow14.txt $OUT/stress_mixed.txt
cat $OUT/log.txt; grep -E "passed|failed|error" $OUT/pytest.txt | tail -2; tail -n 2 $OUT/smoke.txt; grep real $OUT/bench_time.txt
python tools/show_bench.py $OUT/bench.json | cut -c1-260 | grep -v "setup\|kernels ms"
