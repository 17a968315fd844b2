#!/bin/bash
# round-2 GPU session 2: full GPU test tier, then A/B of occupancy variants at 2^20 (no secondaries), then prove / recip256 workloads
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/s2_pytest.txt 2>&1; echo "pytest rc=$?" > $OUT/s2_log.txt
for v in "" _w2 _w3 _w2f3 _w3f3; do
  BPPP_LIB=$PWD/bp_pp_amd/libbppp_hip$v.so timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/s2_bench$v.json 2> $OUT/s2_bench$v.err; echo "bench$v rc=$?" >> $OUT/s2_log.txt
done
timeout 600 python bench.py --workload prove > $OUT/s2_prove.json 2> $OUT/s2_prove.err; echo "prove rc=$?" >> $OUT/s2_log.txt
timeout 900 python bench.py --workload recip256 > $OUT/s2_recip.json 2> $OUT/s2_recip.err; echo "recip rc=$?" >> $OUT/s2_log.txt
tail -12 $OUT/s2_pytest.txt
cat $OUT/s2_log.txt
for v in "" _w2 _w3 _w2f3 _w3f3; do echo "== variant '$v'"; python tools/show_bench.py $OUT/s2_bench$v.json; tail -2 $OUT/s2_bench$v.err; done
head -c 3000 $OUT/s2_prove.json; tail -3 $OUT/s2_prove.err
head -c 3000 $OUT/s2_recip.json; tail -3 $OUT/s2_recip.err
