#!/bin/bash
# round-2 GPU session 1: full GPU test tier (incl. the 2^17 / 2^20 / recip256 scale tests), bench at 2^20 (W=22 and W=20), A/B of the 2-waves/SIMD build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6; echo "host cores: $(nproc)"; free -g | head -2) > $OUT/s1_box.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/s1_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/s1_box.txt
timeout 900 python bench.py --steps 5 --warmup 1 > $OUT/s1_bench_w22.json 2> $OUT/s1_bench_w22.err; echo "bench22 rc=$?" >> $OUT/s1_box.txt
timeout 900 python bench.py --steps 5 --warmup 1 --fb-window-bits 20 --no-cpu-baseline > $OUT/s1_bench_w20.json 2> $OUT/s1_bench_w20.err; echo "bench20 rc=$?" >> $OUT/s1_box.txt
BPPP_LIB=$PWD/bp_pp_amd/libbppp_hip_w2.so timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/s1_bench_w2.json 2> $OUT/s1_bench_w2.err; echo "bench_w2 rc=$?" >> $OUT/s1_box.txt
tail -15 $OUT/s1_pytest.txt
cat $OUT/s1_box.txt
for f in w22 w20 w2; do python tools/show_bench.py $OUT/s1_bench_$f.json 2>/dev/null || head -c 1500 $OUT/s1_bench_$f.json; tail -3 $OUT/s1_bench_$f.err; done
