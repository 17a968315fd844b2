#!/bin/bash
# round-3 session V/W: small-call path with four parts per GLV stream up to one proof per SIMD, two parts up to four per SIMD; prover sums of a stage in one launch
# (BPPP_SPLIT_MAX was a sweep switch of these sessions; removed afterwards)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/${1:-r03_v}; mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_verify.py tests/test_gpu_transcript.py tests/test_gpu_rlc.py tests/test_gpu_group.py tests/test_capi_harness.py tests/test_gpu_prove.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
timeout 600 python tools/latency_breakdown.py 22 > $OUT/latency_w22.txt 2>&1; echo "latency22 rc=$?" >> $OUT/log.txt
BPPP_SPLIT_MAX=16384 timeout 600 python tools/latency_breakdown.py 22 > $OUT/latency_w22_max16384.txt 2>&1; echo "latency22 max16384 rc=$?" >> $OUT/log.txt
cat $OUT/log.txt; tail -n 3 $OUT/pytest.txt; grep -v Warn $OUT/latency_w22.txt; echo --- max16384; grep "verify n" $OUT/latency_w22_max16384.txt
