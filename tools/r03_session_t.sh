#!/bin/bash
# round-3 session T: small prove calls with the next commitments as fixed-base sums
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/${1:-r03_t}; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_prove.py tests/test_gpu_transcript.py tests/test_gpu_group.py tests/test_capi_harness.py tests/test_gpu_verify.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
timeout 600 python tools/latency_breakdown.py 22 > $OUT/latency_w22.txt 2>&1; echo "latency22 rc=$?" >> $OUT/log.txt
cat $OUT/log.txt; tail -n 3 $OUT/pytest.txt; grep "prove n\|kernels" $OUT/latency_w22.txt | tail -8
