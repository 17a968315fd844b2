#!/bin/bash
# round-3 session AM: window tables with a lane per point beside phase 1 for the lane-group batch sizes (4,096 < n <= 16,384)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/${1:-r03_am}; mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_verify.py tests/test_gpu_transcript.py tests/test_gpu_rlc.py tests/test_gpu_group.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
timeout 600 python tools/latency_breakdown.py 22 > $OUT/latency_w22.txt 2>&1; echo "latency rc=$?" >> $OUT/log.txt
BPPP_NO_SPLIT=1 timeout 600 python tools/latency_breakdown.py 22 > $OUT/latency_w22_nosplit.txt 2>&1; echo "latency nosplit rc=$?" >> $OUT/log.txt
cat $OUT/log.txt; grep -E "passed|failed|error" $OUT/pytest.txt | tail -2; grep "verify n" $OUT/latency_w22.txt | grep host; echo --- nosplit; grep "verify n" $OUT/latency_w22_nosplit.txt | grep host
