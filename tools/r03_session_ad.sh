#!/bin/bash
# round-3 session AD: reciprocal (16, 16) calls handed to the u64 kernels
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/${1:-r03_ad}; mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_recip.py tests/test_gpu_group.py tests/test_gpu_transcript.py tests/test_gpu_scale.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
timeout 600 python tools/latency_generic.py > $OUT/latency_generic.txt 2>&1; echo "latency_generic rc=$?" >> $OUT/log.txt
cat $OUT/log.txt; grep -E "passed|failed|error" $OUT/pytest.txt | tail -3; grep -v Warn $OUT/latency_generic.txt | tail -3
