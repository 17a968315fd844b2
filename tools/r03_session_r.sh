#!/bin/bash
# round-3 session R: the small-call path (a lane per window table / per half GLV stream, a wavefront per fixed-base sum)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/${1:-r03_r}; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_verify.py tests/test_gpu_transcript.py tests/test_gpu_rlc.py tests/test_gpu_group.py tests/test_capi_harness.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
timeout 600 python tools/latency_breakdown.py 16 > $OUT/latency_breakdown_w16.txt 2>&1; echo "latency16 rc=$?" >> $OUT/log.txt
timeout 600 python tools/latency_breakdown.py 22 > $OUT/latency_breakdown_w22.txt 2>&1; echo "latency22 rc=$?" >> $OUT/log.txt
BPPP_NO_SPLIT=1 timeout 600 python tools/latency_breakdown.py 22 > $OUT/latency_breakdown_w22_nosplit.txt 2>&1; echo "latency22 nosplit rc=$?" >> $OUT/log.txt
cat $OUT/log.txt; tail -n 3 $OUT/pytest.txt; grep -v Warn $OUT/latency_breakdown_w22.txt; echo ---; grep "verify n" $OUT/latency_breakdown_w22_nosplit.txt; echo; grep "verify n" $OUT/latency_breakdown_w16.txt
