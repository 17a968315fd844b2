#!/bin/bash
# round-3 session AB: prover output in the wire format; GPU tier on the build
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/${1:-r03_ab}; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
cat $OUT/log.txt; grep -E "passed|failed|error" $OUT/pytest.txt | tail -3
