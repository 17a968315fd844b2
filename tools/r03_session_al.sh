#!/bin/bash
# round-3 session AL: next commitments as fixed-base sums up to 2^14 values, fused with the next round's X | R (A/B against BPPP_NO_SPLIT)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/${1:-r03_al}; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_prove.py tests/test_gpu_transcript.py tests/test_gpu_group.py tests/test_capi_harness.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
for LOGN in 11 12 13 14 15; do
  for V in default nosplit; do
    E=BPPP_X=0; [ $V = nosplit ] && E=BPPP_NO_SPLIT=1
    env $E python bench.py --workload prove --total-proofs $((1 << LOGN)) --steps 20 --no-cpu-baseline > $OUT/prove_${LOGN}_$V.json 2> $OUT/prove_${LOGN}_$V.err
    python - $OUT/prove_${LOGN}_$V.json "2^$LOGN $V" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if '"value"' in l:
        d = json.loads(l); print(sys.argv[2], round(d["value"]), d["unit"], round(d["ms_per_step"], 3), "ms", {k: round(v, 2) for k, v in d["kernels_ms_per_step"].items()}, d["proofs_verify"])
PY
  done
done
timeout 600 python tools/latency_breakdown.py 22 > $OUT/latency_w22.txt 2>&1
grep "prove n" $OUT/latency_w22.txt
cat $OUT/log.txt; grep -E "passed|failed|error" $OUT/pytest.txt | tail -2
