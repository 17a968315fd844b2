#!/bin/bash
# round-3 session AK: the prover's sums on the fewest lanes per proof that still fill the chip (64 / 8 / 4 / 1), all through the fused kernels
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/${1:-r03_ak}; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_prove.py tests/test_gpu_transcript.py tests/test_gpu_scale.py tests/test_gpu_group.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
for LOGN in 12 13 14 15 16 17; do
  python bench.py --workload prove --total-proofs $((1 << LOGN)) --steps 20 --no-cpu-baseline > $OUT/prove_$LOGN.json 2> $OUT/prove_$LOGN.err
  python - $OUT/prove_$LOGN.json $LOGN <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if '"value"' in l:
        d = json.loads(l); print("2^" + sys.argv[2], round(d["value"]), d["unit"], round(d["ms_per_step"], 3), "ms", {k: round(v, 2) for k, v in d["kernels_ms_per_step"].items()}, d["proofs_verify"], d["roofline"]["kernel"], d["roofline"]["traffic"])
PY
done
cat $OUT/log.txt; grep -E "passed|failed|error" $OUT/pytest.txt | tail -2
