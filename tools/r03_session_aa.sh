#!/bin/bash
# round-3 session AA: the prover's independent sums of a stage as one launch, at every size (A/B against one launch per sum)
# (BPPP_PROVE_SEPARATE_SUMS was the A/B switch of this session; removed afterwards, the result is in profiles/r03_aa_*)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/${1:-r03_aa}; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_prove.py tests/test_gpu_transcript.py tests/test_gpu_scale.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
for LOGN in 14 17; do
  for V in fused separate; do
    E=BPPP_X=0; [ $V = separate ] && E=BPPP_PROVE_SEPARATE_SUMS=1
    env $E python bench.py --workload prove --total-proofs $((1 << LOGN)) --steps 20 --no-cpu-baseline > $OUT/prove_${LOGN}_$V.json 2> $OUT/prove_${LOGN}_$V.err
    python - $OUT/prove_${LOGN}_$V.json prove_${LOGN}_$V <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if '"value"' in l:
        d = json.loads(l); print(sys.argv[2], round(d["value"]), d["unit"], round(d["ms_per_step"], 3), "ms", {k: round(v, 2) for k, v in d["kernels_ms_per_step"].items()}, d["proofs_verify"])
PY
  done
done
cat $OUT/log.txt; tail -n 3 $OUT/pytest.txt
