#!/bin/bash
# round-3 session K: the u64 prover's MSMs over present terms only (small scalars: reachable windows; structural zeros: skipped)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/r03_k; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_verify.py tests/test_gpu_transcript.py tests/test_gpu_scale.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
for LOGN in 14 17; do
  timeout 600 python bench.py --workload prove --total-proofs $((1 << LOGN)) --steps 10 > $OUT/prove_$LOGN.json 2> $OUT/prove_$LOGN.err; echo "prove $LOGN rc=$?" >> $OUT/log.txt
done
timeout 900 python bench.py --steps 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/log.txt
cat $OUT/log.txt; tail -n 3 $OUT/pytest.txt
for f in $OUT/prove_14.json $OUT/prove_17.json; do python - $f <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if '"value"' in l:
        d = json.loads(l); print(round(d["value"]), d["unit"], round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["kernels_ms_per_step"].items()}, d["proofs_verify"], (d.get("cpu_baseline") or {}).get("byte_identical_to_gpu"))
PY
done
python tools/show_bench.py $OUT/bench.json | grep -v roofline
