#!/bin/bash
# round-3 session L/N: generic WNLA prover (R over odd blocks; next commitments by the verifier relation)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/${1:-r03_l}; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_wnla.py tests/test_gpu_circuit.py tests/test_gpu_recip.py tests/test_gpu_transcript.py tests/test_gpu_scale.py tests/test_gpu_prove.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
timeout 1200 python bench.py --workload recip256 --steps 5 > $OUT/recip256.json 2> $OUT/recip256.err; echo "recip256 rc=$?" >> $OUT/log.txt
cat $OUT/log.txt; tail -n 3 $OUT/pytest.txt
python - $OUT/recip256.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if '"value"' in l:
        d = json.loads(l); print(round(d["value"]), d["unit"], round(d["ms_per_step"], 2), "ms", d["setup_s"], d["prover"], d["cpu_baseline"].get("prover_bytes_equal_oracle"))
PY
