#!/bin/bash
# round-3 session O: 18- and 19-bit signed fixed-base windows (configs[4]'s 769 generators: 92 / 181 GB of tables against 52 GB at
# 16 bits), generic prover with the next commitments by the verifier relation
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/r03_o; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_wnla.py tests/test_gpu_circuit.py tests/test_gpu_recip.py tests/test_gpu_transcript.py tests/test_gpu_scale.py tests/test_gpu_prove.py tests/test_gpu_group.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
for W in 16 18 19; do
  timeout 900 python bench.py --workload recip256 --total-proofs 32768 --fb-window-bits $W --steps 5 --no-cpu-baseline > $OUT/recip256_15_w$W.json 2> $OUT/recip256_15_w$W.err; echo "recip256 2^15 W=$W rc=$?" >> $OUT/log.txt
done
cat $OUT/log.txt; tail -n 3 $OUT/pytest.txt
for W in 16 18 19; do python - $OUT/recip256_15_w$W.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if '"value"' in l:
        d = json.loads(l); print(sys.argv[1].split('/')[-1], round(d["value"]), d["unit"], round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["kernels_ms_per_step"].items()}, "rlc", round(d["rlc_mode"]["value"]), d["setup_s"], d["device_bytes"], d["accept_bits_ok"])
PY
done
