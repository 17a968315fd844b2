#!/bin/bash
# round-3 session Y: forms of the variable-base kernels for batches of about one wavefront per SIMD (2^15 .. 2^17 proofs)
# (BPPP_ROUND_FORM / BPPP_C0VAR_FORM were experiment switches of commit 8d7a956; removed once the sweep was recorded in profiles/r03_y_forms.txt)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/${1:-r03_y}; mkdir -p $OUT
run() {  # name, total, env...
  local name=$1 total=$2; shift 2
  env "$@" python bench.py --total-proofs $total --steps 10 --no-cpu-baseline --no-secondary > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if '"value"' in l:
        d = json.loads(l); k = d["kernels_ms_per_step"]
        print(f"{sys.argv[2]:28s} {d['ms_per_step']:7.3f} ms  c0_var {k['k_verify_c0_var']:.3f} round {k['k_verify_round']:.3f} ok {d['accept_bits_ok']}")
PY
}
for LOGN in 15 16 17; do
  N=$((1 << LOGN))
  run b${LOGN}_default $N BPPP_X=0
  for CF in 2 4; do run b${LOGN}_c0var$CF $N BPPP_C0VAR_FORM=$CF; done
  for RF in 1 2 4 12 14; do run b${LOGN}_round$RF $N BPPP_ROUND_FORM=$RF; done
done
