#!/bin/bash
# round-3 session M: sharded prover entry points (one-device group), prover MSM with one lane per proof from 2^17 values,
# bench.py --workload prove --gpus 2 dry run (two ranks on one device over gloo: control flow only, never a measurement)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/r03_m; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_group.py tests/test_gpu_prove.py tests/test_gpu_scale.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
for LOGN in 14 16 17 18; do
  timeout 600 python bench.py --workload prove --total-proofs $((1 << LOGN)) --steps 10 --no-cpu-baseline > $OUT/prove_$LOGN.json 2> $OUT/prove_$LOGN.err; echo "prove $LOGN rc=$?" >> $OUT/log.txt
done
BPPP_FB_ONE_LANE=0 timeout 600 python bench.py --workload prove --total-proofs $((1 << 17)) --steps 10 --no-cpu-baseline > $OUT/prove_17_l8.json 2> $OUT/prove_17_l8.err; echo "prove 17 (8 lanes) rc=$?" >> $OUT/log.txt
BENCH_ONE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --workload prove --gpus 2 --total-proofs 16384 --steps 3 --warmup 1 > $OUT/prove_dry2.json 2> $OUT/prove_dry2.err; echo "prove dry2 rc=$?" >> $OUT/log.txt
cat $OUT/log.txt; tail -n 3 $OUT/pytest.txt
for f in $OUT/prove_14.json $OUT/prove_16.json $OUT/prove_17.json $OUT/prove_18.json $OUT/prove_17_l8.json $OUT/prove_dry2.json; do python - $f <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if '"value"' in l:
        d = json.loads(l); print(sys.argv[1].split('/')[-1], d["n_gpus"], round(d["value"]), d["unit"], round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["kernels_ms_per_step"].items()}, d["proofs_verify"])
PY
done
