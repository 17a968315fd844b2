#!/bin/bash
# round-3 session AG: two-rank dry runs of the three bench workloads on one device over gloo (control flow of N > 1 only, never a measurement)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/${1:-r03_ag}; mkdir -p $OUT
export BENCH_ONE_DEVICE=1 BENCH_DIST_BACKEND=gloo
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1"
timeout 900 $L --master-port 29541 bench.py --gpus 2 --total-proofs 131072 --steps 3 --warmup 1 > $OUT/dry2_verify.json 2> $OUT/dry2_verify.err; echo "verify dry2 rc=$?" >> $OUT/log.txt
timeout 900 $L --master-port 29542 bench.py --workload prove --gpus 2 --total-proofs 16384 --steps 3 --warmup 1 > $OUT/dry2_prove.json 2> $OUT/dry2_prove.err; echo "prove dry2 rc=$?" >> $OUT/log.txt
timeout 1200 $L --master-port 29543 bench.py --workload recip256 --gpus 2 --total-proofs 16384 --steps 2 --warmup 1 --fb-window-bits 8 > $OUT/dry2_recip.json 2> $OUT/dry2_recip.err; echo "recip dry2 rc=$?" >> $OUT/log.txt
cat $OUT/log.txt
for f in verify prove recip; do python - $OUT/dry2_$f.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if '"value"' in l:
        d = json.loads(l); print(sys.argv[1].split('/')[-1], "n_gpus", d["n_gpus"], round(d["value"]), d["unit"], round(d["ms_per_step"], 2), "ms", d["scaling"], d["config"].get("parallelism"), d.get("accept_bits_ok", d.get("proofs_verify")))
PY
done
