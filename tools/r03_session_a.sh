#!/bin/bash
# round-3 session A on the GPU box: the full GPU test tier (new: sharded forms, fail-closed groups, recip256 with 16-bit windows at
# 2^15), the default bench line (now carrying prove_2pow14 / recip256_2pow15), the recip256 workload at its full 2^18 size on one
# GPU, and a two-rank dry run of its sharded form on one device (gloo; control flow only, never a measurement).
# usage: tools/r03_session_a.sh <tag> [skip-tests]
set -u
TAG=${1:-r03_a}
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8; echo "host cores: $(nproc)"; grep -m1 "model name" /proc/cpuinfo; free -g | head -2) > $OUT/box.txt 2>&1
if [ "${2:-}" != "skip-tests" ]; then
  timeout 2700 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" > $OUT/log.txt
fi
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/log.txt
timeout 1200 python bench.py --workload recip256 > $OUT/recip256.json 2> $OUT/recip256.err; echo "recip256 rc=$?" >> $OUT/log.txt
BENCH_ONE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --workload recip256 --gpus 2 --total-proofs 65536 --steps 3 --no-cpu-baseline > $OUT/recip256_dry2.json 2> $OUT/recip256_dry2.err; echo "dry2 rc=$?" >> $OUT/log.txt
tail -25 $OUT/pytest_gpu.txt 2>/dev/null
cat $OUT/log.txt $OUT/box.txt
python tools/show_bench.py $OUT/bench.json; tail -3 $OUT/bench.err
head -c 3000 $OUT/recip256.json; tail -3 $OUT/recip256.err
head -c 1500 $OUT/recip256_dry2.json; tail -5 $OUT/recip256_dry2.err
