#!/bin/bash
# Multi-GPU readiness kit, one command:   tools/scale_run.sh N [dry] [outdir]
#   (1) bench.py --gpus N under torch.distributed.run (one process per GPU, RCCL = torch's "nccl" backend) -- what the driver's SCALE run does
#   (2) tools/group_run.py --gpus N (one process, bppp_group: a host thread + stream + RCCL communicator per GPU) -- what a C-ABI caller gets
# Both verify the SAME fixed batch, split the same way, and print rccl_nranks, per-rank ms, the max over ranks, the all-reduced reject
# count and accept_bits_ok.  "dry": no second GPU needed -- (1) runs N ranks on device 0 over gloo (BENCH_ONE_DEVICE=1), (2) runs a
# one-device group through a one-rank RCCL communicator; both exercise the N > 1 control flow only and are never a measurement
# (tests/test_gpu_scale.py keeps the dry run alive).
set -u
N=${1:?number of GPUs}; MODE=${2:-real}; REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; OUT=${3:-$REPO/gpurun_out/scale_$N}
cd "$REPO"; mkdir -p "$OUT"
TOTAL=${SCALE_TOTAL_PROOFS:-1048576}; STEPS=${SCALE_STEPS:-5}
WBITS=""
# (dry: N ranks share ONE device -- explicit 16-bit tables, 3 GB per rank, instead of each rank sizing its tables to "the free HBM" at the same moment)
if [ "$MODE" = "dry" ]; then export BENCH_ONE_DEVICE=1 BENCH_DIST_BACKEND=gloo; TOTAL=${SCALE_TOTAL_PROOFS:-131072}; WBITS="--fb-window-bits 16"; fi
PORT=$(python3 -c 'import socket; s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1])')      # a port that is free right now
timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N \
  --total-proofs $TOTAL --steps $STEPS --warmup 1 --no-secondary --no-cpu-baseline $WBITS > $OUT/bench_gpus$N.json 2> $OUT/bench_gpus$N.err
RC=$?; echo "bench.py --gpus $N rc=$RC"
[ $RC -ne 0 ] && { echo "--- tail of bench_gpus$N.err"; grep -v "amdgpu.ids\|hostname of the client socket\|OMP_NUM_THREADS\|^\*\*\*" $OUT/bench_gpus$N.err | tail -60; }
timeout 1800 python tools/group_run.py --gpus $N --total-proofs $TOTAL --steps $STEPS > $OUT/group_gpus$N.json 2> $OUT/group_gpus$N.err
echo "group_run.py --gpus $N rc=$?"
python - $OUT/bench_gpus$N.json $OUT/group_gpus$N.json <<'PY'
import json, sys
for path in sys.argv[1:]:
    for l in open(path):
        if '"value"' not in l:
            continue
        d = json.loads(l)
        r = d.get("ranks") or {}
        print(path.split("/")[-1], {"n_gpus": d.get("n_gpus"), "rccl_nranks": r.get("rccl_nranks", d.get("rccl_nranks")), "backend": r.get("backend"),
                                    "per_rank_ms": r.get("per_rank_ms", d.get("per_rank_kernel_ms")), "max_ms": r.get("max_ms", d.get("max_rank_kernel_ms")),
                                    "ms_per_step": d.get("ms_per_step", d.get("ms_per_step_wall")), "value": round(d["value"]),
                                    "reject_count": d.get("reject_count_all_reduced", d.get("reject_count_on_every_device")), "accept_bits_ok": d.get("accept_bits_ok")})
PY
