#!/bin/bash
# round-3 session F: phase 1 as two kernels (transcript | scalars) against the one-kernel build, phase stamps of the one-kernel build
# (where phase 1's time goes), then the GPU tests that cover phase 1 and a two-rank dry run of the headline bench on one device.
set -u
TAG=${1:-r03_f}
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
  timeout 600 $B > $OUT/bench_split_$rep.json 2> $OUT/bench_split_$rep.err; echo "split $rep rc=$?" >> $OUT/log.txt
  BPPP_LIB=$REPO/bp_pp_amd/libbppp_hip_p1old.so timeout 600 $B > $OUT/bench_one_$rep.json 2> $OUT/bench_one_$rep.err; echo "one $rep rc=$?" >> $OUT/log.txt
done
BPPP_LIB=$REPO/bp_pp_amd/libbppp_hip_pt.so timeout 600 python tools/phase_probe.py > $OUT/phase_probe.txt 2>&1; echo "probe rc=$?" >> $OUT/log.txt
timeout 1500 python -m pytest tests/test_gpu_verify.py tests/test_gpu_transcript.py tests/test_gpu_scale.py tests/test_gpu_rlc.py tests/test_gpu_group.py -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
BENCH_ONE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 2 --steps 3 --fb-window-bits 20 > $OUT/bench_dry2.json 2> $OUT/bench_dry2.err; echo "dry2 rc=$?" >> $OUT/log.txt
cat $OUT/log.txt
python tools/show_bench.py $OUT/bench_split_1.json $OUT/bench_one_1.json $OUT/bench_split_2.json $OUT/bench_one_2.json | grep -v "roofline\|setup"
cat $OUT/phase_probe.txt | tail -25
tail -4 $OUT/pytest_gpu.txt
python tools/show_bench.py $OUT/bench_dry2.json | head -4; tail -3 $OUT/bench_dry2.err
