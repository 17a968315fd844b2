#!/bin/bash
# One GPU-box session: microbenchmarks, GPU parity tests, bench, rocprof.  Everything is logged under gpurun_out/.
# usage: tools/gpu_session.sh [steps...]   steps in {intbench smoke pytest bench rocprof}; default: all
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT="$REPO/gpurun_out"
STEPS="${*:-intbench smoke pytest bench rocprof}"
LOG=$OUT/session.log
echo "== session $(date) steps: $STEPS" > $LOG
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8; echo "host cores: $(nproc)"; grep -m1 "model name" /proc/cpuinfo; free -g | head -2) >> $LOG 2>&1
for s in $STEPS; do
  echo "== $s" >> $LOG
  case $s in
    intbench)
      if [ ! -x tools/intbench ]; then hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o tools/intbench tools/intbench.hip >> $LOG 2>&1; fi
      timeout 300 ./tools/intbench > $OUT/intbench.txt 2>&1; echo "intbench rc=$?" >> $LOG ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $LOG ;;
    pytest)
      timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $LOG ;;
    bench)
      timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $LOG ;;
    rocprof)
      rm -rf $OUT/prof
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/rocprof_bench.json 2> $OUT/rocprof.err); echo "rocprof rc=$?" >> $LOG
      find $OUT/prof -name "*kernel_stats*" >> $LOG
      # keep only the small summaries (traces can be large)
      find $OUT/prof -name "*kernel_trace*" -size +8M -delete ;;
  esac
done
echo "---- tails"
tail -3 $OUT/smoke.txt 2>/dev/null
tail -15 $OUT/pytest_gpu.txt 2>/dev/null
cat $LOG
cat $OUT/intbench.txt 2>/dev/null
cat $OUT/bench.json 2>/dev/null
tail -5 $OUT/bench.err 2>/dev/null
