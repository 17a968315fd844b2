#!/bin/bash
# ONE GPU-box session script: tools/session.sh <tag> <step> [<step> ...]  -> everything under gpurun_out/<tag>/ (log.txt holds the return
# codes).  Run it through gpurun:   gpurun --timeout 3000 -- 'bash tools/session.sh r04_p tests bench prof pmc sq'
# Steps (any order, each bounded by its own timeout):
#   box        rocminfo / host summary (always run first)
#   tests      python -m pytest tests -m gpu          smoke      __graft_entry__.smoke()
#   bench      python bench.py (the default line)      shard17    one GPU's share of the 8-GPU split (2^17 proofs)
#   prove      --workload prove                        recip      --workload recip256
#   prof       rocprofv3 --kernel-trace --stats of the default bench command (no secondaries), plus prove / recip256 (prof_prove, prof_recip)
#   pmc        FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only) + the calibration stream -> pmc/pmc_traffic.json
#   sq         SQ_* counter passes -> sq/pmc_valu.json            (both summaries carry the code-object hash of the library they ran)
#   cc         tools/concurrent_callers.py (single-proof callers through the coalescing front end)      cc_prove   the same with prove_one
#   soak       tests/soak.py at 2^1 .. 2^20, stress_mixed, soak_generic
#   latency    tools/latency_breakdown.py 22
#   dry2       two-rank gloo dry runs of the three workloads on one device (control flow of N > 1 only)
#   cmd:<...>  any other command line (quoted), logged as cmd_<n>.txt
set -u
TAG=${1:?tag}; shift
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
LOG=$OUT/log.txt
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8; python3 tools/hostinfo.py; free -g | head -2) > $OUT/box.txt 2>&1
B="python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-session-rates"
NCMD=0
for STEP in "$@"; do
  echo "== $STEP $(date +%T)" >> $LOG
  case "$STEP" in
    box) ;;
    tests) timeout 3000 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $LOG; tail -5 $OUT/pytest_gpu.txt ;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $LOG; tail -2 $OUT/smoke.txt ;;
    bench) timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $LOG; python tools/show_bench.py $OUT/bench.json; tail -3 $OUT/bench.err ;;
    shard17) timeout 600 python bench.py --total-proofs 131072 --no-cpu-baseline --no-secondary > $OUT/bench_shard17.json 2> $OUT/bench_shard17.err; echo "shard17 rc=$?" >> $LOG; python tools/show_bench.py $OUT/bench_shard17.json ;;
    prove) timeout 600 python bench.py --workload prove > $OUT/prove.json 2> $OUT/prove.err; echo "prove rc=$?" >> $LOG; head -c 500 $OUT/prove.json; echo ;;
    recip) timeout 1200 python bench.py --workload recip256 > $OUT/recip256.json 2> $OUT/recip256.err; echo "recip rc=$?" >> $LOG; python tools/show_bench.py $OUT/recip256.json | head -5 ;;
    prof)
      (cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $REPO/bench.py --no-cpu-baseline --no-secondary --no-session-rates > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?" >> $LOG)
      find $OUT -name "*kernel_trace*" -size +4M -delete ;;
    prof_prove) (cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_prove -- python3 $REPO/bench.py --workload prove --no-cpu-baseline --no-session-rates > /dev/null 2> $OUT/prof_prove.err; echo "rocprof prove rc=$?" >> $LOG); find $OUT -name "*kernel_trace*" -size +4M -delete ;;
    prof_recip) (cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_recip -- python3 $REPO/bench.py --workload recip256 --total-proofs 32768 --no-cpu-baseline --no-session-rates > /dev/null 2> $OUT/prof_recip.err; echo "rocprof recip rc=$?" >> $LOG); find $OUT -name "*kernel_trace*" -size +4M -delete ;;
    pmc)
      [ -x tools/membench ] || hipcc --offload-arch=gfx950 -O3 -w -o tools/membench tools/membench.hip >> $LOG 2>&1
      (cd /tmp
       for C in FETCH_SIZE WRITE_SIZE; do
         timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/cal_$C -- $REPO/tools/membench > $OUT/cal_$C.log 2>&1
         timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/bench_$C -- $B > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err; echo "$C rc=$?" >> $LOG
         timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/prove_$C -- python3 $REPO/bench.py --workload prove --no-cpu-baseline --no-session-rates > /dev/null 2> $OUT/pmc_prove_$C.err
         timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/recip_$C -- python3 $REPO/bench.py --workload recip256 --total-proofs 32768 --no-cpu-baseline --no-session-rates > /dev/null 2> $OUT/pmc_recip_$C.err
       done)
      python3 tools/pmc_summarize.py $OUT/pmc 1048576 k_verify,k_rlc,k_bkt,k_fb,k_decode > $OUT/pmc_summary.txt 2>&1
      python3 tools/pmc_summarize.py $OUT/pmc 16384 k_prove prove >> $OUT/pmc_summary.txt 2>&1
      python3 tools/pmc_summarize.py $OUT/pmc 32768 k_recip,k_wnla,k_msm,k_bkt recip >> $OUT/pmc_summary.txt 2>&1
      python3 tools/pmc_merge.py $OUT/pmc >> $OUT/pmc_summary.txt 2>&1
      find $OUT -name "*counter_collection.csv" -size +8M -delete ;;
    sq)
      (cd /tmp
       timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq/p1 -- $B > $OUT/sq_p1.json 2> $OUT/sq_p1.err; echo "sq1 rc=$?" >> $LOG
       timeout 900 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d $OUT/sq/p2 -- $B > $OUT/sq_p2.json 2> $OUT/sq_p2.err; echo "sq2 rc=$?" >> $LOG
       timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d $OUT/sq/p3 -- python3 $REPO/bench.py --workload prove --no-cpu-baseline --no-session-rates > /dev/null 2> $OUT/sq_p3.err; echo "sq3 rc=$?" >> $LOG)
      python3 tools/sq_summarize.py $OUT/sq > $OUT/sq_summary.txt 2>&1; cut -c1-300 $OUT/sq_summary.txt
      find $OUT -name "*counter_collection.csv" -size +8M -delete ;;
    cc) timeout 600 python tools/concurrent_callers.py --json $OUT/concurrent_callers.json 2>&1 | grep -v amdgpu.ids | tee $OUT/concurrent_callers.txt; echo "cc rc=$?" >> $LOG ;;
    cc_prove) timeout 600 python tools/concurrent_callers.py --prove --json $OUT/concurrent_callers_prove.json 2>&1 | grep -v amdgpu.ids | tee $OUT/concurrent_callers_prove.txt; echo "cc_prove rc=$?" >> $LOG ;;
    soak)
      for S in "200 1" "200 5" "100 10" "60 12" "40 16" "6 20"; do set -- $S; timeout 900 python tests/soak.py $1 $2 > $OUT/soak_2pow$2.txt 2>&1; echo "soak$2 rc=$?" >> $LOG; done
      timeout 900 python tests/stress_mixed.py > $OUT/stress_mixed.txt 2>&1; echo "stress rc=$?" >> $LOG
      timeout 900 python tests/soak_generic.py > $OUT/soak_generic.txt 2>&1; echo "soak_generic rc=$?" >> $LOG
      for f in soak_2pow16 soak_2pow20 stress_mixed soak_generic; do tail -n 2 $OUT/$f.txt; done ;;
    latency) timeout 600 python tools/latency_breakdown.py 22 > $OUT/latency_w22.txt 2>&1; echo "latency rc=$?" >> $LOG ;;
    dry2) bash tools/scale_run.sh 2 dry $OUT >> $LOG 2>&1 ;;
    cmd:*) NCMD=$((NCMD+1)); timeout 1800 bash -c "${STEP#cmd:}" > $OUT/cmd_$NCMD.txt 2>&1; echo "cmd_$NCMD rc=$? : ${STEP#cmd:}" >> $LOG; tail -20 $OUT/cmd_$NCMD.txt ;;
    *) echo "unknown step $STEP" >> $LOG ;;
  esac
done
cat $LOG
du -sh $OUT
