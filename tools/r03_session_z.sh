#!/bin/bash
# round-3 session Z: C0's variable-base sum on two lanes per proof for 2^15 < n <= 2^16; configs[1] before / after on one box
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/${1:-r03_z}; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_verify.py tests/test_gpu_transcript.py tests/test_gpu_rlc.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
for V in default c0var1; do
  E=BPPP_X=0; [ $V = c0var1 ] && E=BPPP_C0VAR_FORM=1
  env $E python bench.py --total-proofs 65536 --steps 20 --no-cpu-baseline > $OUT/b16_$V.json 2> $OUT/b16_$V.err; echo "b16 $V rc=$?" >> $OUT/log.txt
done
cat $OUT/log.txt; tail -n 3 $OUT/pytest.txt
python tools/show_bench.py $OUT/b16_default.json $OUT/b16_c0var1.json | grep -v "roofline\|setup\|prove_2pow14\|recip256"
