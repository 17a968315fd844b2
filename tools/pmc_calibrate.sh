cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; OUT=$PWD/gpurun_out/pmc2; rm -rf $OUT; mkdir -p $OUT; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/cal_$C -- $GRAFT_REPO_ROOT/tools/membench > $OUT/cal_$C.log 2>&1; done
python3 $GRAFT_REPO_ROOT/tools/pmc_summarize.py $OUT | head -60
