cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_w2; mkdir -p $O
VARIANTS="1:0 2:0 2:1" REPS=7 timeout 1500 python tools/probes/recip_parts_probe.py 17 18 > $O/recip_parts.txt 2> $O/recip_parts.err; echo "probe rc=$?" >> $O/log.txt
cat $O/log.txt; cat $O/recip_parts.txt; tail -3 $O/recip_parts.err
