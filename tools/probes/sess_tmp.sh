cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_g1; mkdir -p $O
timeout 900 tools/probes/gatherwin 160 > $O/gatherwin.txt 2>&1; echo "gatherwin rc=$?" >> $O/log.txt
cat $O/log.txt; cat $O/gatherwin.txt
