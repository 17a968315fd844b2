cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_si; mkdir -p $O
ONLY_ENV="default:;si=16:BPPP_SHARED_INV=16;si=4:BPPP_SHARED_INV=4;si=0:BPPP_SHARED_INV=0" REPS=7 timeout 1200 python tools/probes/twin_pace_probe.py 19 > $O/si.txt 2> $O/err.txt; echo "rc=$?" >> $O/log.txt
cat $O/log.txt $O/si.txt; tail -2 $O/err.txt
