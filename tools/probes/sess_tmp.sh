cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_q; mkdir -p $O
EXTRA_ENV="twin pace1:BPPP_TWIN=1,BPPP_PACE=1;twin pace2:BPPP_TWIN=1,BPPP_PACE=2;pace2:BPPP_TWIN=0,BPPP_PACE=2;twin pace2 b:BPPP_TWIN=1,BPPP_PACE=2;twin pace1 b:BPPP_TWIN=1,BPPP_PACE=1" REPS=9 timeout 900 python tools/probes/twin_pace_probe.py 17 98304 > $O/fb_pace.txt 2>&1
