cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_l3; mkdir -p $O
for wl in circuit wnla; do for n in 512 1024 2048 4096 16384; do for NS in 0 1; do
  if [ $NS = 1 ]; then export BPPP_NO_SPLIT=1; else unset BPPP_NO_SPLIT; fi
  timeout 300 python bench.py --workload $wl --total-proofs $n --no-cpu-baseline --steps 10 --fb-window-bits 16 > $O/c.json 2>> $O/err.txt; python - <<P >> $O/sizes.txt
import json
d=json.loads(open("$O/c.json").read().strip().splitlines()[-1]); k=d["kernels_ms_per_step"]
print("$wl n=$n no_split=$NS", round(d["value"]), round(d["ms_per_step"],3), d.get("accept_bits_ok"), {a.replace("k_",""):round(b,3) for a,b in k.items() if b>0.05})
P
done; done; done
unset BPPP_NO_SPLIT
cat $O/sizes.txt
