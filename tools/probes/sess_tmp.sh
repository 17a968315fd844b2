cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_p5; mkdir -p $O
for n in 2048 4096 8192 16384; do for W in 0 1000000; do
  BPPP_GENERIC_FB_WIDE_MAX=$W timeout 600 python bench.py --workload recip256 --total-proofs $n --no-cpu-baseline --steps 10 --fb-window-bits 16 > $O/r.json 2> $O/r.err
  python - <<P >> $O/ab.txt
import json
d=json.loads(open("$O/r.json").read().strip().splitlines()[-1])
k=d["kernels_ms_per_step"]
print("n=$n wide_max=$W", round(d["value"]), round(d["ms_per_step"],3), d.get("accept_bits_ok"), "c0_fixed", round(k["k_recip_c0_fixed"],3), "msm", round(k["k_wnla_msm"],3))
P
done; done
cat $O/ab.txt
