cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_b1; mkdir -p $O
for n in 65536 131072 262144; do for B in 1 0; do
  BPPP_RECIP_BESIDE=$B timeout 900 python bench.py --workload recip256 --total-proofs $n --no-cpu-baseline --steps 6 > $O/r.json 2> $O/r.err
  python - <<P >> $O/ab.txt
import json
d=json.loads(open("$O/r.json").read().strip().splitlines()[-1])
print("n=$n beside=$B", round(d["value"]), round(d["ms_per_step"],3), "timed pass", round(d.get("timing_pass_ms_per_step",0),3), d.get("accept_bits_ok"))
P
done; done
timeout 900 python bench.py --workload recip256 --no-cpu-baseline --steps 6 > $O/default.json 2>> $O/r.err
python - <<P >> $O/ab.txt
import json
d=json.loads(open("$O/default.json").read().strip().splitlines()[-1])
print("default 2^18", round(d["value"]), round(d["ms_per_step"],3), "timed pass", round(d.get("timing_pass_ms_per_step",0),3), d.get("accept_bits_ok"))
P
cat $O/ab.txt
