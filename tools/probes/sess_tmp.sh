cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_w9; mkdir -p $O
p() { python - <<P >> $O/out.txt
import json
d=json.loads(open("$O/x.json").read().strip().splitlines()[-1])
print("$1", round(d["value"]), round(d["ms_per_step"],3), "timed pass", d.get("timing_pass_ms_per_step"))
P
}
for rep in 1 2; do
for W in 1 30; do python bench.py --workload wnla --no-cpu-baseline --warmup $W > $O/x.json 2>> $O/err.txt; p "wnla warmup=$W"; done
for W in 1 30; do python bench.py --workload circuit --no-cpu-baseline --warmup $W > $O/x.json 2>> $O/err.txt; p "circuit warmup=$W"; done
for V in 1 0; do BPPP_NEXT_OVERLAP=$V python bench.py --workload prove --no-cpu-baseline --no-session-rates --warmup 10 > $O/x.json 2>> $O/err.txt; p "prove next_overlap=$V warmup=10"; done
python bench.py --workload prove --no-cpu-baseline --no-session-rates > $O/x.json 2>> $O/err.txt; p "prove default warmup=1"
done
cat $O/out.txt
