O=$GRAFT_REPO_ROOT/gpurun_out/r04_e; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for N in 32768 65536 131072; do
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_$N -- python3 $GRAFT_REPO_ROOT/bench.py --total-proofs $N --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-session-rates > $O/sq_$N.json 2> $O/sq_$N.err
  echo "N=$N rc=$?"
  timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/tc_$N -- python3 $GRAFT_REPO_ROOT/bench.py --total-proofs $N --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-session-rates > $O/tc_$N.json 2> $O/tc_$N.err
  echo "N=$N tc rc=$?"
done
cd $GRAFT_REPO_ROOT
python3 - $O <<'PY'
import csv, glob, os, sys
from collections import defaultdict
O = sys.argv[1]
for d in sorted(glob.glob(os.path.join(O, "sq_*")) + glob.glob(os.path.join(O, "tc_*"))):
    if not os.path.isdir(d): continue
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0]
            if not k.startswith("k_verify"): continue
            a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k in ("k_verify_c0_var", "k_verify_c0_var_small", "k_verify_round", "k_verify_round_small", "k_verify_round_g2", "k_verify_tables", "k_verify_phase1", "k_verify_phase1_small"):
        if k in acc:
            print(os.path.basename(d), k, {c: round(v[0] / v[1]) for c, v in acc[k].items()})
PY
find $O -name "*counter_collection.csv" -size +2M -delete; find $O -name "*kernel_trace*" -size +2M -delete
