cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03_aj
for V in base l4 base l4; do
  E=BPPP_X=0; [ $V = l4 ] && E=BPPP_TRY_L4=1
  env $E python bench.py --workload prove --total-proofs 16384 --steps 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if '\"value\"' in l:
        d=json.loads(l); print('$V', round(d['value']), round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['kernels_ms_per_step'].items()}, d['proofs_verify'])"
done
