cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_v11; mkdir -p $O
(rocm-smi --showuniqueid 2>&1 | grep -E "GPU\[0\]" | head -2) > $O/box_id.txt
F="--no-cpu-baseline --no-secondary --no-session-rates"
for rep in 1 2; do for lib in libbppp_hip_prev.so libbppp_hip.so; do
  BPPP_LIB=$PWD/bp_pp_amd/$lib python bench.py --steps 6 --warmup 1 $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$lib 2^20', round(d['value']), round(d['ms_per_step'],2), d['accept_bits_ok'], {a.replace('k_verify_',''):round(b,2) for a,b in k.items()})" >> $O/ab.txt
done; done
cat $O/box_id.txt $O/ab.txt
