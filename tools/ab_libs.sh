cd ${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  BPPP_LIB=$PWD/bp_pp_amd/$lib python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$lib', round(d['value']), round(d['ms_per_step'],2), d['accept_bits_ok'], {a:round(b,3) for a,b in k.items()})"
done
