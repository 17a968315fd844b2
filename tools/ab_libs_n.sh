# like ab_libs.sh with a batch size: tools/ab_libs_n.sh <total_proofs> lib...
cd ${GRAFT_REPO_ROOT:-/root/repo}
n=$1; shift
for lib in "$@"; do
  BPPP_LIB=$PWD/bp_pp_amd/$lib python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-secondary --total-proofs $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$lib', $n, round(d['value']), round(d['ms_per_step'],3), d['accept_bits_ok'], {a:round(b,3) for a,b in k.items() if 'fixed' in a or 'check' in a})"
done
