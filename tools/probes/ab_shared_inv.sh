cd ${GRAFT_REPO_ROOT:-/root/repo}
F="--no-cpu-baseline --no-secondary --no-session-rates"
for rep in 1 2; do
for n in 262144 524288 1048576; do
for g in 0 4 8 16; do
  BPPP_SHARED_INV=$g python bench.py --steps 8 --warmup 2 $F --total-proofs $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n', $n, 'G', $g, round(d['value']), round(d['ms_per_step'],3), d['accept_bits_ok'])"
done
done
done
