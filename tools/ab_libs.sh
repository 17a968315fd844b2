# A/B of library builds on ONE box: tools/ab_libs.sh lib... (file names under bp_pp_amd/, e.g. libbppp_hip_r04.so libbppp_hip.so).
# Per library: the default verify step (2^20 proofs resident, per-kernel times), 2^16 and 2^17 proofs, prove 2^14.  Run twice round-robin
# (the chip's clocks drift over a session).
cd ${GRAFT_REPO_ROOT:-/root/repo}
F="--no-cpu-baseline --no-secondary --no-session-rates"
for rep in 1 2; do
for lib in "$@"; do
  export BPPP_LIB=$PWD/bp_pp_amd/$lib
  python bench.py --steps 6 --warmup 1 $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$lib 2^20', round(d['value']), round(d['ms_per_step'],2), d['accept_bits_ok'], {a.replace('k_verify_',''):round(b,2) for a,b in k.items()})"
  for n in 65536 131072; do
    python bench.py --steps 10 --warmup 2 $F --total-proofs $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', $n, round(d['value']), round(d['ms_per_step'],3), d['accept_bits_ok'])"
  done
  python bench.py --workload prove $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib prove', d.get('config',{}).get('workload','')[:40], round(d['value']), round(d['ms_per_step'],3))"
done
done
