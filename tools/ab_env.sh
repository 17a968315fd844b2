# A/B of an environment switch on ONE box, same library: tools/ab_env.sh VAR=value [bench args...]  -> the default bench step with and without
# the variable set, twice in turns (2^20 proofs resident, per-kernel times; then 2^16 and 2^17)
cd ${GRAFT_REPO_ROOT:-/root/repo}
KV=$1; shift
F="--no-cpu-baseline --no-secondary --no-session-rates"
for rep in 1 2; do
for mode in off on; do
  if [ $mode = on ]; then export "$KV"; else unset "${KV%%=*}"; fi
  python bench.py --steps 6 --warmup 1 $F "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; c=d['config']; print('$KV $mode 2^20', round(d['value']), round(d['ms_per_step'],2), d['accept_bits_ok'], 'W', c.get('fb_window_bits'), c.get('fb_window_bits_hi'), {a.replace('k_verify_',''):round(b,2) for a,b in k.items()}, 'GB', round(d['device_bytes']/1e9,1))"
  for n in 65536 131072; do
    python bench.py --steps 10 --warmup 2 $F --total-proofs $n "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$KV $mode', $n, round(d['value']), round(d['ms_per_step'],3), d['accept_bits_ok'])"
  done
done
done
