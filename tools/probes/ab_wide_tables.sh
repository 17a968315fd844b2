# one library, BPPP_NO_WIDE_TABLES=1 (two table regions: 24-bit windows for g / g_vec, 22-bit for h_vec; 152 GB) against unset (windows of two
# widths, code 523: 11 additions per scalar for every generator; 210 GB), in turns on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
F="--no-cpu-baseline --no-secondary --no-session-rates"
for rep in 1 2; do
for mode in wide two_regions; do
  if [ $mode = two_regions ]; then export BPPP_NO_WIDE_TABLES=1; else unset BPPP_NO_WIDE_TABLES; fi
  python bench.py --steps 6 --warmup 1 $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; c=d['config']; print('$mode 2^20', round(d['value']), round(d['ms_per_step'],2), d['accept_bits_ok'], 'W', c.get('fb_window_bits'), c.get('fb_window_bits_hi'), {a.replace('k_verify_',''):round(b,2) for a,b in k.items()}, 'GB', round(d['device_bytes']/1e9,1))"
  for n in 65536 131072; do
    python bench.py --steps 10 --warmup 2 $F --total-proofs $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', $n, round(d['value']), round(d['ms_per_step'],3), d['accept_bits_ok'])"
  done
  python bench.py --workload prove --no-cpu-baseline --no-session-rates 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode prove 2^14', round(d['value']), round(d['ms_per_step'],3))"
done
done
