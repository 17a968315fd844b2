cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in 16 20 16 20; do
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --fb-window-bits $w 2>gpurun_out/ab_w_$w.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('W=$w', round(d['value']), round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items()}, d['accept_bits_ok'], d['setup_s'], d['device_bytes'])"
done
tail -3 gpurun_out/ab_w_20.err
