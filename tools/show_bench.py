import sys, json
for l in sys.stdin:
    if '"value"' in l:
        d=json.loads(l); print(d['config']['proofs_per_gpu'], round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})
