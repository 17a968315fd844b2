"""k_wnla_tables_split (and the rest of a generic WNLA verify) by generator-set shape: python tools/probes/wnla_shape_probe.py n  ->
bench_other.measure_wnla at n instances for (ng, nh) = (16, 32) [4 rounds], (4, 16) [2 rounds: the WNLA stage of the circuit statement
mixed_k2], (4, 4) [1 round], (64, 64) [5 rounds]; per-kernel ms.  (Round 6: the circuit verifier's round-point tables take 1.9 ms at 512
instances where the (16, 32) argument's take 0.55 -- is it the number of rounds?)"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench_other

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
args = argparse.Namespace(fb_window_bits=16, steps=10, warmup=1, no_cpu_baseline=True, total_proofs=n, workload="wnla", statement="mixed_k2")
for ng, nh in ((16, 32), (4, 16), (4, 4), (64, 64)):
    r, ok = bench_other.measure_wnla(args, n, ng=ng, nh=nh, cpu_baseline=False)
    k = r["kernels_ms_per_step"]
    print(f"ng {ng:3d} nh {nh:3d} rounds {r['config'].get('rounds')}  n {n}  {r['ms_per_step']:7.3f} ms  ok={ok}", {a.replace('k_wnla_', ''): round(b, 3) for a, b in k.items()}, flush=True)
