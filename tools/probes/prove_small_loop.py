"""20 small prove calls and 20 small verify calls (n = 64) -- the workload for `rocprofv3 --kernel-trace --stats` when looking at
where a small call's time goes kernel by kernel."""
import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import bench
from bp_pp_amd import U64RangeProofProtocol, synth
gens, g, gv, hv = bench.load_generators()
proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=16)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x, s, rnd = synth.bulk_values(n), synth.bulk_blindings(n), synth.bulk_prover_randomness(n)
for _ in range(20):
    P, V, st = proto.prove_batch(x, s, rnd, synth.LABEL)
for _ in range(20):
    acc, _ = proto.verify_batch(V, P, synth.LABEL)
assert acc.all()
