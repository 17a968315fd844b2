"""The library's own remedies for launches that fill the chip only once or twice (plan_core.h: twin, pace), A/B per batch size on one box:
child contexts over one table set, each created under its own BPPP_TWIN / BPPP_PACE setting, timed in turns (round-robin over the
configurations, REPS rounds, median per configuration) so that clock drift hits all alike.

    python tools/probes/twin_pace_probe.py [n ...]      (sizes as log2 or plain numbers; default 17 18 19 20)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from bp_pp_amd import U64RangeProofProtocol, synth

REPS = int(os.environ.get("REPS", "9"))
CONFIGS = [("library default", {}), ("twin=0 pace=0", {"BPPP_TWIN": 0, "BPPP_PACE": 0}), ("twin=1", {"BPPP_TWIN": 1, "BPPP_PACE": 0}),
           ("pace=1", {"BPPP_TWIN": 0, "BPPP_PACE": 1}), ("twin=1 pace=1", {"BPPP_TWIN": 1, "BPPP_PACE": 1})]
if os.environ.get("ONLY_ENV"):           # replaces the list; the FIRST entry is the base of the comparison, e.g. ONLY_ENV="one kernel:BPPP_TABLES_STAGED=0;by stage:BPPP_TABLES_STAGED=1"
    CONFIGS = []
    for item in os.environ["ONLY_ENV"].split(";"):
        name, _, kv = item.partition(":")
        CONFIGS.append((name, dict(x.split("=") for x in kv.split(",")) if kv else {}))
BASE = CONFIGS[0][0] if os.environ.get("ONLY_ENV") else "twin=0 pace=0"
if os.environ.get("EXTRA_ENV"):          # e.g. EXTRA_ENV="twin=1 si=8:BPPP_TWIN=1,BPPP_SHARED_INV=8;..."
    for item in os.environ["EXTRA_ENV"].split(";"):
        name, _, kv = item.partition(":")
        CONFIGS.append((name, dict(x.split("=") for x in kv.split(","))))


def child(parent, env):
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        return parent.clone_shared()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    sizes = [(1 << int(a)) if int(a) < 64 else int(a) for a in sys.argv[1:]] or [1 << 17, 1 << 18, 1 << 19, 1 << 20]
    gens, g, gv, hv = bench.load_generators()
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=int(os.environ.get("FB_WINDOW_BITS", "0")))
    nmax = max(sizes)
    dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 0, nmax)
    dA = torch.zeros(nmax, dtype=torch.uint8, device="cuda"); dS = torch.zeros(nmax, dtype=torch.int32, device="cuda")
    dR = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctxs = [(name, child(proto, env)) for name, env in CONFIGS]
    for n in sizes:
        times = {name: [] for name, _ in ctxs}
        oks, plans = {}, {}
        for name, c in ctxs:         # warm-up + correctness of each configuration at this size
            dA.zero_(); dS.zero_()
            for _ in range(2):
                c.verify_batch_device(synth.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, dR.data_ptr())
            c.synchronize()
            oks[name] = bool((dA[:n].cpu().numpy() == expect[:n]).all()) and not bool(dS[:n].any().item()) and int(dR.item()) == int((expect[:n] == 0).sum())
            plans[name] = c.last_plan()
        inner = max(1, min(16, (1 << 20) // n))
        for _ in range(REPS):
            for name, c in ctxs:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(inner):
                    c.verify_batch_device(synth.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, 0)
                c.synchronize()
                times[name].append((time.perf_counter() - t0) * 1e3 / inner)
        base = float(np.median(times[BASE]))
        for name, _ in ctxs:
            t = np.array(times[name])
            print(f"n={n:8d} {name:18s} median {np.median(t):8.3f} ms  min {t.min():8.3f}  {n / np.median(t) / 1e3:6.3f} M/s  vs {BASE} {np.median(t) / base - 1:+.2%}  ok={oks[name]}  {plans[name]}", flush=True)
    for _, c in ctxs:
        c.close()
    proto.close()


if __name__ == "__main__":
    main()
