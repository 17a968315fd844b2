"""Kernel timeline of ONE verify call of 2^17 proofs, from a rocprofv3 --kernel-trace of this script:
    mode lib     : the library's twin plan (BPPP_TWIN=1 child context)
    mode ctx2    : two child contexts, half a batch each (256-register kernels, one lane per fixed-base sum), called one after the other
    mode one     : one sequence
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 $REPO/tools/probes/twin_trace.py MODE [log2 n]
then   python3 tools/probes/twin_trace.py show OUT   prints the last call's kernels: queue, start (ms after the first), duration."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def show(path):
    import csv, glob
    rows = []
    for f in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows = [r for r in rows if r["Kernel_Name"].startswith(("k_verify", "void k_verify"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # calls are separated by gaps > 2 ms between one kernel's start and every earlier kernel's end
    calls, cur, last_end = [], [], None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if last_end is not None and s - last_end > 1_000_000:
            calls.append(cur); cur = []
        cur.append(r)
        last_end = e if last_end is None else max(last_end, e)
    calls.append(cur)
    call = calls[-1]
    t0 = int(call[0]["Start_Timestamp"])
    print(f"{len(calls)} calls; the last one: {len(call)} kernels, {(max(int(r['End_Timestamp']) for r in call) - t0) / 1e6:.3f} ms")
    for r in call:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  q{r['Queue_Id']:>3s} {r['Kernel_Name'][:34]:34s} start {(s - t0) / 1e6:8.3f}  dur {(e - s) / 1e6:7.3f}  end {(e - t0) / 1e6:8.3f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")


def main():
    mode = sys.argv[1]
    if mode == "show":
        return show(sys.argv[2])
    import numpy as np, torch, bench
    from bp_pp_amd import U64RangeProofProtocol, synth
    n = 1 << int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 17
    gens, g, gv, hv = bench.load_generators()
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=int(os.environ.get("FB_WINDOW_BITS", "0")))
    dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 0, n)
    dA = torch.zeros(n, dtype=torch.uint8, device="cuda"); dS = torch.zeros(n, dtype=torch.int32, device="cuda")

    def child(**env):
        for k, v in env.items():
            os.environ[k] = str(v)
        c = proto.clone_shared()
        for k in env:
            os.environ.pop(k)
        return c
    dummies = []
    if mode.startswith("many"):          # manyK: K more child contexts alive (each used once), then the library's twin plan
        for _ in range(int(mode[4:] or 8)):
            d = child(BPPP_TWIN=0)
            d.verify_batch_device(synth.LABEL, 4096, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, 0)
            d.synchronize()
            dummies.append(d)
        cs = [child(BPPP_TWIN=1, BPPP_TWIN_STREAMS=os.environ.get("TWIN_STREAMS", "1"))]
    elif mode == "lib":
        cs = [child(BPPP_TWIN=1)]
    elif mode == "ctx2":
        cs = [child(BPPP_NO_SMALL_KERNELS=1, BPPP_FB_ONE_LANE=1, BPPP_TWIN=0) for _ in range(2)]
    else:
        cs = [child(BPPP_TWIN=0)]
    m = n // len(cs)
    ts = []
    for it in range(6):
        torch.cuda.synchronize()
        time.sleep(0.01)
        t0 = time.perf_counter()
        for i, c in enumerate(cs):
            c.verify_batch_device(synth.LABEL, m, dV[i * m:].data_ptr(), dP[i * m:].data_ptr(), dA[i * m:].data_ptr(), dS[i * m:].data_ptr(), 0, 0)
        for c in cs:
            c.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(mode, "ms per call:", [round(t, 3) for t in ts], "ok", bool((dA.cpu().numpy() == expect).all()), cs[0].last_plan(), flush=True)
    for c in cs + dummies:
        c.close()
    proto.close()


if __name__ == "__main__":
    main()
