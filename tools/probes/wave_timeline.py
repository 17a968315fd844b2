"""When do the wavefronts of the one-lane kernels start and end, how long does each take, and where do they run?

A -DBPPP_PHASE_TIMING build (tools/build_variants.py pt=-DBPPP_PHASE_TIMING) stamps the 100 MHz real-time counter per sampled wavefront
at marked points of verify_phase1 / verify_tables / verify_c0_var / verify_round, plus HW_ID | XCC_ID at each kernel's first stamp
(verify_ws.h: BPPP_STAMP; up to 4,096 rows: every wavefront up to 2^18 proofs, every 4th at 2^20).

    BPPP_LIB=bp_pp_amd/libbppp_hip_pt.so python tools/probes/wave_timeline.py [log2 n ...]      (default 17 20)

Per kernel: spread of the start times, percentiles of the wavefront durations, first start to last end; mean duration per XCD; how many
of the launch's wavefronts each SIMD hosted; and -- when every wavefront is sampled -- the time-integrated residency: the share of
(SIMD x kernel span) during which 0 / 1 / 2 of them were resident."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from bp_pp_amd import U64RangeProofProtocol, synth, _capi

ROWS = 4096
KERNELS = (("phase1", 0, 8, 24), ("tables", 16, 19, 26), ("c0_var sum", 20, 21, 27), ("round 4 (last launch)", 9, 13, 25))
ROUND_PHASES = (("C_{k-1} to affine", 9, 10), ("transcript + challenge", 10, 11), ("stores, GLV split", 11, 12), ("two-point sum", 12, 13))


def pct(x, ps=(0, 10, 50, 90, 99, 100)):
    return [round(float(np.percentile(x, p)) / 1e5, 3) for p in ps]          # 100 MHz ticks -> ms


def residency(keys, s, e):
    """share of (SIMD x span) with 0 / 1 / 2+ of the launch's wavefronts resident"""
    t0, t1 = int(s.min()), int(e.max())
    span = t1 - t0
    by = {}
    for k, a, b in zip(keys, s, e):
        by.setdefault(int(k), []).append((int(a), int(b)))
    tot = np.zeros(3)
    for iv in by.values():
        ev = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
        cur, last = 0, t0
        for t, d in ev:
            tot[min(cur, 2)] += t - last
            cur += d
            last = t
        tot[0] += t1 - last
    return [round(float(x) / (span * len(by)), 4) for x in tot], len(by)


def main():
    sizes = [1 << int(a) for a in sys.argv[1:]] or [1 << 17, 1 << 20]
    gens, g, gv, hv = bench.load_generators()
    proto = U64RangeProofProtocol(g, gv, hv, device=0, fb_window_bits=int(os.environ.get("FB_WINDOW_BITS", "0")))
    L = _capi.lib()
    L.bppp_debug_read_stamps.argtypes = [C.c_void_p]
    nmax = max(sizes)
    dV, dP, expect, _ = bench.make_resident_batch(torch, proto, synth, 0, nmax)
    for n in sizes:
        dA = torch.zeros(n, dtype=torch.uint8, device="cuda"); dS = torch.zeros(n, dtype=torch.int32, device="cuda")
        proto.enable_timing(True)            # the C0 halves back to back, as in the per-kernel measurements
        for _ in range(3):
            proto.verify_batch_device(synth.LABEL, n, dV.data_ptr(), dP.data_ptr(), dA.data_ptr(), dS.data_ptr(), 0, 0)
        proto.synchronize()
        kt = proto.timings(reset=True)
        proto.enable_timing(False)
        assert bool((dA.cpu().numpy() == expect[:n]).all())
        buf = np.zeros((ROWS, 32), np.uint64)
        assert L.bppp_debug_read_stamps(buf.ctypes.data) == 0
        waves = n // 64
        stride = max(1, (waves + ROWS - 1) // ROWS)
        w = min(ROWS, waves // stride)
        d = buf[:w].astype(np.int64)
        print(f"== n = 2^{n.bit_length() - 1}: {waves} wavefronts per one-lane launch, every {stride}{'st' if stride == 1 else 'th'} sampled ({w} rows); plan: {proto.last_plan()}")
        print("   kernel ms (HIP events, per launch):", {k.replace('k_verify_', ''): round(v['total_ms'] / v['launches'], 3) for k, v in kt.items() if v['launches']}, flush=True)
        for name, a, b, hwslot in KERNELS:
            s, e = d[:, a], d[:, b]
            if not s.any() or not e.any():
                print(f"   {name:22s} (no stamps: this plan runs other kernels for the stage)")
                continue
            t0 = s.min()
            hw = buf[:w, hwslot]
            xcc = (hw >> np.uint64(32)).astype(np.int64) & 15
            simd_key = (xcc << 16) | ((hw.astype(np.int64) >> 8) & 0xFF) << 4 | ((hw.astype(np.int64) >> 4) & 3)
            cu_key = (xcc << 16) | ((hw.astype(np.int64) >> 8) & 0xFF)
            dur = e - s
            print(f"   {name:22s} starts after the first (ms) p0/10/50/90/99/100 {pct(s - t0)}   duration (ms) {pct(dur)}   mean {round(float(dur.mean()) / 1e5, 3)}"
                  f"   first start to last end {round(float(e.max() - t0) / 1e5, 3)} ms")
            per_xcc = {int(x): round(float(dur[xcc == x].mean()) / 1e5, 3) for x in sorted(set(xcc.tolist()))}
            cnt = np.bincount(np.unique(simd_key, return_counts=True)[1])
            cnt_cu = np.bincount(np.unique(cu_key, return_counts=True)[1])
            print(f"      mean duration per XCD (ms): {per_xcc}")
            print(f"      sampled wavefronts per SIMD -> number of SIMDs: { {i: int(c) for i, c in enumerate(cnt) if c} } ({len(set(simd_key.tolist()))} SIMDs seen);"
                  f" per CU -> CUs: { {i: int(c) for i, c in enumerate(cnt_cu) if c} }")
            if stride == 1:
                r, nsimd = residency(simd_key, s, e)
                print(f"      residency over the kernel's span, {nsimd} SIMDs: 0 wavefronts {r[0]:.1%}, 1 wavefront {r[1]:.1%}, 2 wavefronts {r[2]:.1%}")
                print(f"      end times (ms after the first start) p0/10/50/90/99/100 {pct(e - t0)}")
            if name.startswith("round"):
                for pn, pa, pb in ROUND_PHASES:
                    print(f"      {pn:26s} (ms) p0/10/50/90/99/100 {pct(d[:, pb] - d[:, pa])}")
        sys.stdout.flush()
    proto.close()


if __name__ == "__main__":
    main()
