"""Print the essentials of bench.py JSON lines: python tools/show_bench.py file.json [...]  (or stdin)."""
import json, sys
def show(l):
    if '"value"' not in l:
        return
    d = json.loads(l)
    print(d["config"].get("total_proofs_per_step"), "proofs  n_gpus", d["n_gpus"], " value", round(d["value"]), d["unit"], " ms/step", round(d["ms_per_step"], 3),
          " ok", d.get("accept_bits_ok"))
    print("  kernels ms/step:", {k: round(v, 3) for k, v in (d.get("kernels_ms_per_step") or {}).items()})
    r = d.get("roofline") or {}
    print("  roofline:", r.get("kernel"), "avg launch ms", round(r.get("avg_launch_ms", 0), 4), "frac", round(r.get("frac", 0), 5), "traffic", r.get("traffic"))
    for k in ("configs1_2pow16", "shard_2pow17", "rlc_mode", "host_buffer_path", "cpu_baseline", "prove_2pow14", "recip256_2pow15"):
        v = d.get(k)
        if v:
            print(f"  {k}:", round(v["value"]), v.get("unit"), {a: b for a, b in v.items() if a in ("ms_per_step", "one_sequence_ms_per_step_same_moment", "ms_per_batch", "cores", "accept_bits_ok", "accept_bits_equal_exact_mode", "agrees_with_gpu")})
            if isinstance(v.get("ct_prover"), dict):
                print(f"    {k}.ct_prover:", round(v["ct_prover"]["value"]), round(v["ct_prover"]["ms_per_step"], 2), "x%.2f" % v["ct_prover"]["cost_vs_default"], v["ct_prover"].get("byte_identical_to_default"))
            if isinstance(v.get("all_valid"), dict):
                print(f"    {k}.all_valid:", round(v["all_valid"]["value"]), round(v["all_valid"]["ms_per_step"], 2), v["all_valid"].get("all_accepted"))
    cc = d.get("concurrent_callers") or {}
    for k, v in cc.items():
        if k.startswith("threads_"):
            print(f"  concurrent_callers.{k}:", round(v["verifies_per_s"]), "verifies/s  latency ms", v["latency_ms"], "mean batch", v.get("mean_batch"))
    if d.get("pmc_matches_build") is not None:
        print("  pmc_matches_build:", d["pmc_matches_build"])
    if d.get("call_latency"):
        print("  call_latency:", {k: v for k, v in d["call_latency"].items() if k.startswith("n")})
    print("  setup:", d.get("setup_s"), " device GB", round((d.get("device_bytes") or 0) / 1e9, 1))
files = sys.argv[1:]
if files:
    for f in files:
        for l in open(f):
            show(l)
else:
    for l in sys.stdin:
        show(l)
