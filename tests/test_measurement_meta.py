"""CPU tier: the measurement's self-validation (bench.py) and the host facts it reports -- no GPU involved.
  * the committed PMC summaries carry the SHA-256 of the device code they were collected on, and bench.py hands them out only for
    that build (a kernel edit without a re-profile must null `traffic` / `roofline_valu`, not quote stale counters);
  * device_code_sha256 is a function of the library's embedded code objects;
  * tools/hostinfo.py reads affinity, cgroup quota and CPU model without throwing."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pmc_summaries_are_stamped_and_only_trusted_for_their_build(monkeypatch):
    import bench
    tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    va = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu.json")))
    h1, h2 = tr.get("code_object_sha256"), (va.get("_meta") or {}).get("code_object_sha256")
    assert h1 and len(h1) == 64 and h1 == h2                      # both from the same session, on the same device code
    monkeypatch.setattr(bench, "_BUILD_ID", h1)
    assert bench.pmc_file("pmc_traffic.json") is not None and bench.pmc_file("pmc_valu.json") is not None
    assert bench.pmc_traffic("k_verify_round", 1 << 20) > 1e9
    assert bench.valu_roofline("k_verify_round", 12.6, 1 << 20, 1)["frac_of_datasheet"] > 0.5
    m = bench.pmc_matches_build()
    assert m["traffic"] and m["valu"] and m["code_object_sha256"] == h1
    monkeypatch.setattr(bench, "_BUILD_ID", "0" * 64)             # another build: nothing is quoted
    assert bench.pmc_file("pmc_traffic.json") is None and bench.pmc_file("pmc_valu.json") is None
    assert bench.pmc_traffic("k_verify_round", 1 << 20) is None
    assert bench.valu_roofline("k_verify_round", 12.6, 1 << 20, 1) is None
    assert bench.pmc_matches_build() == {"traffic": False, "valu": False, "code_object_sha256": "0" * 64}


def test_device_code_hash_of_the_built_library():
    from bp_pp_amd import _build
    h = _build.device_code_sha256()
    assert h and len(h) == 64 and h == _build.device_code_sha256()
    assert _build.device_code_sha256(os.path.join(ROOT, "README.md")) is None          # not an ELF file
    assert _build.device_code_sha256(os.path.join(ROOT, "no such file")) is None


def test_hostinfo_reads_what_the_box_grants():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import hostinfo
    s = hostinfo.summary()
    assert s["affinity_cpus"] >= 1 and 1 <= s["usable_cpus"] <= s["affinity_cpus"] and isinstance(s["cpu_model"], str)
    q = s["cgroup_cpu_quota"]
    assert q is None or q > 0
    t = hostinfo.throttle_stats()
    assert set(t) == {"nr_periods", "nr_throttled", "throttled_s"} and t["nr_throttled"] >= 0
