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


def test_predicted_scaling_is_total_over_the_share_time():
    """bench.py: `predicted_scaling` = the fixed batch / the time ONE GPU takes for a 1/N share (shard_2pow19 / 18 / 17)."""
    import bench
    shares = {17: {"ms_per_step": 18.0}, 18: {"ms_per_step": 36.5}, 19: {"ms_per_step": 72.0}}
    p = bench.predicted_scaling(1 << 20, 143.0, shares)
    assert set(p) == {"1", "2", "4", "8", "note"}
    assert p["1"]["share"] == 1 << 20 and p["8"]["share"] == 1 << 17 and p["2"]["share"] == 1 << 19
    assert abs(p["8"]["value"] - (1 << 20) / 18.0e-3) < 1.0 and abs(p["8"]["efficiency"] - 143.0 / (8 * 18.0)) < 1e-9
    assert abs(p["4"]["efficiency"] - 143.0 / (4 * 36.5)) < 1e-9
    assert set(bench.predicted_scaling(1 << 20, 143.0, {})) == {"1", "note"}          # nothing measured, nothing predicted


def test_gpus_n_without_a_launcher_starts_the_ranks_as_children(monkeypatch):
    """bench.py --gpus N with no WORLD_SIZE around it: the ranks go out as a CHILD torch.distributed.run (127.0.0.1, a free port, the
    same arguments), never by replacing this process; under rocprofv3 it refuses instead."""
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_")) or k == "LD_PRELOAD":
            monkeypatch.delenv(k)
    assert bench.launch_ranks(4) == 7                                  # the launcher's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("ROCPROFILER_REGISTER_FORCE_LOAD", "1")
    assert bench.launch_ranks(4) == 2
