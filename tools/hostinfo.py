"""What the host side of a measurement really had: CPUs this process may run on (affinity), the cgroup's CPU quota (a container
granted 16 CPUs of a 256-thread host shows os.cpu_count() = 256), CFS throttling counters, the CPU model.  Used by bench.py
(cpu_baseline) and tools/concurrent_callers.py."""
import os


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _cgroup_dirs():
    """Candidate directories of this process's cpu controller (cgroup v2 unified, then v1), most specific first."""
    out = []
    txt = _read("/proc/self/cgroup") or ""
    for line in txt.splitlines():
        parts = line.split(":", 2)
        if len(parts) != 3:
            continue
        _, ctrl, path = parts
        if ctrl == "":
            out += ["/sys/fs/cgroup" + path, "/sys/fs/cgroup"]
        elif "cpu" in ctrl.split(","):
            out += ["/sys/fs/cgroup/cpu" + path, "/sys/fs/cgroup/cpu,cpuacct" + path, "/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"]
    out += ["/sys/fs/cgroup", "/sys/fs/cgroup/cpu"]
    seen, res = set(), []
    for d in out:
        if d not in seen and os.path.isdir(d):
            seen.add(d)
            res.append(d)
    return res


def cpu_quota():
    """CPUs' worth of run time per period the cgroup grants (float), or None when unlimited / unknown."""
    for d in _cgroup_dirs():
        v2 = _read(os.path.join(d, "cpu.max"))
        if v2:
            q, _, p = v2.partition(" ")
            if q != "max":
                try:
                    return float(q) / float(p or 100000)
                except ValueError:
                    pass
            continue
        q, p = _read(os.path.join(d, "cpu.cfs_quota_us")), _read(os.path.join(d, "cpu.cfs_period_us"))
        if q and p:
            try:
                if int(q) > 0:
                    return int(q) / int(p)
            except ValueError:
                pass
    return None


def throttle_stats():
    """{nr_periods, nr_throttled, throttled_s} of the cgroup (zeros when unknown): a run during which nr_throttled grows was
    stalled by the CPU quota, whatever the code under test did."""
    for d in _cgroup_dirs():
        txt = _read(os.path.join(d, "cpu.stat"))
        if not txt:
            continue
        kv = dict(l.split()[:2] for l in txt.splitlines() if len(l.split()) >= 2)
        if "nr_throttled" in kv:
            t = float(kv.get("throttled_usec", 0)) * 1e-6 if "throttled_usec" in kv else float(kv.get("throttled_time", 0)) * 1e-9
            return {"nr_periods": int(kv.get("nr_periods", 0)), "nr_throttled": int(kv["nr_throttled"]), "throttled_s": round(t, 4)}
    return {"nr_periods": 0, "nr_throttled": 0, "throttled_s": 0.0}


def cpu_model():
    for line in (_read("/proc/cpuinfo") or "").splitlines():
        if line.startswith("model name"):
            return line.split(":", 1)[1].strip()
    return "unknown"


def usable_cpus():
    """CPUs a thread pool of this process can actually keep busy: min(affinity, cgroup quota rounded down, at least 1)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = cpu_quota()
    if q is not None:
        n = max(1, min(n, int(q)))
    return n


def summary():
    return {"affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None, "os_cpu_count": os.cpu_count(),
            "cgroup_cpu_quota": cpu_quota(), "usable_cpus": usable_cpus(), "cpu_model": cpu_model()}


if __name__ == "__main__":
    import json
    print(json.dumps(dict(summary(), throttle=throttle_stats())))
