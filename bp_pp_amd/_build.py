"""Builds libbppp_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

The library is several translation units (csrc/*.hip: the host side + kernel groups), compiled in parallel into
bp_pp_amd/_obj/*.o and linked into one shared object.  Staleness is decided per object from hipcc's own dependency
files (-MD), so editing any header a unit includes rebuilds exactly the units that include it."""
from __future__ import annotations

import glob
import os
import shlex
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
SO = os.path.join(HERE, "libbppp_hip.so")
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _obj_of(src: str, objdir: str) -> str:
    return os.path.join(objdir, os.path.basename(src)[:-4] + ".o")


def _deps(obj: str):
    """Prerequisites recorded by hipcc -MD for this object (None when there is no dependency file yet)."""
    d = obj[:-2] + ".d"
    if not os.path.exists(d):
        return None
    txt = open(d).read().replace("\\\n", " ")
    out = []
    for part in txt.split(":", 1)[1:]:
        out += [p for p in shlex.split(part) if not p.startswith("/opt/rocm") and not p.startswith("/usr/")]
    return out


def _flag_stamp(flags) -> str:
    return " ".join(flags)


def _obj_stale(src: str, obj: str, flags) -> bool:
    if not os.path.exists(obj):
        return True
    stamp = obj[:-2] + ".flags"
    if not os.path.exists(stamp) or open(stamp).read() != _flag_stamp(flags):
        return True
    deps = _deps(obj)
    if deps is None:
        return True
    m = os.path.getmtime(obj)
    return any((not os.path.exists(p)) or os.path.getmtime(p) > m for p in [src] + deps)


def _extra_flags():
    return shlex.split(os.environ.get("BPPP_HIPCC_FLAGS", ""))


def needs_build(so: str = SO, objdir: str = OBJ, extra=None) -> bool:
    flags = BASE_FLAGS + (list(extra) if extra is not None else _extra_flags())
    if not os.path.exists(so):
        return True
    so_m = os.path.getmtime(so)
    for src in sources():
        obj = _obj_of(src, objdir)
        if _obj_stale(src, obj, flags) or os.path.getmtime(obj) > so_m:
            return True
    return False


def build(force: bool = False, verbose: bool = False, so: str = SO, objdir: str = OBJ, extra=None, jobs: int | None = None) -> str:
    """Compile what is stale and link `so`.  `extra`: additional hipcc flags (A/B builds pass their own `so` and `objdir`)."""
    extra = list(extra) if extra is not None else _extra_flags()
    flags = BASE_FLAGS + extra
    if not force and not needs_build(so, objdir, extra):
        return so
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(objdir, exist_ok=True)
    todo = [(s, _obj_of(s, objdir)) for s in sources()]
    stale = [(s, o) for s, o in todo if force or _obj_stale(s, o, flags)]

    def compile_one(so_pair):
        src, obj = so_pair
        # -cuid: hipcc names a translation unit's registration symbol after a hash of its PATH and options by default, so the same sources
        # built in another directory gave another device-code hash (device_code_sha256: what the PMC summaries are stamped with)
        cmd = [hipcc] + flags + ["-cuid=bppp-" + os.path.basename(src), "-MD", "-MF", obj[:-2] + ".d", "-c", src, "-o", obj + ".tmp"]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {os.path.basename(src)}:\n{r.stderr[-6000:]}")
        os.replace(obj + ".tmp", obj)
        with open(obj[:-2] + ".flags", "w") as f:
            f.write(_flag_stamp(flags))
        return r.stderr

    jobs = jobs or min(len(stale) or 1, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        logs = list(ex.map(compile_one, stale))
    if verbose:
        for lg in logs:
            print(lg)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-rpath,/opt/rocm/lib", "-o", so + ".tmp"] + [o for _, o in todo]
    subprocess.check_call(link)
    os.replace(so + ".tmp", so)
    return so


def device_code_sha256(so: str = SO) -> str | None:
    """SHA-256 of the gfx950 code objects embedded in the library (the ELF section .hip_fatbin): the identity of the kernels a
    measurement ran.  The PMC summaries under profiles/ carry it, and bench.py only trusts them for the build that is loaded."""
    import hashlib
    import struct
    try:
        with open(so, "rb") as f:
            data = f.read()
    except OSError:
        return None
    if data[:4] != b"\x7fELF" or data[4] != 2:
        return None
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    sec = lambda i: struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize)
    str_off = sec(shstrndx)[4]
    for i in range(shnum):
        name_off, _, _, _, off, size = sec(i)[:6]
        end = data.index(b"\0", str_off + name_off)
        if data[str_off + name_off:end] == b".hip_fatbin":
            return hashlib.sha256(data[off:off + size]).hexdigest()
    return None


def build_tool(name: str) -> str:
    """hipcc one stand-alone measurement program tools/<name>.hip -> tools/<name> (git-ignored; travels to the GPU box)."""
    src, exe = os.path.join(ROOT, "tools", name + ".hip"), os.path.join(ROOT, "tools", name)
    if os.path.exists(exe) and os.path.getmtime(exe) >= os.path.getmtime(src):
        return exe
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-o", exe + ".tmp", src])
    os.replace(exe + ".tmp", exe)
    return exe
