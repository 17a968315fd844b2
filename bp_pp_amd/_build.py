"""Builds libbppp_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "bppp_hip.hip")
HEADERS = [os.path.join(HERE, "csrc", h) for h in ("field.h", "point.h", "merlin.h", "verify_core.h")] + [
    os.path.join(ROOT, "include", "bppp.h")]
SO = os.path.join(HERE, "libbppp_hip.so")


def needs_build() -> bool:
    if not os.path.exists(SO):
        return True
    so_m = os.path.getmtime(SO)
    return any(os.path.getmtime(p) > so_m for p in [SRC] + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-Wl,-rpath,/opt/rocm/lib", "-o", SO + ".tmp", SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    os.replace(SO + ".tmp", SO)
    return SO
