#!/usr/bin/env python3
"""Build A/B variants of libbppp_hip.so with different hipcc -D flags: tools/build_variants.py name=-DFOO=1,-DBAR=2 ...
-> bp_pp_amd/libbppp_hip_<name>.so (objects in bp_pp_amd/_obj_<name>/).  Load one with BPPP_LIB=<path>."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bp_pp_amd import _build
for spec in sys.argv[1:]:
    name, _, flags = spec.partition("=")
    extra = [f for f in flags.split(",") if f]
    so = os.path.join(_build.HERE, f"libbppp_hip_{name}.so")
    _build.build(so=so, objdir=os.path.join(_build.HERE, f"_obj_{name}"), extra=extra)
    print("built", so, extra, flush=True)
