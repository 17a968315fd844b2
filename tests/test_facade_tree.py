"""The Rust facade tree (facade/) cannot be compiled here (no cargo / rustc): what CAN be checked is that its `extern "C"` block is
exactly what include/bppp.h declares -- regenerated from the header -- and that it covers every symbol the .so exports."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ffi_rs_is_generated_from_the_header_and_complete():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_facade_ffi
    src, names = gen_facade_ffi.generate()
    assert open(os.path.join(ROOT, "facade", "src", "ffi.rs")).read() == src, "run python tools/gen_facade_ffi.py"
    from bp_pp_amd import _capi
    assert sorted(names) == sorted(_capi.EXPORTS)


def test_facade_uses_only_declared_entry_points():
    ffi = open(os.path.join(ROOT, "facade", "src", "ffi.rs")).read()
    declared = set(re.findall(r"pub fn (bppp_\w+)", ffi))
    for f in ("gpu.rs", "tstate.rs", "conv.rs", os.path.join("bin", "gen_fixtures.rs")):
        used = set(re.findall(r"\b(bppp_\w+)\s*\(", open(os.path.join(ROOT, "facade", "src", f)).read()))
        assert used <= declared, (f, used - declared)
    gpu = open(os.path.join(ROOT, "facade", "src", "gpu.rs")).read()
    assert "cpu: U64RangeProofProtocol" in gpu and "self.cpu.verify(" in gpu     # the CPU path of the CRATE, in the facade
