"""CPU tier for the drop-in boundary: libbppp_hip.so must load without a GPU, export every function include/bppp.h
declares, and refuse (loudly, no fallback) to compute when there is no gfx950 device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "bppp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bppp_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from bp_pp_amd import _capi
    assert _declared() == sorted(_capi.EXPORTS)


def test_library_loads_and_exports_every_symbol():
    from bp_pp_amd import _build, _capi
    if not os.path.exists(_build.SO):
        pytest.skip("libbppp_hip.so not built yet (python -c 'import __graft_entry__ as g; g.build()')")
    L = _capi.lib()
    for name in _declared():
        assert hasattr(L, name), name
    assert L.bppp_strerror(0) == b"ok"
    assert b"no CPU fallback" in L.bppp_strerror(_capi.ERR_NO_DEVICE)


def test_no_device_means_error_not_fallback():
    import torch
    from bp_pp_amd import _build, _capi
    if not os.path.exists(_build.SO):
        pytest.skip("libbppp_hip.so not built yet")
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present; covered by the -m gpu tests")
    L = _capi.lib()
    ctx = C.c_void_p()
    rc = L.bppp_ctx_create(C.byref(ctx), bytes(64), bytes(16 * 64), bytes(32 * 64), 0, 8)
    assert rc == _capi.ERR_NO_DEVICE and not ctx.value
    assert L.bppp_u64_verify_batch(None, b"x", 1, 1, None, None, None, None) == _capi.ERR_INVALID_ARG
