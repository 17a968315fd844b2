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
    # the other ways to obtain a context or a device group fail the same way: loudly, with no fallback
    grp, dev = C.c_void_p(), (C.c_int * 1)(0)
    assert L.bppp_group_create(C.byref(grp), bytes(64), bytes(16 * 64), bytes(32 * 64), dev, 1, 8) == _capi.ERR_NO_DEVICE and not grp.value
    assert L.bppp_wnla_ctx_create(C.byref(ctx), bytes(64), bytes(64), 1, bytes(64), 1, 0, 8) == _capi.ERR_NO_DEVICE and not ctx.value
    assert L.bppp_ctx_create_from_tables(C.byref(ctx), b"/nonexistent", 0) in (_capi.ERR_NO_DEVICE, _capi.ERR_INVALID_ARG) and not ctx.value
    assert L.bppp_ctx_create_shared(C.byref(ctx), None) == _capi.ERR_INVALID_ARG
    assert L.bppp_group_size(None) == 0


def test_sharded_entry_points_refuse_bad_arguments_without_touching_a_gpu():
    """Every *_sharded* entry point and the group housekeeping calls validate their arguments before anything else: a NULL group (what
    a caller is left with after group creation failed for want of a GPU) is BPPP_ERR_INVALID_ARG, never a crash.  No GPU needed."""
    import numpy as np
    from bp_pp_amd import _build, _capi
    if not os.path.exists(_build.SO):
        pytest.skip("libbppp_hip.so not built yet")
    L = _capi.lib()
    E = _capi.ERR_INVALID_ARG
    buf = np.zeros(4096, np.uint8)
    p = buf.ctypes.data
    rej = C.c_int32(7)
    arr = (C.c_void_p * 1)(C.c_void_p(p))
    seed = bytes(32)
    assert L.bppp_group_set_option(None, b"rlc_superchunk", 256) == E
    assert L.bppp_u64_verify_batch_sharded(None, b"x", 1, 1, p, p, p, p, C.byref(rej)) == E
    assert L.bppp_u64_verify_batch_rlc_sharded(None, b"x", 1, 1, p, p, p, p, C.byref(rej), seed) == E
    assert L.bppp_u64_verify_batch_sec1_sharded(None, b"x", 1, 1, p, p, p, p, C.byref(rej)) == E
    assert L.bppp_u64_verify_batch_transcript_sharded(None, 1, p, 1, p, p, p, p, p, C.byref(rej)) == E
    assert L.bppp_reciprocal_verify_batch_sharded(None, b"x", 1, 1, 16, 16, p, p, 4, 2, 1, p, p, C.byref(rej)) == E
    assert L.bppp_reciprocal_verify_batch_rlc_sharded(None, b"x", 1, 1, 16, 16, p, p, 4, 2, 1, p, p, C.byref(rej), seed) == E
    assert L.bppp_u64_verify_batch_sharded_device(None, b"x", 1, 1, arr, arr, arr, arr, arr) == E
    assert L.bppp_u64_verify_batch_rlc_sharded_device(None, b"x", 1, 1, arr, arr, arr, arr, arr, seed) == E
    assert L.bppp_u64_verify_batch_sec1_sharded_device(None, b"x", 1, 1, arr, arr, arr, arr, arr) == E
    assert L.bppp_u64_verify_batch_transcript_sharded_device(None, 1, arr, 1, arr, arr, arr, arr, arr, arr) == E
    assert L.bppp_reciprocal_verify_batch_sharded_device(None, b"x", 1, 1, 16, 16, arr, arr, 4, 2, 1, arr, arr, arr) == E
    assert L.bppp_reciprocal_verify_batch_rlc_sharded_device(None, b"x", 1, 1, 16, 16, arr, arr, 4, 2, 1, arr, arr, arr, seed) == E
    assert L.bppp_u64_prove_batch_sharded(None, b"x", 1, 1, p, p, p, p, p, p) == E
    assert L.bppp_u64_prove_batch_sharded_device(None, b"x", 1, 1, arr, arr, arr, arr, arr, arr) == E
    grp, dev = C.c_void_p(), (C.c_int * 2)(0, 0)
    assert L.bppp_wnla_group_create(C.byref(grp), bytes(64), bytes(64), 1, bytes(64), 1, dev, 2, 8) == E and not grp.value   # a device twice
    assert L.bppp_wnla_group_create(C.byref(grp), bytes(64), bytes(64), 1, bytes(64), 1, dev, 0, 8) == E                       # no devices
    assert L.bppp_group_ctx(None, 0) is None and L.bppp_group_size(None) == 0
    L.bppp_group_destroy(None)                                                                                                  # a no-op


def test_header_is_plain_c_and_a_c_program_links(tmp_path):
    """The boundary is a C ABI: include/bppp.h must compile as C99 (no C++-isms, no torch / HIP types), and a C program must link
    against libbppp_hip.so and get BPPP_ERR_NO_DEVICE -- not a crash, not a fallback -- on a machine without a gfx950 device."""
    import shutil
    import subprocess
    import torch
    from bp_pp_amd import _build
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("gcc not available")
    src = tmp_path / "capi_smoke.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "bppp.h"
int main(void) {
    unsigned char g[64], gv[16 * 64], hv[32 * 64];
    bppp_ctx* ctx = NULL;
    size_t rounds = 0, nl = 0, nn = 0;
    memset(g, 0, sizeof g); memset(gv, 0, sizeof gv); memset(hv, 0, sizeof hv);
    bppp_wnla_proof_shape(32, 16, &rounds, &nl, &nn);
    if (rounds != 4 || nl != 2 || nn != 1) return 10;                 /* the u64 proof's shape (wnla.rs:126) */
    printf("%d %s\n", bppp_ctx_create(&ctx, g, gv, hv, 0, 8), bppp_strerror(BPPP_ERR_NO_DEVICE));
    return ctx != NULL;
}
''')
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
    if not os.path.exists(_build.SO):
        pytest.skip("libbppp_hip.so not built yet")
    exe = tmp_path / "capi_smoke"
    subprocess.check_call([gcc, "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), _build.SO,
                           "-Wl,-rpath," + os.path.dirname(_build.SO), "-Wl,-rpath,/opt/rocm/lib"])
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present; the no-device path is a CPU-tier check")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split()[0] == "-1" and "no CPU fallback" in out.stdout


def test_labels_longer_than_merlin_can_frame_are_refused():
    """merlin frames a label's length as a u32 (`Transcript::append_message` asserts that); round 4 cast `label_len` to u32 unchecked and
    would have hashed a truncated length.  The ABI answers BPPP_ERR_INVALID_ARG before it reads a byte of the label.  No GPU needed: the
    transcript helpers run on the host."""
    from bp_pp_amd import _build, _capi
    if not os.path.exists(_build.SO):
        pytest.skip("libbppp_hip.so not built yet")
    L = _capi.lib()
    E = _capi.ERR_INVALID_ARG
    state = C.create_string_buffer(203)
    too_long = C.c_size_t(1 << 32)
    assert L.bppp_transcript_new(b"x", too_long, state) == E
    assert L.bppp_transcript_new(b"u64 range proof", 15, state) == 0
    assert L.bppp_transcript_append_message(state, b"x", too_long, b"m", 1) == E
    out = C.create_string_buffer(32)
    assert L.bppp_transcript_challenge_bytes(state, b"x", too_long, out, 32) == E
    # the single-proof entry points build their transcript first: same answer, with or without a device
    acc, st = C.c_uint8(0), C.c_int32(0)
    assert L.bppp_u64_verify_one(None, b"x", too_long, bytes(64), bytes(928), C.byref(acc), C.byref(st)) == E
