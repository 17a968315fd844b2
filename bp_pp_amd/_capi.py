"""ctypes binding of include/bppp.h (libbppp_hip.so).  No CPU fallback: loading or context creation fails loudly
when the HIP extension or a gfx950 device is missing."""
from __future__ import annotations

import ctypes as C
import os

from . import _build

OK = 0
ERR_NO_DEVICE, ERR_INVALID_ARG, ERR_HIP, ERR_ENCODING, ERR_NOMEM, ERR_RCCL, ERR_CLOSED = -1, -2, -3, -4, -5, -6, -7
ST_BAD_ENCODING, ST_DEGENERATE = 1, 2
POINT_BYTES, SCALAR_BYTES, U64_PROOF_BYTES, U64_TRACE_BYTES = 64, 32, 928, 704

# every symbol include/bppp.h declares (tests/test_capi_symbols.py checks the header against this list and the .so)
EXPORTS = [
    "bppp_ctx_create", "bppp_wnla_ctx_create", "bppp_wnla_ctx_create_budget", "bppp_wnla_commit_batch", "bppp_wnla_verify_batch", "bppp_wnla_verify_batch_device", "bppp_circuit_verify_batch_device", "bppp_reciprocal_verify_batch", "bppp_reciprocal_verify_batch_device", "bppp_reciprocal_verify_batch_rlc",
    "bppp_reciprocal_verify_batch_rlc_device", "bppp_reciprocal_prove_batch", "bppp_msm_batch", "bppp_wnla_proof_shape", "bppp_wnla_prove_batch", "bppp_circuit_create", "bppp_circuit_destroy", "bppp_circuit_verify_batch", "bppp_circuit_prove_batch", "bppp_ctx_destroy", "bppp_ctx_set_stream", "bppp_ctx_synchronize", "bppp_ctx_set_option", "bppp_u64_verify_batch", "bppp_u64_verify_batch_device", "bppp_u64_verify_batch_rlc_device", "bppp_u64_verify_batch_rlc", "bppp_u64_verify_batch_sec1", "bppp_u64_verify_batch_sec1_device",
    "bppp_u64_commit_value_batch", "bppp_u64_prove_batch", "bppp_u64_prove_batch_device", "bppp_ctx_enable_timing", "bppp_ctx_get_timings", "bppp_ctx_device_bytes",
    "bppp_strerror", "bppp_last_error",
    "bppp_u64_verify_batch_transcript", "bppp_u64_verify_batch_transcript_device", "bppp_u64_prove_batch_transcript",
    "bppp_u64_prove_batch_transcript_device", "bppp_wnla_verify_batch_transcript", "bppp_reciprocal_verify_batch_transcript",
    "bppp_circuit_verify_batch_transcript", "bppp_wnla_prove_batch_transcript", "bppp_reciprocal_prove_batch_transcript",
    "bppp_circuit_prove_batch_transcript", "bppp_transcript_new",
    "bppp_transcript_append_message", "bppp_transcript_challenge_bytes",
    "bppp_derive_generators", "bppp_ctx_save_tables", "bppp_ctx_create_from_tables", "bppp_ctx_create_shared",
    "bppp_shard_range", "bppp_group_create", "bppp_group_destroy", "bppp_group_size", "bppp_group_ctx", "bppp_u64_verify_batch_sharded",
    "bppp_u64_verify_batch_sharded_device",
    "bppp_wnla_group_create", "bppp_group_set_option", "bppp_u64_verify_batch_rlc_sharded", "bppp_u64_verify_batch_rlc_sharded_device",
    "bppp_u64_verify_batch_sec1_sharded", "bppp_u64_verify_batch_sec1_sharded_device", "bppp_u64_verify_batch_transcript_sharded",
    "bppp_u64_verify_batch_transcript_sharded_device", "bppp_reciprocal_verify_batch_sharded", "bppp_reciprocal_verify_batch_rlc_sharded",
    "bppp_reciprocal_verify_batch_sharded_device", "bppp_reciprocal_verify_batch_rlc_sharded_device",
    "bppp_u64_prove_batch_sharded", "bppp_u64_prove_batch_sharded_device",
    "bppp_u64_prove_batch_sec1", "bppp_u64_prove_batch_sec1_device",
    "bppp_u64_verify_one", "bppp_u64_verify_one_transcript", "bppp_u64_prove_one", "bppp_u64_prove_one_transcript",
    "bppp_ctx_get_coalesce_stats", "bppp_ctx_get_option", "bppp_u64_plan", "bppp_plan_describe", "bppp_reciprocal_verify_one", "bppp_reciprocal_verify_one_transcript",
]

_lib = None


class BpppError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        msg = lib().bppp_strerror(code).decode() if _lib is not None else str(code)
        super().__init__(f"bppp error {code}: {msg}" + (f" ({detail})" if detail else ""))


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so (soname libamdhip64.so.7, the soname this library
    links against); two copies in one process means two ROCr instances, and the second one to initialise finds "no ROCm-capable
    device".  If torch is installed but not imported yet, its copy is loaded first (by path, without importing torch), so that this
    library's DT_NEEDED entry and a later `import torch` both resolve to it -- whatever the order of the imports."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("BPPP_NO_TORCH_PRELOAD"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for d in (spec.submodule_search_locations or []) if spec else []:
        p = os.path.join(d, "lib", "libamdhip64.so")
        if os.path.exists(p):
            try:
                C.CDLL(p, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


def lib():
    """Load libbppp_hip.so (after torch's copy of the HIP runtime, if torch is installed: _preload_torch_hip_runtime), so that one
    runtime serves both and device pointers stay interchangeable."""
    global _lib
    if _lib is not None:
        return _lib
    so = os.environ.get("BPPP_LIB", _build.SO)      # A/B builds of the same ABI (tools/), default: the in-tree build
    if not os.path.exists(so):
        raise ImportError(f"{_build.SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the bp_pp_amd product path)")
    _preload_torch_hip_runtime()
    L = C.CDLL(so)
    vp, sz, i32, u8p = C.c_void_p, C.c_size_t, C.c_int, C.c_char_p
    L.bppp_ctx_create.argtypes = [C.POINTER(vp), u8p, u8p, u8p, i32, i32]
    L.bppp_ctx_create.restype = i32
    L.bppp_wnla_ctx_create.argtypes = [C.POINTER(vp), u8p, u8p, sz, u8p, sz, i32, i32]
    if "BPPP_LIB" not in os.environ or hasattr(L, "bppp_wnla_ctx_create_budget"):
        L.bppp_wnla_ctx_create_budget.argtypes = [C.POINTER(vp), u8p, u8p, sz, u8p, sz, i32, i32, C.c_uint64]
    L.bppp_wnla_commit_batch.argtypes = [vp, sz, vp, vp, vp, sz, vp, sz, vp, vp]
    L.bppp_wnla_verify_batch.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp, sz, vp, vp, vp, sz, vp, sz, vp, vp]
    L.bppp_reciprocal_verify_batch.argtypes = [vp, u8p, sz, sz, sz, sz, vp, vp, sz, sz, sz, vp, vp]
    if "BPPP_LIB" not in os.environ or hasattr(L, "bppp_wnla_verify_batch_device"):
        L.bppp_wnla_verify_batch_device.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp, sz, vp, vp, vp, sz, vp, sz, vp, vp]
        L.bppp_circuit_verify_batch_device.argtypes = [vp, vp, u8p, sz, sz, vp, vp, sz, sz, sz, vp, vp]
    L.bppp_reciprocal_verify_batch_device.argtypes = [vp, u8p, sz, sz, sz, sz, vp, vp, sz, sz, sz, vp, vp]
    L.bppp_reciprocal_verify_batch_rlc_device.argtypes = [vp, u8p, sz, sz, sz, sz, vp, vp, sz, sz, sz, vp, vp, u8p]
    L.bppp_reciprocal_verify_batch_rlc.argtypes = [vp, u8p, sz, sz, sz, sz, vp, vp, sz, sz, sz, vp, vp, u8p]
    L.bppp_wnla_proof_shape.argtypes = [sz, sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    L.bppp_wnla_proof_shape.restype = None
    L.bppp_wnla_prove_batch.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp, vp, sz, vp, sz, vp, vp, vp, vp, vp]
    L.bppp_msm_batch.argtypes = [vp, sz, sz, vp, vp, vp, vp]
    L.bppp_circuit_create.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.bppp_circuit_destroy.argtypes = [vp]
    L.bppp_circuit_destroy.restype = None
    L.bppp_circuit_verify_batch.argtypes = [vp, vp, u8p, sz, sz, vp, vp, sz, sz, sz, vp, vp]
    L.bppp_circuit_prove_batch.argtypes = [vp, vp, u8p, sz, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.bppp_reciprocal_prove_batch.argtypes = [vp, u8p, sz, sz, sz, sz, vp, vp, vp, vp, vp, vp, vp, vp]
    L.bppp_ctx_destroy.argtypes = [vp]
    L.bppp_ctx_destroy.restype = None
    L.bppp_ctx_set_stream.argtypes = [vp, vp]
    L.bppp_ctx_synchronize.argtypes = [vp]
    L.bppp_ctx_set_option.argtypes = [vp, u8p, C.c_long]
    L.bppp_ctx_get_option.argtypes = [vp, u8p]
    L.bppp_ctx_get_option.restype = C.c_long
    if "BPPP_LIB" not in os.environ or hasattr(L, "bppp_u64_plan"):      # (an A/B library of an earlier round lacks the plan exports)
        L.bppp_u64_plan.argtypes = [i32, sz, i32, i32]
        L.bppp_u64_plan.restype = C.c_long
        L.bppp_plan_describe.argtypes = [C.c_long, i32, C.c_char_p, sz]
    L.bppp_u64_verify_batch.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp]
    L.bppp_u64_verify_batch_device.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp, vp, vp]
    L.bppp_u64_verify_batch_rlc_device.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp, vp, u8p]
    L.bppp_u64_verify_batch_rlc.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp, u8p]
    L.bppp_u64_verify_batch_sec1.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp]
    L.bppp_u64_verify_batch_sec1_device.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp, vp, vp]
    L.bppp_u64_commit_value_batch.argtypes = [vp, sz, vp, vp, vp]
    L.bppp_u64_prove_batch.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp, vp, vp]
    L.bppp_u64_prove_batch_device.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp, vp, vp]
    L.bppp_ctx_enable_timing.argtypes = [vp, i32]
    L.bppp_ctx_get_timings.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64), i32]
    L.bppp_ctx_device_bytes.argtypes = [vp]
    L.bppp_ctx_device_bytes.restype = sz
    L.bppp_u64_verify_batch_transcript.argtypes = [vp, sz, vp, sz, vp, vp, vp, vp, vp]
    L.bppp_u64_verify_batch_transcript_device.argtypes = [vp, sz, vp, sz, vp, vp, vp, vp, vp, vp]
    L.bppp_u64_prove_batch_transcript.argtypes = [vp, sz, vp, sz, vp, vp, vp, vp, vp, vp, vp]
    L.bppp_u64_prove_batch_transcript_device.argtypes = [vp, sz, vp, sz, vp, vp, vp, vp, vp, vp, vp]
    L.bppp_wnla_verify_batch_transcript.argtypes = [vp, sz, vp, sz, vp, vp, vp, vp, sz, vp, vp, vp, sz, vp, sz, vp, vp, vp]
    L.bppp_reciprocal_verify_batch_transcript.argtypes = [vp, sz, vp, sz, sz, sz, vp, vp, sz, sz, sz, vp, vp, vp]
    L.bppp_circuit_verify_batch_transcript.argtypes = [vp, vp, sz, vp, sz, vp, vp, sz, sz, sz, vp, vp, vp]
    L.bppp_wnla_prove_batch_transcript.argtypes = [vp, sz, vp, sz, vp, vp, vp, vp, vp, sz, vp, sz, vp, vp, vp, vp, vp, vp]
    L.bppp_reciprocal_prove_batch_transcript.argtypes = [vp, sz, vp, sz, sz, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.bppp_circuit_prove_batch_transcript.argtypes = [vp, vp, sz, vp, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.bppp_transcript_new.argtypes = [u8p, sz, vp]
    L.bppp_transcript_append_message.argtypes = [vp, u8p, sz, u8p, sz]
    L.bppp_transcript_challenge_bytes.argtypes = [vp, u8p, sz, vp, sz]
    L.bppp_derive_generators.argtypes = [u8p, sz, sz, sz, vp]
    L.bppp_ctx_save_tables.argtypes = [vp, u8p]
    L.bppp_ctx_create_from_tables.argtypes = [C.POINTER(vp), u8p, i32]
    L.bppp_ctx_create_shared.argtypes = [C.POINTER(vp), vp]
    L.bppp_shard_range.argtypes = [sz, i32, i32, C.POINTER(sz), C.POINTER(sz)]
    L.bppp_shard_range.restype = None
    L.bppp_group_create.argtypes = [C.POINTER(vp), u8p, u8p, u8p, C.POINTER(i32), i32, i32]
    L.bppp_group_destroy.argtypes = [vp]
    L.bppp_group_destroy.restype = None
    L.bppp_group_size.argtypes = [vp]
    L.bppp_group_ctx.argtypes = [vp, i32]
    L.bppp_group_ctx.restype = vp
    L.bppp_u64_verify_batch_sharded.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp, C.POINTER(C.c_int32)]
    L.bppp_u64_verify_batch_sharded_device.argtypes = [vp, u8p, sz, sz, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    pvp = C.POINTER(vp)
    L.bppp_wnla_group_create.argtypes = [C.POINTER(vp), u8p, u8p, sz, u8p, sz, C.POINTER(i32), i32, i32]
    L.bppp_group_set_option.argtypes = [vp, u8p, C.c_long]
    L.bppp_u64_verify_batch_rlc_sharded.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp, C.POINTER(C.c_int32), u8p]
    L.bppp_u64_verify_batch_rlc_sharded_device.argtypes = [vp, u8p, sz, sz, pvp, pvp, pvp, pvp, pvp, u8p]
    L.bppp_u64_verify_batch_sec1_sharded.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp, C.POINTER(C.c_int32)]
    L.bppp_u64_verify_batch_sec1_sharded_device.argtypes = [vp, u8p, sz, sz, pvp, pvp, pvp, pvp, pvp]
    L.bppp_u64_verify_batch_transcript_sharded.argtypes = [vp, sz, vp, sz, vp, vp, vp, vp, vp, C.POINTER(C.c_int32)]
    L.bppp_u64_verify_batch_transcript_sharded_device.argtypes = [vp, sz, pvp, sz, pvp, pvp, pvp, pvp, pvp, pvp]
    L.bppp_u64_prove_batch_sec1.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp, vp, vp]
    L.bppp_u64_prove_batch_sec1_device.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp, vp, vp]
    L.bppp_u64_prove_batch_sharded.argtypes = [vp, u8p, sz, sz, vp, vp, vp, vp, vp, vp]
    L.bppp_u64_prove_batch_sharded_device.argtypes = [vp, u8p, sz, sz, pvp, pvp, pvp, pvp, pvp, pvp]
    L.bppp_reciprocal_verify_batch_sharded.argtypes = [vp, u8p, sz, sz, sz, sz, vp, vp, sz, sz, sz, vp, vp, C.POINTER(C.c_int32)]
    L.bppp_reciprocal_verify_batch_rlc_sharded.argtypes = [vp, u8p, sz, sz, sz, sz, vp, vp, sz, sz, sz, vp, vp, C.POINTER(C.c_int32), u8p]
    L.bppp_reciprocal_verify_batch_sharded_device.argtypes = [vp, u8p, sz, sz, sz, sz, pvp, pvp, sz, sz, sz, pvp, pvp, pvp]
    L.bppp_reciprocal_verify_batch_rlc_sharded_device.argtypes = [vp, u8p, sz, sz, sz, sz, pvp, pvp, sz, sz, sz, pvp, pvp, pvp, u8p]
    L.bppp_u64_verify_one.argtypes = [vp, u8p, sz, vp, vp, vp, vp]
    L.bppp_u64_verify_one_transcript.argtypes = [vp, vp, vp, vp, vp, vp]
    L.bppp_u64_prove_one.argtypes = [vp, u8p, sz, C.c_uint64, vp, vp, vp, vp, vp]
    L.bppp_u64_prove_one_transcript.argtypes = [vp, vp, C.c_uint64, vp, vp, vp, vp, vp]
    L.bppp_ctx_get_coalesce_stats.argtypes = [vp, i32, C.POINTER(C.c_uint64)]
    L.bppp_reciprocal_verify_one.argtypes = [vp, u8p, sz, sz, sz, vp, vp, sz, sz, sz, vp, vp]
    L.bppp_reciprocal_verify_one_transcript.argtypes = [vp, vp, sz, sz, vp, vp, sz, sz, sz, vp, vp]
    L.bppp_strerror.argtypes = [i32]
    L.bppp_strerror.restype = C.c_char_p
    L.bppp_last_error.restype = C.c_char_p
    _lib = L
    return L


def check(rc: int):
    if rc < 0:
        raise BpppError(rc, lib().bppp_last_error().decode())
    return rc
