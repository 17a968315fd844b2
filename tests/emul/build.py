"""Builds the TEST-ONLY host emulation of the device code (tests/emul/bppp_emul.cpp) with g++."""
import ctypes as C
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# BPPP_EMUL_SANITIZE=1: AddressSanitizer + UBSan build of the same translation unit (tools/sanitize_emul.sh preloads libasan and
# runs the emulation tests on it) -- the device code's indexing and integer arithmetic checked on the CPU, where sanitizers exist
SANITIZE = os.environ.get("BPPP_EMUL_SANITIZE") == "1"
SO = os.path.join(HERE, "libbppp_emul_san.so" if SANITIZE else "libbppp_emul.so")


def load():
    src = os.path.join(HERE, "bppp_emul.cpp")
    deps = [src] + glob.glob(os.path.join(ROOT, "bp_pp_amd", "csrc", "*.h"))
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"] if SANITIZE else ["-O2"]
        subprocess.check_call(["g++", *flags, "-shared", "-fPIC", "-std=c++17", "-pthread", "-o", SO, src])
    L = C.CDLL(SO)
    vp, sz, i32, cp = C.c_void_p, C.c_size_t, C.c_int, C.c_char_p
    L.emul_fb_table_entries.restype = sz
    L.emul_fb_table_entries.argtypes = [i32, i32]
    L.emul_fb_build.argtypes = [cp, i32, i32, vp]
    L.emul_fb_digit.argtypes = [i32, cp, i32, vp, vp, vp]
    L.emul_fb_msm.argtypes = [vp, i32, i32, i32, cp, vp]
    L.emul_fb_msm_lanes.argtypes = [vp, i32, i32, i32, cp, vp, vp]
    L.emul_fb_msm_lanes_nl.argtypes = [vp, i32, i32, i32, cp, vp, vp, i32]
    L.emul_fb_msm_mixed.argtypes = [vp, i32, i32, vp, i32, i32, i32, cp, vp, i32]
    L.emul_straus.argtypes = [i32, cp, cp, vp]
    L.emul_inv.argtypes = [i32, cp, vp, vp]
    L.emul_straus_affine.argtypes = [i32, cp, cp, vp, vp]
    L.emul_straus_glv.argtypes = [i32, cp, cp, vp]
    L.emul_glv_split.argtypes = [cp, vp, vp, vp, vp]
    L.emul_pt_op.argtypes = [i32, cp, cp, vp]
    L.emul_merlin_kat.argtypes = [cp, sz, cp, sz, vp, sz]
    L.emul_u64_verify_batch.argtypes = [vp, i32, cp, sz, sz, vp, vp, vp, vp, vp]
    L.emul_u64_verify_batch_transcript.argtypes = [vp, i32, sz, vp, sz, vp, vp, vp, vp, vp]
    L.emul_u64_bucket_stage.argtypes = [vp, i32, cp, sz, sz, vp, vp, cp, C.c_uint32, vp, vp]
    L.emul_u64_verify_batch_rlc.argtypes = [vp, i32, cp, sz, sz, vp, vp, cp, vp, vp, vp, vp, vp]
    L.emul_u64_prove_batch.argtypes = [vp, i32, cp, sz, sz, vp, vp, vp, vp, vp, vp]
    L.emul_u64_prove_batch_transcript.argtypes = [vp, i32, sz, vp, sz, vp, vp, vp, vp, vp, vp, vp]
    L.emul_set_transcripts.argtypes = [vp, sz, vp]
    L.emul_set_transcripts.restype = None
    L.emul_set_rlc.argtypes = [cp, vp]
    L.emul_set_rlc.restype = None
    L.emul_set_generic_slow_rounds.argtypes = [i32]
    L.emul_set_generic_slow_rounds.restype = None
    L.emul_sec1_expand.argtypes = [sz, vp, vp, vp, vp]
    L.emul_sec1_compress.argtypes = [sz, vp, vp, vp, vp]
    L.emul_sec1_compress.restype = None
    L.emul_wnla_run.argtypes = [i32, vp, i32, i32, i32, cp, sz, sz, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp]
    L.emul_recip_verify.argtypes = [vp, i32, i32, i32, i32, i32, cp, sz, sz, vp, vp, i32, i32, i32, vp, vp]
    L.emul_circuit_verify.argtypes = [vp, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, cp, sz, sz, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.emul_wnla_prove.argtypes = [vp, i32, i32, i32, cp, sz, sz, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.emul_circuit_prove.argtypes = [vp, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, cp, sz, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.emul_recip_prove.argtypes = [vp, i32, i32, i32, i32, i32, cp, sz, sz, vp, vp, vp, vp, vp, vp, vp, vp]
    L.emul_set_rlc_superchunk.argtypes = [C.c_uint32, vp]
    L.emul_set_rlc_superchunk.restype = None
    L.emul_group_verify.argtypes = [i32, i32, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, cp, sz, sz, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.emul_group_missing_ranks.argtypes = [i32, i32, vp]
    L.emul_set_rlc_chunk.argtypes = [i32]
    L.emul_set_rlc_chunk.restype = None
    L.emul_set_final_scalars_lg.argtypes = [i32]
    L.emul_set_final_scalars_lg.restype = None
    L.emul_set_shared_inv.argtypes = [i32]
    L.emul_set_shared_inv.restype = None
    L.emul_fb_shape.argtypes = [i32, i32, i32, vp]
    L.emul_fb_shape.restype = None
    L.emul_fe_batch_inv.argtypes = [i32, sz, cp, vp, i32]
    L.emul_plan_rlc.argtypes = [C.c_uint, i32, i32, C.c_double, vp]
    L.emul_plan_rlc.restype = None
    L.emul_trace_begin.argtypes = []
    L.emul_trace_begin.restype = None
    L.emul_trace_end.argtypes = [vp, sz]
    L.emul_trace_end.restype = sz
    L.emul_group_prove.argtypes = [i32, i32, vp, i32, cp, sz, sz, vp, vp, vp, vp, vp, vp]
    L.emul_straus_split.argtypes = [i32, i32, cp, cp, vp, vp, vp]
    L.emul_set_prove_next_by_msm.argtypes = [i32]
    L.emul_set_prove_next_by_msm.restype = None
    L.emul_set_prove_ct.argtypes = [i32]
    L.emul_set_prove_ct.restype = None
    L.emul_fb_lookup_both.argtypes = [vp, i32, i32, cp, cp, vp, vp]
    pvp = C.POINTER(vp)
    L.emul_coalesce_run.argtypes = [i32, vp, i32, i32, sz, pvp, pvp, vp, sz, C.c_long, i32, i32, i32, i32, vp, vp, sz, vp]
    return L
