"""secp256k1 known answers from an implementation that shares nothing with this repository: the OpenSSL command-line tool
(tests/golden/openssl_secp256k1.json, made by tests/golden/make_openssl_vectors.py).  k256 0.13.3 -- the arithmetic under the
reference -- is not in /root/reference (Cargo.lock:411), so this is how the curve layer is pinned: both oracles, the device code
(compiled for the host) and, on the GPU tier, the library itself reproduce OpenSSL's k*G (affine and the 33-byte SEC1 form the
transcripts hash) and the x coordinate of k*P."""
import ctypes as C
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import bppp_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "openssl_secp256k1.json")


@pytest.fixture(scope="module")
def vec():
    with open(GOLD) as f:
        return json.load(f)


def test_python_oracle_reproduces_openssl(vec):
    for v in vec["mul_g"]:
        P = O.pt_mul(O.G, int(v["k"], 16))
        assert O.pt_to_xy64(P).hex() == v["xy"] and O.pt_to_bytes(P).hex() == v["sec1"]
        assert O.pt_from_bytes(bytes.fromhex(v["sec1"])) == P                       # decompression (sqrt) too
    for v in vec["ecdh"]:
        Q = O.pt_mul(O.pt_from_xy64(bytes.fromhex(v["peer_xy"])), int(v["k"], 16))
        assert O.pt_to_xy64(Q)[:32].hex() == v["x"]


def test_c_oracle_reproduces_openssl(vec, oracle_c):
    for v in vec["mul_g"]:
        assert oracle_c.point_mul(None, bytes.fromhex(v["k"])).hex() == v["xy"]
    for v in vec["ecdh"]:
        assert oracle_c.point_mul(bytes.fromhex(v["peer_xy"]), bytes.fromhex(v["k"]))[:32].hex() == v["x"]


@pytest.mark.parametrize("W", [4, 10])
def test_device_code_reproduces_openssl(vec, W):
    """fixed-base tables + signed-window sums (k*G), GLV + shared-doubling window sums (k*P), SEC1 expansion: the exact
    __host__ __device__ functions of bp_pp_amd/csrc, compiled for the host."""
    from emul.build import load
    L = load()
    G = bytes.fromhex(vec["mul_g"][0]["xy"])
    assert int(vec["mul_g"][0]["k"], 16) == 1
    tab = np.zeros(L.emul_fb_table_entries(1, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(G, 1, W, tab.ctypes.data) == 0
    out, fb = C.create_string_buffer(64), C.c_int()
    for v in vec["mul_g"]:
        assert L.emul_fb_msm(tab.ctypes.data, W, 0, 1, bytes.fromhex(v["k"]), out) == 0 and out.raw.hex() == v["xy"]
        assert L.emul_fb_msm_lanes(tab.ctypes.data, W, 0, 1, bytes.fromhex(v["k"]), out, C.byref(fb)) == 0 and out.raw.hex() == v["xy"]
    for v in vec["ecdh"]:
        assert L.emul_straus_glv(1, bytes.fromhex(v["peer_xy"]), bytes.fromhex(v["k"]), out) == 0 and out.raw[:32].hex() == v["x"]
        assert L.emul_straus_affine(1, bytes.fromhex(v["peer_xy"]), bytes.fromhex(v["k"]), out, C.byref(fb)) == 0 and out.raw[:32].hex() == v["x"]
    # SEC1 compressed -> affine on the device code (the wire-format path): y recovered by the square root
    from bp_pp_amd import wire
    n = len(vec["mul_g"])
    c33 = np.frombuffer(b"".join(bytes.fromhex(v["sec1"]) for v in vec["mul_g"]), np.uint8).reshape(n, 33).copy()
    one = bytes.fromhex(vec["mul_g"][0]["sec1"])
    p525 = np.frombuffer((one * 13 + bytes(96)) * n, np.uint8).reshape(n, 525).copy()
    c64, p928 = np.zeros((n, 64), np.uint8), np.zeros((n, 928), np.uint8)
    L.emul_sec1_expand(n, c33.ctypes.data, p525.ctypes.data, c64.ctypes.data, p928.ctypes.data)
    assert [bytes(r).hex() for r in c64] == [v["xy"] for v in vec["mul_g"]] and bytes(p928[0, :64]).hex() == vec["mul_g"][0]["xy"]
    for v in vec["mul_g"]:
        assert wire.decompress_point(bytes.fromhex(v["sec1"])).hex() == v["xy"] and wire.compress_point(bytes.fromhex(v["xy"])).hex() == v["sec1"]


def test_the_committed_vectors_are_what_openssl_says_here(vec):
    """Re-derive a sample with the OpenSSL on this machine (the fixture is data; this shows where it came from)."""
    if not shutil.which("openssl"):
        pytest.skip("no openssl command-line tool here")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_openssl_vectors", os.path.join(os.path.dirname(GOLD), "make_openssl_vectors.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    for v in vec["mul_g"][::7]:
        assert M.pub(int(v["k"], 16), "uncompressed")[-64:].hex() == v["xy"]
    v = vec["ecdh"][3]
    # the peer's SubjectPublicKeyInfo again from its point: same header as any uncompressed secp256k1 key
    hdr = M.pub(1, "uncompressed")[:-64]
    assert M.ecdh_x(int(v["k"], 16), hdr + bytes.fromhex(v["peer_xy"])).hex() == v["x"]


@pytest.mark.gpu
def test_gpu_library_reproduces_openssl(vec):
    """bppp_msm_batch on a context whose generators are G and the fixture's peer points: k*G and x(k*P) through the GPU's
    fixed-base path; bppp_u64_verify_batch_sec1's decompression is covered by test_gpu_verify.py."""
    import torch
    if torch.cuda.device_count() == 0:
        pytest.fail("needs a GPU")
    from bp_pp_amd.wnla import WeightNormLinearArgument
    G = bytes.fromhex(vec["mul_g"][0]["xy"])
    peers = sorted({v["peer_xy"] for v in vec["ecdh"]})
    for W in (8, 16):
        w = WeightNormLinearArgument(G, [bytes.fromhex(p) for p in peers], [G], device=0, fb_window_bits=W)
        try:
            ks = np.frombuffer(b"".join(bytes.fromhex(v["k"]) for v in vec["mul_g"]), np.uint8).reshape(-1, 1, 32)
            out, st = w.msm_batch([0], ks)
            assert not st.any() and [bytes(o).hex() for o in out] == [v["xy"] for v in vec["mul_g"]]
            for v in vec["ecdh"]:
                out, st = w.msm_batch([1 + peers.index(v["peer_xy"])], np.frombuffer(bytes.fromhex(v["k"]), np.uint8).reshape(1, 1, 32))
                assert not st.any() and bytes(out[0])[:32].hex() == v["x"]
        finally:
            w.close()
