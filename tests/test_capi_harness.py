"""tools/capi_harness.c: a plain-C caller of libbppp_hip.so that sees nothing but include/bppp.h (what a cgo / Rust-FFI / JNI binding
sees).  CPU tier: the header is valid C99 under -Wall -Wextra -Werror, the harness links against the built library, and without a
GPU it stops at bppp_ctx_create with BPPP_ERR_NO_DEVICE (no CPU fallback).  GPU tier: its verdicts over the committed golden proofs,
prover output, transcript variants and the one-device group are the ones the fixture states."""
import json
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    from bp_pp_amd import _build
    _build.build()
    libdir = os.path.join(ROOT, "bp_pp_amd")
    exe = str(tmp_path_factory.mktemp("capi") / "capi_harness")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tools", "capi_harness.c"), "-o", exe, "-L" + libdir, "-lbppp_hip", "-Wl,-rpath," + libdir], check=True)
    return exe


@pytest.fixture(scope="module")
def fixture_file(tmp_path_factory):
    with open(os.path.join(ROOT, "tests", "golden", "u64_golden.json")) as f:
        gold = json.load(f)
    label, cases = bytes.fromhex(gold["label"]), gold["cases"]
    blob = b"BPPPFIX1" + struct.pack("<I", len(label)) + label + struct.pack("<I", len(cases)) + bytes.fromhex(gold["generators"])
    blob += b"".join(bytes.fromhex(c["commitment"]) for c in cases) + b"".join(bytes.fromhex(c["proof"]) for c in cases)
    blob += b"".join(struct.pack("<Q", int(c["x"])) for c in cases) + b"".join(bytes.fromhex(c["s"]) for c in cases)
    blob += b"".join(bytes.fromhex(c["rnd"]) for c in cases)
    path = str(tmp_path_factory.mktemp("capi_fx") / "fixture.bin")
    with open(path, "wb") as f:
        f.write(blob)
    return path, gold


def _run(exe, *args):
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600)
    return r.returncode, dict(line.split(" ", 1) for line in r.stdout.splitlines() if " " in line), r.stderr


def test_header_is_plain_c_and_harness_fails_loudly_without_a_gpu(harness, fixture_file):
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present: the GPU-tier test covers the harness")
    rc, out, err = _run(harness, fixture_file[0])
    assert out == {"null_ctx_rc": "-2"}                       # BPPP_ERR_INVALID_ARG needs no device
    assert rc == 11 and "bppp_ctx_create" in err and "-1" in err  # BPPP_ERR_NO_DEVICE: nothing computes on the CPU


@pytest.mark.gpu
@pytest.mark.parametrize("wbits", [8, 16])
def test_c_caller_gets_the_fixture_verdicts(harness, fixture_file, wbits):
    path, gold = fixture_file
    n = len(gold["cases"])
    rc, out, err = _run(harness, path, str(wbits))
    assert rc == 0, err
    ones, alt = "1" * n, "".join("0" if i % 2 == 0 else "1" for i in range(n))
    assert [int(c["accept"]) for c in gold["cases"]] == [1] * n
    assert out["null_ctx_rc"] == "-2" and out["empty_rc"] == "0"
    assert out["verify"] == ones and out["status"].split() == ["0"] * n
    assert out["prove_same_proofs"] == "1" and out["prove_same_commitments"] == "1"
    assert out["sec1_same_x"] == "1" and out["verify_sec1"] == ones
    assert out["verify_transcript"] == ones and out["verify_other_transcript"] == "0" * n
    assert out["verify_flipped"] == out["verify_clone"] == out["verify_group"] == alt
    assert out["shard_range"] == "0 %d" % n and out["group_size"] == "1" and out["reject_count"] == str((n + 1) // 2)
    # the transcript handed back for proof 0 is the reference's `t` after verify: the oracle's next challenge from it agrees
    import bppp_oracle as O
    from transcript_cases import ser
    gens = bytes.fromhex(gold["generators"])
    pts = [O.pt_from_xy64(gens[64 * i:64 * i + 64]) for i in range(49)]
    proto = O.U64RangeProofProtocol(pts[0], pts[1:17], pts[17:49])
    t = O.Transcript(bytes.fromhex(gold["label"]))
    c0 = gold["cases"][0]
    assert proto.verify(O.pt_from_xy64(bytes.fromhex(c0["commitment"])), O.u64_proof_from_bytes(bytes.fromhex(c0["proof"])), t)
    assert len(ser(t)) == 203 and t.challenge_bytes(b"next", 32).hex() == out["next_challenge"]
    assert out["done"] == "1"
