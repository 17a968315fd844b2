"""Generates tests/golden/openssl_secp256k1.json with the OpenSSL command-line tool (3.0.2 in this image): an implementation of
secp256k1 that shares nothing with this repository.  k256 is absent from /root/reference (Cargo.lock:411), so the arithmetic under
the reference is pinned here against published behaviour of the curve instead: for seeded and edge-case scalars k,
  * `openssl ec -pubout` gives k*G in SEC1 uncompressed and compressed form (fixed-base multiplication, affine conversion and the
    33-byte encoding the transcripts hash: transcript.rs:7, k256 `to_encoded_point(true)`);
  * `openssl pkeyutl -derive` gives the x coordinate of k*P for a peer point P (variable-base multiplication).
Only the OUTPUTS are committed; the tests compare the oracles, the device code (host emulation) and the GPU library with them.
Run:  python tests/golden/make_openssl_vectors.py"""
import hashlib
import json
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
LAMBDA = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72      # the GLV eigenvalue: k = lambda exercises phi


def _der(k: int) -> bytes:   # SEC1 ECPrivateKey { version 1, privateKey, [0] secp256k1 }
    return bytes.fromhex("302e0201010420") + k.to_bytes(32, "big") + bytes.fromhex("a00706052b8104000a")


def _run(args, data_files):
    with tempfile.TemporaryDirectory() as d:
        argv = []
        for a in args:
            if a in data_files:
                p = os.path.join(d, a)
                with open(p, "wb") as f:
                    f.write(data_files[a])
                argv.append(p)
            else:
                argv.append(a)
        r = subprocess.run(["openssl"] + argv, capture_output=True, check=True)
        return r.stdout


def pub(k: int, form: str) -> bytes:
    out = _run(["ec", "-inform", "DER", "-in", "k.der", "-pubout", "-outform", "DER", "-conv_form", form], {"k.der": _der(k)})
    return out


def ecdh_x(k: int, peer_spki: bytes) -> bytes:
    return _run(["pkeyutl", "-derive", "-inkey", "k.der", "-keyform", "DER", "-peerkey", "p.der", "-peerform", "DER"],
                {"k.der": _der(k), "p.der": peer_spki})


def main():
    sc = lambda tag, i: int.from_bytes(hashlib.shake_256(b"bppp-openssl-vectors" + tag + bytes([i])).digest(40), "big") % (N - 1) + 1
    ks = [1, 2, 3, 15, 16, 17, 2**19, 2**20 - 1, 2**20, 2**64 - 1, 2**128, 2**255, N - 1, N - 2, (N - 1) // 2, LAMBDA, N - LAMBDA] + \
         [sc(b"k", i) for i in range(24)]
    mul_g = []
    for k in ks:
        u, c = pub(k, "uncompressed"), pub(k, "compressed")
        assert u[-65] == 4 and c[-33] in (2, 3)
        mul_g.append({"k": "%064x" % k, "xy": u[-64:].hex(), "sec1": c[-33:].hex()})
    ecdh = []
    peers = [2, N - 1, LAMBDA] + [sc(b"peer", i) for i in range(9)]
    mults = [2, 3, N - 1, 2**128 - 1, LAMBDA] + [sc(b"m", i) for i in range(11)]
    for i, a in enumerate(peers):
        spki = pub(a, "uncompressed")
        for k in (mults[i], mults[(i + 5) % len(mults)]):
            if k * a % N == 0:
                continue
            ecdh.append({"k": "%064x" % k, "peer_xy": spki[-64:].hex(), "x": ecdh_x(k, spki).hex()})
    ver = subprocess.run(["openssl", "version"], capture_output=True, text=True).stdout.strip()
    doc = {"about": "secp256k1 known answers from the OpenSSL CLI (tests/golden/make_openssl_vectors.py); xy = affine big-endian x||y, "
                    "sec1 = 33-byte compressed, x = affine x of k*peer", "openssl": ver, "mul_g": mul_g, "ecdh": ecdh}
    with open(os.path.join(HERE, "openssl_secp256k1.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote openssl_secp256k1.json:", len(mul_g), "k*G,", len(ecdh), "k*P  (", ver, ")")


if __name__ == "__main__":
    main()
